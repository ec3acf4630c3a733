#!/usr/bin/env python3
"""Which 16-bit roundings of the step's forward make the logit error?  BASELINE config 1 (ViT-S/16, r = 64, bs 8) on the fp64 CPU
oracle with ONE group of values rounded to the operand type at a time, then all together, over the eight input batches of the
parity fixtures (tests/golden/g5_cfg1_vits.npz + g5_cfg1_seeds.npz): max|logits - exact| / max|exact|, max and mean over batches.

    python3 tools/rounding_sites.py [fp16|bf16]  > profiles/r04_rounding_sites_fp16.md

Sites (what the HIP path stores or feeds in 16 bits): W = the frozen Linear weights as GEMM operands; xhat = LayerNorm output (GEMM
operand); qkv = stored q, k, v; P = softmax probabilities before P x V; o = attention output; h = GELU output; branch = the proj / fc2
outputs added into the fp32 residual stream; patch = patch-embedding operands."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from oracle import apla_oracle as O
    from test_engine_gpu import build_classifier
    dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == "fp16") else torch.bfloat16
    rd = lambda t: t.to(dt).double()  # noqa: E731
    tp = dict(img_size=[224], patch_size=16, pretrained_type="dinov2", is_memory_efficient=True,
              block_conf=dict(has_layerscale=True, layerscale_init_values=1.0))
    model = build_classifier("vit_small", 64, 10, tp, seed=0)
    p = {(k[len("backbone."):] if k.startswith("backbone.") else k): (v.detach().double() if v.is_floating_point() else v.detach().clone()) for k, v in model.state_dict().items()}
    cfg = dict(patch=16, depth=12, heads=6, r=64)
    gs = np.load(os.path.join(ROOT, "tests/golden/g5_cfg1_seeds.npz"))
    seeds = [0] + [int(s) for s in gs["seeds"]]
    lin0, att0, gelu0, ln0, pe0 = O.linear_fwd, O.attention_fwd, O.gelu_fwd, O.layernorm_fwd, O.patch_embed
    on = set()

    def lin(x, W, b):
        if W.shape[0] == 10:                       # the head: fp32 in the product
            return lin0(x, W, b)
        kind = {(1152, 384): "qkv", (1536, 384): "fc1", (384, 1536): "fc2"}.get(tuple(W.shape), "proj")
        xin = x
        if kind in ("qkv", "fc1") and "xhat" in on:
            xin = rd(x)
        if kind == "proj" and "o" in on:
            xin = rd(x)
        if kind == "fc2" and "h" in on:
            xin = rd(x)
        y = lin0(xin, rd(W) if "W" in on else W, b)
        if kind == "qkv" and "qkv" in on:
            y = rd(y)
        if kind in ("proj", "fc2") and "branch" in on:
            y = rd(y)
        return y

    def att(qkv, H, scale, return_attn=False):
        if "P" not in on:
            return att0(qkv, H, scale, return_attn)
        B, N, D3 = qkv.shape
        D = D3 // 3
        q, k, v = qkv.reshape(B, N, 3, H, D // H).permute(2, 0, 3, 1, 4)
        s = (q @ k.transpose(-1, -2)) * scale
        m = s.max(-1, keepdim=True).values
        e = torch.exp(s - m)
        o = (rd(e) @ v) / e.sum(-1, keepdim=True)   # the kernel multiplies the 16-bit rounded un-normalised probabilities
        lse = (m + torch.log(e.sum(-1, keepdim=True))).squeeze(-1)
        o = o.transpose(1, 2).reshape(B, N, D)
        return (o, lse, None) if return_attn else (o, lse)

    def pe(images, W, b, patch):
        if "patch" in on:
            return pe0(rd(images), rd(W), b, patch)
        return pe0(images, W, b, patch)

    O.linear_fwd, O.attention_fwd, O.patch_embed = lin, att, pe
    sites = ["W", "xhat", "qkv", "P", "o", "h", "branch", "patch"]
    batches = []
    for sd in seeds:
        g = torch.Generator().manual_seed(sd)
        batches.append(torch.randn(8, 3, 224, 224, generator=g).double())
    on.clear()
    exact = [O.vit_forward(im, p, cfg, keep_ctx=False)[0] for im in batches]
    print(f"# Logit error of BASELINE config 1 by rounding site ({'fp16' if dt == torch.float16 else 'bf16'} roundings on the fp64 oracle, eight batches)\n")
    print("| rounded | max over batches | mean over batches |")
    print("|---|---:|---:|")
    rows = [[s] for s in sites] + [sites, [s for s in sites if s != "branch"], [s for s in sites if s not in ("branch", "qkv")],
                                   [s for s in sites if s not in ("branch", "xhat")], [s for s in sites if s != "h"], [s for s in sites if s != "W"]]
    for row in rows:
        on.clear()
        on.update(row)
        errs = []
        for im, ex in zip(batches, exact):
            lg, _ = O.vit_forward(im, p, cfg, keep_ctx=False)
            errs.append(float((lg - ex).abs().max() / ex.abs().max()))
        name = "all" if len(row) == len(sites) else ("only " + row[0] if len(row) == 1 else "all but " + ", ".join(s for s in sites if s not in row))
        print(f"| {name} | {max(errs):.2e} | {sum(errs) / len(errs):.2e} |", flush=True)
    O.linear_fwd, O.attention_fwd, O.gelu_fwd, O.layernorm_fwd, O.patch_embed = lin0, att0, gelu0, ln0, pe0


if __name__ == "__main__":
    main()
