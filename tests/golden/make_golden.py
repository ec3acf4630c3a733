"""Generate golden vectors from the ACTUAL reference code (build container only).

    python tests/golden/make_golden.py

Imports the reference's ``apla/*.py`` and ``utils/transformers/vit.py`` in place from
/root/reference through ``_ref_shim`` and writes plain-array fixtures (``*.npz`` + ``*.json``)
next to this file.  Fixtures are data only: inputs, parameters and the reference's outputs.
Groups follow SURVEY.md §8c (G1..G8).
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_shim import Cfg, REF_SRC, load_reference  # noqa: E402

vit, attn_mod, avit, mem_mod = load_reference()
torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy().copy()  # copy: later optimizer steps must not alias saved arrays


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1e6:.2f} MB")


def tensor_digest(t: torch.Tensor) -> str:
    return hashlib.sha256(npy(t.contiguous()).tobytes()).hexdigest()[:16]


# ---------------------------------------------------------------- G1: index selection
def g1():
    out = {}
    for seed in (0, 7, 123):
        for D in (384, 768, 1024, 1536):
            torch.manual_seed(seed)
            # exactly what APLA_Attention.__init__ draws when indices=None (appla_attn.py:26)
            m = attn_mod.APLA_Attention(Cfg(partial_size=8), D, num_heads=D // 64, qkv_bias=True)
            torch.manual_seed(seed)
            assert torch.equal(m.inds, torch.randperm(D))
            out[f"seed{seed}_D{D}"] = npy(m.inds).astype(np.int32)
    save("g1_indices.npz", **out)


# ---------------------------------------------------------------- G2 + G8: weight split / inds JSON
def g2_g8():
    torch.manual_seed(11)
    D, H, r, L = 64, 2, 8, 2
    model = vit.VisionTransformer(img_size=[32], patch_size=16, embed_dim=D, depth=L, num_heads=H, qkv_bias=True,
                                  block_conf=Cfg(has_layerscale=False))
    for blk in model.blocks:  # make proj bias non-zero so the bias split is visible
        torch.nn.init.normal_(blk.attn.proj.bias, std=0.1)
    full_w = [npy(b.attn.proj.weight).copy() for b in model.blocks]
    full_b = [npy(b.attn.proj.bias).copy() for b in model.blocks]
    # random-sampling path
    torch.manual_seed(5)
    m1 = avit.build_apla(Cfg(partial_size=r), model, "apla_attn")
    arrs = {}
    for i, b in enumerate(m1.blocks):
        arrs[f"rand_full_w{i}"] = full_w[i]
        arrs[f"rand_full_b{i}"] = full_b[i]
        arrs[f"rand_inds{i}"] = npy(b.attn.inds).astype(np.int32)
        arrs[f"rand_w1_{i}"] = npy(b.attn.proj_weight1)
        arrs[f"rand_w2_{i}"] = npy(b.attn.proj_weight2)
        arrs[f"rand_b1_{i}"] = npy(b.attn.proj_bias1)
        arrs[f"rand_b2_{i}"] = npy(b.attn.proj_bias2)
    # inds_path path (apla_vit.py:20-24)
    torch.manual_seed(11)
    model = vit.VisionTransformer(img_size=[32], patch_size=16, embed_dim=D, depth=L, num_heads=H, qkv_bias=True,
                                  block_conf=Cfg(has_layerscale=False))
    g = torch.Generator().manual_seed(3)
    inds_dict = {f"block_{i}": torch.randperm(D, generator=g)[:r].tolist() for i in range(L)}
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(inds_dict, f)
    m2 = avit.build_apla(Cfg(partial_size=r, inds_path=f.name), model, "apla_attn", is_multi_gpu=True)
    os.unlink(f.name)
    for i, b in enumerate(m2.blocks):
        arrs[f"json_trainable{i}"] = np.asarray(inds_dict[f"block_{i}"], dtype=np.int32)
        arrs[f"json_inds{i}"] = npy(b.attn.inds).astype(np.int32)
        arrs[f"json_w1_{i}"] = npy(b.attn.proj_weight1)
        arrs[f"json_full_w{i}"] = npy(model.blocks[i].attn.proj_weight1) * 0  # placeholder, filled below
    # keep the pre-swap weights for the json model as well
    torch.manual_seed(11)
    model0 = vit.VisionTransformer(img_size=[32], patch_size=16, embed_dim=D, depth=L, num_heads=H, qkv_bias=True,
                                   block_conf=Cfg(has_layerscale=False))
    for i, b in enumerate(model0.blocks):
        arrs[f"json_full_w{i}"] = npy(b.attn.proj.weight)
    save("g2_g8_split.npz", **arrs)


# ---------------------------------------------------------------- G3: module forward/backward
def make_module(D, H, r, seed, bias_std=0.05):
    torch.manual_seed(seed)
    m = attn_mod.APLA_Attention(Cfg(partial_size=r), D, num_heads=H, qkv_bias=True)
    with torch.no_grad():
        for n, p_ in m.named_parameters():
            if p_.ndim == 2:
                p_.normal_(std=0.08)
            else:
                p_.normal_(std=bias_std)
    return m


def g3():
    arrs = {}
    for tag, (B, N, D, H, r) in {"tiny": (2, 17, 64, 2, 8), "mid": (2, 197, 128, 2, 32)}.items():
        m = make_module(D, H, r, seed=21)
        torch.manual_seed(22)
        x = torch.randn(B, N, D, requires_grad=True)
        y, attn = m(x)
        loss = y.square().mean()
        loss.backward()
        for k, v in m.state_dict().items():
            arrs[f"{tag}.{k}"] = npy(v) if v.dtype != torch.int64 else npy(v).astype(np.int32)
        arrs[f"{tag}.x"] = npy(x)
        arrs[f"{tag}.y"] = npy(y)
        arrs[f"{tag}.attn"] = npy(attn)
        arrs[f"{tag}.dx"] = npy(x.grad)
        arrs[f"{tag}.dW1"] = npy(m.proj_weight1.grad)
        arrs[f"{tag}.db1"] = npy(m.proj_bias1.grad)
        assert m.proj_weight2.grad is None and m.qkv.weight.grad is None
        arrs[f"{tag}.meta"] = np.asarray([B, N, D, H, r], dtype=np.int32)
    save("g3_module.npz", **arrs)


# ---------------------------------------------------------------- G4: block forward/backward (GELU-MLP and SwiGLU)
def g4():
    arrs = {}
    for tag, (B, N, D, H, r, swiglu) in {"gelu": (2, 23, 64, 2, 8, False), "swiglu": (2, 23, 64, 2, 8, True)}.items():
        torch.manual_seed(31)
        blk = vit.Block(dim=D, num_heads=H, mlp_ratio=4., qkv_bias=True,
                        norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6),
                        conf=Cfg(has_layerscale=True, layerscale_init_values=1.0), use_swiglu=swiglu)
        with torch.no_grad():
            for n, p_ in blk.named_parameters():
                if "gamma" in n:
                    p_.uniform_(0.5, 1.5)
                elif "norm" in n and n.endswith("weight"):
                    p_.uniform_(0.8, 1.2)
                elif p_.ndim == 2:
                    p_.normal_(std=0.08)
                else:
                    p_.normal_(std=0.05)
        holder = torch.nn.Module()
        holder.blocks = torch.nn.ModuleList([blk])
        torch.manual_seed(32)
        avit.build_apla(Cfg(partial_size=r), holder, "apla_attn")
        blk = holder.blocks[0]
        torch.manual_seed(33)
        x = torch.randn(B, N, D, requires_grad=True)
        out = blk(x)
        out.square().mean().backward()
        for k, v in blk.state_dict().items():
            arrs[f"{tag}.blocks.0.{k}"] = npy(v) if v.dtype != torch.int64 else npy(v).astype(np.int32)
        arrs[f"{tag}.x"] = npy(x)
        arrs[f"{tag}.out"] = npy(out)
        arrs[f"{tag}.dx"] = npy(x.grad)
        arrs[f"{tag}.dW1"] = npy(blk.attn.proj_weight1.grad)
        arrs[f"{tag}.db1"] = npy(blk.attn.proj_bias1.grad)
        n_with_grad = sum(1 for p_ in blk.parameters() if p_.grad is not None)
        assert n_with_grad == 2, n_with_grad
        arrs[f"{tag}.meta"] = np.asarray([B, N, D, H, r, int(swiglu)], dtype=np.int32)
    save("g4_block.npz", **arrs)


# ---------------------------------------------------------------- G5: whole-model training step
def g9_lr_schedule():
    """Per-iteration learning rates produced by the reference's own scheduler classes (utils/_utils.py LinearWarmup and
    MixedLRScheduler, wired as in defaults/wrappers.py:255-303).  utils/_utils.py cannot be imported as a module here
    (torchvision / timm are absent), so the two class definitions are taken from its AST and executed in this process;
    only the resulting number sequences are stored."""
    import ast, json, warnings
    from torch.optim.lr_scheduler import _LRScheduler, CosineAnnealingLR
    src = open(os.path.join(REF_SRC, "utils", "_utils.py")).read()
    tree = ast.parse(src)
    class _LRSchedulerCompat(_LRScheduler):  # torch >= 2.7 dropped the `verbose` positional the reference still passes
        def __init__(self, optimizer, last_epoch=-1, verbose=False):
            super().__init__(optimizer, last_epoch)

    ns = {"_LRScheduler": _LRSchedulerCompat, "warnings": warnings, "print_ddp": lambda *a, **k: None, "torch": torch}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name in ("LinearWarmup", "MixedLRScheduler"):
            exec(compile(ast.Module(body=[node], type_ignores=[]), "ref_utils", "exec"), ns)
    LinearWarmup, Mixed = ns["LinearWarmup"], ns["MixedLRScheduler"]

    def run(max_lr, types, steps_per_epoch, epochs, n, warm=None, cos_eta=1e-6):
        prm = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([prm], lr=max_lr)
        scheds, warmup_iters = [], 0
        for t in types:  # same order and parameter rules as defaults/wrappers.py:255-303
            if t == "LinearWarmup":
                sc = LinearWarmup(opt, max_lr=max_lr, steps_per_epoch=steps_per_epoch, **warm)
                warmup_iters = sc.warmup_iters
            else:
                T_max = steps_per_epoch * epochs - (warmup_iters if "LinearWarmup" in types else 0)
                sc = CosineAnnealingLR(opt, T_max=T_max, eta_min=cos_eta)
            scheds.append(sc)
        mixed = Mixed(scheds, list(types), steps_per_epoch)
        lrs = []
        for _ in range(n):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            mixed.step(None, None)
        return lrs

    cases = {
        "shipped_warmup500": dict(max_lr=5e-4, types=["LinearWarmup"], steps_per_epoch=100, epochs=3, n=700,
                                  warm=dict(warmup_iters=500, warmup_epochs=0)),
        "warmup_epochs2": dict(max_lr=1e-3, types=["LinearWarmup"], steps_per_epoch=7, epochs=5, n=40,
                               warm=dict(warmup_iters=0, warmup_epochs=2)),
        "warmup_then_cosine": dict(max_lr=1e-3, types=["LinearWarmup", "CosineAnnealingLR"], steps_per_epoch=10, epochs=6,
                                   n=60, warm=dict(warmup_iters=12, warmup_epochs=0)),
        "cosine_only": dict(max_lr=2e-3, types=["CosineAnnealingLR"], steps_per_epoch=8, epochs=4, n=32, warm=None),
    }
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, c in cases.items():
            out[name] = dict(config={k: v for k, v in c.items()}, lr=run(**c))
    with open(os.path.join(HERE, "g9_lr_schedule.json"), "w") as f:
        json.dump(out, f)
    print("g9_lr_schedule.json", {k: len(v["lr"]) for k, v in out.items()})


def g10_ssl_losses():
    """DINO CLS-token loss and iBOT patch loss of the reference (self_supervised/dinov2/loss/*.py, pure torch: loaded from
    the files in place), run for two iterations so the centre EMA is exercised: inputs, teacher distributions, losses,
    gradients wrt the student logits, centres."""
    import importlib.util
    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF_SRC, "self_supervised", "dinov2", "loss", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    DINOLoss, iBOT = load("dino_clstoken_loss").DINOLoss, load("ibot_patch_loss").iBOTPatchLoss
    g = torch.Generator().manual_seed(11)
    K, n, n_local = 512, 6, 3
    arrs = {"meta": np.array([K, n, n_local])}
    dino = DINOLoss(K, student_temp=0.1, center_momentum=0.9)
    for it in range(2):
        teacher = torch.randn(2 * n, K, generator=g) * 2
        tprobs = dino.softmax_center_teacher(teacher, teacher_temp=0.05).view(2, n, K)
        dino.update_center(teacher)
        s_glob = (torch.randn(2 * n, K, generator=g)).requires_grad_(True)
        s_loc = (torch.randn(n_local * n, K, generator=g)).requires_grad_(True)
        loss_g = dino(student_output_list=[s_glob], teacher_out_softmaxed_centered_list=[tprobs.flatten(0, 1)])
        loss_l = dino(student_output_list=s_loc.chunk(n_local), teacher_out_softmaxed_centered_list=tprobs)
        (loss_g + loss_l).backward()
        arrs.update({f"dino{it}.teacher": npy(teacher), f"dino{it}.tprobs": npy(tprobs), f"dino{it}.s_glob": npy(s_glob),
                     f"dino{it}.s_loc": npy(s_loc), f"dino{it}.loss_g": npy(loss_g), f"dino{it}.loss_l": npy(loss_l),
                     f"dino{it}.ds_glob": npy(s_glob.grad), f"dino{it}.ds_loc": npy(s_loc.grad)})
    dino.apply_center_update()
    arrs["dino.center"] = npy(dino.center)
    ibot = iBOT(K, student_temp=0.1, center_momentum=0.9)
    B, N = 4, 9
    for it in range(2):
        masks = torch.rand(B, N, generator=g) < 0.4
        masks[0, 0] = True
        n_masked = int(masks.sum())
        t_tok = torch.randn(n_masked, K, generator=g) * 2
        tprobs = ibot.softmax_center_teacher(t_tok.unsqueeze(0), teacher_temp=0.05).squeeze(0)
        ibot.update_center(t_tok.unsqueeze(0))
        s_tok = torch.randn(n_masked + 3, K, generator=g).requires_grad_(True)   # padded buffer as in models.py ("upperbound")
        t_pad = torch.cat([tprobs, torch.zeros(3, K)])
        mw = (1 / masks.sum(-1).clamp(min=1.0)).unsqueeze(-1).expand_as(masks)[masks]
        loss = ibot.forward_masked(s_tok, t_pad, student_masks_flat=masks, n_masked_patches=n_masked, masks_weight=mw)
        loss.backward()
        arrs.update({f"ibot{it}.masks": npy(masks), f"ibot{it}.t_tok": npy(t_tok), f"ibot{it}.tprobs": npy(tprobs),
                     f"ibot{it}.s_tok": npy(s_tok), f"ibot{it}.loss": npy(loss), f"ibot{it}.ds": npy(s_tok.grad)})
    # dense form: (B, N, K) tokens with a mask
    s3, t3 = torch.randn(B, N, K, generator=g).requires_grad_(True), torch.softmax(torch.randn(B, N, K, generator=g), -1)
    m3 = torch.rand(B, N, generator=g) < 0.5
    l3 = ibot(s3, t3, m3)
    l3.backward()
    arrs.update({"ibotd.s": npy(s3), "ibotd.t": npy(t3), "ibotd.m": npy(m3), "ibotd.loss": npy(l3), "ibotd.ds": npy(s3.grad)})
    ibot.apply_center_update()
    arrs["ibot.center"] = npy(ibot.center)
    save("g10_ssl_losses.npz", **arrs)


def g11_ssl_head_koleo():
    """DINOHead (dinov2/layers/dino_head.py) and KoLeoLoss (dinov2/loss/koleo_loss.py) of the reference, loaded from the files
    in place: parameters, inputs, outputs and gradients."""
    import importlib.util
    def load(sub, name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF_SRC, "self_supervised", "dinov2", sub, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    DINOHead = load("layers", "dino_head").DINOHead
    KoLeo = load("loss", "koleo_loss").KoLeoLoss
    torch.manual_seed(21)
    head = DINOHead(in_dim=128, out_dim=512, hidden_dim=256, bottleneck_dim=128)
    with torch.no_grad():
        for p_ in head.mlp.parameters():
            p_.add_(torch.randn_like(p_) * 0.05)     # biases are zero-initialised: make them count
    g = torch.Generator().manual_seed(22)
    x = torch.randn(37, 128, generator=g).requires_grad_(True)
    w = torch.randn(37, 512, generator=g)
    out = head(x)
    (out * w).sum().backward()
    arrs = {"head.x": npy(x), "head.w": npy(w), "head.out": npy(out), "head.dx": npy(x.grad)}
    for k, v in head.state_dict().items():
        arrs["head.p." + k] = npy(v)
    for k, v in head.named_parameters():
        arrs["head.g." + k] = npy(v.grad)
    xk = torch.randn(16, 64, generator=g).requires_grad_(True)
    lk = KoLeo()(xk)
    lk.backward()
    arrs.update({"koleo.x": npy(xk), "koleo.loss": npy(lk), "koleo.dx": npy(xk.grad)})
    save("g11_ssl_head_koleo.npz", **arrs)


# ---------------------------------------------------------------- G12: DINOv2-APLA self-supervised step
def _load_ref_dinov2():
    """self_supervised/dinov2/{layers,loss,dinov2_vits,dinov2_utils} of the reference as a package, files in place."""
    import importlib
    import types
    import warnings
    warnings.filterwarnings("ignore", message=".*xFormers.*")
    for name, sub in (("self_supervised", ("self_supervised",)), ("self_supervised.dinov2", ("self_supervised", "dinov2"))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF_SRC, *sub)]
            sys.modules[name] = m
    imp = lambda n: importlib.import_module("self_supervised.dinov2." + n)  # noqa: E731
    return imp("layers"), imp("loss"), imp("dinov2_vits"), imp("dinov2_utils")


def g12_ssl_step(tag="apla", partial_size=32):
    """Two iterations of the DINOv2-APLA step (self_supervised/dinov2/models.py:207-453 + trainer.py:106-141) on a tiny
    ViT/14, every numerical component being the reference's own class: DinoVisionTransformer (+ build_apla with an inds
    JSON, 'apla_attn_mem_eff'; or partial_size 'full' with is_multi_gpu), DINOHead, DINOLoss, iBOTPatchLoss, KoLeoLoss,
    MaskingGenerator / collate_data_and_cast (under random.seed), CosineScheduler, torch AdamW with the reference's two
    parameter groups, clip_grad_norm_, the EMA teacher update.  The glue between them follows DINOv2.forward line by
    line with ONE difference forced by the missing xformers: the student runs one dense pass per crop resolution instead
    of the packed block-diagonal pass (the same function: attention never crosses crops), and the head inputs are
    concatenated with torch.cat instead of BlockDiagonalMask.from_tensor_list (the head is token-wise).  The dense fallback of
    APLA_MemEffAttention returns APLA_Attention's (x, attn) pair; the block needs x, so the pair is unpacked here."""
    import copy
    import random
    from functools import partial
    layers, loss_mod, vits, du = _load_ref_dinov2()
    D, depth, heads, patch, pre, gsz, lsz = 128, 2, 2, 14, 70, 56, 28
    K, hid, bott, B, n_local = 512, 256, 128, 3, 8   # kernel granularity: GEMM N % 128, K % 64
    torch.manual_seed(31)
    def backbone():
        return vits.DinoVisionTransformer(img_size=pre, patch_size=patch, embed_dim=D, depth=depth, num_heads=heads, init_values=1.0,
                                          ffn_layer="mlp", block_chunks=0, num_register_tokens=0, interpolate_antialias=False,
                                          interpolate_offset=0.1, block_fn=partial(layers.NestedTensorBlock, attn_class=layers.MemEffAttention))
    student_bb, teacher_bb = backbone(), backbone()
    with torch.no_grad():   # "pretrained" weights: make every frozen tensor count (zero biases / unit scales hide mistakes)
        for n_, p_ in student_bb.named_parameters():
            if n_.endswith("gamma"):
                p_.uniform_(0.5, 1.5)
            elif p_.ndim == 1 or "token" in n_:
                p_.normal_(std=0.1) if "norm" not in n_ or n_.endswith("bias") else p_.uniform_(0.8, 1.2)
            elif "pos_embed" in n_:
                p_.normal_(std=0.2)
            else:
                p_.normal_(std=0.06)
    teacher_bb.load_state_dict(copy.deepcopy(student_bb.state_dict()))
    g = torch.Generator().manual_seed(32)
    if partial_size == "full":
        cfg = Cfg(partial_size="full")
        inds = None
    else:
        inds = {f"block_{i}": torch.randperm(D, generator=g)[:partial_size].tolist() for i in range(depth)}
        tmp = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False)
        json.dump(inds, tmp)
        tmp.close()
        cfg = Cfg(partial_size=partial_size, inds_path=tmp.name)
    multi = partial_size == "full"
    student_bb = avit.build_apla(cfg, student_bb, "apla_attn_mem_eff", is_multi_gpu=multi)
    teacher_bb = avit.build_apla(cfg, teacher_bb, "apla_attn_mem_eff", is_multi_gpu=multi)
    if not multi:   # without xformers APLA_MemEffAttention falls back to APLA_Attention.forward, which returns (x, attn): keep x
        for bb_ in (student_bb, teacher_bb):
            for blk in bb_.blocks:
                blk.attn.forward = (lambda x, attn_bias=None, _f=blk.attn.forward: _f(x)[0])
    head = partial(layers.DINOHead, in_dim=D, out_dim=K, hidden_dim=hid, bottleneck_dim=bott, nlayers=3)
    student = torch.nn.ModuleDict({"backbone": student_bb, "dino_head": head()})
    teacher = torch.nn.ModuleDict({"backbone": teacher_bb, "dino_head": head()})
    with torch.no_grad():
        for p_ in student["dino_head"].mlp.parameters():
            p_.add_(torch.randn(p_.shape, generator=g) * 0.05)
    for k in student.keys():
        teacher[k].load_state_dict(student[k].state_dict())
    for p_ in teacher.parameters():
        p_.requires_grad = False
    dino_loss, ibot_loss, koleo = loss_mod.DINOLoss(K), loss_mod.iBOTPatchLoss(K), loss_mod.KoLeoLoss()
    arrs = {"meta": np.array([D, depth, heads, patch, pre, gsz, lsz, K, hid, bott, B, n_local])}
    for k_, v_ in student.state_dict().items():
        arrs["init." + k_] = npy(v_)
    if inds is not None:
        arrs["inds"] = np.array([inds[f"block_{i}"] for i in range(depth)], dtype=np.int32)

    # optimizer: defaults/wrappers.py:205-221 groups, AdamW lr/wd per iteration from the schedules (trainer.py:96-121)
    reg, noreg = [], []
    for n_, p_ in student.named_parameters():
        if p_.requires_grad:
            (noreg if n_.endswith(".bias") or p_.ndim == 1 else reg).append(p_)
    opt = torch.optim.AdamW([{"params": reg}, {"params": noreg, "weight_decay": 0.0}], lr=1e-3, weight_decay=0.04)
    total_iters = 6
    lr_s = du.CosineScheduler(base_value=1e-3, final_value=1e-6, total_iters=total_iters, warmup_iters=2, start_warmup_value=0)
    wd_s = du.CosineScheduler(base_value=0.04, final_value=1e-4, total_iters=total_iters)
    mom_s = du.CosineScheduler(base_value=0.9, final_value=1.0, total_iters=total_iters)
    tt_s = du.CosineScheduler(base_value=0.07, final_value=0.07, total_iters=3, warmup_iters=3, start_warmup_value=0.04)
    arrs["sched.lr"], arrs["sched.wd"], arrs["sched.mom"] = lr_s.schedule, wd_s.schedule, mom_s.schedule
    arrs["sched.tt"] = np.array([tt_s[i] for i in range(total_iters)])

    n_tok = (gsz // patch) ** 2
    mask_gen = du.MaskingGenerator(input_size=(gsz // patch, gsz // patch), max_num_patches=0.5 * gsz // patch * gsz // patch)
    random.seed(5)
    n_global, dino_w, ibot_w, koleo_w, clip = 2, 1.0, 1.0, 0.1, 3.0
    for it in range(1, 3):   # trainer iterations start at 1
        samples = [([torch.randn(3, gsz, gsz, generator=g) for _ in range(2)] + [torch.randn(3, lsz, lsz, generator=g) for _ in range(n_local)],
                    torch.tensor(0)) for _ in range(B)]
        data = du.collate_data_and_cast(samples, n_global_crops=2, n_local_crops=n_local, mask_ratio_tuple=(0.1, 0.5),
                                        mask_probability=0.5, dtype=torch.float32, n_tokens=n_tok, mask_generator=mask_gen)["images"]
        glob, loc, masks = data["collated_global_crops"], data["collated_local_crops"], data["collated_masks"]
        idx, mw, upper = data["mask_indices_list"], data["masks_weight"], data["upperbound"]
        n_masked = idx.shape[0]
        lr, wd, ttemp, mom = lr_s[it], wd_s[it], tt_s[it], mom_s[it]
        for gp in opt.param_groups:
            gp["lr"] = lr
        opt.param_groups[0]["weight_decay"] = wd
        opt.zero_grad()
        with torch.no_grad():   # models.py:231-300
            tout = teacher["backbone"](glob, is_training=True)
            tcls = tout["x_norm_clstoken"].chunk(2)
            tcls = torch.cat((tcls[1], tcls[0]))
            tpatch = tout["x_norm_patchtokens"]
            ncls = tcls.shape[0]
            buf = tpatch.new_zeros(upper + ncls, D)
            buf[:ncls].copy_(tcls)
            torch.index_select(tpatch.flatten(0, 1), dim=0, index=idx, out=buf[ncls:ncls + n_masked])
            after = teacher["dino_head"](buf)
            tcls_h, tpatch_h = after[:ncls], after[ncls:ncls + n_masked]
            t_dino = dino_loss.softmax_center_teacher(tcls_h, teacher_temp=ttemp).view(2, -1, K)
            dino_loss.update_center(tcls_h)
            tpatch_h = tpatch_h.unsqueeze(0)
            t_ibot = ibot_loss.softmax_center_teacher(tpatch_h[:, :n_masked], teacher_temp=ttemp).squeeze(0)
            ibot_loss.update_center(tpatch_h[:n_masked])
        sg = student["backbone"](glob, masks=masks, is_training=True)      # dense pass per resolution (see docstring)
        sl = student["backbone"](loc, is_training=True)
        s_loc_cls, s_glob_cls = sl["x_norm_clstoken"], sg["x_norm_clstoken"]
        pbuf = sg["x_norm_patchtokens"].new_zeros(upper, D)
        pbuf[:n_masked].copy_(torch.index_select(sg["x_norm_patchtokens"].flatten(0, 1), dim=0, index=idx))
        outs = student["dino_head"](torch.cat([s_loc_cls, s_glob_cls, pbuf]))
        o_loc, o_glob, o_patch = outs[:s_loc_cls.shape[0]], outs[s_loc_cls.shape[0]:s_loc_cls.shape[0] + s_glob_cls.shape[0]], \
            outs[s_loc_cls.shape[0] + s_glob_cls.shape[0]:][:n_masked]
        n_loc_terms, n_glob_terms = max(n_local * n_global, 1), (n_global - 1) * n_global
        ld = {}
        ld["dino_local_crops_loss"] = dino_loss(student_output_list=o_loc.chunk(n_local),
                                                teacher_out_softmaxed_centered_list=t_dino) / (n_glob_terms + n_loc_terms)
        total = dino_w * ld["dino_local_crops_loss"]
        ld["dino_global_crops_loss"] = dino_loss(student_output_list=[o_glob], teacher_out_softmaxed_centered_list=[t_dino.flatten(0, 1)]) \
            * 2 / (n_glob_terms + n_loc_terms)
        total = total + dino_w * ld["dino_global_crops_loss"]
        kl = koleo_w * sum(koleo(p_) for p_ in s_glob_cls.chunk(2))
        total = total + kl
        ld["koleo_loss"] = kl / 2
        ib = ibot_loss.forward_masked(o_patch, t_ibot, student_masks_flat=masks, n_masked_patches=n_masked, masks_weight=mw) * 2 * (1.0 / n_global)
        ld["ibot_loss"] = ib / 2
        total = total + ibot_w * ib
        total.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(student.parameters(), clip)
        grads = {n_: npy(p_.grad) for n_, p_ in student.named_parameters() if p_.grad is not None}
        if it == 1:   # freeze_last_layer for the first "epoch": trainer.py:84-90 (cancel_gradients sets .grad = None)
            for n_, p_ in student.named_parameters():
                if "dino_head.last_layer" in n_:
                    p_.grad = None
        opt.step()
        with torch.no_grad():   # models.py:443-453
            sp = [p_ for k in student.keys() for p_ in student[k].parameters()]
            tp = [p_ for k in student.keys() for p_ in teacher[k].parameters()]
            torch._foreach_mul_(tp, mom)
            torch._foreach_add_(tp, sp, alpha=1 - mom)
        pre_ = f"it{it}."
        arrs.update({pre_ + "glob": npy(glob), pre_ + "loc": npy(loc), pre_ + "masks": npy(masks), pre_ + "mask_indices": npy(idx),
                     pre_ + "masks_weight": npy(mw), pre_ + "upperbound": np.array(upper), pre_ + "hyper": np.array([lr, wd, ttemp, mom]),
                     pre_ + "loss": npy(total), pre_ + "gnorm": npy(gnorm), pre_ + "t_dino": npy(t_dino[:, :, :64]),
                     pre_ + "s_glob_cls": npy(s_glob_cls), pre_ + "s_loc_cls": npy(s_loc_cls)})
        for k_, v_ in ld.items():
            arrs[pre_ + "ld." + k_] = npy(v_)
        for k_, v_ in grads.items():
            arrs[pre_ + "g." + k_] = v_          # AFTER clipping (what the optimizer consumed)
        for k_, p_ in student.named_parameters():
            if p_.requires_grad:
                arrs[pre_ + "student." + k_] = npy(p_)
        for (k_, p_), (_, ps) in zip(teacher.named_parameters(), student.named_parameters()):
            if ps.requires_grad:
                arrs[pre_ + "teacher." + k_] = npy(p_)
    dino_loss.apply_center_update()
    ibot_loss.apply_center_update()
    arrs["dino.center"], arrs["ibot.center"] = npy(dino_loss.center), npy(ibot_loss.center)
    arrs["trainable"] = np.array([n_ for n_, p_ in student.named_parameters() if p_.requires_grad])
    save(f"g12_ssl_step_{tag}.npz", **arrs)


def classifier_step(model, fc, images, labels, lr=1e-4, wd=1e-5, clip=1.0):
    """defaults/trainer.py:106-151 (no AMP) with the param groups of defaults/wrappers.py:205-221."""
    named = [(n, p_) for n, p_ in list(model.named_parameters()) + [("fc." + n, p_) for n, p_ in fc.named_parameters()]
             if p_.requires_grad]
    reg = [p_ for n, p_ in named if not (n.endswith(".bias") or p_.ndim == 1)]
    noreg = [p_ for n, p_ in named if (n.endswith(".bias") or p_.ndim == 1)]
    opt = torch.optim.AdamW([{"params": reg}, {"params": noreg, "weight_decay": 0.}], lr=lr, weight_decay=wd)
    opt.zero_grad()
    logits = fc(model(images))
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    grads = {n: p_.grad.clone() for n, p_ in named}
    all_params = list(model.parameters()) + list(fc.parameters())
    gnorm = torch.nn.utils.clip_grad_norm_(all_params, clip)
    opt.step()
    return logits, loss, grads, gnorm, dict(named)


def g5_tiny():
    """Small ViT with every tensor dumped; pretrain grid 4x4 (img 64) fed 48x48 images so that the
    bicubic pos-embed interpolation (vit.py:421-437) is exercised."""
    torch.manual_seed(41)
    D, L, H, r, C = 64, 2, 2, 8, 5
    model = vit.VisionTransformer(img_size=[64], patch_size=16, embed_dim=D, depth=L, num_heads=H, qkv_bias=True,
                                  norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6),
                                  block_conf=Cfg(has_layerscale=True, layerscale_init_values=1.0),
                                  is_memory_efficient=True)
    with torch.no_grad():
        for n, p_ in model.named_parameters():
            if "gamma" in n:
                p_.uniform_(0.5, 1.5)
            elif p_.ndim >= 2:
                p_.normal_(std=0.08)
            elif "norm" in n and n.endswith("weight"):
                p_.uniform_(0.8, 1.2)
            else:
                p_.normal_(std=0.05)
    model = avit.build_apla(Cfg(partial_size=r), model, "apla_attn")
    model.fc = torch.nn.Identity()
    fc = torch.nn.Linear(D, C)
    g = torch.Generator().manual_seed(42)
    images = torch.randn(3, 3, 48, 48, generator=g)
    labels = torch.randint(0, C, (3,), generator=g)
    arrs = {}
    for k, v in model.state_dict().items():
        arrs["p." + k] = npy(v) if v.dtype != torch.int64 else npy(v).astype(np.int32)
    arrs["p.fc.weight"] = npy(fc.weight).copy()
    arrs["p.fc.bias"] = npy(fc.bias).copy()
    logits, loss, grads, gnorm, named = classifier_step(model, fc, images, labels)
    arrs["images"], arrs["labels"] = npy(images), npy(labels).astype(np.int32)
    arrs["logits"], arrs["loss"], arrs["gnorm"] = npy(logits), npy(loss), npy(gnorm)
    for n, gr in grads.items():
        arrs["g." + n] = npy(gr)
    for n, p_ in named.items():
        arrs["after." + n] = npy(p_)
    arrs["meta"] = np.asarray([D, L, H, r, C, 16], dtype=np.int32)
    assert len(grads) == 2 * L + 2
    save("g5_tiny_model.npz", **arrs)


def g5_cfg1():
    """BASELINE config 1: ViT-S/16, r=64, C=10, bs=8, weights by construction recipe under
    torch.manual_seed(0).  Weights are too large to commit: we store per-tensor
    digests so the test can prove it rebuilt identical weights, plus the reference outputs."""
    torch.manual_seed(0)
    model = vit.vit_small(pretrained=False, img_size=[224], patch_size=16, pretrained_type="dinov2",
                          is_memory_efficient=True, block_conf=Cfg(has_layerscale=True, layerscale_init_values=1.0))
    model = avit.build_apla(Cfg(partial_size=64), model, "apla_attn")
    model.fc = torch.nn.Identity()
    fc = torch.nn.Linear(384, 10)  # defaults/models.py:65, created after the backbone
    digests = {k: tensor_digest(v) for k, v in model.state_dict().items()}
    digests["fc.weight"], digests["fc.bias"] = tensor_digest(fc.weight), tensor_digest(fc.bias)
    g = torch.Generator().manual_seed(0)
    images = torch.randn(8, 3, 224, 224, generator=g)
    labels = torch.randint(0, 10, (8,), generator=g)
    logits, loss, grads, gnorm, named = classifier_step(model, fc, images, labels)
    arrs = dict(logits=npy(logits), loss=npy(loss), gnorm=npy(gnorm))
    for i in (0, 5, 11):
        arrs[f"inds{i}"] = npy(model.blocks[i].attn.inds).astype(np.int32)
        for nm in ("proj_weight1", "proj_bias1"):
            arrs[f"g.blocks.{i}.attn.{nm}"] = npy(grads[f"blocks.{i}.attn.{nm}"])
            arrs[f"after.blocks.{i}.attn.{nm}"] = npy(named[f"blocks.{i}.attn.{nm}"])
    arrs["g.fc.weight"], arrs["g.fc.bias"] = npy(grads["fc.weight"]), npy(grads["fc.bias"])
    arrs["after.fc.weight"], arrs["after.fc.bias"] = npy(named["fc.weight"]), npy(named["fc.bias"])
    n_train = sum(p_.numel() for p_ in named.values())
    assert n_train == 299530, n_train
    save("g5_cfg1_vits.npz", **arrs)
    with open(os.path.join(HERE, "g5_cfg1_digests.json"), "w") as f:
        json.dump(dict(digests=digests, n_trainable=n_train, torch=torch.__version__), f, indent=1)


def g5_cfg1_seeds(seeds=(1, 2, 3, 4, 5, 6, 7)):
    """The same BASELINE config-1 model (weights under torch.manual_seed(0), as g5_cfg1) on further input batches: reference logits
    and loss per input seed, so that the 16-bit parity numbers of the product are a maximum over batches, not one sample."""
    torch.manual_seed(0)
    model = vit.vit_small(pretrained=False, img_size=[224], patch_size=16, pretrained_type="dinov2",
                          is_memory_efficient=True, block_conf=Cfg(has_layerscale=True, layerscale_init_values=1.0))
    model = avit.build_apla(Cfg(partial_size=64), model, "apla_attn")
    model.fc = torch.nn.Identity()
    fc = torch.nn.Linear(384, 10)
    logits, losses = [], []
    with torch.no_grad():
        for sd in seeds:
            g = torch.Generator().manual_seed(sd)
            images = torch.randn(8, 3, 224, 224, generator=g)
            labels = torch.randint(0, 10, (8,), generator=g)
            out = fc(model(images))
            logits.append(npy(out))
            losses.append(npy(torch.nn.functional.cross_entropy(out, labels)))
    save("g5_cfg1_seeds.npz", seeds=np.array(seeds), logits=np.stack(logits), loss=np.stack(losses))


# ---------------------------------------------------------------- G13: kNN vote and the metric objects
def _ref_knn_predict():
    """``Trainer.knn_predict`` (defaults/trainer.py:392-455) as the reference file defines it: the method's own source is taken from
    the file IN PLACE (ast) and compiled with the names it uses — importing ``defaults.trainer`` itself needs torchvision and wandb,
    which are not installed (the method touches neither; it does not use ``self``)."""
    import ast
    import torch.nn.functional as F
    src = open(os.path.join(REF_SRC, "defaults/trainer.py")).read()
    tree = ast.parse(src)
    fn = next(n for c in tree.body if isinstance(c, ast.ClassDef) and c.name == "Trainer" for n in c.body
              if isinstance(n, ast.FunctionDef) and n.name == "knn_predict")
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"torch": torch, "F": F}
    exec(compile(mod, os.path.join(REF_SRC, "defaults/trainer.py"), "exec"), ns)
    return ns["knn_predict"]


def _ref_metrics():
    """utils/metrics.py loaded in place; its ``from ._utils import *`` (torchvision, timm) is replaced by the handful of names the
    file uses, ``easydict`` (not installed) by a dict with attribute access."""
    import types
    from copy import deepcopy
    import _ref_shim
    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed
    u = types.ModuleType("utils._utils")
    u.torch, u.np, u.nn, u.deepcopy = torch, np, torch.nn, deepcopy
    u.synchronize = lambda: None
    u.dist_gather = lambda x, **_k: [x]
    u.__all__ = ["torch", "np", "nn", "deepcopy", "synchronize", "dist_gather"]
    sys.modules["utils._utils"] = u
    return _ref_shim._load("utils.metrics", os.path.join(REF_SRC, "utils/metrics.py"), package="utils")


def g13_knn_metrics():
    import torch.nn.functional as F
    knn = _ref_knn_predict()
    M = _ref_metrics()
    g = torch.Generator().manual_seed(13)
    arrs = {}
    # kNN: single-label and multi-label votes
    B, D, N, C, Cm, k, t = 16, 32, 200, 7, 5, 20, 0.1
    feat = F.normalize(torch.randn(B, D, generator=g), dim=1)
    bank = F.normalize(torch.randn(N, D, generator=g), dim=1).t().contiguous()
    lab = torch.randint(0, C, (N,), generator=g)
    labm = (torch.rand(Cm, N, generator=g) < 0.3).float()
    arrs.update(knn_feature=npy(feat), knn_bank=npy(bank), knn_labels=npy(lab), knn_labels_multi=npy(labm),
                knn_k=np.array(k), knn_t=np.array(t), knn_classes=np.array(C),
                knn_scores=npy(knn(None, feat, bank, lab, k, t, classes=C)),
                knn_scores_multi=npy(knn(None, feat.clone(), bank, labm, k, t, classes=Cm, multi_label=True)))
    # single-label metric object: three batches of logits
    for tag, n_cls in (("mc", 7), ("bin", 2)):
        logits = torch.randn(300, n_cls, generator=g) * 2
        truths = torch.randint(0, n_cls, (300,), generator=g)
        logits[torch.arange(300), truths] += 1.5          # better than chance
        m = M.ClassificationMetrics(n_classes=n_cls, mode="val")
        for lo in (0, 100, 200):
            m.add_preds(logits[lo:lo + 100], truths[lo:lo + 100])
        res = m.get_values(use_dist=False, do_reset=False, return_conf_matrix=True)
        arrs[f"{tag}_logits"], arrs[f"{tag}_truths"] = npy(logits), npy(truths)
        arrs[f"{tag}_confusion"] = np.asarray(res["confusion_matrix"])
        for key in ("accuracy", "mean_per_class_accuracy", "quadratic_kappa", "roc_auc", "recall"):
            arrs[f"{tag}_{key}"] = np.array(float(res["val_" + key]))
    # multi-label metric object
    logits = torch.randn(300, Cm, generator=g) * 2
    truths = (torch.rand(300, Cm, generator=g) < 0.35).float()
    logits += (truths * 2 - 1) * 1.0
    mm = M.MultiLabelClassificationMetrics(n_classes=Cm, mode="val")
    for lo in (0, 150):
        mm.add_preds(logits[lo:lo + 150], truths[lo:lo + 150])
    res = mm.get_value(use_dist=False)
    arrs["ml_logits"], arrs["ml_truths"] = npy(logits), npy(truths)
    for key in ("accuracy", "mAP", "precision", "recall", "f1", "roc_auc"):
        arrs[f"ml_{key}"] = np.array(float(res["val_" + key]))
    save("g13_knn_metrics.npz", **arrs)


if __name__ == "__main__":
    g1()
    g2_g8()
    g3()
    g4()
    g5_tiny()
    g5_cfg1()
    g5_cfg1_seeds()
    g9_lr_schedule()
    g10_ssl_losses()
    g11_ssl_head_koleo()
    g12_ssl_step("apla", 32)
    g12_ssl_step("full", "full")
    g13_knn_metrics()
