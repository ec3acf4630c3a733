"""Drop-in module path on the GPU: the reference's plugin surface — APLA_Attention(...).forward(x) ->
(x, attn) and APLA_MemEffAttention.forward(x, attn_bias) -> x — running the HIP kernels through torch.autograd, checked
against the fp64 CPU oracle (which the reference goldens pin, tests/test_oracle_golden.py).

Tolerances: activations / input gradients are products of bf16 MFMA GEMMs with fp32 accumulation -> relative L2 error
<= 1e-2 against fp64; dW1/db1 contract bf16-rounded operands over ~400 tokens -> <= 2e-2.
"""
import pytest
import torch

from conftest import rel_err
from oracle import apla_oracle as O

pytestmark = pytest.mark.gpu

ACT_TOL, GRAD_TOL = 1e-2, 2e-2


def make_module(cls, dim, heads, r, seed):
    from apla_amd.models import AttrDict
    torch.manual_seed(seed)
    m = cls(AttrDict(partial_size=r), dim, num_heads=heads, qkv_bias=True)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape) * (0.05 if p.ndim > 1 else 0.1))
    return m


def oracle_params(m):
    return {"a." + k: (v.detach().cpu() if k == "inds" else v.detach().double().cpu()) for k, v in m.state_dict().items()}


@pytest.mark.parametrize("B,N,dim,heads,r", [(2, 197, 256, 4, 64), (3, 50, 128, 2, 128), (1, 257, 384, 6, 192), (2, 197, 128, 2, 8),
                                             (2, 64, 256, 4, 100)])
def test_apla_attention_module_fwd_bwd(B, N, dim, heads, r):
    from apla_amd.apla import APLA_Attention
    m = make_module(APLA_Attention, dim, heads, r, seed=5)
    p = oracle_params(m)
    x = torch.randn(B, N, dim, generator=torch.Generator().manual_seed(6))
    m = m.cuda()
    m.return_attn_matrix = True
    xg = x.cuda().requires_grad_(True)
    y, attn = m(xg)
    loss = y.float().square().mean()
    loss.backward()

    yref, aref, ctx = O.apla_attention_fwd(x.double(), p, "a.", heads, r, return_attn=True)
    dyref = 2.0 * yref / yref.numel()
    dxref, dW1ref, db1ref = O.apla_attention_bwd(dyref, ctx, p, "a.", heads)
    assert rel_err(y.detach().cpu(), yref) < ACT_TOL
    assert rel_err(attn.cpu(), aref) < ACT_TOL
    assert rel_err(xg.grad.cpu(), dxref) < GRAD_TOL
    assert rel_err(m.proj_weight1.grad.cpu(), dW1ref) < GRAD_TOL
    assert rel_err(m.proj_bias1.grad.cpu(), db1ref) < GRAD_TOL
    # frozen parameters get no gradient (apla_vit.py freeze policy; appla_attn.py:37-45)
    assert m.proj_weight2.grad is None and m.qkv.weight.grad is None


@pytest.mark.parametrize("crops", [[(2, 257), (8, 50)], [(1, 197)], [(3, 33), (2, 129), (1, 64)]])
def test_mem_eff_attention_block_diagonal(crops):
    """dinov2 nested-tensor path: crops of different sizes packed into [1, total, C] with a BlockDiagonalMask must give,
    crop by crop, what the dense module gives on each crop batch — forward and backward (appla_attn_mem_eff.py:27-67)."""
    from apla_amd.apla import APLA_MemEffAttention
    from apla_amd.nested import BlockDiagonalMask
    dim, heads, r = 256, 4, 64
    m = make_module(APLA_MemEffAttention, dim, heads, r, seed=7)
    p = oracle_params(m)
    m = m.cuda()
    g = torch.Generator().manual_seed(8)
    xs = [torch.randn(b, n, dim, generator=g) for b, n in crops]
    xs_dev = [x.cuda().requires_grad_(True) for x in xs]
    mask, packed = BlockDiagonalMask.from_tensor_list(xs_dev)
    out = m(packed, attn_bias=mask)
    assert out.shape == packed.shape
    outs = mask.split(out)
    loss = sum(o.float().square().mean() for o in outs)
    loss.backward()
    gW = m.proj_weight1.grad.clone()

    dW_ref = torch.zeros_like(p["a.proj_weight1"])
    for x, xd, o in zip(xs, xs_dev, outs):
        yref, _, ctx = O.apla_attention_fwd(x.double(), p, "a.", heads, r)
        dxref, dW1ref, _ = O.apla_attention_bwd(2.0 * yref / yref.numel(), ctx, p, "a.", heads)
        dW_ref += dW1ref
        assert rel_err(o.detach().cpu(), yref) < ACT_TOL
        assert rel_err(xd.grad.cpu(), dxref) < GRAD_TOL
    assert rel_err(gW.cpu(), dW_ref) < GRAD_TOL

    # and the packed path equals the module's own dense path on each crop batch to bf16 rounding of the same kernels
    m.zero_grad()
    dense = [m(x.detach().cuda()) for x in xs]
    for a, b in zip(dense, outs):
        assert rel_err(a.detach().cpu(), b.detach().cpu().double()) < 2e-3


def test_mem_eff_attention_rejects_foreign_bias():
    from apla_amd.apla import APLA_MemEffAttention
    m = make_module(APLA_MemEffAttention, 128, 2, 64, seed=9).cuda()
    with pytest.raises(TypeError):
        m(torch.zeros(1, 10, 128, device="cuda"), attn_bias=torch.zeros(10, 10))


def test_fused_block_loop_equals_block_by_block():
    """VisionTransformer.run_blocks (fp32 residual stream, residual adds fused into the following LayerNorm, frozen
    LayerScale folded into the APLA projection / fc2) against the plain composition of the same modules block by block
    (x + ls1(attn(norm1 x)); x + ls2(mlp(norm2 x)), vit.py:279-288 — the path the goldens G4/G5 pin through the oracle),
    with LayerScale != 1: outputs, input gradient and the APLA gradients (incl. the gamma row scales of dW1/db1)."""
    from functools import partial
    from apla_amd import functional as AF
    from apla_amd.apla import build_apla
    from apla_amd.models import AttrDict
    from apla_amd.vit import VisionTransformer
    torch.manual_seed(11)
    bb = VisionTransformer(img_size=[32], patch_size=16, embed_dim=128, depth=3, num_heads=2, qkv_bias=True,
                           norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), block_conf=dict(has_layerscale=True, layerscale_init_values=1.0))
    with torch.no_grad():
        for n, p in bb.named_parameters():
            if n.endswith("gamma"):
                p.uniform_(0.5, 1.5)
            elif p.ndim >= 2 and "pos_embed" not in n and "token" not in n:
                p.normal_(std=0.06)
            elif p.ndim == 1:
                p.normal_(std=0.1) if n.endswith("bias") else p.uniform_(0.8, 1.2)
    build_apla(AttrDict(partial_size=40), bb, "apla_attn")     # a rank that is not a multiple of 64: padded dW rows + row scales
    bb = bb.cuda()
    x = torch.randn(3, 21, 128, generator=torch.Generator().manual_seed(12)).cuda()

    def plain(xin):
        h = xin.float()
        for blk in bb.blocks:
            h = h + blk.ls1(blk.attn(AF.layer_norm(h, blk.norm1))[0]).float()
            h = h + blk.ls2(blk.mlp(AF.layer_norm(h, blk.norm2))).float()
        return h, AF.layer_norm(h, bb.norm)

    outs = {}
    for name, fn in (("fused", bb.run_blocks), ("plain", plain)):
        xg = x.clone().requires_grad_(True)
        pre, nrm = fn(xg)
        (nrm.float().square().mean() + 0.1 * pre.float().square().mean()).backward()
        outs[name] = (pre.detach(), nrm.detach(), xg.grad.clone(),
                      {n: p.grad.clone() for n, p in bb.named_parameters() if p.grad is not None})
        bb.zero_grad()
    f, p = outs["fused"], outs["plain"]
    assert rel_err(f[0].cpu(), p[0].double().cpu()) < 5e-3 and rel_err(f[1].float().cpu(), p[1].double().cpu()) < 1e-2
    assert rel_err(f[2].cpu(), p[2].double().cpu()) < GRAD_TOL
    assert set(f[3]) == set(p[3]) and len(f[3]) == 6
    for n in f[3]:
        assert rel_err(f[3][n].cpu(), p[3][n].double().cpu()) < GRAD_TOL, n


def test_dropout_and_stochastic_depth_on_the_module_path():
    """drop_rate / drop_path_rate > 0 (main.py:101-111 can set them; utils/transformers/vit.py:74-93, 152-168, 257-288): the ViT's
    module path applies pos_drop, proj_drop, both Mlp dropouts and per-sample DropPath through the HIP kernels; eval mode is the
    deterministic network; a seed repeats a training pass; gradients flow through the masks; attention-probability dropout and the fused
    step refuse."""
    from apla_amd import functional as AF
    from apla_amd.vit import Block, DropPath, VisionTransformer
    torch.manual_seed(0)
    kw = dict(img_size=[32], patch_size=16, embed_dim=128, depth=3, num_heads=2, qkv_bias=True)
    vit = VisionTransformer(drop_rate=0.2, drop_path_rate=0.3, **kw).cuda()
    ref = VisionTransformer(**kw).cuda()
    ref.load_state_dict(vit.state_dict())
    for net in (vit, ref):          # the APLA situation: everything frozen but the attention projections
        for n, p_ in net.named_parameters():
            p_.requires_grad_(".attn.proj." in n)
    assert isinstance(vit.blocks[2].drop_path, DropPath) and abs(vit.blocks[2].drop_path.drop_prob - 0.3) < 1e-6
    assert isinstance(vit.blocks[0].drop_path, torch.nn.Identity)          # linspace(0, rate, depth)[0] = 0
    x = torch.randn(6, 3, 32, 32, device="cuda")
    vit.eval(), ref.eval()
    with torch.no_grad():
        assert torch.equal(vit(x), ref(x))                                  # eval: every dropout is the identity
    vit.train()
    torch.manual_seed(5)
    a = vit(x)
    torch.manual_seed(5)
    b = vit(x)
    c = vit(x)
    assert torch.equal(a, b) and not torch.equal(a, c)                      # the masks follow torch's seed
    with torch.no_grad():
        e = ref(x)
    assert not torch.equal(a.detach(), e) and float((a.detach().float() - e.float()).abs().mean()) < float(e.float().abs().mean())
    vit(x).float().square().mean().backward()
    got = [p_.grad for n, p_ in vit.named_parameters() if ".attn.proj.weight" in n]
    assert len(got) == 3 and all(g_ is not None and bool(torch.isfinite(g_).all()) and float(g_.abs().max()) > 0 for g_ in got)

    # DropPath per sample: the branch of a sample is either dropped or scaled by 1 / keep_prob (vit.py:74-82)
    t = torch.randn(64, 5, 128, device="cuda")
    y = AF.drop_path(t, 0.25, True)
    ratio = (y / t).reshape(64, -1)
    per = ratio[:, 0]
    assert bool(((per == 0) | ((per - 1 / 0.75).abs() < 1e-6)).all()) and 0 < int((per == 0).sum()) < 64
    assert float((ratio - per[:, None]).abs().max()) < 1e-6
    assert AF.drop_path(t, 0.25, False) is t and AF.dropout(t, 0.5, False) is t

    # the expectation is preserved: mean over many draws of dropout(x) ~ x
    acc = torch.zeros_like(t)
    for _ in range(200):
        acc += AF.dropout(t, 0.5, True)
    assert float((acc / 200 - t).abs().mean()) < 0.12 * float(t.abs().mean()) + 0.05

    blk = Block(128, 2, attn_drop=0.1).cuda().train()
    with pytest.raises(NotImplementedError):
        blk(torch.randn(2, 5, 128, device="cuda"))


@pytest.mark.parametrize("half", [torch.bfloat16, torch.float16])
def test_sixteen_bit_gradient_stream_against_the_fp32_one(half):
    """run_blocks carries the gradient of the fp32 residual stream in 16 bits through an autograd proxy (AF.ResidualStream; the fused
    engine's gradient stream); tokens that require grad take the fp32-gradient route.  Same forward bits; gradients of the trainable
    projection rows equal up to the 16-bit rounding of the stream between blocks; the stream's exit (x_prenorm) still carries gradient."""
    from apla_amd import ops as OPS
    from apla_amd.apla import build_apla
    from apla_amd.models import AttrDict
    from apla_amd.vit import VisionTransformer
    torch.manual_seed(0)
    vit = VisionTransformer(img_size=[32], patch_size=16, embed_dim=128, depth=4, num_heads=2, qkv_bias=True)
    build_apla(AttrDict(partial_size=64), vit, "apla_attn")
    vit = vit.cuda().train()
    tokens = torch.randn(6, 5, 128, device="cuda")
    res = {}
    with OPS.use_half(half):
        for mode in ("stream", "fp32"):
            vit.zero_grad(set_to_none=True)
            x = tokens.clone().requires_grad_(mode == "fp32")
            x_pre, x_norm = vit.run_blocks(x)
            assert x_pre.dtype == torch.float32 and x_norm.dtype == half
            (x_norm.float().square().mean() + 0.1 * x_pre.square().mean()).backward()
            res[mode] = (x_pre.detach().clone(), x_norm.detach().clone(),
                         {n: p.grad.detach().float().clone() for n, p in vit.named_parameters() if p.grad is not None})
    assert torch.equal(res["stream"][0], res["fp32"][0]) and torch.equal(res["stream"][1], res["fp32"][1])
    assert set(res["stream"][2]) == set(res["fp32"][2]) and len(res["stream"][2]) == 8        # proj_weight1 / proj_bias1 of four blocks
    for n, g_ in res["fp32"][2].items():
        assert rel_err(res["stream"][2][n].cpu(), g_.double().cpu()) < (2e-2 if half == torch.bfloat16 else 3e-3), n


def test_apla_attention_with_attention_dropout_trains_and_is_the_identity_in_eval_mode():
    """`main.py --adr` (src/main.py:109-111 -> attn_drop_rate -> APLA_Attention(attn_drop=)): in training mode the module's forward goes
    through the dropout attention kernels (different outputs for different draws, gradients for every trainable tensor, the expected
    value of the output close to the undropped one); in evaluation mode the dropout is the identity (bitwise the p = 0 module)."""
    from apla_amd.apla.appla_attn import APLA_Attention
    D, H, B, N = 128, 2, 4, 197
    att, ref = make_module(APLA_Attention, D, H, 16, seed=3).cuda(), make_module(APLA_Attention, D, H, 16, seed=3).cuda()
    att.attn_drop.p = 0.2                      # (what APLA_Attention(attn_drop=0.2) sets: nn.Dropout(attn_drop), appla_attn.py:46)
    assert torch.equal(att.inds, ref.inds) and torch.equal(att.proj_weight1, ref.proj_weight1)
    x = torch.randn(B, N, D, device="cuda")
    att.eval(), ref.eval()
    assert torch.equal(att(x)[0], ref(x)[0])
    att.train(), ref.train()
    torch.manual_seed(1)
    y1 = att(x)[0]
    y2 = att(x)[0]
    torch.manual_seed(1)
    y1b = att(x)[0]
    assert torch.equal(y1, y1b) and not torch.equal(y1, y2)          # the seed comes from torch's CPU generator
    y0 = ref(x)[0]
    mean = torch.stack([att(x)[0].float() for _ in range(48)]).mean(0)
    assert float((mean - y0.float()).abs().max()) < 0.35 * float((y1.float() - y0.float()).abs().max())   # E[dropout(attn)] = attn
    xg = x.clone().requires_grad_(True)
    att(xg)[0].float().square().mean().backward()
    assert att.proj_weight1.grad is not None and att.proj_bias1.grad is not None and xg.grad is not None
    assert float(att.proj_weight1.grad.abs().max()) > 0 and torch.isfinite(xg.grad).all()
    # the second return value is the attention matrix AFTER attn_drop (appla_attn.py:58, :83): the same draw as the output — a fraction
    # p of exact zeros, rows that sum to ~1 in expectation, and attn @ v reproduces what the projection was fed
    att.return_attn_matrix = True
    torch.manual_seed(1)
    y, attn = att(x)
    assert torch.equal(y, y1) and attn.shape == (B, H, N, N)
    assert abs(float((attn == 0).float().mean()) - 0.2) < 0.01 and abs(float(attn.sum(-1).mean()) - 1.0) < 0.01
    att.eval()
    _, attn_eval = att(x)
    assert float((attn_eval.sum(-1) - 1).abs().max()) < 1e-3 and float((attn_eval == 0).float().mean()) < 1e-3
    att.return_attn_matrix = False
