"""Building blocks of the DINOv2-APLA step on the GPU: DINOHead and KoLeoLoss against goldens produced by the
REFERENCE modules (G11); the packed multi-crop backbone forward against one dense pass per resolution; the EMA teacher
update.  bf16 GEMM tolerances as in tests/test_modules_gpu.py."""
import copy

import pytest
import torch

from conftest import load_golden, rel_err, t

pytestmark = pytest.mark.gpu


def test_dino_head_matches_reference():
    from apla_amd.ssl import DINOHead
    g = load_golden("g11_ssl_head_koleo.npz")
    head = DINOHead(in_dim=128, out_dim=512, hidden_dim=256, bottleneck_dim=128)
    sd = {k[len("head.p."):]: t(g[k]) for k in g.files if k.startswith("head.p.")}
    assert set(sd) == set(head.state_dict())                    # same parameter names as the reference module
    head.load_state_dict(sd)
    head = head.cuda()
    x = t(g["head.x"]).cuda().requires_grad_(True)
    out = head(x)
    assert rel_err(out.float().cpu(), g["head.out"]) < 1e-2
    (out.float() * t(g["head.w"]).cuda()).sum().backward()
    assert rel_err(x.grad.cpu(), g["head.dx"]) < 2e-2
    for name, p in head.named_parameters():
        assert rel_err(p.grad.cpu(), g["head.g." + name]) < 2e-2, name


def test_koleo_matches_reference():
    from apla_amd.ssl import KoLeoLoss
    g = load_golden("g11_ssl_head_koleo.npz")
    x = t(g["koleo.x"]).cuda().requires_grad_(True)
    loss = KoLeoLoss()(x)
    loss.backward()
    assert abs(float(loss) - float(g["koleo.loss"])) < 1e-5 and rel_err(x.grad.cpu(), g["koleo.dx"]) < 1e-4


@pytest.mark.parametrize("G,B,D,dtype", [(2, 64, 768, torch.float32), (1, 7, 1024, torch.float32), (3, 130, 384, torch.float32),
                                          (2, 1, 64, torch.float32), (2, 64, 768, None), (1, 300, 4096, torch.float32)])
def test_koleo_kernel_against_the_oracle_in_groups(G, B, D, dtype):
    """apla_koleo_fwd / _bwd (one launch each for all groups) against oracle/ssl_oracle.py:_koleo per chunk, as models.py:410-413 sums it:
    loss, the neighbour choice (through the gradient) and the gradient; a one-row group (its own neighbour), ragged B, D up to the limit,
    the 16-bit input type, a close pair of rows."""
    from apla_amd import ops
    from apla_amd.ssl import KoLeoLoss
    from oracle.ssl_oracle import _koleo
    dtype = dtype or ops.half()
    torch.manual_seed(G * 1000 + B)
    x = torch.randn(G * B, D)
    if B > 4:
        x[3] = x[1] + 0.02 * torch.randn(D)     # a close pair: each is the other's neighbour
    x = x.to(dtype)
    xd = x.cuda().requires_grad_(True)
    loss = KoLeoLoss().grouped(xd, G)
    (loss * 0.37).backward()
    xr = x.float().requires_grad_(True)
    ref = sum(_koleo(c) for c in xr.chunk(G))
    (ref * 0.37).backward()
    assert abs(float(loss) - float(ref)) < 2e-5 * max(1.0, abs(float(ref)))
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    assert xd.grad.dtype == dtype and rel_err(xd.grad.float().cpu(), xr.grad) < tol
    out, _ = ops.koleo_fwd(xd.detach(), G)
    assert abs(float(out[:G].sum()) - float(out[G])) < 1e-5 and all(int(v) == 0 for v in ops._KOLEO_TICKET.values())


def test_koleo_kernel_on_clamped_norms_and_a_duplicate():
    """The degenerate inputs of koleo_loss.py:17-45: a row whose norm is under eps (F.normalize's clamp: divided by eps, its gradient
    goes through 1 / eps and not through the norm), a zero row (at distance 1 from every unit row: the loss is defined, the neighbour is
    not) and an exact duplicate (distance ||1e-8|| = 1e-8 sqrt(D): the largest loss term)."""
    from apla_amd.ssl import KoLeoLoss
    from oracle.ssl_oracle import _koleo
    torch.manual_seed(5)
    x = torch.randn(6, 64)
    x[4] = 1e-10 * torch.randn(64)
    xd = x.cuda().requires_grad_(True)
    loss = KoLeoLoss()(xd)
    loss.backward()
    xr = x.clone().requires_grad_(True)
    ref = _koleo(xr)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-5
    keep = [0, 1, 2, 3, 5]
    assert rel_err(xd.grad[keep].cpu(), xr.grad[keep]) < 1e-4 and rel_err(xd.grad[4].cpu(), xr.grad[4]) < 1e-4
    with torch.no_grad():
        x[4] = 0.0
        assert abs(float(KoLeoLoss()(x.cuda())) - float(_koleo(x))) < 1e-5
        x[4] = x[2]
        assert abs(float(KoLeoLoss()(x.cuda())) - float(_koleo(x))) < 1e-4


def test_koleo_neighbour_when_rows_nearly_coincide():
    """CLS tokens at initialisation: rows 1e-4 apart after normalisation, where every fp32 inner product rounds to 1 and the reference's
    argmax follows its GEMM's rounding.  The kernel searches by distance, so it must find the float64 nearest neighbour (and the loss of
    that choice), in both input types."""
    from apla_amd import ops
    torch.manual_seed(9)
    base = torch.randn(1, 768)
    x = (base + 3e-3 * torch.randn(64, 768)).to(ops.half()).float()     # the values the 16-bit input carries
    xn = torch.nn.functional.normalize(x.double(), dim=-1)
    dist = torch.cdist(xn, xn)
    dist.fill_diagonal_(float("inf"))
    want = -torch.log(dist.min(dim=1).values + 1e-8).mean()
    for dt in (torch.float32, ops.half()):
        out, (idx, d, _) = ops.koleo_fwd(x.to(dt).cuda(), 1)
        assert torch.equal(idx.cpu().long(), dist.argmin(dim=1)) and abs(float(out[0]) - float(want)) < 1e-3 * abs(float(want))


def _student(seed=0):
    from functools import partial
    from apla_amd.apla import build_apla
    from apla_amd.models import AttrDict
    from apla_amd.ssl import DinoVisionTransformer
    torch.manual_seed(seed)
    bb = DinoVisionTransformer(img_size=[64], patch_size=16, embed_dim=128, depth=2, num_heads=2, qkv_bias=True,
                               norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), block_conf=dict(has_layerscale=True, layerscale_init_values=1.0))
    with torch.no_grad():
        bb.mask_token.normal_(std=0.5)
        for n, p in bb.named_parameters():
            if p.ndim >= 2 and "pos_embed" not in n and "token" not in n:
                p.normal_(std=0.06)
    build_apla(AttrDict(partial_size=64), bb, "apla_attn_mem_eff")
    return bb.cuda()


def test_packed_multicrop_forward_equals_one_pass_per_resolution():
    """forward_features_list([global crops, local crops], [masks, None]) packs 2x17 + 4x5 token sequences into one block-
    diagonal pass per block; it must agree with running each resolution alone through the same modules (dense attention),
    forward and backward (dinov2_vits.py:249-267, block.py:254-288)."""
    bb = _student()
    g = torch.Generator().manual_seed(1)
    glob, loc = torch.randn(2, 3, 64, 64, generator=g).cuda(), torch.randn(4, 3, 32, 32, generator=g).cuda()
    masks = (torch.rand(2, 16, generator=g) < 0.4).cuda()
    outs = bb.forward_features_list([glob, loc], [masks, None])
    assert outs[0]["x_norm_clstoken"].shape == (2, 128) and outs[0]["x_norm_patchtokens"].shape == (2, 16, 128)
    assert outs[1]["x_norm_clstoken"].shape == (4, 128) and outs[1]["x_norm_patchtokens"].shape == (4, 4, 128)
    loss = sum(o["x_norm_clstoken"].float().square().mean() + o["x_norm_patchtokens"].float().square().mean() for o in outs)
    loss.backward()
    packed_grads = {n: p.grad.clone() for n, p in bb.named_parameters() if p.grad is not None}
    bb.zero_grad()
    dense = [bb.forward_features_dict(glob, masks), bb.forward_features_dict(loc, None)]
    for a, b in zip(outs, dense):
        for key in ("x_norm_clstoken", "x_norm_patchtokens"):
            assert rel_err(a[key].detach().float().cpu(), b[key].detach().double().cpu()) < 3e-3, key
    loss2 = sum(o["x_norm_clstoken"].float().square().mean() + o["x_norm_patchtokens"].float().square().mean() for o in dense)
    loss2.backward()
    for n, p in bb.named_parameters():
        if p.grad is not None:
            assert rel_err(packed_grads[n].float().cpu(), p.grad.double().cpu()) < 2e-2, n
    assert not bb.mask_token.requires_grad     # build_apla freezes everything but the selected projection rows (apla_vit.py:51-59)


@pytest.mark.parametrize("half", [torch.bfloat16, torch.float16])
def test_packed_tokens_kernel_against_the_torch_route(half):
    """pack_tokens (apla_assemble_tokens_masked into the packed residual stream) against prepare_tokens_with_masks of the reference
    (dinov2_vits.py:210-222: mask-token replacement, class token, resized position table) evaluated in fp32 on the kernel's own
    16-bit patch rows — exact up to fp32 rounding; and the mask bookkeeping of the packed batch."""
    from apla_amd import ops as OPS
    bb = _student()
    with torch.no_grad():
        bb.cls_token.normal_(std=0.3)
        bb.pos_embed.normal_(std=0.3)
    g = torch.Generator().manual_seed(2)
    glob, loc = torch.randn(3, 3, 64, 64, generator=g).cuda(), torch.randn(5, 3, 32, 32, generator=g).cuda()
    masks = (torch.rand(3, 16, generator=g) < 0.5).cuda()
    masks[1] = False
    with OPS.use_half(half), torch.no_grad():
        mask, packed = bb.pack_tokens([glob, loc], [masks, None])
        assert packed.dtype == torch.float32 and packed.shape == (1, 3 * 17 + 5 * 5, 128)
        assert mask.seqlens == [17] * 3 + [5] * 5 and mask.runs() == [(3, 17), (5, 5)] and mask._batch_sizes == [3, 5]
        row = 0
        for x, m in ((glob, masks), (loc, None)):
            pt = bb.patch_embed(x).float()
            if m is not None:
                pt = torch.where(m.unsqueeze(-1), bb.mask_token.float().unsqueeze(0), pt)
            ref = torch.cat((bb.cls_token.float().expand(x.shape[0], -1, -1), pt), dim=1) + bb.interpolate_pos_encoding(pt.shape[1]).float()
            got = packed[0, row:row + ref.shape[0] * ref.shape[1]].reshape(ref.shape)
            assert float((got - ref).abs().max()) < 1e-6
            row += ref.shape[0] * ref.shape[1]
        # the list forward takes this route when the tokens are frozen (they are under APLA), the torch route otherwise
        outs = bb.forward_features_list([glob, loc], [masks, None])
        bb.cls_token.requires_grad_(True)
        outs2 = bb.forward_features_list([glob, loc], [masks, None])
        bb.cls_token.requires_grad_(False)
        for a, b in zip(outs, outs2):
            assert rel_err(a["x_norm_patchtokens"].float().cpu(), b["x_norm_patchtokens"].double().cpu()) < (2e-2 if half == torch.bfloat16 else 3e-3)


def test_update_teacher_ema_touches_only_trainable_tensors():
    from apla_amd.ssl import update_teacher
    student = _student(seed=0)
    teacher = copy.deepcopy(student)
    with torch.no_grad():
        for p in student.parameters():
            if p.requires_grad:
                p.add_(1.0)
    before = {n: p.detach().clone() for n, p in teacher.named_parameters()}
    n = update_teacher(student, teacher, 0.9)
    sp = dict(student.named_parameters())
    assert n == sum(p.requires_grad for p in student.parameters()) and n > 0
    for name, p in teacher.named_parameters():
        if sp[name].requires_grad:
            assert torch.allclose(p, before[name] * 0.9 + sp[name] * 0.1)
        else:
            assert torch.equal(p, before[name])
