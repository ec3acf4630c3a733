"""Kernel-level parity: every HIP kernel (through the C-ABI, apla_amd.ops) against the CPU oracle evaluated in fp64 on
the SAME bf16-rounded inputs.  Tolerances are stated per test: bf16 outputs carry one rounding (2^-9 relative) on top
of fp32 accumulation-order differences; fp32 outputs are compared much tighter."""
import math

import pytest
import torch

from conftest import col_err, rel_err
from oracle import apla_oracle as O

pytestmark = pytest.mark.gpu

BF16_OUT = 6e-3   # max-abs error / max-abs reference for a bf16-rounded output of O(1) dynamic range
F32_OUT = 2e-5


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from apla_amd import ops as _ops
    return _ops


def dev(t, dtype=None):
    return t.to("cuda", dtype) if dtype else t.to("cuda")


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def bf(t):
    """round to bf16 and return (bf16 tensor, its exact fp64 value)"""
    b = t.to(torch.bfloat16)
    return b, b.double()


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(300, 256, 192), (1000, 384, 768), (128, 128, 64), (1, 128, 64), (25216, 768, 768)])
def test_gemm_store_bias(ops, M, N, K):
    a, ad = bf(rnd(M, K, seed=1))
    w, wd = bf(rnd(N, K, scale=K ** -0.5, seed=2))
    bias = rnd(N, seed=3)
    ref = ad @ wd.t() + bias.double()
    out = ops.gemm_nt(dev(a), dev(w), dev(bias))
    assert rel_err(out.cpu(), ref) < BF16_OUT and col_err(out.cpu(), ref) < 1.5 * BF16_OUT
    out32 = ops.gemm_nt(dev(a), dev(w), dev(bias), out_dtype=torch.float32)
    assert rel_err(out32.cpu(), ref) < F32_OUT and col_err(out32.cpu(), ref) < 4 * F32_OUT
    nob = ops.gemm_nt(dev(a), dev(w), None, out_dtype=torch.float32)
    assert rel_err(nob.cpu(), ad @ wd.t()) < F32_OUT


def test_gemm_strided_a_and_tail(ops):
    """A given as a column slice of a wider buffer (lda > K), M not a multiple of the tile."""
    M, N, K = 333, 128, 128
    big, bigd = bf(rnd(M, 3 * K, seed=4))
    w, wd = bf(rnd(N, K, scale=K ** -0.5, seed=5))
    A = dev(big)[:, K:2 * K]
    out = ops.gemm_nt(A, dev(w), None, out_dtype=torch.float32)
    assert rel_err(out.cpu(), bigd[:, K:2 * K] @ wd.t()) < F32_OUT


@pytest.mark.parametrize("M,N,K", [(1000, 256, 192), (25216, 768, 768), (4096, 3072, 768)])
def test_gemm_output_columns_of_very_different_magnitude(ops, M, N, K):
    """Every output column against ITS OWN scale (col_err): the weight rows span four decades, so a column that came out wrong but small
    would pass the whole-array comparison of the tests above.  Plain store (both output types), GELU + GELU', the MUL epilogue, on
    whichever schedule the shape dispatches to (25 216 x 768 x 768 = wide 4-wave kernel, 4 096 x 3 072 x 768 = persistent 4-wave kernel)."""
    a, ad = bf(rnd(M, K, seed=11))
    decades = 10.0 ** (-4.0 * torch.arange(N).double() / (N - 1))           # column j of the output is scaled by 10^(-4 j / (N-1))
    w, wd = bf((rnd(N, K, scale=K ** -0.5, seed=12).double() * decades[torch.randperm(N, generator=torch.Generator().manual_seed(13))][:, None]).float())
    base = ad @ wd.t()
    out = ops.gemm_nt(dev(a), dev(w), None)
    assert col_err(out.cpu(), base) < 1.5 * BF16_OUT, col_err(out.cpu(), base)
    out32 = ops.gemm_nt(dev(a), dev(w), None, out_dtype=torch.float32)
    assert col_err(out32.cpu(), base) < 4 * F32_OUT, col_err(out32.cpu(), base)
    g = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    h = ops.gemm_nt(dev(a), dev(w), None, epilogue=ops.EPI_GELU, aux_out=g)
    # (GELU of a tiny pre-activation is x / 2 and GELU' is 1 / 2: both keep the column's relative scale)
    assert col_err(h.cpu(), O.gelu_fwd(base)) < 1.5 * BF16_OUT and col_err(g.cpu(), O.gelu_grad(base)) < 1.5 * BF16_OUT
    gm, gmd = bf(rnd(M, N, seed=14))
    mul = ops.gemm_nt(dev(a), dev(w), None, epilogue=ops.EPI_MUL, aux_in=dev(gm))
    assert col_err(mul.cpu(), base * gmd) < 1.5 * BF16_OUT


def test_gemm_gelu(ops):
    M, N, K = 515, 256, 128
    a, ad = bf(rnd(M, K, seed=6))
    w, wd = bf(rnd(N, K, scale=2 * K ** -0.5, seed=7))
    bias = rnd(N, scale=0.3, seed=8)
    pre = ad @ wd.t() + bias.double()
    g = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    h = ops.gemm_nt(dev(a), dev(w), dev(bias), epilogue=ops.EPI_GELU, aux_out=g)
    assert rel_err(h.cpu(), O.gelu_fwd(pre)) < BF16_OUT
    assert rel_err(g.cpu(), O.gelu_grad(pre)) < BF16_OUT


@pytest.mark.parametrize("res_dtype", [torch.float32, torch.bfloat16])
def test_gemm_residual(ops, res_dtype):
    M, N, K = 260, 128, 192
    a, ad = bf(rnd(M, K, seed=9))
    w, wd = bf(rnd(N, K, scale=K ** -0.5, seed=10))
    bias = rnd(N, seed=11)
    res = rnd(M, N, seed=12).to(res_dtype)
    ref = res.double() + ad @ wd.t() + bias.double()
    r_dev = dev(res)
    out = ops.gemm_nt(dev(a), dev(w), dev(bias), epilogue=ops.EPI_RESIDUAL, aux_in=r_dev, out_dtype=res_dtype)
    assert rel_err(out.cpu(), ref) < (F32_OUT if res_dtype == torch.float32 else BF16_OUT)
    # in place (aux_in aliases out)
    ops.gemm_nt(dev(a), dev(w), dev(bias), epilogue=ops.EPI_RESIDUAL, aux_in=r_dev, out=r_dev)
    assert rel_err(r_dev.cpu(), ref) < (F32_OUT if res_dtype == torch.float32 else BF16_OUT)


def test_gemm_mul(ops):
    M, N, K = 130, 256, 64
    a, ad = bf(rnd(M, K, seed=13))
    w, wd = bf(rnd(N, K, scale=K ** -0.5, seed=14))
    g, gd = bf(rnd(M, N, seed=15))
    out = ops.gemm_nt(dev(a), dev(w), None, epilogue=ops.EPI_MUL, aux_in=dev(g))
    assert rel_err(out.cpu(), (ad @ wd.t()) * gd) < BF16_OUT


def test_gemm_swiglu_fwd_bwd(ops):
    M, Fh, K = 200, 128, 64  # hidden Fh; interleaved weight has 2*Fh rows
    a, ad = bf(rnd(M, K, seed=16))
    w12, w12d = bf(rnd(2 * Fh, K, scale=2 * K ** -0.5, seed=17))  # rows [w1; w2]
    b12 = rnd(2 * Fh, scale=0.2, seed=18)
    inter = torch.stack([w12[:Fh], w12[Fh:]], 1).reshape(2 * Fh, K).contiguous()
    binter = torch.stack([b12[:Fh], b12[Fh:]], 1).reshape(2 * Fh).contiguous()
    x12 = ad @ w12d.t() + b12.double()
    x1, x2 = x12[:, :Fh], x12[:, Fh:]
    saved = torch.empty(M, 2 * Fh, device="cuda", dtype=torch.bfloat16)
    h = ops.gemm_nt(dev(a), dev(inter), dev(binter), epilogue=ops.EPI_SWIGLU, aux_out=saved)
    assert rel_err(h.cpu(), O.silu(x1) * x2) < BF16_OUT
    assert rel_err(saved.cpu().double().reshape(M, Fh, 2)[:, :, 0], x1) < BF16_OUT
    # backward: dh = g @ w3t^T ; dx12 interleaved
    Dout = 64
    gup, gupd = bf(rnd(M, Dout, seed=19))
    w3t, w3td = bf(rnd(Fh, Dout, scale=Dout ** -0.5, seed=20))  # [Fh, Dout] = w3^T layout for the NT kernel
    dh = gupd @ w3td.t()
    sv = saved.cpu().double().reshape(M, Fh, 2)
    s1, s2 = sv[:, :, 0], sv[:, :, 1]
    ref = torch.stack([dh * s2 * O.silu_grad(s1), dh * O.silu(s1)], 2).reshape(M, 2 * Fh)
    dx12 = ops.gemm_nt(dev(gup), dev(w3t), None, epilogue=ops.EPI_SWIGLU_BWD, aux_in=saved)
    assert rel_err(dx12.cpu(), ref) < BF16_OUT


def test_gemm_rejects_bad_shapes(ops):
    from apla_amd._lib import AplaHipError
    a = torch.zeros(16, 96, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(128, 96, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(AplaHipError):
        ops.gemm_nt(a, w)  # K % 64 != 0
    with pytest.raises(AplaHipError):
        ops.gemm_nt(torch.zeros(4, 64), torch.zeros(128, 64))  # CPU tensors: no CPU path


# ------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("res_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,D", [(37, 64), (1001, 384), (515, 768), (64, 1536)])
def test_layernorm_fwd_bwd(ops, res_dtype, M, D):
    x = (rnd(M, D, seed=21) * 1.5 + 0.3).to(res_dtype)
    gamma, beta = 1 + 0.2 * rnd(D, seed=22), 0.1 * rnd(D, seed=23)
    xd = x.double()
    yref, mref, rref = O.layernorm_fwd(xd, gamma.double(), beta.double(), 1e-6)
    y, mean, rstd = ops.layernorm_fwd(dev(x), dev(gamma), dev(beta), 1e-6)
    assert rel_err(y.cpu(), yref) < BF16_OUT and col_err(y.cpu(), yref) < 1.5 * BF16_OUT
    assert rel_err(mean.cpu(), mref) < 1e-5 and rel_err(rstd.cpu(), rref) < 1e-5
    y32, _, _ = ops.layernorm_fwd(dev(x), dev(gamma), dev(beta), 1e-6, out_dtype=torch.float32)
    assert rel_err(y32.cpu(), yref) < F32_OUT and col_err(y32.cpu(), yref) < 4 * F32_OUT
    dy, dyd = bf(rnd(M, D, seed=24))
    dres = rnd(M, D, seed=25).to(res_dtype)
    ref = dres.double() + O.layernorm_bwd_dx(dyd, xd, gamma.double(), mref, rref)
    r = min(64, D // 2)
    inds = torch.randperm(D, generator=torch.Generator().manual_seed(1)).int()
    dx, gathered = ops.layernorm_bwd(dev(dy), dev(x), dev(gamma), mean, rstd, dres=dev(dres), inds=dev(inds), r=r)
    tol = F32_OUT * 5 if res_dtype == torch.float32 else BF16_OUT
    assert rel_err(dx.cpu(), ref) < tol and col_err(dx.cpu(), ref) < 2 * tol
    assert rel_err(gathered.cpu(), ref[:, inds[:r].long()]) < BF16_OUT and col_err(gathered.cpu(), ref[:, inds[:r].long()]) < 1.5 * BF16_OUT
    dx2, none = ops.layernorm_bwd(dev(dy), dev(x), dev(gamma), mean, rstd)
    assert none is None
    assert rel_err(dx2.cpu(), ref - dres.double()) < tol
    g2 = ops.gather_cols(dx, dev(inds), r)
    assert torch.equal(g2.cpu(), dx.cpu()[:, inds[:r].long()].to(torch.bfloat16))


def test_layernorm_cls_rows_strided(ops):
    """final norm on CLS rows only: row m at offset m*N*D of the residual buffer (vit.py:416-419)."""
    B, N, D = 5, 7, 128
    x = rnd(B * N, D, seed=26)
    gamma, beta = 1 + 0.2 * rnd(D, seed=27), 0.1 * rnd(D, seed=28)
    xdev = dev(x)
    y, mean, rstd = ops.layernorm_fwd(xdev, dev(gamma), dev(beta), 1e-6, out_dtype=torch.float32, rows=B, row_stride=N * D)
    yref, mref, rref = O.layernorm_fwd(x.double()[::N], gamma.double(), beta.double(), 1e-6)
    assert rel_err(y.cpu(), yref) < F32_OUT
    dy = rnd(B, D, seed=29)
    out = torch.zeros(B * N, D, device="cuda")
    ops.layernorm_bwd(dev(dy), xdev, dev(gamma), mean, rstd, out=out, rows=B, row_stride=N * D)
    ref = torch.zeros(B * N, D, dtype=torch.float64)
    ref[::N] = O.layernorm_bwd_dx(dy.double(), x.double()[::N], gamma.double(), mref, rref)
    assert rel_err(out.cpu(), ref) < 5 * F32_OUT


@pytest.mark.parametrize("M,D,N", [(6 * 7, 128, 7), (515, 768, 5), (4 * 197, 384, 197)])
def test_layernorm_normalised_row_mode(ops, M, D, N):
    """What the fused step runs since round 3 (apla_layernorm_bwd_ex): the forward writes xhat itself (gamma / beta are folded into
    the next frozen Linear: W' = W diag(gamma), b' = b + W beta), in place on ONE residual buffer; the backward reads that 16-bit
    xhat + rstd instead of the fp32 row, takes dy as the gradient of xhat, and can treat the incoming residual gradient as
    non-zero in every N-th row only.  Checked against the oracle's LayerNorm (vit.py:279-288) with the fold undone."""
    x = rnd(M, D, seed=31) * 1.5 + 0.3
    add, addd = bf(rnd(M, D, seed=32) * 0.5)
    gamma, beta = 1 + 0.2 * rnd(D, seed=33), 0.1 * rnd(D, seed=34)
    W = rnd(64, D, seed=35) * D ** -0.5
    xd = x.double() + addd
    yref, mref, rref = O.layernorm_fwd(xd, gamma.double(), beta.double(), 1e-6)
    res = dev(x)
    xh, _, rstd = ops.layernorm_fwd(res, None, None, 1e-6, add=dev(add), x_out=res)          # in place
    assert rel_err(res.cpu(), xd) < 1e-6 and rel_err(rstd.cpu(), rref) < 1e-5
    xhat_ref = (xd - mref[:, None]) * rref[:, None]
    assert rel_err(xh.cpu(), xhat_ref) < BF16_OUT
    # the fold: LN(x) W^T = xhat (W diag(gamma))^T + W beta
    lhs = yref @ W.double().t()
    rhs = xhat_ref @ (W.double() * gamma.double()[None, :]).t() + (W.double() @ beta.double())[None, :]
    assert rel_err(rhs, lhs) < 1e-12
    # backward: dy_hat = dy * gamma is what the dX GEMM on the folded weight delivers
    dy, dyd = bf(rnd(M, D, seed=36))
    ref_dx = O.layernorm_bwd_dx(dyd, xd, torch.ones(D, dtype=torch.float64), mref, rref)      # gradient of xhat given
    dres = torch.zeros(M, D)
    dres[::N] = rnd(M, D, seed=37)[::N]
    G = dev(rnd(M, D, seed=38).to(torch.bfloat16))      # garbage everywhere ...
    G[::N] = dev(dres[::N].to(torch.bfloat16))         # ... except the rows that carry gradient
    dresd = G.cpu().double()
    inds = torch.randperm(D, generator=torch.Generator().manual_seed(2)).int()
    r = 64
    out, gathered = ops.layernorm_bwd(dev(dy), xh, None, None, rstd, dres=G, out=G, dres_period=N, inds=dev(inds), r=r)
    ref = ref_dx.clone()
    ref[::N] += dresd[::N]
    assert rel_err(out.cpu(), ref) < 1.5 * BF16_OUT          # xhat rounded to 16 bits enters the projection term
    assert rel_err(gathered.cpu(), ref[:, inds[:r].long()]) < 1.5 * BF16_OUT
    # strided rows with a compact xhat (the CLS-only last block): x_row_stride = D, gradient rows N*D apart
    B = M // N
    xh_c = xh.view(B, N * D)[:, :D].contiguous()
    out2 = torch.zeros(M, D, device="cuda")
    ops.layernorm_bwd(dev(dy[::N].contiguous()), xh_c, None, None, rstd[::N].contiguous(), out=out2, rows=B, row_stride=N * D, x_row_stride=D)
    ref2 = torch.zeros(M, D, dtype=torch.float64)
    ref2[::N] = ref_dx[::N]
    assert rel_err(out2.cpu(), ref2) < 1.5 * BF16_OUT


@pytest.mark.parametrize("M,D,N", [(6 * 7, 128, 7), (3 * 197, 768, 197), (4 * 50, 1024, 50), (2 * 33, 1536, 33)])
def test_layernorm_regularisation_forms(ops, M, D, N):
    """Round 6: the LayerNorm pair with a branch's regularisation inside.  Forward (apla_layernorm_fwd_dp / _drop): x_new = x +
    scale[sample] * dropout(add) with the keep decision of element m * row_stride + c drawn in the kernel from {seed, step} in device
    memory — against the oracle's philox_keep_mask (the definition of apla_dropout_fwd), dense rows and every N-th row of a larger tensor
    (the index stride is the full tensor's).  Backward (apla_layernorm_bwd_dp / _drop): dx = dres + dy_scale[sample] * LN_bwd(dy), a second
    copy of dx through the NEXT consumer's mask x its branch's factor, and the gathered columns taken from that copy.  Widths 128 .. 1536
    (every chunk-count instantiation)."""
    B = M // N
    seed, step, stride, site, p = 0x5EED_1234_ABCD_0001, 3, 17, 5, 0.25
    rng = torch.tensor([seed, step], dtype=torch.int64, device="cuda")
    off = step * stride + site
    inv = 1.0 / (1.0 - float(torch.tensor(p, dtype=torch.float32)))
    x = rnd(M, D, seed=61) * 1.5 + 0.3
    add, addd = bf(rnd(M, D, seed=62) * 0.5)
    scale = torch.tensor([0.0, 1.25] * B)[:B].float()
    keep = torch.from_numpy(O.philox_keep_mask(M * D, p, seed, off)).reshape(M, D)
    xd = x.double() + scale.double().repeat_interleave(N)[:, None] * (addd * keep.double() * inv)
    _, mref, rref = O.layernorm_fwd(xd, torch.ones(D, dtype=torch.float64), torch.zeros(D, dtype=torch.float64), 1e-6)
    res = dev(x)
    xh, _, rstd = ops.layernorm_fwd(res, None, None, 1e-6, add=dev(add), x_out=res, add_scale=dev(scale), scale_period=N,
                                    drop=(rng, stride, site, p, D))
    assert rel_err(res.cpu(), xd) < 1e-6 and rel_err(rstd.cpu(), rref) < 1e-5
    xhat_ref = (xd - mref[:, None]) * rref[:, None]
    assert rel_err(xh.cpu(), xhat_ref) < BF16_OUT
    # every N-th row only (the CLS rows of the final norm): the mask index is the row's position in the FULL tensor
    res2 = dev(x)
    xn2 = torch.empty(B, D, device="cuda", dtype=torch.bfloat16)
    m2, r2 = torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    ops.layernorm_fwd(res2, None, None, 1e-6, out=xn2, mean=m2, rstd=r2, rows=B, row_stride=N * D, add=dev(add), x_out=res2, D=D,
                      add_scale=dev(scale), scale_period=1, drop=(rng, stride, site, p, N * D))
    assert rel_err(res2.cpu()[::N], xd[::N]) < 1e-6 and torch.equal(res2.cpu()[1::N], x[1::N])
    # backward: stream, masked copy (other site), gathered columns
    dy, dyd = bf(rnd(M, D, seed=63))
    G0, G0d = bf(rnd(M, D, seed=64))
    dys = torch.tensor([2.0, 0.0, 1.0] * B)[:B].float()
    msc = torch.tensor([1.0, 0.0, 4.0 / 3.0] * B)[:B].float()
    ref_dx = O.layernorm_bwd_dx(dyd, xd, torch.ones(D, dtype=torch.float64), mref, rref)
    ref = G0d + dys.double().repeat_interleave(N)[:, None] * ref_dx
    site2 = 9
    keep2 = torch.from_numpy(O.philox_keep_mask(M * D, p, seed, step * stride + site2)).reshape(M, D)
    inds = torch.randperm(D, generator=torch.Generator().manual_seed(3)).int()
    r = 64
    G, Gm = dev(G0), torch.zeros(M, D, device="cuda", dtype=torch.bfloat16)
    out, gathered = ops.layernorm_bwd(dev(dy), xh, None, None, rstd, dres=G, out=G, inds=dev(inds), r=r, dy_scale=dev(dys), scale_period=N,
                                      masked=Gm, mask_scale=dev(msc), drop=(rng, stride, site2, p))
    assert rel_err(out.cpu(), ref) < 1.5 * BF16_OUT
    want = out.cpu().double() * keep2.double() * inv * msc.double().repeat_interleave(N)[:, None]      # the mask acts on the stored stream
    assert rel_err(Gm.cpu(), want) < BF16_OUT and bool(((Gm.cpu() == 0) | keep2).all())
    assert torch.equal(gathered.cpu(), Gm.cpu()[:, inds[:r].long()])
    # without a mask: the factor of the gathered columns is gather_scale
    G = dev(G0)
    out3, g3 = ops.layernorm_bwd(dev(dy), xh, None, None, rstd, dres=G, out=G, inds=dev(inds), r=r, dy_scale=dev(dys), gather_scale=dev(msc), scale_period=N)
    assert torch.equal(out3, out)
    assert rel_err(g3.cpu(), ref[:, inds[:r].long()] * msc.double().repeat_interleave(N)[:, None]) < 1.5 * BF16_OUT


@pytest.mark.parametrize("M,N,K,img", [(300, 256, 128, False), (5000, 3072, 768, True), (9000, 768, 256, True), (1000, 384 * 4, 384, False)])
def test_gemm_gelu_with_dropout_in_the_epilogue(ops, M, N, K, img):
    """apla_gemm_nt_gelu_drop (round 6): fc1 + GELU + GELU' with Mlp.drop after the activation inside the epilogue — both outputs equal the
    plain launch's outputs through ONE keep mask, the oracle's for element m * N + n, whatever layout the outputs have."""
    a = dev(bf(rnd(M, K, seed=71))[0])
    w = dev(bf(rnd(N, K, scale=K ** -0.5, seed=72))[0])
    bias = dev(rnd(N, seed=73))
    seed, step, stride, site, p = 0x0DDB_A11_5EED, 2, 56, 10, 0.3
    rng = torch.tensor([seed, step], dtype=torch.int64, device="cuda")
    g_ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    old = ops.set_gemm_variant(15)
    try:
        h_ref = ops.gemm_nt(a, w, bias, epilogue=ops.EPI_GELU, aux_out=g_ref)
    finally:
        ops.set_gemm_variant(old)
    keep = torch.from_numpy(O.philox_keep_mask(M * N, p, seed, step * stride + site)).reshape(M, N).cuda()
    inv = 1.0 / (1.0 - float(torch.tensor(p, dtype=torch.float32)))
    if img:
        h, g = torch.zeros(N // 32, M, 32, device="cuda", dtype=torch.bfloat16), torch.zeros(N // 32, M, 32, device="cuda", dtype=torch.bfloat16)
    else:
        h, g = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16), torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt(a, w, bias, epilogue=ops.EPI_GELU, aux_out=g, out=h, drop=(rng, stride, site, p))
    if img:
        h, g = h.permute(1, 0, 2).reshape(M, N), g.permute(1, 0, 2).reshape(M, N)
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.01
    assert bool(((h == 0) | keep).all()) and bool(((g == 0) | keep).all())
    # kept entries: the plain outputs times 1 / (1 - p), rounded once from fp32 (the plain outputs were rounded before the scaling: one ulp)
    hk, gk = h[keep].float(), g[keep].float()
    assert rel_err(hk.cpu(), (h_ref[keep].float() * inv).cpu()) < 2 * 2 ** -8 and rel_err(gk.cpu(), (g_ref[keep].float() * inv).cpu()) < 2 * 2 ** -8


# ------------------------------------------------------------------------------------------- attention
@pytest.fixture
def attn_variant(request):
    """0 = auto, 1 = always the key-blocked kernels, 2 = the one-workgroup-per-head short-sequence kernels, 3 = the persistent
    kernels (one workgroup per CU walking its heads) wherever they apply."""
    from apla_amd import ops
    old = ops.set_attn_variant(request.param)
    yield request.param
    ops.set_attn_variant(old)


@pytest.mark.parametrize("attn_variant", [0, 1, 2, 3], indirect=True)
@pytest.mark.parametrize("B,N,H", [(2, 197, 2), (1, 64, 1), (3, 300, 3), (1, 1, 1), (2, 129, 2), (1, 1370, 1), (2, 256, 3),
                                   (1, 33, 2), (2, 224, 1), (2, 257, 2), (3, 288, 1), (5, 225, 2), (7, 50, 3), (3, 32, 1), (2, 65, 1)])
def test_attention_fwd_bwd(ops, attn_variant, B, N, H):
    D = 64 * H
    scale = 64 ** -0.5
    qkv, qkvd = bf(rnd(B, N, 3 * D, seed=31))
    oref, lref, aref = O.attention_fwd(qkvd, H, scale, return_attn=True)
    o, lse = ops.attn_fwd(dev(qkv).reshape(B * N, 3 * D), B, N, H, scale)
    assert rel_err(o.cpu().reshape(B, N, D), oref) < BF16_OUT
    assert float((lse.cpu().double() - lref).abs().max()) < 2e-4
    do, dod = bf(rnd(B, N, D, seed=32))
    # reference backward evaluated at the bf16-rounded o that the kernel actually saved
    o_saved = o.cpu().double().reshape(B, N, D)
    dref = O.attention_bwd(dod, qkvd, o_saved, lref, H, scale)
    dqkv = ops.attn_bwd(dev(qkv).reshape(B * N, 3 * D), o, dev(do).reshape(B * N, D), lse, B, N, H, scale)
    got = dqkv.cpu().reshape(B, N, 3 * D)
    for i, nm in enumerate(("dq", "dk", "dv")):
        e = rel_err(got[..., i * D:(i + 1) * D], dref[..., i * D:(i + 1) * D])
        assert e < 2e-2, (nm, e)   # P and dS are rounded to bf16 before the second product
    attn = ops.attn_probs(dev(qkv).reshape(B * N, 3 * D), lse, B, N, H, scale)
    assert rel_err(attn.cpu(), aref) < 1e-4


@pytest.mark.parametrize("attn_variant", [0, 1], indirect=True)
@pytest.mark.parametrize("seqlens,H", [([257, 257, 50, 50, 50, 50], 2), ([1, 64, 33, 200, 7], 1), ([300, 17, 129], 2),
                                        ([50] * 16, 3), ([197], 2)])
def test_attention_block_diagonal_packed(ops, attn_variant, seqlens, H):
    """Packed crops with a block-diagonal mask (dinov2 nested-tensor path, appla_attn_mem_eff.py:40-42): every sequence
    must match the dense per-sequence attention of the oracle; tolerance as in test_attention_fwd_bwd."""
    D, total = 64 * H, sum(seqlens)
    scale = 64 ** -0.5
    qkv, qkvd = bf(rnd(total, 3 * D, seed=35))
    oref, lref = O.attention_varlen_fwd(qkvd, seqlens, H, scale)
    cu = torch.tensor([0] + list(torch.tensor(seqlens).cumsum(0)), dtype=torch.int32, device="cuda")
    o, lse = ops.attn_varlen_fwd(dev(qkv), cu, max(seqlens), H, scale)
    assert rel_err(o.cpu(), oref) < BF16_OUT
    assert float((lse.cpu().double() - lref).abs().max()) < 2e-4
    do, dod = bf(rnd(total, D, seed=36))
    dref = O.attention_varlen_bwd(dod, qkvd, o.cpu().double(), lref, seqlens, H, scale)
    got = ops.attn_varlen_bwd(dev(qkv), o, dev(do), lse, cu, max(seqlens), H, scale).cpu()
    for i, nm in enumerate(("dq", "dk", "dv")):
        e = rel_err(got[:, i * D:(i + 1) * D], dref[:, i * D:(i + 1) * D])
        assert e < 2e-2, (nm, e)
    # a uniform batch expressed as a packed one gives bit-identical results to the uniform entry point
    if len(set(seqlens)) == 1:
        B, N = len(seqlens), seqlens[0]
        o2, lse2 = ops.attn_fwd(dev(qkv), B, N, H, scale)
        assert torch.equal(o2, o) and torch.equal(lse2.permute(1, 0, 2).reshape(H, total), lse)


@pytest.mark.parametrize("half", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("runs,H", [([(6, 257), (24, 50)], 2), ([(3, 197)], 3), ([(2, 33), (260, 50), (1, 300)], 1)])
def test_attention_packed_as_runs_of_uniform_batches(ops, runs, H, half):
    """The multi-crop batch as the backbone runs it (functional.attention_core_varlen with the mask's runs): one uniform launch per
    run of equal-length sequences (260 x 50 tokens takes the persistent backward).  Forward and the three gradients against the
    oracle's per-sequence attention, and against the one-launch packed kernels on the same operands."""
    from apla_amd import functional as AF
    from apla_amd import ops as OPS
    from apla_amd.nested import BlockDiagonalMask
    seqlens = [n for c, n in runs for _ in range(c)]
    mask = BlockDiagonalMask(seqlens)
    assert mask.runs() == [tuple(r) for r in runs]
    D, total, scale = 64 * H, sum(seqlens), 64 ** -0.5
    x = rnd(total, 3 * D, seed=37)
    dox = rnd(total, D, seed=38)
    with OPS.use_half(half):
        qkv = x.to(half)
        qkvd = qkv.double()
        oref, lref = O.attention_varlen_fwd(qkvd, seqlens, H, scale)
        q = dev(qkv).reshape(1, total, 3 * D).requires_grad_(True)
        o = AF.attention_core_varlen(q, mask.cu_seqlens("cuda"), mask.max_seqlen, H, scale, runs=mask.runs())
        tol_o = BF16_OUT if half == torch.bfloat16 else BF16_OUT / 8
        assert o.shape == (1, total, D) and rel_err(o[0].detach().cpu(), oref) < tol_o
        do = dox.to(half)
        dref = O.attention_varlen_bwd(do.double(), qkvd, o[0].detach().cpu().double(), lref, seqlens, H, scale)
        o.backward(dev(do).reshape(1, total, D))
        got = q.grad[0].cpu()
        for i, nm in enumerate(("dq", "dk", "dv")):
            e = rel_err(got[:, i * D:(i + 1) * D], dref[:, i * D:(i + 1) * D])
            assert e < (2e-2 if half == torch.bfloat16 else 3e-3), (nm, e)
        # the packed one-launch path on the same operands: same forward bits (same kernel body per sequence)
        q2 = dev(qkv).reshape(1, total, 3 * D).requires_grad_(True)
        o2 = AF.attention_core_varlen(q2, mask.cu_seqlens("cuda"), mask.max_seqlen, H, scale)
        if max(seqlens) <= 288:
            assert torch.equal(o2, o)
        # more runs than MAX_ATTN_RUNS: the packed launch, whatever `runs` says
        many = [(1, n) for n in seqlens]
        if len(many) > AF.MAX_ATTN_RUNS:
            o3 = AF.attention_core_varlen(dev(qkv).reshape(1, total, 3 * D), mask.cu_seqlens("cuda"), mask.max_seqlen, H, scale, runs=many)
            assert torch.equal(o3, o2)
        with pytest.raises(ValueError):
            AF.attention_core_varlen(dev(qkv).reshape(1, total, 3 * D), mask.cu_seqlens("cuda"), mask.max_seqlen, H, scale, runs=[(1, total - 1)])


@pytest.mark.parametrize("B,N,H", [(24, 197, 12), (128, 197, 12), (70, 129, 4), (40, 224, 8), (300, 65, 1),
                                   (128, 257, 12), (30, 257, 12), (64, 256, 6), (50, 280, 8), (70, 225, 4), (1, 257, 2),
                                   (100, 50, 6), (700, 33, 2), (2100, 17, 1), (64, 64, 4), (3000, 1, 1), (5, 50, 2)])
def test_attention_persistent_forward_walks_its_heads(ops, B, N, H):
    """attn_fwd_persist_kernel with MORE heads than workgroups (one workgroup per CU walks its heads: double-buffered K / V fed by the
    loader wave, q rows requested a head ahead, O stored one head late) — the small cases of test_attention_fwd_bwd give every
    workgroup exactly one head.  225..280 tokens: attn_fwd_persist_blocks_kernel (four waves walking the 8 or 9 query blocks); up to
    128 tokens several small workgroups share a CU (the grid is the kernel's occupancy x CUs).  Against the one-workgroup-per-head kernel (itself checked against the oracle there): one maximum per
    row instead of four running ones, so equal to rounding — O within 4 ulp of the largest 16-bit output, lse within 2e-6 — and, on a sample
    of heads spread over the walk, against the fp64 oracle."""
    from apla_amd import ops as OPS
    D, scale = 64 * H, 64 ** -0.5
    qkv, _ = bf(rnd(B * N, 3 * D, seed=39))
    q = dev(qkv)
    old = OPS.set_attn_variant(2)
    try:
        o_ref, lse_ref = ops.attn_fwd(q, B, N, H, scale)
        OPS.set_attn_variant(0)
        if B * H >= 256:
            assert ops.attn_kernel_name("fwd", B, N, H) == ("attn_fwd_persist_kernel" if N <= 224 else "attn_fwd_persist_blocks_kernel")
        else:
            OPS.set_attn_variant(3)      # fewer heads than CUs: the launch would take the one-workgroup-per-head kernel
        o, lse = ops.attn_fwd(q, B, N, H, scale)
        o2, lse2 = ops.attn_fwd(q, B, N, H, scale)
    finally:
        OPS.set_attn_variant(old)
    assert torch.equal(o, o2) and torch.equal(lse, lse2)                     # reproducible
    assert float((lse - lse_ref).abs().max()) < 2e-6 * max(1.0, float(lse_ref.abs().max()))
    d, top = (o.float() - o_ref.float()).abs(), float(o_ref.float().abs().max())
    # an output is a sum of ~N rounded products: the two kernels round P against different maxima, so single outputs differ by a few
    # units in the last place of the LARGEST outputs (not of their own magnitude: small outputs are sums with cancellation)
    assert float(d.max()) < 4 * 2 ** -9 * top and float(d.mean()) < 2 ** -12 * top and float((d > 0).float().mean()) < 0.4
    for b in sorted({0, B // 3, B - 1}):                                       # first, middle and last sequences of the walk
        oref, lref = O.attention_fwd(qkv[b * N:(b + 1) * N].double().reshape(1, N, 3 * D), H, scale)
        assert rel_err(o[b * N:(b + 1) * N].cpu().reshape(1, N, D), oref) < BF16_OUT
        assert float((lse[b].cpu().double() - lref[0]).abs().max()) < 2e-4


@pytest.mark.parametrize("B,N,H,p", [(2, 197, 2, 0.1), (1, 64, 1, 0.5), (3, 300, 2, 0.25), (2, 129, 3, 0.1), (1, 33, 1, 0.9), (2, 257, 2, 0.3)])
def test_attention_probability_dropout_fwd_bwd(ops, B, N, H, p):
    """dropout INSIDE the attention (appla_attn.py:56-58, main.py --adr) — forward and the three gradients against the
    oracle's dense masked softmax, with the mask the oracle regenerates from (seed, offset) through its own Philox4x32-10
    (oracle/apla_oracle.py:philox_attn_keep_mask; the generator itself is pinned by the Random123 known-answer vectors).  The mask
    is never stored: the forward and the two backward kernels each rebuild it in their own tiling, so agreement of all three with ONE
    oracle mask checks the counter mapping of every kernel.  lse stays the softmax's own."""
    D, scale, seed, offset = 64 * H, 64 ** -0.5, 0x1234_5678_9ABC_DEF1 + N, 7
    qkv, qkvd = bf(rnd(B, N, 3 * D, seed=51))
    do, dod = bf(rnd(B, N, D, seed=52))
    keep = O.philox_attn_keep_mask(B, H, N, p, seed, offset)
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.02
    oref, lref, dref = O.attention_dropout_fwd_bwd(qkvd, H, scale, keep, float(torch.tensor(p, dtype=torch.float32)), dod)
    o, lse = ops.attn_fwd_dropout(dev(qkv).reshape(B * N, 3 * D), B, N, H, scale, p, seed, offset)
    assert rel_err(o.cpu().reshape(B, N, D), oref) < BF16_OUT
    assert float((lse.cpu().double() - lref).abs().max()) < 2e-4
    # a different offset is a different mask; the same arguments the same bits
    o2, _ = ops.attn_fwd_dropout(dev(qkv).reshape(B * N, 3 * D), B, N, H, scale, p, seed, offset + 1)
    o3, _ = ops.attn_fwd_dropout(dev(qkv).reshape(B * N, 3 * D), B, N, H, scale, p, seed, offset)
    assert not torch.equal(o2, o) and torch.equal(o3, o)
    # backward evaluated at the 16-bit o the kernel saved (delta = rowsum(dO o O))
    _, _, dref = O.attention_dropout_fwd_bwd(qkvd, H, scale, keep, float(torch.tensor(p, dtype=torch.float32)), dod)
    got = ops.attn_bwd_dropout(dev(qkv).reshape(B * N, 3 * D), o, dev(do).reshape(B * N, D), lse, B, N, H, scale, p, seed, offset).cpu().reshape(B, N, 3 * D)
    for i, nm in enumerate(("dq", "dk", "dv")):
        e = rel_err(got[..., i * D:(i + 1) * D], dref[..., i * D:(i + 1) * D])
        assert e < 2e-2, (nm, e)
    with pytest.raises(Exception):
        ops.attn_fwd_dropout(dev(qkv).reshape(B * N, 3 * D), B, N, H, scale, 1.0, seed)
    # the matrix the module returns in training mode (appla_attn.py:58, :83): the softmax after the SAME dropout — zero exactly where the
    # oracle's mask drops, softmax / (1 - p) elsewhere
    t = qkvd.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    aref = torch.softmax((t[0] @ t[1].transpose(-2, -1)) * scale, -1) * keep.double() / (1.0 - float(torch.tensor(p, dtype=torch.float32)))
    a = ops.attn_probs(dev(qkv).reshape(B * N, 3 * D), lse, B, N, H, scale, p, seed, offset).cpu()
    assert torch.equal(a != 0, keep & (a != 0)) and float(((a == 0) != ~keep).float().mean()) < 1e-3    # (an underflowed kept entry may be 0)
    assert float((a.double() - aref).abs().max()) < 2e-3 * float(aref.abs().max())
    a0 = ops.attn_probs(dev(qkv).reshape(B * N, 3 * D), lse, B, N, H, scale).cpu()
    assert float((a0.sum(-1) - 1).abs().max()) < 1e-3


@pytest.mark.parametrize("attn_variant", [0, 1], indirect=True)
def test_attention_online_softmax_rescale_branch(ops, attn_variant):
    """Force the running max to jump at a later key block (guide rule 26): one key spikes against every query."""
    B, N, H = 1, 200, 1
    scale = 64 ** -0.5
    qkv = rnd(B, N, 192, seed=33)
    qkv[0, 150, 64:128] = qkv[0, :, 0:64].mean(0) * 0 + 6.0 * torch.sign(qkv[0, 3, 0:64])  # big key in block 2
    qkvb, qkvd = bf(qkv)
    oref, lref = O.attention_fwd(qkvd, H, scale)
    o, lse = ops.attn_fwd(dev(qkvb).reshape(N, 192), B, N, H, scale)
    assert rel_err(o.cpu().reshape(B, N, 64), oref) < BF16_OUT
    assert float((lse.cpu().double() - lref).abs().max()) < 1e-3


# ------------------------------------------------------------------------------------------- APLA dW / pack
@pytest.mark.parametrize("M,r,D", [(1000, 64, 128), (197 * 8, 64, 384), (64, 128, 256), (25216, 192, 768), (300, 16384, 256)])
def test_proj_dw(ops, M, r, D):
    dyg, dygd = bf(rnd(M, r, seed=41))
    x, xd = bf(rnd(M, D, seed=42))
    scale = (0.5 + torch.rand(r, generator=torch.Generator().manual_seed(43))) * 10.0 ** (-3.0 * torch.rand(r, generator=torch.Generator().manual_seed(44)))
    ref_w = scale.double()[:, None] * (dygd.t() @ xd)
    ref_b = scale.double() * dygd.sum(0)
    dW = torch.full((r, D), 7.0, device="cuda")
    db = torch.full((r,), 7.0, device="cuda")
    ops.proj_dw(dev(dyg), dev(x), dW, db, row_scale=dev(scale))
    assert rel_err(dW.cpu(), ref_w) < 5e-5 and rel_err(db.cpu(), ref_b) < 5e-5
    # every trainable row of dW (a column of dW^T) and every column against its own scale
    assert col_err(dW.cpu().t(), ref_w.t()) < 2e-4 and col_err(dW.cpu(), ref_w) < 2e-4
    ops.proj_dw(dev(dyg), dev(x), dW, db, row_scale=dev(scale), accumulate=True)
    assert rel_err(dW.cpu(), 2 * ref_w) < 5e-5 and rel_err(db.cpu(), 2 * ref_b) < 5e-5
    # determinism: bitwise identical on a re-run
    dW2, db2 = torch.empty_like(dW), torch.empty_like(db)
    ops.proj_dw(dev(dyg), dev(x), dW2, db2, row_scale=dev(scale))
    dW3, db3 = torch.empty_like(dW), torch.empty_like(db)
    ops.proj_dw(dev(dyg), dev(x), dW3, db3, row_scale=dev(scale))
    assert torch.equal(dW2, dW3) and torch.equal(db2, db3)


@pytest.mark.parametrize("M,r,D,nb", [(1000, 64, 128, 2), (3001, 192, 768, 5), (700, 128, 256, 8), (130, 64, 128, 3)])
def test_proj_dw_batched(ops, M, r, D, nb):
    """Several projections' weight gradients in one launch pair (apla_proj_dw_batched): every layer against the fp64 oracle,
    with a row scale on some layers only, accumulate, and bitwise reproducibility."""
    dygs, xs, refs_w, refs_b, scales = [], [], [], [], []
    for l in range(nb):
        dyg, dygd = bf(rnd(M, r, seed=300 + l))
        x, xd = bf(rnd(M, D, seed=400 + l))
        sc = None if l % 2 else 0.5 + torch.rand(r, generator=torch.Generator().manual_seed(500 + l))
        scd = torch.ones(r, dtype=torch.float64) if sc is None else sc.double()
        dygs.append(dev(dyg)), xs.append(dev(x)), scales.append(None if sc is None else dev(sc))
        refs_w.append(scd[:, None] * (dygd.t() @ xd)), refs_b.append(scd * dygd.sum(0))
    dWs = [torch.full((r, D), 3.0, device="cuda") for _ in range(nb)]
    dbs = [torch.full((r,), 3.0, device="cuda") for _ in range(nb)]
    ops.proj_dw_batched(dygs, xs, dWs, dbs, row_scale=scales)
    for l in range(nb):
        assert rel_err(dWs[l].cpu(), refs_w[l]) < 5e-5 and rel_err(dbs[l].cpu(), refs_b[l]) < 5e-5, l
    ops.proj_dw_batched(dygs, xs, dWs, dbs, row_scale=scales, accumulate=True)
    for l in range(nb):
        assert rel_err(dWs[l].cpu(), 2 * refs_w[l]) < 5e-5 and rel_err(dbs[l].cpu(), 2 * refs_b[l]) < 5e-5, l
    dW2 = [torch.empty_like(t) for t in dWs]
    db2 = [torch.empty_like(t) for t in dbs]
    dW3 = [torch.empty_like(t) for t in dWs]
    db3 = [torch.empty_like(t) for t in dbs]
    ops.proj_dw_batched(dygs, xs, dW2, db2, row_scale=scales)
    ops.proj_dw_batched(dygs, xs, dW3, db3, row_scale=scales)
    assert all(torch.equal(a, b) for a, b in zip(dW2, dW3)) and all(torch.equal(a, b) for a, b in zip(db2, db3))
    with pytest.raises(ValueError):
        ops.proj_dw_batched(dygs * 9, xs * 9, dWs * 9, dbs * 9)


def test_pack_proj_rows_equals_reference_scatter(ops):
    """Weight-side scatter == the reference's activation-side scatter (appla_attn.py:64-79)."""
    D, r, M = 128, 32, 50
    torch.manual_seed(5)
    W, b = rnd(D, D, scale=0.1, seed=44), rnd(D, scale=0.1, seed=45)
    inds = O.sample_indices(D)
    W1, W2, b1, b2 = O.split_proj(W, b, inds, r)
    gamma = 0.5 + torch.rand(D, generator=torch.Generator().manual_seed(46))
    # frozen part prepared on the host once; trainable rows packed by the kernel
    Wn = torch.zeros(D, D)
    Wn[inds[r:]] = gamma[inds[r:], None] * W2
    bn = torch.zeros(D)
    bn[inds[r:]] = gamma[inds[r:]] * b2
    Wnat, WnatT, bnat = dev(Wn, torch.bfloat16), dev(Wn.t().contiguous(), torch.bfloat16), dev(bn)
    ops.pack_proj_rows(dev(W1), dev(b1), dev(inds.int()), dev(gamma), Wnat, WnatT, bnat)
    full = (gamma[:, None] * W).to(torch.bfloat16)
    assert torch.equal(Wnat.cpu(), full)
    assert torch.equal(WnatT.cpu(), full.t())
    assert rel_err(bnat.cpu(), gamma * b) < 1e-6
    x, xd = bf(rnd(M, D, seed=47))
    y = ops.gemm_nt(dev(x), Wnat, bnat, out_dtype=torch.float32)
    ref = gamma.double() * O.apla_proj_fwd(xd, W1.to(torch.bfloat16).double() * 1.0, b1.double(), W2.double(), b2.double(), inds)
    assert rel_err(y.cpu(), ref) < 1e-2  # gamma*W rounded once to bf16 vs W rounded then scaled


# ------------------------------------------------------------------------------------------- optimizer
def test_adamw_clip_matches_oracle(ops):
    n = 5000
    p0, g0 = rnd(n, seed=51), rnd(n, scale=0.05, seed=52)
    mask = (torch.arange(n) % 3 != 0)
    p = {"w": p0[mask].clone(), "b.bias": p0[~mask].clone()}
    g = {"w": g0[mask].clone() * 0.5, "b.bias": g0[~mask].clone() * 0.5}
    p["w"] = p["w"].reshape(-1, 1)  # ndim 2 -> decayed
    g["w"] = g["w"].reshape(-1, 1)
    state = {}
    P, G = dev(p0.clone()), dev(g0.clone())
    m, v = torch.zeros_like(P), torch.zeros_like(P)
    ws = torch.zeros(512, device="cuda")
    decay = dev(mask.to(torch.uint8))
    for step in (1, 2, 3):
        gn = O.clip_grad_norm(g, 1.0)
        O.adamw_step(p, g, state, lr=1e-2, wd=0.1)
        ops.adamw_step(P, G, m, v, decay, lr=1e-2, weight_decay=0.1, step=step, max_norm=1.0, grad_scale=0.5, norm_ws=ws)
        assert abs(float(ws[1]) - float(gn)) < 1e-5 * float(gn)
        got = P.cpu()
        assert rel_err(got[mask], p["w"].flatten()) < 2e-6 and rel_err(got[~mask], p["b.bias"]) < 2e-6
        g = {"w": (g0[mask] * 0.5).reshape(-1, 1).clone(), "b.bias": (g0[~mask] * 0.5).clone()}
        G.copy_(dev(g0))


# ------------------------------------------------------------------------------------------- front end / head
@pytest.mark.parametrize("S,patch,D", [(32, 16, 128), (28, 14, 128)])
def test_patch_embed_tokens(ops, S, patch, D):
    B = 3
    img = rnd(B, 3, S, S, seed=61)
    Wc, bc = rnd(D, 3, patch, patch, scale=0.05, seed=62), rnd(D, scale=0.1, seed=63)
    Np = (S // patch) ** 2
    cls, pos = rnd(1, 1, D, scale=0.1, seed=64), rnd(1, Np + 1, D, scale=0.1, seed=65)
    K = 3 * patch * patch
    Kp = (K + 63) // 64 * 64
    Wp = torch.zeros(D, Kp)
    Wp[:, :K] = Wc.reshape(D, K)
    cols = ops.patchify(dev(img), patch, Kp)
    patches = ops.gemm_nt(cols, dev(Wp, torch.bfloat16), dev(bc))
    tok = ops.assemble_tokens(patches, dev(cls.flatten()), dev(pos[0].contiguous()), B, Np)
    p = {"patch_embed.proj.weight": Wc.to(torch.bfloat16).double(), "patch_embed.proj.bias": bc.double(),
         "cls_token": cls.double(), "pos_embed": pos.double()}
    ref = O.embed_tokens(img.to(torch.bfloat16).double(), p, patch)
    assert rel_err(tok.cpu().reshape(B, Np + 1, D), ref) < BF16_OUT


def test_head_and_cross_entropy(ops):
    B, D, C = 37, 96, 1000
    xn, W, b = rnd(B, D, seed=71), rnd(C, D, scale=0.05, seed=72), rnd(C, scale=0.1, seed=73)
    labels = torch.randint(0, C, (B,), generator=torch.Generator().manual_seed(74))
    logits = ops.sgemm_small(dev(xn), dev(W), trans_b=True, bias=dev(b))
    ref_logits = xn.double() @ W.double().t() + b.double()
    assert rel_err(logits.cpu(), ref_logits) < 1e-5
    loss, dlogits, _ = ops.cross_entropy(logits, dev(labels.int()))
    rl, rdl = O.cross_entropy_fwd_bwd(ref_logits, labels)
    assert abs(float(loss) - float(rl)) < 1e-5 and rel_err(dlogits.cpu(), rdl) < 1e-5
    dW = ops.sgemm_small(dlogits, dev(xn), trans_a=True)
    dxn = ops.sgemm_small(dlogits, dev(W))
    db = ops.colsum(dlogits)
    assert rel_err(dW.cpu(), rdl.t() @ xn.double()) < 1e-5
    assert rel_err(dxn.cpu(), rdl @ W.double()) < 1e-5
    assert rel_err(db.cpu(), rdl.sum(0)) < 1e-5


@pytest.mark.parametrize("M,N", [(4879, 1024), (7, 260), (1, 4)])
def test_colsum_16bit_rows(ops, M, N):
    """apla_colsum_h16: fp32 column sums of a 16-bit matrix (the iBOT centre update, ibot_patch_loss.py:123-135), strided rows too."""
    x, xd = bf(rnd(M, N + 12, seed=131))
    X = dev(x)[:, :N]
    got = ops.colsum(X).cpu().double()
    assert rel_err(got, xd[:, :N].sum(0)) < 1e-5


# ------------------------------------------------------------------------------------------- GEMM schedule variants
@pytest.mark.parametrize("variant", [4, 9, 14, 15, 16, 1])
@pytest.mark.parametrize("M,N,K", [(25216, 768, 768), (1000, 512, 256), (333, 256, 128), (161, 256, 192)])
def test_gemm_variants_all_epilogues(ops, variant, M, N, K):
    """Every main-loop schedule (apla_gemm_nt_ex flags, pinned through ops.set_gemm_variant) must give the same results for every epilogue, including the
    persistent multi-tile-per-workgroup case (M = 25216) and row tails."""
    from apla_amd._lib import lib
    a, ad = bf(rnd(M, K, seed=81))
    w, wd = bf(rnd(N, K, scale=K ** -0.5, seed=82))
    bias = rnd(N, seed=83)
    base = ad @ wd.t() + bias.double()
    res32 = rnd(M, N, seed=84)
    g, gd = bf(rnd(M, N, seed=85))
    old = ops.set_gemm_variant(variant)
    try:
        A, W, Bv = dev(a), dev(w), dev(bias)
        assert rel_err(ops.gemm_nt(A, W, Bv).cpu(), base) < BF16_OUT
        assert rel_err(ops.gemm_nt(A, W, Bv, out_dtype=torch.float32).cpu(), base) < F32_OUT
        gp = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        h = ops.gemm_nt(A, W, Bv, epilogue=ops.EPI_GELU, aux_out=gp)
        assert rel_err(h.cpu(), O.gelu_fwd(base)) < BF16_OUT and rel_err(gp.cpu(), O.gelu_grad(base)) < BF16_OUT
        r = ops.gemm_nt(A, W, Bv, epilogue=ops.EPI_RESIDUAL, aux_in=dev(res32), out_dtype=torch.float32)
        assert rel_err(r.cpu(), base + res32.double()) < F32_OUT
        rb = ops.gemm_nt(A, W, Bv, epilogue=ops.EPI_RESIDUAL, aux_in=dev(res32, torch.bfloat16), out_dtype=torch.bfloat16)
        assert rel_err(rb.cpu(), base + res32.to(torch.bfloat16).double()) < BF16_OUT
        mlt = ops.gemm_nt(A, W, None, epilogue=ops.EPI_MUL, aux_in=dev(g))
        assert rel_err(mlt.cpu(), (ad @ wd.t()) * gd) < BF16_OUT
    finally:
        ops.set_gemm_variant(old)


@pytest.mark.parametrize("M,N,K", [(25216, 768, 768), (2500, 512, 256), (4001, 256, 128)])
def test_gemm_k_panel_images(ops, M, N, K):
    """Operands given as K-panel images (apla_gemm_nt_ex flags bits 16 / 17, ops.k_panels) give the row-major results bit for bit;
    the image itself is the documented permutation; a problem the ping-pong kernel does not cover refuses an image."""
    from apla_amd._lib import AplaHipError
    a, _ = bf(rnd(M, K, seed=91))
    w, _ = bf(rnd(N, K, scale=K ** -0.5, seed=92))
    bias = dev(rnd(N, seed=93))
    A, W = dev(a), dev(w)
    Wi, Ai = ops.k_panels(W), ops.k_panels(A)
    assert torch.equal(Wi.cpu(), w.view(N, K // 32, 32).permute(1, 0, 2).contiguous())
    assert torch.equal(ops.k_panels(A[:, : K]).cpu(), a.view(M, K // 32, 32).permute(1, 0, 2).contiguous())
    assert ops.gemm_panel_ok(M, N, K)
    ref = ops.gemm_nt(A, W, bias)
    assert torch.equal(ops.gemm_nt(A, Wi, bias), ref)
    assert torch.equal(ops.gemm_nt(Ai, Wi, bias), ref)
    assert torch.equal(ops.gemm_nt(Ai, W, bias), ref)
    ref32 = ops.gemm_nt(A, W, bias, out_dtype=torch.float32)
    assert torch.equal(ops.gemm_nt(A, Wi, bias, out_dtype=torch.float32), ref32)
    g0, g1 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    h0 = ops.gemm_nt(A, W, bias, epilogue=ops.EPI_GELU, aux_out=g0)
    h1 = ops.gemm_nt(A, Wi, bias, epilogue=ops.EPI_GELU, aux_out=g1)
    assert torch.equal(h0, h1) and torch.equal(g0, g1)
    # the output written as an image (GELU / MUL epilogues of the 4-wave kernel) = k_panels of the row-major output, also when it
    # feeds the next GEMM as its A operand
    hi_ = torch.empty(N // 32, M, 32, device="cuda", dtype=torch.bfloat16)
    g2 = torch.empty_like(g0)
    ops.gemm_nt(A, W, bias, epilogue=ops.EPI_GELU, aux_out=g2, out=hi_)
    assert torch.equal(hi_, ops.k_panels(h0)) and torch.equal(g2, g0)
    ops.gemm_nt(A, W, bias, epilogue=ops.EPI_GELU_FWD, out=hi_.zero_())
    assert torch.equal(hi_, ops.k_panels(h0))
    mref = ops.gemm_nt(A, W, None, epilogue=ops.EPI_MUL, aux_in=g0)
    ops.gemm_nt(A, W, None, epilogue=ops.EPI_MUL, aux_in=g0, out=hi_.zero_())
    assert torch.equal(hi_, ops.k_panels(mref))
    # gelu' as an image between the two epilogues that own it
    gi = torch.empty(N // 32, M, 32, device="cuda", dtype=torch.bfloat16)
    h3 = ops.gemm_nt(A, W, bias, epilogue=ops.EPI_GELU, aux_out=gi)
    assert torch.equal(h3, h0) and torch.equal(gi, ops.k_panels(g0))
    assert torch.equal(ops.gemm_nt(A, W, None, epilogue=ops.EPI_MUL, aux_in=gi), mref)
    ops.gemm_nt(A, W, None, epilogue=ops.EPI_MUL, aux_in=gi, out=hi_.zero_())
    assert torch.equal(hi_, ops.k_panels(mref))
    w2, _ = bf(rnd(256, N, scale=N ** -0.5, seed=94))
    W2 = dev(w2)
    assert torch.equal(ops.gemm_nt(hi_, ops.k_panels(W2)), ops.gemm_nt(mref, W2))
    # the SwiGLU epilogues (ViT-g, vit.py:131-149): C is N/2 resp. 2N wide, written as an image just the same
    x12 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    hs = ops.gemm_nt(A, W, bias, epilogue=ops.EPI_SWIGLU, aux_out=x12)
    hs_i = torch.empty(N // 64, M, 32, device="cuda", dtype=torch.bfloat16)
    x12b = torch.empty_like(x12)
    ops.gemm_nt(A, W, bias, epilogue=ops.EPI_SWIGLU, aux_out=x12b, out=hs_i)
    assert torch.equal(hs_i, ops.k_panels(hs)) and torch.equal(x12b, x12)
    x12w = dev(bf(rnd(M, 2 * N, seed=95))[0])
    db = ops.gemm_nt(A, W, None, epilogue=ops.EPI_SWIGLU_BWD, aux_in=x12w)
    db_i = torch.empty(2 * N // 32, M, 32, device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt(A, W, None, epilogue=ops.EPI_SWIGLU_BWD, aux_in=x12w, out=db_i)
    assert torch.equal(db_i, ops.k_panels(db))
    with pytest.raises(AplaHipError):
        ops.gemm_nt(A, W, bias, out=hi_)          # plain STORE has no output image
    # not covered: an operand epilogue, N not a multiple of 256, fewer than four 32-wide K-steps
    assert not ops.gemm_panel_ok(M, 128, K) and not ops.gemm_panel_ok(M, N, K, ops.EPI_MUL) and not ops.gemm_panel_ok(1000, N, K)
    assert ops.gemm_panel_ok(50000, N, K, ops.EPI_GELU) and not ops.gemm_panel_ok(M, N, K, ops.EPI_GELU)
    assert ops.gemm_out_image_ok(50000, N, K, ops.EPI_GELU) and ops.gemm_out_image_ok(M, N, K, ops.EPI_GELU)
    with pytest.raises(AplaHipError):
        ops.gemm_nt(A, Wi, None, epilogue=ops.EPI_MUL, aux_in=g0)
    with pytest.raises(AplaHipError):
        ops.gemm_nt(dev(a[:, :64].contiguous()), ops.k_panels(dev(w[:, :64].contiguous())), bias)   # K = 64 < 128


@pytest.mark.parametrize("reserve", [8, 40])
def test_gemm_with_reserved_cus_is_bitwise_the_same(ops, reserve):
    """The data-parallel step's GEMM launches leave CUs free for the RCCL kernels of the overlapped all-reduce (apla_gemm_nt_ex flags
    bits 20-27, ops.reserved_cus): fewer persistent workgroups walk the same tiles — both kernels, multi-round tile counts."""
    M, K = 25216, 768
    a, _ = bf(rnd(M, K, seed=101))
    A = dev(a)
    for N, epi in ((2304, ops.EPI_STORE), (768, ops.EPI_STORE), (3072, ops.EPI_GELU), (3072, ops.EPI_MUL)):
        w, _ = bf(rnd(N, K, scale=K ** -0.5, seed=102))
        W, bias = dev(w), dev(rnd(N, seed=103))
        kw = {}
        if epi == ops.EPI_GELU:
            kw = dict(aux_out=torch.empty(M, N, device="cuda", dtype=torch.bfloat16))
        elif epi == ops.EPI_MUL:
            kw = dict(aux_in=dev(bf(rnd(M, N, seed=104))[0]))
        ref = ops.gemm_nt(A, W, bias, epilogue=epi, **kw).clone()
        aux_ref = kw["aux_out"].clone() if epi == ops.EPI_GELU else None
        with ops.reserved_cus(reserve):
            got = ops.gemm_nt(A, W, bias, epilogue=epi, **kw)
        assert torch.equal(got, ref), (N, epi)
        if aux_ref is not None:
            assert torch.equal(kw["aux_out"], aux_ref)


@pytest.mark.parametrize("M,N,K", [(25216, 768, 768), (9000, 512, 256), (4001, 256, 128), (161, 256, 192)])
def test_gemm_wide_4wave_kernel(ops, M, N, K):
    """gemm_w4.hip (schedule 16: 160 x 256 tiles, 80 x 128 wave tiles, two workgroups per CU): STORE, GELU and GELU_FWD, row-major or
    K-panel-image operands, row-major or image outputs, with CUs reserved, on both ring depths — the bits of the other schedules
    (same K order, same epilogue arithmetic); the automatic rule picks it for short-K STORE problems and the forward-only GELU."""
    a, ad = bf(rnd(M, K, seed=111))
    w, wd = bf(rnd(N, K, scale=K ** -0.5, seed=112))
    bias = dev(rnd(N, seed=113))
    A, W = dev(a), dev(w)
    base = ad @ wd.t() + bias.cpu().double()
    old = ops.set_gemm_variant(15)
    try:
        ref = ops.gemm_nt(A, W, bias).clone()
        g_ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        h_ref = ops.gemm_nt(A, W, bias, epilogue=ops.EPI_GELU, aux_out=g_ref).clone()
        ops.set_gemm_variant(16)
        assert ops.gemm_kernel_name(M, N, K, ops.EPI_STORE).startswith("gemm_w4_kernel<STORE")
        assert rel_err(ref.cpu(), base) < BF16_OUT
        for Ai, Wi in ((A, W), (A, ops.k_panels(W)), (ops.k_panels(A), ops.k_panels(W))):
            for exp in (0, 4, 7):      # default, three-stage ring, no priority
                ops._GEMM_EXP = exp
                assert torch.equal(ops.gemm_nt(Ai, Wi, bias), ref), exp
                g = torch.zeros_like(g_ref)
                assert torch.equal(ops.gemm_nt(Ai, Wi, bias, epilogue=ops.EPI_GELU, aux_out=g), h_ref) and torch.equal(g, g_ref), exp
                assert torch.equal(ops.gemm_nt(Ai, Wi, bias, epilogue=ops.EPI_GELU_FWD), h_ref), exp
            ops._GEMM_EXP = 0
            hi_ = torch.zeros(N // 32, M, 32, device="cuda", dtype=torch.bfloat16)
            gi = torch.zeros(N // 32, M, 32, device="cuda", dtype=torch.bfloat16)
            ops.gemm_nt(Ai, Wi, bias, epilogue=ops.EPI_GELU, aux_out=gi, out=hi_)
            assert torch.equal(hi_, ops.k_panels(h_ref)) and torch.equal(gi, ops.k_panels(g_ref))
            ops.gemm_nt(Ai, Wi, bias, epilogue=ops.EPI_GELU_FWD, out=hi_.zero_())
            assert torch.equal(hi_, ops.k_panels(h_ref))
        with ops.reserved_cus(8):
            assert torch.equal(ops.gemm_nt(A, W, bias), ref)
        assert torch.equal(ops.gemm_nt(A, W, None), ops.gemm_nt(A, W, torch.zeros_like(bias)))   # no bias piece in the stream
    finally:
        ops._GEMM_EXP = 0
        ops.set_gemm_variant(old)
    if M >= 8192:
        assert ops.gemm_kernel_name(M, N, K, ops.EPI_STORE).startswith("gemm_w4_kernel<STORE" if K <= 1024 else "gemm_pp2_kernel")
        assert ops.gemm_kernel_name(M, N, K, ops.EPI_GELU_FWD).startswith("gemm_tp_kernel<GELU_FWD" if K >= 704 else "gemm_w4_kernel<GELU_FWD")
        assert ops.gemm_kernel_name(M, N, 3072, ops.EPI_STORE).startswith("gemm_pp2_kernel<STORE")
        assert ops.gemm_kernel_name(M, N, K, ops.EPI_GELU).startswith("gemm_persist_kernel<GELU")


@pytest.mark.parametrize("M,N,K", [(25216, 768, 768), (25216, 3072, 768), (9000, 512, 1024), (4001, 256, 832), (161, 256, 768), (1, 256, 3072),
                                   (4001, 512, 704)])
def test_gemm_tile_alternating_kernel(ops, M, N, K):
    """gemm_tp.hip (schedule 17: 160 x 256 tiles, ONE 8-wave workgroup per CU whose two wave groups swap the compute and the service
    role per tile — the service group issues every LDS-DMA and runs the previous tile's epilogue in slices under the partner's MFMAs;
    accumulators in the AGPR half of the register file): STORE and GELU_FWD with row-major outputs, GELU, GELU_FWD and MUL with image outputs,
    row-major or K-panel-image operands, with and without bias, with CUs reserved — the bits of the other schedules (same K order, same
    epilogue arithmetic); one tile per workgroup, several tiles per workgroup (role swaps), row tails, a single row."""
    a, ad = bf(rnd(M, K, seed=141))
    w, wd = bf(rnd(N, K, scale=K ** -0.5, seed=142))
    bias = dev(rnd(N, seed=143))
    A, W = dev(a), dev(w)
    base = ad @ wd.t() + bias.cpu().double()
    old = ops.set_gemm_variant(15)
    try:
        ref = ops.gemm_nt(A, W, bias).clone()
        ref_nb = ops.gemm_nt(A, W, None).clone()
        g_ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        h_ref = ops.gemm_nt(A, W, bias, epilogue=ops.EPI_GELU, aux_out=g_ref).clone()
        h_nb = ops.gemm_nt(A, W, None, epilogue=ops.EPI_GELU_FWD).clone()
        g_in = dev(bf(rnd(M, N, seed=144))[0])
        m_ref = ops.gemm_nt(A, W, bias, epilogue=ops.EPI_MUL, aux_in=g_in).clone()
        m_nb = ops.gemm_nt(A, W, None, epilogue=ops.EPI_MUL, aux_in=g_in).clone()
        assert rel_err(ref.cpu(), base) < BF16_OUT
        ops.set_gemm_variant(17)
        assert ops.gemm_kernel_name(M, N, K, ops.EPI_GELU, out_image=True, aux_image=True).startswith("gemm_tp_kernel<GELU")
        assert ops.gemm_kernel_name(M, N, K, ops.EPI_STORE).startswith("gemm_tp_kernel<STORE")
        assert not ops.gemm_kernel_name(M, N, K, ops.EPI_GELU).startswith("gemm_tp_kernel")      # row-major two-output GELU: not instantiated
        hi_ = torch.zeros(N // 32, M, 32, device="cuda", dtype=torch.bfloat16)
        gi = torch.zeros(N // 32, M, 32, device="cuda", dtype=torch.bfloat16)
        g_img = ops.k_panels(g_in)
        mul_tp = K // 32 >= 23          # the MUL form needs two K-steps of load lead beside its twenty half-slices
        assert ops.gemm_kernel_name(M, N, K, ops.EPI_MUL, out_image=True, aux_image=True).startswith("gemm_tp_kernel<MUL") == mul_tp
        for Ai, Wi in ((A, W), (A, ops.k_panels(W)), (ops.k_panels(A), ops.k_panels(W))):
            for rep in range(2):    # (twice: a race would not repeat itself)
                assert torch.equal(ops.gemm_nt(Ai, Wi, bias), ref), rep
                assert torch.equal(ops.gemm_nt(Ai, Wi, bias, epilogue=ops.EPI_GELU_FWD), h_ref), rep
                ops.gemm_nt(Ai, Wi, bias, epilogue=ops.EPI_GELU, aux_out=gi.zero_(), out=hi_.zero_())
                assert torch.equal(hi_, ops.k_panels(h_ref)) and torch.equal(gi, ops.k_panels(g_ref)), rep
                ops.gemm_nt(Ai, Wi, bias, epilogue=ops.EPI_GELU_FWD, out=hi_.zero_())
                assert torch.equal(hi_, ops.k_panels(h_ref)), rep
                if mul_tp:      # round 6: (A W^T + b) * X with X and the output as images — the second operand read two half-slices ahead
                    ops.gemm_nt(Ai, Wi, bias, epilogue=ops.EPI_MUL, aux_in=g_img, out=hi_.zero_())
                    assert torch.equal(hi_, ops.k_panels(m_ref)), rep
        with ops.reserved_cus(8):
            assert torch.equal(ops.gemm_nt(A, W, bias), ref)
        with ops.reserved_cus(150):      # few workgroups, many tiles each
            ops.gemm_nt(A, W, bias, epilogue=ops.EPI_GELU, aux_out=gi.zero_(), out=hi_.zero_())
            assert torch.equal(hi_, ops.k_panels(h_ref)) and torch.equal(gi, ops.k_panels(g_ref))
            assert torch.equal(ops.gemm_nt(A, W, bias), ref)
            if mul_tp:
                ops.gemm_nt(A, W, bias, epilogue=ops.EPI_MUL, aux_in=g_img, out=hi_.zero_())
                assert torch.equal(hi_, ops.k_panels(m_ref))
        if mul_tp:
            ops.gemm_nt(A, W, None, epilogue=ops.EPI_MUL, aux_in=g_img, out=hi_.zero_())
            assert torch.equal(hi_, ops.k_panels(m_nb))
        assert torch.equal(ops.gemm_nt(A, W, None), ref_nb)   # no bias piece in the stream
        ops.gemm_nt(A, W, None, epilogue=ops.EPI_GELU_FWD, out=hi_.zero_())
        assert torch.equal(hi_, ops.k_panels(h_nb))
    finally:
        ops._GEMM_EXP = 0
        ops.set_gemm_variant(old)


@pytest.mark.parametrize("M,N,K", [(25216, 3072, 768), (1000, 256, 2304), (130, 512, 256), (1, 128, 128), (90000, 128, 192)])
def test_gemm_loader_wave_kernel(ops, M, N, K):
    """gemm_lw.hip (schedule 18, round 6: the 4-wave persistent kernel with its LDS-DMA issue moved to one or two LOADER waves; measured
    1.7-1.9 x slower than the product kernel — profiles/r06_a_lw_gemm.md — and kept as a selectable schedule for that record): STORE,
    GELU_FWD, GELU and MUL with row-major and image outputs, with and without bias, one and two loader waves, CUs reserved — the bits
    of schedule 15 (same tiles, same K order, same epilogue code); row tails, one row, many tiles per workgroup."""
    a, ad = bf(rnd(M, K, seed=161))
    w, wd = bf(rnd(N, K, scale=K ** -0.5, seed=162))
    bias = dev(rnd(N, seed=163))
    A, W = dev(a), dev(w)
    g_in = dev(bf(rnd(M, N, seed=164))[0])
    old = ops.set_gemm_variant(15)
    try:
        ref = ops.gemm_nt(A, W, bias).clone()
        ref_nb = ops.gemm_nt(A, W, None).clone()
        g_ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        h_ref = ops.gemm_nt(A, W, bias, epilogue=ops.EPI_GELU, aux_out=g_ref).clone()
        m_ref = ops.gemm_nt(A, W, bias, epilogue=ops.EPI_MUL, aux_in=g_in).clone()
        if M <= 30000:
            assert rel_err(ref.cpu(), ad @ wd.t() + bias.cpu().double()) < BF16_OUT
        ops.set_gemm_variant(18)
        assert ops.gemm_kernel_name(M, N, K, ops.EPI_GELU).startswith("gemm_lw_kernel<GELU")
        hi_ = torch.zeros(N // 32, M, 32, device="cuda", dtype=torch.bfloat16)
        gi = torch.zeros(N // 32, M, 32, device="cuda", dtype=torch.bfloat16)
        for loaders in (0, 1):          # GemmParams::exp bit 0: two loader waves per workgroup
            ops._GEMM_EXP = loaders
            for rep in range(2):        # (twice: a race would not repeat itself)
                assert torch.equal(ops.gemm_nt(A, W, bias), ref), (loaders, rep)
                assert torch.equal(ops.gemm_nt(A, W, bias, epilogue=ops.EPI_GELU_FWD), h_ref), (loaders, rep)
                g2 = torch.zeros_like(g_ref)
                assert torch.equal(ops.gemm_nt(A, W, bias, epilogue=ops.EPI_GELU, aux_out=g2), h_ref) and torch.equal(g2, g_ref), (loaders, rep)
                assert torch.equal(ops.gemm_nt(A, W, bias, epilogue=ops.EPI_MUL, aux_in=g_in), m_ref), (loaders, rep)
                ops.gemm_nt(A, W, bias, epilogue=ops.EPI_GELU, aux_out=gi.zero_(), out=hi_.zero_())
                assert torch.equal(hi_, ops.k_panels(h_ref)) and torch.equal(gi, ops.k_panels(g_ref)), (loaders, rep)
            assert torch.equal(ops.gemm_nt(A, W, None), ref_nb)   # no bias piece in the stream
            with ops.reserved_cus(150):      # few workgroups, many tiles each
                assert torch.equal(ops.gemm_nt(A, W, bias), ref)
    finally:
        ops._GEMM_EXP = 0
        ops.set_gemm_variant(old)


@pytest.mark.parametrize("M,N,K", [(27400, 768, 1536), (27400, 768, 768), (1000, 256, 2304), (130, 512, 256), (257, 256, 3072)])
def test_gemm_small_row_tiles(ops, M, N, K):
    """The 256-row tile of the ping-pong kernel and the 128-row tile of the wide 4-wave kernel (bf16 STORE; chosen where the tile count
    falls just past a multiple of the CU count — the first two shapes: 86 x 3 tiles of 320 rows = 258 -> two rounds; the self-supervised
    step's N = 768 launches): forced on and off on every shape (row tails, fewer rows than a tile), row-major and image operands, with
    and without bias, CUs reserved — the bits of the tall tiles and of the 4-wave kernel."""
    a, ad = bf(rnd(M, K, seed=151))
    w, wd = bf(rnd(N, K, scale=K ** -0.5, seed=152))
    bias = dev(rnd(N, seed=153))
    A, W = dev(a), dev(w)
    old = ops.set_gemm_variant(15)
    try:
        ref, ref_nb = ops.gemm_nt(A, W, bias).clone(), ops.gemm_nt(A, W, None).clone()
        assert rel_err(ref.cpu(), ad @ wd.t() + bias.cpu().double()) < BF16_OUT
        for variant, small in ((9, "gemm_pp2_kernel<STORE,bf16,256 rows>"), (16, "gemm_w4_kernel<STORE,bf16,128 rows>")):
            ops.set_gemm_variant(variant)
            for exp in (5, 6):
                ops._GEMM_EXP = exp
                assert (ops.gemm_kernel_name(M, N, K) == small) == (exp == 5)
                for Ai, Wi in ((A, W), (A, ops.k_panels(W)), (ops.k_panels(A), ops.k_panels(W))):
                    for rep in range(2):
                        assert torch.equal(ops.gemm_nt(Ai, Wi, bias), ref), (variant, exp, rep)
                assert torch.equal(ops.gemm_nt(A, W, None), ref_nb)
                with ops.reserved_cus(150):
                    assert torch.equal(ops.gemm_nt(A, W, bias), ref)
            ops._GEMM_EXP = 0
        ops.set_gemm_variant(0)
        if M == 27400:   # the automatic rule takes the small tile here, on whichever of the two kernels it picks
            assert ops.gemm_kernel_name(M, N, K) == ("gemm_w4_kernel<STORE,bf16,128 rows>" if K <= 1024 else "gemm_pp2_kernel<STORE,bf16,256 rows>")
            assert ops.gemm_kernel_name(25216, 768, 3072) == "gemm_pp2_kernel<STORE,bf16>" and ops.gemm_kernel_name(25216, 2304, 768) == "gemm_w4_kernel<STORE,bf16>"
            assert torch.equal(ops.gemm_nt(A, W, bias), ref)
    finally:
        ops._GEMM_EXP = 0
        ops.set_gemm_variant(old)


@pytest.mark.parametrize("M,N,K", [(5581, 256, 8192), (300, 512, 4096), (161, 256, 256)])
def test_gemm_split_k(ops, M, N, K):
    """apla_gemm_nt_splitk (few tiles, long K: the prototype layer's input gradient of the self-supervised step): the K axis cut into
    parts on the wide 4-wave kernel, fp32 partials summed in a fixed order — equal to the one-pass kernels up to fp32 summation order,
    and the same bits on every call."""
    a, ad = bf(rnd(M, K, seed=121))
    w, wd = bf(rnd(N, K, scale=K ** -0.5, seed=122))
    bias = rnd(N, seed=123)
    A, W, Bv = dev(a), dev(w), dev(bias)
    base = ad @ wd.t()
    o32 = ops.gemm_nt_splitk(A, W, out_dtype=torch.float32)
    assert rel_err(o32.cpu(), base) < F32_OUT
    o16 = ops.gemm_nt_splitk(A, W, Bv)
    assert o16.dtype == torch.bfloat16 and rel_err(o16.cpu(), base + bias.double()) < BF16_OUT
    assert torch.equal(ops.gemm_nt_splitk(A, W, out_dtype=torch.float32), o32)
    assert rel_err(o32.cpu(), ops.gemm_nt(A, W, out_dtype=torch.float32).cpu().double()) < 1e-5
    assert ops.gemm_splitk_wanted(5581, 256, 65536) and not ops.gemm_splitk_wanted(25216, 768, 3072)


def test_gemm_gelu_images_on_the_pingpong_kernel(ops):
    """Above 40 000 rows the two-output GELU runs on the ping-pong kernel (the student fc1 of the self-supervised step, config 3's
    fc1): its line-store epilogue writes h and gelu' as K-panel images too, with row-major or image operands — same bits."""
    M, N, K = 40100, 512, 256
    a, _ = bf(rnd(M, K, seed=97))
    w, _ = bf(rnd(N, K, scale=K ** -0.5, seed=98))
    bias = dev(rnd(N, seed=99))
    A, W = dev(a), dev(w)
    assert ops.gemm_kernel_name(M, N, K, ops.EPI_GELU).startswith("gemm_pp2_kernel<GELU")
    assert ops.gemm_kernel_name(M, N, K, ops.EPI_GELU, out_image=True, aux_image=True).startswith("gemm_pp2_kernel<GELU")
    g0 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    h0 = ops.gemm_nt(A, W, bias, epilogue=ops.EPI_GELU, aux_out=g0)
    for Ai, Wi in ((A, W), (A, ops.k_panels(W)), (ops.k_panels(A), ops.k_panels(W))):
        hi_ = torch.zeros(N // 32, M, 32, device="cuda", dtype=torch.bfloat16)
        gi = torch.zeros(N // 32, M, 32, device="cuda", dtype=torch.bfloat16)
        ops.gemm_nt(Ai, Wi, bias, epilogue=ops.EPI_GELU, aux_out=gi, out=hi_)
        assert torch.equal(hi_, ops.k_panels(h0)) and torch.equal(gi, ops.k_panels(g0))
        g1 = torch.zeros_like(g0)                 # image output, row-major gelu'
        ops.gemm_nt(Ai, Wi, bias, epilogue=ops.EPI_GELU, aux_out=g1, out=hi_.zero_())
        assert torch.equal(hi_, ops.k_panels(h0)) and torch.equal(g1, g0)
    # the MUL epilogue (4-wave kernel) reads that gelu' image
    m0 = ops.gemm_nt(A, W, None, epilogue=ops.EPI_MUL, aux_in=g0)
    assert torch.equal(ops.gemm_nt(A, W, None, epilogue=ops.EPI_MUL, aux_in=gi), m0)


@pytest.mark.parametrize("B,N,H", [(3, 197, 2), (2, 5, 1), (1, 300, 3)])
def test_attention_bwd_cls_only(ops, B, N, H):
    """dO non-zero only at token 0 (last ViT block): the rank-1 kernel equals the general backward."""
    D = 64 * H
    scale = 64 ** -0.5
    qkv, qkvd = bf(rnd(B, N, 3 * D, seed=91))
    o, lse = ops.attn_fwd(dev(qkv).reshape(B * N, 3 * D), B, N, H, scale)
    do_cls, dod = bf(rnd(B, D, seed=92))
    do_full = torch.zeros(B, N, D, dtype=torch.float64)
    do_full[:, 0] = dod
    ref = O.attention_bwd(do_full, qkvd, o.cpu().double().reshape(B, N, D), lse.cpu().double(), H, scale)
    got = ops.attn_bwd_cls(dev(qkv).reshape(B * N, 3 * D), o, dev(do_cls), lse, B, N, H, scale).cpu().reshape(B, N, 3 * D)
    for i, nm in enumerate(("dq", "dk", "dv")):
        assert rel_err(got[..., i * D:(i + 1) * D], ref[..., i * D:(i + 1) * D]) < 1e-2, nm
    assert float(got[:, 1:, :D].abs().max()) == 0.0 if N > 1 else True


@pytest.mark.parametrize("L,D,r", [(3, 256, 64), (2, 768, 192), (2, 1536, 512), (1, 128, 8)])
def test_pack_proj_rows_batched_all_layouts(ops, L, D, r):
    """The per-step re-scatter of the trainable projection rows (appla_attn.py:62-79 folded into the weight layout): natural-order
    weight, its transpose, both K-panel images and the bias, for every block in one launch — trainable rows replaced (LayerScale
    folded), frozen entries untouched.  (D % 32 == 0: the panel-oriented kernel; D = 128 with one k slice.)"""
    g = torch.Generator().manual_seed(5)
    stride = r * D + r
    flat = torch.randn(L * stride, generator=g)
    inds = torch.stack([torch.randperm(D, generator=g) for _ in range(L)]).int()
    gamma = 0.5 + torch.rand(L, D, generator=g)
    W0 = torch.randn(L, D, D, generator=g).to(torch.bfloat16)          # frozen content before the call
    b0 = torch.randn(L, D, generator=g)
    Wnat, WnatT, bnat = dev(W0.clone()), dev(W0.transpose(1, 2).contiguous()), dev(b0.clone())
    Wp = torch.stack([ops.k_panels(Wnat[l]) for l in range(L)]).contiguous()
    WTp = torch.stack([ops.k_panels(WnatT[l]) for l in range(L)]).contiguous()
    ops.pack_proj_rows_batched(dev(flat), stride, dev(inds), dev(gamma), Wnat, WnatT, bnat, r, Wp, WTp)
    ref, bref = W0.clone().float(), b0.clone()
    for l in range(L):
        W1 = flat[l * stride:l * stride + r * D].view(r, D)
        b1 = flat[l * stride + r * D:(l + 1) * stride]
        rows = inds[l, :r].long()
        ref[l, rows] = (gamma[l, rows, None] * W1).to(torch.bfloat16).float()
        bref[l, rows] = gamma[l, rows] * b1
    assert torch.equal(Wnat.cpu().float(), ref) and torch.equal(WnatT.cpu().float(), ref.transpose(1, 2))
    assert torch.allclose(bnat.cpu(), bref, rtol=1e-6, atol=0)
    for l in range(L):
        assert torch.equal(Wp[l], ops.k_panels(Wnat[l])) and torch.equal(WTp[l], ops.k_panels(WnatT[l]))


@pytest.mark.parametrize("B,N,H", [(3, 197, 2), (2, 5, 1), (1, 1370, 3), (2, 257, 4)])
def test_attention_fwd_cls_only(ops, B, N, H):
    """The CLS-query forward of the last block (one query per head against all keys) equals row 0 of the full attention."""
    D = 64 * H
    scale = 64 ** -0.5
    qkv, qkvd = bf(rnd(B, N, 3 * D, seed=96))
    o_ref, lse_ref = O.attention_fwd(qkvd, H, scale)[:2]
    o = torch.zeros(B * N, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(B, H, N, device="cuda")
    ops.attn_fwd_cls(dev(qkv).reshape(B * N, 3 * D), B, N, H, scale, o=o, lse=lse)
    assert rel_err(o.cpu().reshape(B, N, D)[:, 0], o_ref[:, 0]) < BF16_OUT
    assert rel_err(lse.cpu()[:, :, 0], lse_ref[:, :, 0]) < 1e-5
    assert float(o.cpu().reshape(B, N, D)[:, 1:].abs().max()) == 0.0 if N > 1 else True


# ------------------------------------------------------------------------------------------- dynamic loss scaling
def test_adamw_dynamic_loss_scaling_matches_gradscaler_semantics(ops):
    """apla_adamw_step_dynamic == scaler.unscale_ + clip_grad_norm_ + scaler.step(AdamW) + scaler.update() of
    torch.cuda.amp.GradScaler (defaults/trainer.py:129-138), replayed on the CPU with torch.optim.AdamW: finite steps
    update with bias corrections counted over steps actually taken, an overflowing step is skipped and halves the scale,
    `growth_interval` consecutive finite steps double it."""
    n, world = 5000, 2
    g = torch.Generator().manual_seed(7)
    p0 = torch.randn(n, generator=g)
    decay = (torch.rand(n, generator=g) < 0.7)
    pr = torch.nn.Parameter(p0[decay].clone())       # decayed group
    pn = torch.nn.Parameter(p0[~decay].clone())      # not decayed
    opt = torch.optim.AdamW([{"params": [pr]}, {"params": [pn], "weight_decay": 0.0}], lr=1e-2, weight_decay=0.1)
    scale, tracker, interval = 1024.0, 0, 3
    params = p0.clone().cuda()
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    ws = torch.zeros(512, device="cuda")
    st = ops.new_scaler_state("cuda", init_scale=scale)
    # ... and the real class beside the replay: a torch.amp.GradScaler driving a one-tensor optimizer through the same overflow pattern
    twin = torch.nn.Parameter(torch.zeros(8, device="cuda"))
    twin_opt = torch.optim.SGD([twin], lr=0.0)
    tsc = torch.amp.GradScaler("cuda", init_scale=scale, growth_factor=2.0, backoff_factor=0.5, growth_interval=interval)
    for it in range(9):
        true_grad = torch.randn(n, generator=g) * (3.0 if it % 2 else 0.05)     # some steps clip, some do not
        scaled = true_grad * scale * world                                       # what backward + SUM all-reduce leave behind
        if it in (2, 6):
            scaled[17] = float("inf") if it == 2 else float("nan")
        grads = scaled.clone().cuda()
        assert float(tsc.scale(torch.ones((), device="cuda"))) == scale
        twin.grad = torch.full((8,), float("inf") if it == 2 else (float("nan") if it == 6 else 1.0), device="cuda")
        tsc.unscale_(twin_opt)
        tsc.step(twin_opt)
        tsc.update()
        ops.adamw_step_dynamic(params, grads, m, v, decay.to(torch.uint8).cuda(), st, it & 1, lr=1e-2, weight_decay=0.1,
                               max_norm=1.0, grad_scale=1.0 / world, growth_interval=interval, norm_ws=ws)
        # reference
        unscaled = scaled / (scale * world)
        if torch.isfinite(unscaled).all():
            pr.grad, pn.grad = unscaled[decay].clone(), unscaled[~decay].clone()
            gn = torch.nn.utils.clip_grad_norm_([pr, pn], 1.0)
            opt.step()
            tracker += 1
            if tracker == interval:
                scale, tracker = scale * 2.0, 0
            assert abs(float(ws[1]) - float(gn)) < 1e-4 * float(gn)
            assert float(st[7]) == 0.0
        else:
            scale, tracker = scale * 0.5, 0
            assert float(st[7]) == 1.0
        ref = torch.empty(n)
        ref[decay], ref[~decay] = pr.detach(), pn.detach()
        assert rel_err(params.cpu(), ref) < 1e-5, it
        assert float(st[6]) == scale and float(st[3 * ((it + 1) & 1)]) == scale and float(st[3 * ((it + 1) & 1) + 1]) == tracker
        assert float(tsc.get_scale()) == scale and int(tsc.state_dict()["_growth_tracker"]) == tracker     # torch's own bookkeeping agrees


# ------------------------------------------------------------------------------------------- C-ABI error conventions
def test_cabi_error_codes_and_messages(ops):
    """SURVEY §8b error conventions at the raw C-ABI: 0 OK, -22 (EINVAL) bad shape / alignment / null pointers with a
    message in apla_last_error(), and the other build's dtype code is refused — nothing is launched on a bad call."""
    from apla_amd._lib import APLA_BF16, APLA_F16, APLA_F32, lib
    L = lib()
    a = torch.zeros(256, 128, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(128, 128, device="cuda", dtype=torch.bfloat16)
    out = torch.zeros(256, 128, device="cuda", dtype=torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream

    def gemm(M=256, N=128, K=128, aptr=None, dtype=APLA_BF16, epi=0):
        return L.apla_gemm_nt(a.data_ptr() if aptr is None else aptr, 128, w.data_ptr(), 128, None, out.data_ptr(), 128,
                              M, N, K, epi, dtype, None, 0, None, 0, s)
    assert gemm() == 0
    for bad in (dict(N=100), dict(K=72), dict(M=0), dict(aptr=a.data_ptr() + 2), dict(dtype=APLA_F16), dict(epi=1)):
        rc = gemm(**bad)
        assert rc == (-38 if "dtype" in bad else -22), (bad, rc)   # -ENOSYS: unsupported dtype; -EINVAL: shapes / pointers
        assert len(L.apla_last_error()) > 0
    assert L.apla_operand_dtype() == APLA_BF16 and L.apla_version() >= 100
    # layernorm: feature size not a multiple of the vector width (D % 4), and the fp16 code on the bf16 build
    x = torch.zeros(8, 104, device="cuda")
    g = torch.ones(104, device="cuda")
    y = torch.zeros(8, 104, device="cuda", dtype=torch.bfloat16)
    st = torch.zeros(8, device="cuda")
    rc = L.apla_layernorm_fwd(x.data_ptr(), APLA_F32, 104, g.data_ptr(), g.data_ptr(), y.data_ptr(), APLA_BF16, 104,
                              st.data_ptr(), st.data_ptr(), 8, 102, 1e-6, None, 0, None, 0, s)
    assert rc == -22 and b"D%4" in L.apla_last_error()
    x = torch.zeros(8, 128, device="cuda")
    g = torch.ones(128, device="cuda")
    y = torch.zeros(8, 128, device="cuda", dtype=torch.bfloat16)
    rc = L.apla_layernorm_fwd(x.data_ptr(), APLA_F32, 128, g.data_ptr(), g.data_ptr(), y.data_ptr(), APLA_F16, 128,
                              st.data_ptr(), st.data_ptr(), 8, 128, 1e-6, None, 0, None, 0, s)
    assert rc in (-22, -38)
    torch.cuda.synchronize()


def test_cross_entropy_soft_targets(ops):
    """Probability targets (timm Mixup / label smoothing under the reference's advanced_aug): loss and dlogits vs the fp64 oracle,
    and one-hot targets reproduce the class-id path."""
    B, C = 37, 1000
    logits = rnd(B, C, scale=3.0, seed=51)
    g = torch.Generator().manual_seed(52)
    y1, y2 = torch.randint(0, C, (B,), generator=g), torch.randint(0, C, (B,), generator=g)
    lam = torch.rand(B, 1, generator=g)
    tgt = torch.full((B, C), 0.1 / C)
    tgt += 0.9 * (lam * torch.nn.functional.one_hot(y1, C) + (1 - lam) * torch.nn.functional.one_hot(y2, C))   # mixup + smoothing
    loss_ref, dl_ref = O.cross_entropy_soft_fwd_bwd(logits.double(), tgt.double())
    loss, dl, _ = ops.cross_entropy(dev(logits), dev(tgt))
    assert abs(float(loss) - float(loss_ref)) < 1e-5 * abs(float(loss_ref)) and rel_err(dl.cpu(), dl_ref) < F32_OUT
    hard = torch.nn.functional.one_hot(y1, C).float()
    l1, d1, _ = ops.cross_entropy(dev(logits), dev(hard))
    l2, d2, _ = ops.cross_entropy(dev(logits), dev(y1.int()))
    assert abs(float(l1) - float(l2)) < 1e-6 * float(l2) and rel_err(d1.cpu(), d2.cpu().double()) < 1e-6


@pytest.mark.parametrize("B,N", [(128, 197), (96, 50), (40, 256), (64, 257), (48, 288), (30, 225), (23, 32), (60, 129), (90, 96),
                                 (512, 50), (300, 17), (120, 64), (70, 65), (400, 1)])
def test_attn_bwd_fused_short_sequence_kernel_equals_split_kernels_at_full_occupancy(ops, B, N):
    """The one-workgroup-per-head backward (N <= 288) and the persistent backward (N <= 256, and 257 = eight blocks + one token: resident
    workgroups of one wave per 32-row block + a loader wave walking their heads, every load one phase ahead of its use; up to 96 tokens
    two to four of them per CU) against the query-/key-blocked kernels on identical inputs, with enough
    heads to fill the chip several times over: bitwise equal (same products in the same order), delta included, and
    reproducible.  (A first version of the fused kernel let the 64-float lse DMA pieces spill into the delta rows next to them:
    invisible at the small batches of the parity tests above, a race at full occupancy.)"""
    from apla_amd._lib import lib
    H = 12
    g = torch.Generator(device="cuda").manual_seed(N)
    qkv = torch.randn(B * N, 3 * 64 * H, device="cuda", generator=g).to(torch.bfloat16)
    do = torch.randn(B * N, 64 * H, device="cuda", generator=g).to(torch.bfloat16)
    scale = 64 ** -0.5
    o, lse = ops.attn_fwd(qkv, B, N, H, scale)
    old = ops.set_attn_variant(0)
    try:
        res = {}
        for v in (2, 3, 0, 1):
            ops.set_attn_variant(v)
            d1, d2 = torch.zeros(B, H, N, device="cuda"), torch.zeros(B, H, N, device="cuda")
            a1 = ops.attn_bwd(qkv, o, do, lse, B, N, H, scale, delta=d1).clone()
            a2 = ops.attn_bwd(qkv, o, do, lse, B, N, H, scale, delta=d2).clone()
            assert torch.equal(a1, a2) and torch.equal(d1, d2), v
            res[v] = (a1, d1.clone())
    finally:
        ops.set_attn_variant(old)
    for v in (2, 3, 0):
        if N == 257 and v != 2:
            # 8 blocks + 1: the persistent kernel takes token 256 as rank-1 corrections in vector arithmetic (round 6) — other products,
            # another summation order: equal to the blocked kernels within the rounding of the 16-bit outputs, delta to fp32 rounding,
            # and the same bits from the automatic choice as from the pinned variant
            assert ops.attn_kernel_name("bwd", B, N, H).startswith("attn_bwd_persist_kernel")
            assert torch.equal(res[v][0], res[3][0]) and torch.equal(res[v][1], res[3][1]), v
            assert rel_err(res[v][0].float().cpu(), res[1][0].float().cpu()) < 1e-2, v
            assert float((res[v][1] - res[1][1]).abs().max()) < 1e-4 * float(res[1][1].abs().max()), v
            continue
        assert torch.equal(res[v][0], res[1][0]) and torch.equal(res[v][1], res[1][1]), v
    # ... and against the fp64 oracle (appla_attn.py:56-60 differentiated by hand, oracle/apla_oracle.py:attention_bwd), as the forward's
    # test does: all twelve heads of the first, a middle and the last sequence of the walk — at (128, 197) and (64, 257) the launch is
    # the persistent / one-workgroup-per-head kernel with every CU busy (the bitwise chain above was anchored to
    # kernels the oracle had only seen at <= 9 heads).  dq, dk, dv and delta = rowsum(dO * O).
    dqkv, delta = res[0]
    D = 64 * H
    for b in sorted({0, B // 3, B - 1}):
        rows = slice(b * N, (b + 1) * N)
        q64, o64, do64 = (t[rows].cpu().double().reshape(1, N, -1) for t in (qkv, o, do))
        ref = O.attention_bwd(do64, q64, o64, lse[b].cpu().double().reshape(1, H, N), H, scale)
        got = dqkv[rows].cpu().double().reshape(1, N, 3 * D)
        for part, name in ((slice(0, D), "dq"), (slice(D, 2 * D), "dk"), (slice(2 * D, 3 * D), "dv")):
            # (one token: the softmax is 1, dq = dk = 0 exactly in the oracle — the scale is then the gradient's as a whole)
            den = max(float(ref[..., part].abs().max()), 1e-3 * float(ref.abs().max()))
            assert float((got[..., part] - ref[..., part]).abs().max()) / den < BF16_OUT, (b, name)
        dref = (do64.reshape(N, H, 64) * o64.reshape(N, H, 64)).sum(-1).t()
        assert float((delta[b].cpu().double() - dref).abs().max()) < 1e-4 * max(1.0, float(dref.abs().max())), b


@pytest.mark.parametrize("M,N,K", [(128, 768, 768), (128, 768, 3072), (128, 3072, 768), (37, 128, 256), (300, 192, 1536)])
def test_gemm_nt_small_few_rows(ops, M, N, K):
    """The split-K few-row GEMM (CLS-only tail of the last block) against fp64 on the same bf16 operands, every epilogue it
    implements, strided A rows (the CLS rows of a [B*N, D] buffer are N*D elements apart) and a strided output."""
    g = torch.Generator().manual_seed(M + N + K)
    stride = 3 * K                                                  # rows of A sit `stride` elements apart
    abuf = (torch.randn(M, stride, generator=g) * 0.5).to(torch.bfloat16).cuda()
    a = abuf[:, :K]
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    ws = ops.gemm_small_workspace(M, N, K, "cuda")
    assert ws is not None and ws.numel() * 4 == (K // (64 if K <= 1536 else 128)) * M * N * 4
    ref = a.double() @ w.double().t() + bias.double()
    out = ops.gemm_nt_small(a, w, bias, workspace=ws, out_dtype=torch.float32)
    assert rel_err(out.cpu(), ref.cpu()) < F32_OUT
    obuf = torch.zeros(M, 2 * N, device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt_small(a, w, bias, workspace=ws, out=obuf[:, :N])
    assert rel_err(obuf[:, :N].float().cpu(), ref.cpu()) < 6e-3 and float(obuf[:, N:].abs().max()) == 0.0
    assert torch.equal(ops.gemm_nt_small(a, w, bias, workspace=ws), obuf[:, :N])               # reproducible
    gp = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    h = ops.gemm_nt_small(a, w, bias, workspace=ws, epilogue=ops.EPI_GELU, aux_out=gp)
    r = ref.float()
    phi = 0.5 * (1 + torch.erf(r / math.sqrt(2)))
    assert rel_err(h.float().cpu(), (r * phi).cpu()) < 6e-3
    assert rel_err(gp.float().cpu(), (phi + r * torch.exp(-0.5 * r * r) / math.sqrt(2 * math.pi)).cpu()) < 6e-3
    mul = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    out = ops.gemm_nt_small(a, w, None, workspace=ws, epilogue=ops.EPI_MUL, aux_in=mul)
    assert rel_err(out.float().cpu(), ((ref - bias.double()) * mul.double()).cpu()) < 6e-3
    res = torch.randn(M, N, generator=g).cuda()
    out = ops.gemm_nt_small(a, w, bias, workspace=ws, epilogue=ops.EPI_RESIDUAL, aux_in=res, out_dtype=torch.float32)
    assert rel_err(out.cpu(), (ref + res.double()).cpu()) < F32_OUT
    if N % 128 == 0 and K % 64 == 0:   # same operands through the tiled kernel: equal up to the fp32 summation order
        assert rel_err(ops.gemm_nt(a, w, bias, out_dtype=torch.float32).cpu(), ops.gemm_nt_small(a, w, bias, workspace=ws, out_dtype=torch.float32).cpu()) < 1e-5
    with pytest.raises(Exception):
        ops.gemm_nt_small(a, w, bias, workspace=ws[:16])


@pytest.mark.parametrize("M,D,r", [(394, 256, 64), (257, 384, 40), (64, 128, 128)])
def test_apla_proj_fwd_bwd_operators(ops, M, D, r):
    """apla_proj_fwd / apla_proj_bwd — the projection as one forward and one backward operator of the C-ABI — called
    through ctypes with raw pointers, against the reference formulas of appla_attn.py:62-79 in float64: two linears scattered
    to their output columns; dX through both parts, dW1 / db1 for the r trainable rows only."""
    from apla_amd._lib import check, lib
    g = torch.Generator().manual_seed(M + r)
    x = (torch.randn(M, D, generator=g) * 0.5).to(torch.bfloat16)
    dy = torch.randn(M, D, generator=g).to(torch.bfloat16)
    W1, W2 = torch.randn(r, D, generator=g) * 0.05, torch.randn(D - r, D, generator=g) * 0.05
    b1, b2 = torch.randn(r, generator=g) * 0.1, torch.randn(D - r, generator=g) * 0.1
    inds = torch.randperm(D, generator=g)
    xd, dyd = x.double(), dy.double()
    yref = torch.empty(M, D, dtype=torch.float64)
    yref[:, inds[:r]] = xd @ W1.double().t() + b1.double()
    yref[:, inds[r:]] = xd @ W2.double().t() + b2.double()
    Wn = torch.zeros(D, D, dtype=torch.float64)
    Wn[inds[:r]], Wn[inds[r:]] = W1.double(), W2.double()
    dxref = dyd @ Wn
    dW1ref, db1ref = dyd[:, inds[:r]].t() @ xd, dyd[:, inds[:r]].sum(0)
    # device state: natural-order merged weight built by apla_pack_proj_rows from the module's parameters
    Wnf = torch.zeros(D, D)
    Wnf[inds[r:]] = W2
    bn = torch.zeros(D)
    bn[inds[r:]] = b2
    Wnat, WnatT, bnat = Wnf.to(torch.bfloat16).cuda(), Wnf.t().contiguous().to(torch.bfloat16).cuda(), bn.cuda()
    i32 = inds.int().cuda()
    ops.pack_proj_rows(W1.cuda(), b1.cuda(), i32, None, Wnat, WnatT, bnat)
    xg, dyg = x.cuda(), dy.cuda()
    y = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    check(lib().apla_proj_fwd(xg.data_ptr(), Wnat.data_ptr(), bnat.data_ptr(), i32.data_ptr(), y.data_ptr(), M, D, r, s), "apla_proj_fwd")
    assert rel_err(y.float().cpu(), yref) < 6e-3
    nbytes = lib().apla_proj_workspace_bytes(M, D, r)
    assert nbytes > 0
    ws = torch.empty(nbytes, device="cuda", dtype=torch.uint8)
    dx = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    dW1, db1 = torch.empty(r, D, device="cuda"), torch.empty(r, device="cuda")
    check(lib().apla_proj_bwd(dyg.data_ptr(), xg.data_ptr(), WnatT.data_ptr(), i32.data_ptr(), dx.data_ptr(), dW1.data_ptr(), db1.data_ptr(),
                              ws.data_ptr(), nbytes, M, D, r, 0, s), "apla_proj_bwd")
    torch.cuda.synchronize()
    assert rel_err(dx.float().cpu(), dxref) < 6e-3
    assert rel_err(dW1.cpu(), dW1ref) < 2e-5 * 50 and rel_err(db1.cpu(), db1ref) < 1e-4   # fp32 accumulation of bf16 products
    assert lib().apla_proj_bwd(dyg.data_ptr(), xg.data_ptr(), WnatT.data_ptr(), i32.data_ptr(), dx.data_ptr(), dW1.data_ptr(), db1.data_ptr(),
                               ws.data_ptr(), 16, M, D, r, 0, s) == -22      # workspace too small: -EINVAL, nothing launched


@pytest.mark.parametrize("M,N,K,variant", [(2500, 512, 256, 4), (300, 256, 128, 4), (300, 256, 128, 1), (128, 768, 768, -1)])
def test_gemm_gelu_forward_only_epilogue(ops, M, N, K, variant):
    """APLA_EPI_GELU_FWD (the no-grad forward: evaluation, EMA teacher) writes the same h as APLA_EPI_GELU bit for bit and saves
    nothing: ping-pong (M >= 2048), persistent, simple and few-row kernels."""
    from apla_amd._lib import lib
    g = torch.Generator().manual_seed(M + N)
    a = (torch.randn(M, K, generator=g) * 0.7).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    gp = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    if variant < 0:
        ws = ops.gemm_small_workspace(M, N, K, "cuda")
        h_ref = ops.gemm_nt_small(a, w, bias, workspace=ws, epilogue=ops.EPI_GELU, aux_out=gp)
        h = ops.gemm_nt_small(a, w, bias, workspace=ws, epilogue=ops.EPI_GELU_FWD)
    else:
        old = ops.set_gemm_variant(variant)
        try:
            h_ref = ops.gemm_nt(a, w, bias, epilogue=ops.EPI_GELU, aux_out=gp)
            h = ops.gemm_nt(a, w, bias, epilogue=ops.EPI_GELU_FWD)
        finally:
            ops.set_gemm_variant(old)
        with pytest.raises(ValueError):
            ops.gemm_nt(a, w, bias, epilogue=ops.EPI_GELU_FWD, aux_out=gp)
    assert torch.equal(h, h_ref)
    r = (a.double() @ w.double().t() + bias.double()).float()
    assert rel_err(h.float().cpu(), (r * 0.5 * (1 + torch.erf(r / math.sqrt(2)))).cpu()) < 6e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("p,n", [(0.1, 8 * 1237), (0.5, 3 * 197 * 768), (0.0, 64)])
def test_dropout_kernels_and_the_oracle_mask(ops, dtype, p, n):
    """apla_dropout_fwd / _bwd: the keep mask is the oracle's counter-based mask bit for bit (Philox4x32-10 pinned by the Random123
    vectors), y = keep ? x / (1 - p) : 0 in the tensor's own rounding, the backward applies the same mask and scale."""
    import numpy as np
    x = rnd(n, seed=161).to(dtype)
    seed, offset = 0x1234_5678_9ABC_DEF0, 3
    y, keep = ops.dropout_fwd(dev(x), p, seed, offset)
    want = torch.from_numpy(O.philox_keep_mask(n, p, seed, offset))
    assert torch.equal(keep.cpu().bool(), want)
    inv = torch.tensor(1.0, dtype=torch.float32) / (torch.tensor(1.0, dtype=torch.float32) - torch.tensor(p, dtype=torch.float32))
    ref = torch.where(want, (x.float() * inv), torch.zeros(())).to(dtype)
    assert torch.equal(y.cpu(), ref)
    dy = rnd(n, seed=162).to(dtype)
    dx = ops.dropout_bwd(dev(dy), keep, p)
    assert torch.equal(dx.cpu(), torch.where(want, dy.float() * inv, torch.zeros(())).to(dtype))
    if p > 0:
        assert abs(float(want.float().mean()) - (1 - p)) < 0.02
        y2, keep2 = ops.dropout_fwd(dev(x), p, seed, offset + 1)
        assert not torch.equal(keep2, keep)            # another offset, another mask
    sc = torch.tensor([0.0, 2.0, 1.0], device="cuda")
    xs = dev(rnd(3, 5, 64, seed=163).to(dtype))
    assert torch.equal(ops.scale_samples(xs, sc).float().cpu(), (xs.float() * sc[:, None, None]).to(dtype).float().cpu())


@pytest.mark.parametrize("K,D", [(65536, 256), (515, 64), (3, 1024), (192, 1024), (128, 36)])
def test_weight_norm_kernels(ops, K, D):
    """apla_weight_norm_fwd / _bwd against torch.nn.utils.weight_norm's own arithmetic in float64 (dinov2 dino_head.py:27-28: the
    prototype layer W = v g / ||v||, dim = 0): W in the 16-bit operand type, dv and dg from an fp32 dW."""
    v = rnd(K, D, seed=171) * 0.3
    g = (rnd(K, 1, seed=172) * 0.1 + 1.0)
    dW = rnd(K, D, seed=173)
    vd, gd = v.double().requires_grad_(True), g.double().requires_grad_(True)
    Wd = vd * (gd / vd.norm(dim=1, keepdim=True))
    (Wd * dW.double()).sum().backward()
    W, norm = ops.weight_norm_fwd(dev(v), dev(g).reshape(-1))
    assert W.dtype == torch.bfloat16 and rel_err(W.cpu(), Wd.detach()) < BF16_OUT
    assert rel_err(norm.cpu(), v.double().norm(dim=1)) < 1e-6
    dv, dg = ops.weight_norm_bwd(dev(dW), dev(v), dev(g).reshape(-1), norm)
    assert rel_err(dv.cpu(), vd.grad) < 2e-5 and rel_err(dg.cpu(), gd.grad.reshape(-1)) < 2e-5
    dv2, none = ops.weight_norm_bwd(dev(dW), dev(v), dev(g).reshape(-1), norm, want_dg=False)
    assert none is None and torch.equal(dv2, dv)
    # the launch that writes W^T beside W (what the dX GEMM reads): the same bits, transposed; None where the kernel does not apply
    W2, norm2, WT = ops.weight_norm_fwd(dev(v), dev(g).reshape(-1), transposed=True)
    assert torch.equal(W2, W) and torch.equal(norm2, norm)
    assert (WT is None) == (K % 64 != 0) and (WT is None or (tuple(WT.shape) == (D, K) and torch.equal(WT, W.t().contiguous())))
