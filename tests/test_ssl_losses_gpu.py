"""DINO CLS-token loss and iBOT patch loss on the HIP kernels against goldens produced by the REFERENCE
classes (tests/golden/make_golden.py::g10_ssl_losses): two iterations each so the centre EMA is covered; losses within
1e-5 relative, gradients and teacher distributions within 1e-5 max-abs/max-abs (everything is fp32)."""
import pytest
import torch

from conftest import load_golden, rel_err, t

pytestmark = pytest.mark.gpu
TOL = 1e-5


def test_dino_loss_matches_reference():
    from apla_amd.ssl import DINOLoss
    g = load_golden("g10_ssl_losses.npz")
    K, n, n_local = [int(v) for v in g["meta"]]
    dino = DINOLoss(K, student_temp=0.1, center_momentum=0.9).cuda()
    for it in range(2):
        teacher = t(g[f"dino{it}.teacher"]).cuda()
        tprobs = dino.softmax_center_teacher(teacher, teacher_temp=0.05).view(2, n, K)
        dino.update_center(teacher)
        assert rel_err(tprobs.cpu(), g[f"dino{it}.tprobs"]) < TOL
        s_glob = t(g[f"dino{it}.s_glob"]).cuda().requires_grad_(True)
        s_loc = t(g[f"dino{it}.s_loc"]).cuda().requires_grad_(True)
        loss_g = dino(student_output_list=[s_glob], teacher_out_softmaxed_centered_list=[tprobs.flatten(0, 1)])
        loss_l = dino(student_output_list=s_loc.chunk(n_local), teacher_out_softmaxed_centered_list=tprobs)
        (loss_g + loss_l).backward()
        assert abs(float(loss_g) - float(g[f"dino{it}.loss_g"])) < TOL * abs(float(g[f"dino{it}.loss_g"]))
        assert abs(float(loss_l) - float(g[f"dino{it}.loss_l"])) < TOL * abs(float(g[f"dino{it}.loss_l"]))
        assert rel_err(s_glob.grad.cpu(), g[f"dino{it}.ds_glob"]) < TOL
        assert rel_err(s_loc.grad.cpu(), g[f"dino{it}.ds_loc"]) < TOL
    dino.apply_center_update()
    assert rel_err(dino.center.cpu(), g["dino.center"]) < TOL


def test_ibot_loss_matches_reference():
    from apla_amd.ssl import iBOTPatchLoss
    g = load_golden("g10_ssl_losses.npz")
    K = int(g["meta"][0])
    ibot = iBOTPatchLoss(K, student_temp=0.1, center_momentum=0.9).cuda()
    for it in range(2):
        masks = t(g[f"ibot{it}.masks"]).bool().cuda()
        n_masked = int(masks.sum())
        t_tok = t(g[f"ibot{it}.t_tok"]).cuda()
        tprobs = ibot.softmax_center_teacher(t_tok.unsqueeze(0), teacher_temp=0.05).squeeze(0)
        ibot.update_center(t_tok.unsqueeze(0))
        assert rel_err(tprobs.cpu(), g[f"ibot{it}.tprobs"]) < TOL
        s_tok = t(g[f"ibot{it}.s_tok"]).cuda().requires_grad_(True)
        t_pad = torch.cat([tprobs, torch.zeros(3, K, device="cuda")])
        mw = (1 / masks.sum(-1).clamp(min=1.0)).unsqueeze(-1).expand_as(masks)[masks]
        loss = ibot.forward_masked(s_tok, t_pad, student_masks_flat=masks, n_masked_patches=n_masked, masks_weight=mw)
        loss.backward()
        assert abs(float(loss) - float(g[f"ibot{it}.loss"])) < TOL * abs(float(g[f"ibot{it}.loss"]))
        assert rel_err(s_tok.grad.cpu(), g[f"ibot{it}.ds"]) < TOL     # rows past n_masked get exactly zero gradient
        assert float(s_tok.grad[n_masked:].abs().max()) == 0.0
    s3 = t(g["ibotd.s"]).cuda().requires_grad_(True)
    l3 = ibot(s3, t(g["ibotd.t"]).cuda(), t(g["ibotd.m"]).bool().cuda())
    l3.backward()
    assert abs(float(l3) - float(g["ibotd.loss"])) < TOL * abs(float(g["ibotd.loss"]))
    assert rel_err(s3.grad.cpu(), g["ibotd.ds"]) < TOL
    ibot.apply_center_update()
    assert rel_err(ibot.center.cpu(), g["ibot.center"]) < TOL


def test_distill_ce_at_shipped_width_and_16bit_student():
    """K = 65 536 prototypes (the shipped head width), student logits in bf16 as the head GEMM produces them."""
    from apla_amd.ssl.losses import distill_ce, softmax_center
    R, K = 24, 65536
    g = torch.Generator().manual_seed(0)
    s = (torch.randn(R, K, generator=g) * 3).to(torch.bfloat16)
    teacher = torch.randn(R, K, generator=g) * 2
    center = torch.randn(K, generator=g) * 0.1
    tp = softmax_center(teacher.cuda(), center.cuda(), 0.04)
    ref_tp = torch.softmax((teacher.double() - center.double()) / 0.04, -1)
    assert rel_err(tp.cpu(), ref_tp) < 1e-4
    sd = s.cuda().requires_grad_(True)
    loss = distill_ce(sd, tp, 0.1, None, 1.0 / R)
    loss.backward()
    s64 = s.double().requires_grad_(True)
    ref = -(ref_tp * torch.log_softmax(s64 / 0.1, -1)).sum(-1).mean()
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-4 * abs(float(ref))
    assert rel_err(sd.grad.float().cpu(), s64.grad) < 4e-3    # gradient returned in the student's dtype (bf16)


@pytest.mark.parametrize("R,K,sdt,xdt", [(24, 65536, torch.bfloat16, torch.bfloat16), (700, 1024, torch.float32, torch.float32),
                                         (300, 65536, torch.bfloat16, torch.bfloat16), (260, 9000, torch.bfloat16, torch.bfloat16),   # one-pass row kernel
                                         (5, 512, torch.bfloat16, torch.float32)])
def test_centered_teacher_inside_the_cross_entropy(R, K, sdt, xdt):
    """iBOT's targets left as (teacher logits, centre, temperature) and computed inside the cross-entropy kernel
    (apla_distill_ce_centered; iBOTPatchLoss.softmax_center_teacher(lazy=True) + forward_masked): loss, gradient and the row
    weights as the two-kernel path (apla_softmax_center + apla_distill_ce) and as the fp64 formula."""
    from apla_amd.ssl import iBOTPatchLoss
    from apla_amd.ssl.losses import CenteredTeacher
    g = torch.Generator().manual_seed(3)
    s = (torch.randn(R, K, generator=g) * 3).to(sdt)
    x = (torch.randn(R, K, generator=g) * 2).to(xdt)
    center = torch.randn(1, 1, K, generator=g) * 0.1
    w = torch.rand(R, generator=g)
    w[-1] = 0.0                                       # a row that must get exactly zero gradient
    ibot = iBOTPatchLoss(K, student_temp=0.1).cuda()
    ibot.center = center.cuda()
    masks = torch.ones(4, 8, dtype=torch.bool).cuda()
    lazy = ibot.softmax_center_teacher(x.cuda().unsqueeze(0), 0.05, lazy=True).squeeze(0)
    eager = ibot.softmax_center_teacher(x.cuda().unsqueeze(0), 0.05).squeeze(0)
    assert isinstance(lazy, CenteredTeacher) and torch.is_tensor(eager)
    assert rel_err(lazy.probs().cpu(), eager.cpu()) == 0.0
    s1, s2 = s.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
    l1 = ibot.forward_masked(s1, lazy, student_masks_flat=masks, n_masked_patches=R, masks_weight=w.cuda())
    l2 = ibot.forward_masked(s2, eager, student_masks_flat=masks, n_masked_patches=R, masks_weight=w.cuda())
    (3.0 * l1).backward()
    (3.0 * l2).backward()
    tp = torch.softmax((x.double() - center.double().view(1, K)) / 0.05, -1)
    s64 = s.double().requires_grad_(True)
    ref = -((tp * torch.log_softmax(s64 / 0.1, -1)).sum(-1) * w.double()).sum() / 4
    (3.0 * ref).backward()
    tol_g = 8e-3 if sdt == torch.bfloat16 else 2e-5   # bf16: rounded by the kernel and again after the scaling by the upstream gradient (2 x 2^-8)
    assert abs(float(l1) - float(ref)) < 1e-4 * abs(float(ref)) and abs(float(l1) - float(l2)) < 1e-5 * abs(float(l2))
    assert rel_err(s1.grad.float().cpu(), s64.grad) < tol_g and rel_err(s1.grad.float().cpu(), s2.grad.float().cpu().double()) < tol_g
    assert float(s1.grad[-1].abs().max()) == 0.0


@pytest.mark.parametrize("R,K,pad", [(3, 1000, 24), (600, 2056, 0), (17, 8, 8)])
def test_wide_loss_kernels_ragged_widths_and_strided_rows(R, K, pad):
    """apla_softmax_center / apla_distill_ce (8 prototypes per access, rows split over several workgroups): widths that are not a
    multiple of a workgroup pass (2 048), more rows than the split threshold (512), rows with a pitch larger than K."""
    from apla_amd.ssl.losses import distill_ce, softmax_center
    g = torch.Generator().manual_seed(5)
    s = (torch.randn(R, K + pad, generator=g) * 3).to(torch.bfloat16)
    x = torch.randn(R, K + pad, generator=g) * 2
    center = torch.randn(K, generator=g) * 0.1
    tp = softmax_center(x.cuda()[:, :K], center.cuda(), 0.07)
    ref_tp = torch.softmax((x[:, :K].double() - center.double()) / 0.07, -1)
    assert rel_err(tp.cpu(), ref_tp) < 1e-4
    sd = s.cuda().requires_grad_(True)
    w = torch.rand(R, generator=g)
    loss = distill_ce(sd[:, :K], tp, 0.1, w.cuda(), 0.5)
    loss.backward()
    s64 = s.double().requires_grad_(True)
    ref = -0.5 * ((ref_tp * torch.log_softmax(s64[:, :K] / 0.1, -1)).sum(-1) * w.double()).sum()
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-4 * abs(float(ref))
    assert rel_err(sd.grad.float().cpu()[:, :K], s64.grad[:, :K]) < 4e-3
    assert float(sd.grad[:, K:].abs().max()) == 0.0 if pad else True
