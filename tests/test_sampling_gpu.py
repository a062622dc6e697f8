"""GPU parity of the fused curve-sampling / splat-attribute HIP kernels against the PyTorch restatement of
scene/gaussian_curve_model.py:70-122,180-198 (oracle/torch_ref.py), forward and autograd backward.
Tolerance 1e-4 relative (north star); rot_to_quat_batch has a discontinuous candidate selection (argmax of q_abs),
so a small outlier budget covers samples that sit on a tie."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as TR
from util import S, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _curves(B, seed, with_lines=True):
    c = S.make_curves(B, seed)
    g = torch.Generator().manual_seed(seed + 7)
    c["width"] = c["width"] + 0.3 * torch.randn(B, 1, generator=g)
    c["opacity"] = c["opacity"] + torch.randn(B, 1, generator=g)
    if with_lines:
        c["is_bezier"] = torch.rand(B, generator=g) > 0.3
    return c


@pytest.mark.parametrize("B,seed,lines", [(417, 1, False), (1000, 2, True), (5, 3, True), (16667, 4, False)])
def test_sample_curves_forward_backward(B, seed, lines):
    from curve_gaussian_amd.ops.curve_sampling import sample_curves
    c = _curves(B, seed, lines)
    m = 12
    # --- oracle (float64 autograd is the truth for gradients; float32 forward for values)
    ref = TR.prepare_scaling_rot(c["curve_points"], c["width"], c["is_bezier"], m)
    cp64 = c["curve_points"].double().requires_grad_(True)
    w64 = c["width"].double().requires_grad_(True)
    r64 = TR.prepare_scaling_rot(cp64, w64, c["is_bezier"], m)
    g = torch.Generator().manual_seed(seed + 100)
    gx, gr, gs = torch.randn(B * m, 3, generator=g), torch.randn(B * m, 4, generator=g), torch.randn(B * m, 3, generator=g)
    (r64[0] * gx.double()).sum().add((r64[1] * gr.double()).sum()).add((r64[2] * gs.double()).sum()).backward()
    # --- HIP
    cp = c["curve_points"].to(DEV).requires_grad_(True)
    w = c["width"].to(DEV).requires_grad_(True)
    xyz, rot, scl = sample_curves(cp, w, c["is_bezier"].to(DEV), m)
    ((xyz * gx.to(DEV)).sum() + (rot * gr.to(DEV)).sum() + (scl * gs.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    assert_close("xyz", xyz.detach().cpu().numpy(), ref[0].numpy(), rel=1e-6, outlier_frac=0)
    assert_close("scaling", scl.detach().cpu().numpy(), ref[2].numpy(), rel=1e-4, outlier_frac=0)  # dist = |B(t)-B(t-h)| cancels ~3 digits in f32
    assert_close("rotation", rot.detach().cpu().numpy(), ref[1].numpy(), rel=1e-4, outlier_frac=2e-3)
    assert_close("dL_dcurve_points", cp.grad.cpu().numpy(), cp64.grad.numpy(), rel=1e-4, outlier_frac=2e-3, abs_floor=1e-6)
    assert_close("dL_dwidth", w.grad.cpu().numpy(), w64.grad.numpy(), rel=1e-4, outlier_frac=0, abs_floor=1e-6)


@pytest.mark.parametrize("case", ["mixed", "bezier", "lines"])
def test_sample_curves_matches_the_reference_generated_fixture(case):
    """k_sample_f12 / k_sample_f3 / k_sample_bwd against tests/golden/prepare_scaling_rot.npz -- outputs and autograd gradients of the
    REFERENCE's own GaussianCurveModel.prepare_scaling_rot (gaussian_curve_model.py:180-198; tests/golden/make_model_golden.py imports
    it), not of the restatement: mixed Bezier / straight curves, all Bezier, all straight.  Tolerances of the test above."""
    import os
    from curve_gaussian_amd.ops.curve_sampling import sample_curves
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prepare_scaling_rot.npz"))
    t = lambda k: torch.from_numpy(z[f"{case}_{k}"])
    cp = t("curve_points").to(DEV).requires_grad_(True)
    w = t("width").to(DEV).requires_grad_(True)
    xyz, rot, scl = sample_curves(cp, w, t("is_bezier").to(DEV), 12)
    ((xyz * t("cot_xyz").to(DEV)).sum() + (rot * t("cot_rotation").to(DEV)).sum() + (scl * t("cot_scaling").to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    assert_close("xyz", xyz.detach().cpu().numpy(), z[f"{case}_xyz"], rel=1e-6, outlier_frac=0)
    assert_close("scaling", scl.detach().cpu().numpy(), z[f"{case}_scaling"], rel=1e-4, outlier_frac=0)
    assert_close("rotation", rot.detach().cpu().numpy(), z[f"{case}_rotation"], rel=1e-4, outlier_frac=2e-3)
    # (the file's gradients also carry the opacity functional: it does not reach curve_points / width)
    assert_close("dL_dcurve_points", cp.grad.cpu().numpy(), z[f"{case}_grad_curve_points"], rel=1e-4, outlier_frac=2e-3, abs_floor=1e-6)
    assert_close("dL_dwidth", w.grad.cpu().numpy(), z[f"{case}_grad_width"], rel=1e-4, outlier_frac=0, abs_floor=1e-6)


def test_sample_curves_none_grads_and_reentry():
    from curve_gaussian_amd.ops.curve_sampling import sample_curves
    c = _curves(300, 9, False)
    cp = c["curve_points"].to(DEV).requires_grad_(True)
    w = c["width"].to(DEV).requires_grad_(True)
    xyz, rot, scl = sample_curves(cp, w, None, 12)
    loss = xyz.sum()  # only xyz participates: rotation / scaling grads arrive as None
    g1 = torch.autograd.grad(loss, cp, retain_graph=True)[0]
    g2 = torch.autograd.grad(loss, cp)[0]
    assert torch.equal(g1, g2)
    # d(sum xyz)/dP_k = sum_i c_k(t_i): Bernstein weights sum to 1 per sample
    np.testing.assert_allclose(g1.sum(dim=(1, 2)).cpu().numpy(), 36.0, rtol=1e-5)


@pytest.mark.parametrize("use_mask", [False, True])
def test_splat_attributes_forward_backward(use_mask):
    from curve_gaussian_amd.ops.curve_sampling import splat_attributes
    B, m = 700, 12
    P = B * m
    c = _curves(B, 21, True)
    g = torch.Generator().manual_seed(5)
    xyz, rot, scl = TR.prepare_scaling_rot(c["curve_points"], c["width"], c["is_bezier"], m)
    rot = rot + 0.05 * torch.randn(P, 4, generator=g)  # make it properly un-normalised
    mask_logit = torch.randn(B, m, 1, generator=g) * 3 if use_mask else None
    thr = 0.3
    cam = S.make_camera((1.7, -0.9, 1.2), (0.5, 0.5, 0.5), (0, 0, 1), 64, 64)
    gr, go, gsc, ga = (torch.randn(P, 4, generator=g), torch.randn(P, 1, generator=g), torch.randn(P, 3, generator=g),
                       torch.randn(P, 4, generator=g))

    def ref(dt):
        r = rot.detach().clone().to(dt).requires_grad_(True)
        ol = c["opacity"].detach().clone().to(dt).requires_grad_(True)
        s = scl.detach().clone().to(dt).requires_grad_(True)
        ml = mask_logit.detach().clone().to(dt).requires_grad_(True) if use_mask else None
        rn = torch.nn.functional.normalize(r)
        op = torch.sigmoid(ol.unsqueeze(1).expand(-1, m, -1).reshape(-1, 1))
        so = s
        if use_mask:
            sg = torch.sigmoid(ml)
            mk = ((sg > thr).to(dt) - sg).detach() + sg
            so = s * mk.view(-1, 1)
            op = op * mk.view(-1, 1)
        am = TR.build_all_map(r, xyz.to(dt), cam.camera_center.to(dt), cam.world_view_transform.to(dt))
        loss = (rn * gr.to(dt)).sum() + (op * go.to(dt)).sum() + (so * gsc.to(dt)).sum() + (am * ga.to(dt)).sum()
        loss.backward()
        return (rn, op, so, am), (r.grad, ol.grad, s.grad, ml.grad if use_mask else None)

    (rn, op, so, am), _ = ref(torch.float32)
    _, (g_r, g_ol, g_s, g_ml) = ref(torch.float64)
    r = rot.detach().to(DEV).requires_grad_(True)
    ol = c["opacity"].detach().to(DEV).requires_grad_(True)
    s = scl.detach().to(DEV).requires_grad_(True)
    ml = mask_logit.detach().to(DEV).requires_grad_(True) if use_mask else None
    h = splat_attributes(r, xyz.to(DEV), ol, s, cam.camera_center.to(DEV), cam.world_view_transform.to(DEV), m, ml, thr)
    ((h[0] * gr.to(DEV)).sum() + (h[1] * go.to(DEV)).sum() + (h[2] * gsc.to(DEV)).sum() + (h[3] * ga.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    for name, a, b in [("rot_n", h[0], rn), ("opacity", h[1], op), ("scales", h[2], so), ("all_map", h[3], am)]:
        assert_close(name, a.detach().cpu().numpy(), b.detach().numpy(), rel=1e-5, outlier_frac=1e-4)
    assert_close("g_rot", r.grad.cpu().numpy(), g_r.numpy(), abs_floor=1e-6)
    assert_close("g_opacity_logit", ol.grad.cpu().numpy(), g_ol.numpy(), abs_floor=1e-6)
    assert_close("g_scaling", s.grad.cpu().numpy(), g_s.numpy(), abs_floor=1e-6)
    if use_mask:
        assert_close("g_mask_logit", ml.grad.cpu().numpy(), g_ml.numpy(), abs_floor=1e-6)


def test_ops_replay_correctly_inside_a_hip_graph():
    """Every op is stream-ordered and capturable; in particular the library's scratch clears are kernel nodes (a
    captured hipMemsetAsync only cleared on the FIRST replay on this runtime -- later replays of the global-norm
    reductions then summed onto stale partials).  Replay several times with changing inputs and compare with eager."""
    from curve_gaussian_amd.ops.curve_sampling import sample_curves
    c = S.make_curves(300, 5)
    cp = c["curve_points"].to(DEV).clone().requires_grad_(True)
    w = c["width"].to(DEV).clone().requires_grad_(True)
    isb = c["is_bezier"].to(DEV)
    gx = torch.randn(300 * 12, 3, device=DEV)
    gr = torch.randn(300 * 12, 4, device=DEV)

    def body():
        xyz, rot, scl = sample_curves(cp, w, isb, 12)
        ((xyz * gx).sum() + (rot * gr).sum() + scl.sum()).backward()
        return xyz, rot, scl

    for p in (cp, w):
        p.grad = torch.zeros_like(p)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        outs = body()
    for it in range(4):
        with torch.no_grad():
            cp.add_(0.01 * torch.randn_like(cp))
        for p in (cp, w):
            p.grad.zero_()
        graph.replay()
        torch.cuda.synchronize()
        got = [o.detach().clone() for o in outs] + [cp.grad.clone(), w.grad.clone()]
        cp2 = cp.detach().clone().requires_grad_(True)
        w2 = w.detach().clone().requires_grad_(True)
        xyz, rot, scl = sample_curves(cp2, w2, isb, 12)
        ((xyz * gx).sum() + (rot * gr).sum() + scl.sum()).backward()
        for name, a, b in zip(("xyz", "rot", "scl", "g_cp", "g_w"), got, (xyz, rot, scl, cp2.grad, w2.grad)):
            assert torch.isfinite(a).all(), (it, name)
            np.testing.assert_allclose(a.cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=f"replay {it} {name}")
