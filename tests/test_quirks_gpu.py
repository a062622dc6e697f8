"""GPU: the HIP rasterizer against the hand-computed expectations of tests/quirk_cases.py (SURVEY App. B quirks 4, 5, 7, 9,
12) -- the same closed forms oracle/raster_ref.c is held to in tests/test_quirks_cpu.py, so the unpinned oracle and the
product are tied to pencil arithmetic independently of each other."""
import math

import numpy as np
import pytest
import torch

import quirk_cases as Q
import test_quirks_cpu as QC
from util import hip_settings

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _hip(sp, antialiasing=False, cov3D=None, dimg=None):
    """-> (color, radii, invdepth, all_map, saved state, grads dict or None) through GaussianRasterizer (the operator API)."""
    from curve_gaussian_amd.diff_cur_rasterization import GaussianRasterizer
    cam = Q.camera()
    ins = {k: v.to(DEV).requires_grad_(True) for k, v in sp.items()}
    m2d = torch.zeros(ins["means3D"].shape[0], 3, device=DEV, requires_grad=True)
    rast = GaussianRasterizer(hip_settings(cam, torch.zeros(3), DEV, antialiasing=antialiasing))
    kw = dict(scales=ins["scales"], rotations=ins["rotations"]) if cov3D is None else dict(cov3D_precomp=cov3D.to(DEV))
    color, radii, invd, amap = rast(means3D=ins["means3D"], means2D=m2d, opacities=ins["opacities"],
                                    colors_precomp=ins["colors"], all_map=ins["all_map"], **kw)
    grads = None
    if dimg is not None:
        (color * torch.from_numpy(dimg).to(DEV)).sum().backward()
        grads = {"dL_dmeans2D": m2d.grad.cpu().numpy(), "dL_dopacity": ins["opacities"].grad.cpu().numpy()}
    torch.cuda.synchronize()
    return color.detach().cpu().numpy(), radii.cpu().numpy(), grads


def test_quirk4_dilation_and_antialiasing_rescale():
    s = 2.0 * math.sqrt(0.3) / Q.focal()
    sp = Q.splats([(16, 16)], 2.0, s, 0.8)
    a, ra, _ = _hip(sp)
    b, _, _ = _hip(sp, antialiasing=True)
    QC.check_quirk4(a, b, s)
    assert ra[0] == 3


def test_quirk5_eigenvalue_floor_and_det_zero_drop():
    _, r, _ = _hip(Q.splats([(16, 16)], 2.0, 1e-7, 0.8))
    assert r[0] == 3
    sp = Q.splats([(15.5, 15.5)], Q.focal(), 0.0, 0.8)
    sp["means3D"][0, 2] = torch.tensor(np.float32(Q.focal()))
    cov = torch.tensor([[-0.3, 0.0, 0.0, 0.5, 0.0, 0.0]], dtype=torch.float32)
    img, r, _ = _hip(sp, cov3D=cov)
    assert r[0] == 0 and img.max() == 0
    cov[0, 0] = 0.25
    img, r, _ = _hip(sp, cov3D=cov)
    assert r[0] > 0 and img.max() > 0.5


def _final_state(sp):
    """final_T / n_contrib of a forward through the C ABI's saved image buffer (cgs_rasterize_forward)."""
    from curve_gaussian_amd.diff_cur_rasterization import _C
    cam = Q.camera().to(DEV)
    tf = math.tan(Q.FOV * 0.5)
    t = {k: v.to(DEV) for k, v in sp.items()}
    e = torch.empty(0, device=DEV)
    out = _C.rasterize_gaussians(torch.zeros(3, device=DEV), t["means3D"], t["colors"], t["opacities"], t["scales"],
                                 t["rotations"], 1.0, e, t["all_map"], cam.world_view_transform, cam.full_proj_transform, tf, tf,
                                 Q.H, Q.W, e, 0, cam.camera_center, False, False, True, False)
    num, color, radii, geom, binb, img = out[:6]
    torch.cuda.synchronize()
    npix = Q.H * Q.W
    raw = img.cpu().numpy()
    final_T = raw[:4 * npix].view(np.float32).reshape(Q.H, Q.W)
    off = (4 * npix + 127) & ~127
    word = raw[off:off + 4 * npix].view(np.uint32).reshape(Q.H, Q.W)
    # bits 0..30: the reference's n_contrib; bit 31: the pixel terminated (csrc/composite.h, NCONTRIB_TERMINATED)
    _final_state.terminated = (word >> 31).astype(bool)
    return color.cpu().numpy(), final_T, word & 0x7fffffff


def test_quirk7_transmittance_stop_excludes_the_splat():
    sp = QC.quirk7_scene()
    color, final_T, n_contrib = _final_state(sp)
    QC.check_quirk7_forward(color, final_T, n_contrib)
    # the saved word also says WHICH pixels the T < 1e-4 test stopped: the stacked pixel, not its neighbour that ran to the end
    term = _final_state.terminated
    assert term[16, 16] and not term[16, 17]
    d = np.zeros((1, Q.H, Q.W), np.float32)
    d[0, 16, 16] = 1.0
    _, _, g = _hip(sp, dimg=d)
    QC.check_quirk7_backward(g["dL_dopacity"].reshape(-1))


def test_quirk9_means2D_gradient_is_in_ndc_units():
    s = 2.0 * math.sqrt(0.7) / Q.focal()
    d = np.zeros((1, Q.H, Q.W), np.float32)
    d[0, 16, 15] = 1.0
    _, _, g = _hip(Q.splats([(16, 16)], 2.0, s, 0.8), dimg=d)
    QC.check_quirk9(g["dL_dmeans2D"], 1.0)


def test_quirk12_masked_splat_stays_in_the_pipeline():
    sp = Q.splats([(16, 16), (8, 8)], 2.0, [0.0, 0.1], [0.0, 0.8])
    img, r, _ = _hip(sp)
    assert r[0] == 3 and r[1] > 3 and float(img[0, 16, 16]) == 0.0 and float(img[0, 8, 8]) > 0.7
    # and through the curve model: render(use_mask=True) with every sample of curve 0 masked out keeps radii > 0 for them
    from curve_gaussian_amd.gaussian_renderer import PipelineParams, render
    from curve_gaussian_amd.scene import GaussianCurveModel
    from curve_gaussian_amd import synthetic as S
    c = S.make_curves(6, 3)
    mask = torch.full((6, 12, 1), 4.0)
    mask[0] = -4.0
    cam = S.make_camera((0.5, -1.7, 0.9), (0.5, 0.5, 0.5), (0, 0, 1), 96, 128).to(DEV)
    gm = GaussianCurveModel(0, 12, device=DEV).create_from_curves(c["curve_points"], c["width"] + 1.0, c["opacity"], mask, c["is_bezier"])
    for route in (None, False):
        on = render(cam, gm, PipelineParams(), torch.zeros(3, device=DEV), use_mask=True, mask_thr=0.5, fused=route)
        off = render(cam, gm, PipelineParams(), torch.zeros(3, device=DEV), use_mask=False, fused=route)
        vis = off["radii"][:12] > 0
        assert vis.any() and (on["radii"][:12][vis] >= 2).all() and (on["radii"][:12][vis] <= 3).all()
        assert torch.equal(on["radii"][12:], off["radii"][12:])
        assert on["render"].sum() < off["render"].sum()
