"""Hand-computed expectations for the reference quirks of SURVEY.md Appendix B that had no direct test (4, 5, 7, 9, 12).

Every scene is small enough for pencil arithmetic: an identity camera (view matrix = I, looking down +z), splats with the
identity quaternion and a flat (z-scale 0) or isotropic covariance, so that the projected covariance is DIAGONAL and known in
closed form.  The expected numbers below come from those closed forms evaluated in float64 -- NOT from any rasterizer
restatement (oracle/torch_ref.dense_render is not used) -- and are compared with BOTH oracle/raster_ref.c
(tests/test_quirks_cpu.py) and the HIP rasterizer (tests/test_quirks_gpu.py).

Reference lines: forward.cu:219-247 (dilation, antialiasing rescale, det == 0 drop, eigenvalue floor),
forward.cu:350-403 (alpha test, transmittance stop before blending, n_contrib), backward.cu:542-543,576-578,663-664 (NDC
scaling of dL_dmean2D, skip of positions behind the last contributor), gaussian_renderer/__init__.py:72-76 (mask)."""
import math

import numpy as np
import torch

from curve_gaussian_amd import synthetic as S

H = W = 32
FOV = 0.6911


def camera():
    cam = S.make_camera((0.0, 0.0, 0.0), (0.0, 0.0, 1.0), (0.0, -1.0, 0.0), H, W, FOV, FOV)
    assert torch.allclose(cam.world_view_transform, torch.eye(4)), "the pencil arithmetic below assumes view = identity"
    return cam


def focal():
    return W / (2.0 * math.tan(FOV * 0.5))


def world_x_for_pixel(px, z):
    """World x (= view x) whose projection lands on pixel centre px: pix = ((ndc + 1) W - 1) / 2, ndc = x / (z tan)."""
    ndc = (2.0 * px + 1.0) / W - 1.0
    return ndc * z * math.tan(FOV * 0.5)


def splats(centres_px, z, scale_xy, opacity, scale_z=0.0):
    """P splats at pixel centres `centres_px` [(px, py), ...], depths z[i]; identity rotation; scales (s, s, scale_z)."""
    P = len(centres_px)
    z = np.broadcast_to(np.asarray(z, np.float64), (P,))
    m = np.zeros((P, 3), np.float64)
    for i, (px, py) in enumerate(centres_px):
        m[i] = (world_x_for_pixel(px, z[i]), world_x_for_pixel(py, z[i]), z[i])
    s = np.broadcast_to(np.asarray(scale_xy, np.float64), (P,))
    sc = np.stack([s, s, np.full(P, scale_z)], 1)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
    return {"means3D": t(m), "scales": t(sc), "rotations": t(np.tile([1.0, 0.0, 0.0, 0.0], (P, 1))),
            "opacities": t(np.broadcast_to(np.asarray(opacity, np.float64), (P,)).reshape(P, 1)),
            "colors": torch.ones(P, 1), "all_map": t(np.tile([0.0, 0.0, 0.0, 1.0], (P, 1)))}


def cov_diag(scale_xy, z):
    """Diagonal entry of the dilated 2D covariance of a flat isotropic splat on pixel (cx, cy): the Jacobian's third column
    multiplies Sigma_zz = 0, so cov2D = (f s / z)^2 I + 0.3 I  (forward.cu:84-113,219-222)."""
    return (focal() * scale_xy / z) ** 2 + 0.3


def radius_of(a):
    """forward.cu:240-244 for cov2D = a I: mid = a, det = a^2, the radicand mid^2 - det = 0 is floored at 0.1."""
    return math.ceil(3.0 * math.sqrt(a + math.sqrt(0.1)))


def alpha_at(op, a, dx, dy):
    return op * math.exp(-0.5 * (dx * dx + dy * dy) / a)
