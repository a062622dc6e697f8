"""Synthetic stand-ins for the BASELINE.json configs.

Real ABC / Replica scans are not available offline, so tests and bench.py use seeded synthetic curve sets and
camera rigs with the reference's shapes and conventions:

* curves: ``_curve_points [B,4,3]``, ``_width [B,1]`` (log), ``_opacity [B,1]`` (logit), ``_mask [B,m,1]``
  (layout of /root/reference/scene/gaussian_curve_model.py:54-64,153-171);
* cameras: ``world_view_transform`` / ``full_proj_transform`` / ``camera_center`` built exactly as
  /root/reference/scene/cameras.py:59-66 with ``getWorld2View2`` / ``getProjectionMatrix`` of
  /root/reference/utils/graphics_utils.py:38-71 (row-major torch tensors holding the TRANSPOSED math matrices).

All randomness comes from ``torch.Generator(device="cpu")`` so CPU oracle and GPU runs see identical inputs.
"""
import math
from dataclasses import dataclass
from typing import List

import numpy as np
import torch

# name -> (B curves, views, H, W, seed, room_scale)
CONFIGS = {
    "cfg1": (417, 50, 800, 800, 1, False),
    "cfg2": (4167, 50, 1600, 1600, 2, False),
    "cfg3": (16667, 64, 1600, 1600, 3, False),
    "cfg4": (25000, 200, 680, 1200, 4, True),
    "cfg5": (83334, 256, 2048, 2048, 5, False),
}
N_GAUSSIANS = 12  # arguments/__init__.py:50


@dataclass
class SynthCamera:
    """Duck-types the attributes of scene.cameras.Camera that gaussian_renderer.render() reads."""
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor  # [4,4] = getWorld2View2(R,T)^T
    full_proj_transform: torch.Tensor   # [4,4] = (P V)^T
    camera_center: torch.Tensor         # [3]
    image_name: str = "synthetic"

    def to(self, device):
        return SynthCamera(self.image_height, self.image_width, self.FoVx, self.FoVy,
                           self.world_view_transform.to(device), self.full_proj_transform.to(device),
                           self.camera_center.to(device), self.image_name)


def projection_matrix(znear: float, zfar: float, fovX: float, fovY: float) -> torch.Tensor:
    """utils/graphics_utils.py:51-71 (getProjectionMatrix), same float32 evaluation."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = torch.zeros(4, 4)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def world2view(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """utils/graphics_utils.py:38-49 (getWorld2View2 with translate=0, scale=1)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def make_camera(eye, target, up, H: int, W: int, fovx: float = 0.6911, fovy: float = 0.6911,
                znear: float = 0.01, zfar: float = 100.0) -> SynthCamera:
    """Look-at camera in the reference's (COLMAP-style, +z forward, +y down) convention."""
    eye = np.asarray(eye, np.float64)
    fwd = np.asarray(target, np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    upv = np.asarray(up, np.float64)
    right = np.cross(fwd, upv)
    if np.linalg.norm(right) < 1e-6:
        right = np.cross(fwd, np.array([1.0, 0.0, 0.0]))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w_rot = np.stack([right, down, fwd], axis=1)  # columns = camera axes in world
    R = c2w_rot                                      # reference stores R = c2w rotation (transposed inside world2view)
    T = -c2w_rot.T @ eye                             # w2c translation
    wv = torch.tensor(world2view(R, T)).transpose(0, 1).contiguous()
    proj = projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1)
    full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wv.inverse()[3, :3].contiguous()
    return SynthCamera(H, W, fovx, fovy, wv, full, center)


def fibonacci_cameras(n: int, H: int, W: int, radius: float = 1.8, center=(0.5, 0.5, 0.5)) -> List[SynthCamera]:
    cams = []
    c = np.asarray(center, np.float64)
    golden = math.pi * (3.0 - math.sqrt(5.0))
    for i in range(n):
        z = 1.0 - 2.0 * (i + 0.5) / n
        r = math.sqrt(max(0.0, 1.0 - z * z))
        th = golden * i
        eye = c + radius * np.array([r * math.cos(th), r * math.sin(th), z])
        cams.append(make_camera(eye, c, (0.0, 0.0, 1.0), H, W))
    return cams


def room_cameras(n: int, H: int, W: int, seed: int) -> List[SynthCamera]:
    """cfg4: cameras inside the box [-3,3]x[-1.5,1.5]x[-3,3] looking at random interior points."""
    g = torch.Generator(device="cpu").manual_seed(seed + 1000)
    lo = torch.tensor([-2.5, -1.0, -2.5])
    hi = torch.tensor([2.5, 1.0, 2.5])
    cams = []
    for _ in range(n):
        eye = (lo + (hi - lo) * torch.rand(3, generator=g)).numpy()
        tgt = (lo + (hi - lo) * torch.rand(3, generator=g)).numpy()
        cams.append(make_camera(eye, tgt, (0.0, 1.0, 0.0), H, W))
    return cams


def make_curves(B: int, seed: int, room_scale: bool = False, m: int = N_GAUSSIANS):
    """Curve parameters with the distribution of SURVEY.md section 8d."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    if room_scale:
        lo = torch.tensor([-3.0, -1.5, -3.0])
        hi = torch.tensor([3.0, 1.5, 3.0])
        lmin, lmax = 0.1, 0.5
    else:
        lo = torch.full((3,), -0.05)
        hi = torch.full((3,), 1.05)
        lmin, lmax = 0.02, 0.10
    P0 = lo + (hi - lo) * torch.rand(B, 3, generator=g)
    u = torch.randn(B, 3, generator=g)
    u = u / u.norm(dim=1, keepdim=True)
    L = lmin + (lmax - lmin) * torch.rand(B, 1, generator=g)
    P3 = P0 + L * u
    P1 = P0 + (P3 - P0) / 3.0 + torch.randn(B, 3, generator=g) * (0.1 * L)
    P2 = P0 + (P3 - P0) * (2.0 / 3.0) + torch.randn(B, 3, generator=g) * (0.1 * L)
    curve_points = torch.stack([P0, P1, P2, P3], dim=1).contiguous()
    width = torch.full((B, 1), math.log(5e-3))
    opacity = torch.full((B, 1), math.log(0.6 / 0.4))
    mask = torch.ones(B, m, 1)
    is_bezier = torch.ones(B, dtype=torch.bool)
    return dict(curve_points=curve_points, width=width, opacity=opacity, mask=mask, is_bezier=is_bezier)


def make_config(name: str, n_views: int = None):
    B, V, H, W, seed, room = CONFIGS[name]
    if n_views is not None:
        V = n_views
    curves = make_curves(B, seed, room)
    cams = room_cameras(V, H, W, seed) if room else fibonacci_cameras(V, H, W)
    return curves, cams


def random_splats(P: int, seed: int, scale_range=(0.002, 0.02), box=(-0.05, 1.05)):
    """Generic (non-curve) splat cloud used by rasterizer parity tests: random anisotropic scales, random
    UN-normalised-then-normalised quaternions, random opacities and direction maps."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    means = box[0] + (box[1] - box[0]) * torch.rand(P, 3, generator=g)
    ls = math.log(scale_range[0]) + (math.log(scale_range[1]) - math.log(scale_range[0])) * torch.rand(P, 3, generator=g)
    scales = torch.exp(ls)
    q = torch.randn(P, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    opac = 0.05 + 0.9 * torch.rand(P, 1, generator=g)
    amap = torch.randn(P, 4, generator=g)
    amap[:, 3] = 1.0
    colors = torch.rand(P, 1, generator=g)
    return dict(means3D=means, scales=scales, rotations=q, opacities=opac, all_map=amap, colors=colors)
