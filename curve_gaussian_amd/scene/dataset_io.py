"""On-disk formats either side of the hot path, host-side Python like the reference's:

* EMAP scan reader  -- ``readEMAP`` (/root/reference/scene/dataset_readers.py:290-329): ``meta_data.json`` with
  ``height, width, frames[{rgb_path, camtoworld 4x4, intrinsics}]`` and one edge map per frame under
  ``edge_DexiNed/`` or ``edge_PidiNet/``; cameras built as ``loadCam`` + ``Camera`` do (utils/camera_utils.py:22-67,
  scene/cameras.py:18-66): R = (w2c rotation)^T, T = w2c translation, FoV from the intrinsics, image = PIL -> [C,H,W]
  in [0,1] (``PILtoTorch``, utils/general_utils.py:91-97), ``world_view_transform`` / ``full_proj_transform`` /
  ``camera_center`` exactly as the synthetic cameras (curve_gaussian_amd.synthetic).
* EMAP scan writer  -- the inverse, used to put synthetic scans on disk (BASELINE cfg1 "plumbing" case).
* ``parametric_edges.json`` / ``edge_points.ply`` writer -- ``extract_curves`` (train.py:250-293) with
  ``get_parametric_edge(visible_checking=False)`` -> ``process_geometry_data``
  (edge_extraction/extract_para_edge.py:60-129, 252-256): curves as 4x3 control points, lines as 6 floats, and the
  edge point cloud sampled every 5 mm of Simpson-rule arc length (edge_extraction/extract_uitl.py:291-330).
  ``merge_endpoints`` (opt.merge_endpoints_flag) and the visibility check are not reproduced.
* ``point_cloud.ply`` splat snapshot -- ``save_ply`` (scene/gaussian_model.py:267-280, 383-400; scene/__init__.py:96).

Parity: the camera arithmetic is pinned by tests/golden/emap_camera.npz (reference graphics_utils imported by
tests/golden/make_golden.py); the edge_extraction modules cannot be imported here (cv2 / point_cloud_utils missing), so
the sampling is restated and unpinned."""
import json
import math
import os
from dataclasses import dataclass
from typing import List, NamedTuple

import numpy as np
import torch

from ..synthetic import projection_matrix, world2view

DETECTOR_DIRS = {"DexiNed": "edge_DexiNed", "PidiNet": "edge_PidiNet"}


def focal2fov(focal, pixels):
    """utils/graphics_utils.py:103-104"""
    return 2 * math.atan(pixels / (2 * focal))


def fov2focal(fov, pixels):
    """utils/graphics_utils.py:100-101"""
    return pixels / (2 * math.tan(fov / 2))


@dataclass
class EdgeCamera:
    """The attributes of scene.cameras.Camera that render() and the train step read."""
    uid: int
    image_name: str
    R: np.ndarray
    T: np.ndarray
    K: np.ndarray
    FoVx: float
    FoVy: float
    image_height: int
    image_width: int
    original_image: torch.Tensor        # [C,H,W] in [0,1] (the edge map; train.py uses channel 0)
    world_view_transform: torch.Tensor  # [4,4]
    full_proj_transform: torch.Tensor   # [4,4]
    camera_center: torch.Tensor         # [3]
    znear: float = 0.01
    zfar: float = 100.0

    def to(self, device):
        c = EdgeCamera(**{**self.__dict__})
        c.original_image = self.original_image.to(device)
        c.world_view_transform = self.world_view_transform.to(device)
        c.full_proj_transform = self.full_proj_transform.to(device)
        c.camera_center = self.camera_center.to(device)
        return c


def camera_from_emap_frame(uid, name, camtoworld, intrinsics, image_chw, znear=0.01, zfar=100.0, orig_size=None):
    """dataset_readers.py:303-324 + cameras.py:53-66 for one frame.  `orig_size` = (width, height) of the image file the
    intrinsics belong to: readEMAP computes the field of view from the ORIGINAL size (:320-321) and loadCam only then
    rescales the pixels (camera_utils.py:28-42), so a down-scaled image keeps its field of view."""
    c2w = np.array(camtoworld, dtype=np.float64)
    K = np.array(intrinsics, dtype=np.float64)
    w2c = np.linalg.inv(c2w)
    R = np.transpose(w2c[:3, :3])   # stored transposed "due to glm in the CUDA code" (:306)
    T = w2c[:3, 3]
    H, W = int(image_chw.shape[1]), int(image_chw.shape[2])
    ow, oh = orig_size if orig_size is not None else (W, H)
    fovy = focal2fov(K[1, 1], oh)
    fovx = focal2fov(K[0, 0], ow)
    wv = torch.tensor(world2view(R, T)).transpose(0, 1).contiguous()
    proj = projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1)
    full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wv.inverse()[3, :3].contiguous()
    return EdgeCamera(uid, name, R, T, K, fovx, fovy, H, W, image_chw.clamp(0.0, 1.0), wv, full, center, znear, zfar)


def _pil_to_chw(img, resolution=None):
    """utils/general_utils.py:91-97 (PILtoTorch)."""
    if resolution is not None:
        img = img.resize(resolution)
    a = torch.from_numpy(np.array(img)) / 255.0
    return a.permute(2, 0, 1) if a.dim() == 3 else a.unsqueeze(-1).permute(2, 0, 1)


def read_emap(path, transformsfile="meta_data.json", detector="DexiNed", max_width=1600) -> List[EdgeCamera]:
    """readEMAP + loadCam: edge maps are opened as RGBA (:319), the first three channels become the image; images wider
    than `max_width` are rescaled to it (camera_utils.py:28-42, resolution == -1)."""
    from PIL import Image
    if detector not in DETECTOR_DIRS:
        raise ValueError(f"Detector {detector} not supported")   # :317
    with open(os.path.join(path, transformsfile)) as f:
        meta = json.load(f)
    cams = []
    for idx, frame in enumerate(meta["frames"]):
        edge_path = os.path.join(path, DETECTOR_DIRS[detector], frame["rgb_path"])
        image = Image.open(edge_path).convert("RGBA")
        ow, oh = image.size
        res = None
        if ow > max_width:
            scale = ow / max_width
            res = (int(ow / scale), int(oh / scale))
        rgb = torch.cat([_pil_to_chw(im, res) for im in image.split()[:3]], dim=0).float()
        name = os.path.splitext(os.path.basename(frame["rgb_path"]))[0]
        cams.append(camera_from_emap_frame(idx, name, frame["camtoworld"], frame["intrinsics"], rgb, orig_size=(ow, oh)))
    return cams


class BasicPointCloud(NamedTuple):
    """utils/graphics_utils.py:17-20"""
    points: np.ndarray
    colors: np.ndarray
    normals: np.ndarray


SH_C0 = 0.28209479177387814   # utils/sh_utils.py:24


def RGB2SH(rgb):
    """utils/sh_utils.py:114-115"""
    return (rgb - 0.5) / SH_C0


def SH2RGB(sh):
    """utils/sh_utils.py:117-118"""
    return sh * SH_C0 + 0.5


def grid_point_cloud(num_pts_per_axis: int = 15, rng=None) -> BasicPointCloud:
    """The initial point cloud of the EMAP loader (dataset_readers.py:404-412,439-441, init_random_init): a regular
    num_pts_per_axis^3 grid over [-0.05, 1.05]^3 (meshgrid in 'xy' order, as numpy's default), near-black random colours.
    `rng`: a numpy Generator / RandomState for the colours (the reference uses the global numpy state)."""
    x = np.linspace(-0.05, 1.05, num_pts_per_axis)
    xx, yy, zz = np.meshgrid(x, x, x)
    xyz = np.vstack([xx.ravel(), yy.ravel(), zz.ravel()]).T
    rnd = (rng.random if rng is not None else np.random.random)((xyz.shape[0], 3))
    return BasicPointCloud(points=xyz, colors=SH2RGB(rnd / 255.0), normals=np.zeros((xyz.shape[0], 3)))


def write_emap(path, cameras, edge_maps, detector="DexiNed"):
    """Writes a scan in the EMAP layout (meta_data.json + <detector dir>/<i>_colors.png) from cameras that carry
    world_view_transform / FoVx / FoVy (e.g. curve_gaussian_amd.synthetic cameras) and [1,H,W] or [H,W] edge maps in
    [0,1].  Inverse of read_emap up to the 8-bit quantisation of the images."""
    from PIL import Image
    os.makedirs(os.path.join(path, DETECTOR_DIRS[detector]), exist_ok=True)
    frames = []
    H = W = None
    for i, (cam, em) in enumerate(zip(cameras, edge_maps)):
        em = em.detach().cpu().float()
        em = em[0] if em.dim() == 3 else em
        H, W = int(em.shape[0]), int(em.shape[1])
        w2c = cam.world_view_transform.detach().cpu().double().numpy().T      # stored transposed (cameras.py:59)
        c2w = np.linalg.inv(w2c)
        fx, fy = fov2focal(cam.FoVx, W), fov2focal(cam.FoVy, H)
        K = [[fx, 0.0, W / 2.0, 0.0], [0.0, fy, H / 2.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]]
        rgb_path = f"{i}_colors.png"
        Image.fromarray((em.clamp(0, 1) * 255.0).round().to(torch.uint8).numpy(), mode="L").save(
            os.path.join(path, DETECTOR_DIRS[detector], rgb_path))
        frames.append({"rgb_path": rgb_path, "camtoworld": c2w.tolist(), "intrinsics": K})
    with open(os.path.join(path, "meta_data.json"), "w") as f:
        json.dump({"height": H, "width": W, "frames": frames}, f)


# ------------------------------------------------------------------------------------------ parametric edges out
def bezier_curve_length(control_points, num_samples=100):
    """Simpson-rule arc length of a cubic Bezier (edge_extraction/extract_uitl.py:291-330), vectorised."""
    P = np.asarray(control_points, dtype=np.float64)
    d = 3.0 * (P[1:] - P[:-1])                                   # control points of the derivative (degree 2)

    def speed(t):
        t = np.asarray(t, dtype=np.float64)[:, None]
        v = (1 - t) ** 2 * d[0] + 2 * (1 - t) * t * d[1] + t ** 2 * d[2]
        return np.linalg.norm(v, axis=1)

    h = 1.0 / num_samples
    odd = speed(np.arange(1, num_samples, 2) * h).sum()
    even = speed(np.arange(2, num_samples - 1, 2) * h).sum()
    ends = speed(np.array([0.0, 1.0])).sum()
    return (ends + 4 * odd + 2 * even) * h / 3.0


def sample_edge_points(curves_ctl_pts, lines_end_pts, sample_resolution=0.005):
    """process_geometry_data :107-129: points every `sample_resolution` of arc length along curves and lines."""
    pts = []
    coeff = np.array([[-1, 3, -3, 1], [3, -6, 3, 0], [-3, 3, 0, 0], [1, 0, 0, 0]], dtype=np.float64)
    for curve in np.asarray(curves_ctl_pts, dtype=np.float64).reshape(-1, 4, 3):
        n = int(bezier_curve_length(curve, 100) // sample_resolution)
        t = np.linspace(0, 1, n)
        U = np.array([t ** 3, t ** 2, t, np.ones_like(t)])
        pts.extend(U.T.dot(coeff).dot(curve).tolist())
    for line in np.asarray(lines_end_pts, dtype=np.float64).reshape(-1, 2, 3):
        n = int(np.linalg.norm(line[0] - line[1]) // sample_resolution)
        t = np.linspace(0, 1, n)
        pts.extend((np.outer(t, line[1] - line[0]) + line[0]).tolist())
    return np.array(pts, dtype=np.float32).reshape(-1, 3)


def extract_curves(gaussians):
    """train.py:252-256,266-273: Bezier curves as [n,12], line segments as [n,6] (first and last control point)."""
    cp = gaussians.get_curve_points.detach()
    isb = gaussians.is_bezier.bool()
    bez = cp[isb].reshape(-1, 12).cpu().numpy()
    lines = cp[~isb][:, [0, -1], :].reshape(-1, 6).cpu().numpy()
    return {"lines_end_pts": lines.tolist() if len(lines) > 0 else [],
            "curves_ctl_pts": bez.tolist() if len(bez) > 0 else []}


def write_parametric_edges(gaussians, model_path):
    """Writes parametric_edges.json (the evaluation input, train.py:287-293) and edge_points.ply (ASCII, :277-285)."""
    os.makedirs(model_path, exist_ok=True)
    merged = extract_curves(gaussians)
    curves = np.array(merged["curves_ctl_pts"]).reshape(-1, 12).reshape(-1, 4, 3)
    lines = np.array(merged["lines_end_pts"]).reshape(-1, 6)
    edge_dict = {"curves_ctl_pts": curves.tolist(), "lines_end_pts": lines.tolist()}
    pts = sample_edge_points(curves, lines)
    with open(os.path.join(model_path, "parametric_edges.json"), "w") as f:
        json.dump(edge_dict, f)
    with open(os.path.join(model_path, "edge_points.ply"), "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty double x\nproperty double y\nproperty double z\n"
                "end_header\n" % len(pts))
        for p in pts:
            f.write("%.10g %.10g %.10g\n" % (p[0], p[1], p[2]))
    return edge_dict, pts


# ------------------------------------------------------------------------------------------ splat snapshot (PLY)
def ply_attribute_names(n_dc, n_rest, n_scale=3, n_rot=4):
    """scene/gaussian_model.py:267-280 (construct_list_of_attributes)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)]
    names += ["opacity"] + [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)]
    return names


def save_ply(gaussians, path):
    """Splat snapshot in the 3DGS point_cloud.ply layout the reference writes (scene/gaussian_model.py:383-400, called
    from scene/__init__.py:96): one vertex per splat with float32 properties x y z | nx ny nz (zeros) | f_dc_* |
    f_rest_* | opacity (logit) | scale_* | rot_* (raw, un-normalised), binary little-endian -- written directly (plyfile
    is not a dependency here).  The curve model keeps opacity per curve: it is expanded to its m splats, and `_scaling`
    holds (segment length, exp(width), exp(width)) as sampled, exactly what the reference's tensors of these names hold."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    xyz = gaussians._xyz.detach().float().cpu().numpy()
    P = xyz.shape[0]
    feats = gaussians.get_features.detach().float().cpu().numpy().reshape(P, -1)      # [P, (D+1)^2] (one channel)
    f_dc, f_rest = feats[:, :1], feats[:, 1:]
    op = gaussians.get_opacity.detach().float().cpu().numpy().reshape(P, 1)
    op = np.clip(op, 1e-12, 1.0 - 1e-7)
    logit = np.log(op / (1.0 - op))                                                     # inverse_sigmoid
    scale = gaussians._scaling.detach().float().cpu().numpy().reshape(P, -1)
    rot = gaussians._rotation.detach().float().cpu().numpy().reshape(P, -1)
    cols = np.concatenate([xyz, np.zeros_like(xyz), f_dc, f_rest, logit, scale, rot], axis=1).astype("<f4")
    names = ply_attribute_names(f_dc.shape[1], f_rest.shape[1], scale.shape[1], rot.shape[1])
    assert cols.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(cols).tobytes())
    return names


def read_ply_vertices(path):
    """Minimal reader of the float32 binary little-endian vertex table save_ply writes -> dict name -> [P] array."""
    with open(path, "rb") as f:
        blob = f.read()
    head, body = blob.split(b"end_header\n", 1)
    lines = head.decode("ascii").splitlines()
    if lines[0] != "ply" or "format binary_little_endian 1.0" not in lines:
        raise ValueError("read_ply_vertices: not a binary little-endian PLY")
    n = int([ln for ln in lines if ln.startswith("element vertex")][0].split()[-1])
    names = [ln.split()[-1] for ln in lines if ln.startswith("property float")]
    table = np.frombuffer(body, dtype="<f4", count=n * len(names)).reshape(n, len(names))
    return {nm: table[:, i] for i, nm in enumerate(names)}
