#!/usr/bin/env python
"""Freezes the rasterizer ORACLE (oracle/raster_ref.c) on four small scenes -> tests/golden/raster_{small,ties,opaque,room}.npz.

What these files are, and what they are not.  The reference rasterizer ships no tests or vectors and cannot be built here
(nvcc / cub / GLM absent), so nothing can pin the oracle to the reference's binary output: these fixtures do NOT change the
"parity unpinned" status of K1-K10.  They pin the oracle AND the kernels to a point in time: the oracle is
rebuilt from source on every box, so without a frozen copy the two could drift together unnoticed.  `-m "not gpu"` tests compare
today's oracle build with the file, `-m gpu` tests compare the HIP path with the file.

    python tests/golden/make_raster_golden.py          # rewrites the four files (do this ONLY for a deliberate oracle change)

Each file holds: the splat tensors, camera matrices, background and upstream gradients (inputs); colour / inverse depth /
all_map / radii / final_T / n_contrib / num_rendered / tile ranges / per-tile lists (forward, reference binning: no tile culling);
all nine gradients of rasterize_gaussians_backward plus dL/dconic (backward).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from curve_gaussian_amd import synthetic as S  # noqa: E402
from oracle import raster as ORA  # noqa: E402
from util import oracle_forward, tanfov  # noqa: E402


def scenes():
    """name -> (splats, camera, background).  Sizes keep the four files near 1 MB together."""
    out = {}
    # small: generic anisotropic cloud, grey background
    sp = S.random_splats(300, 301, scale_range=(0.01, 0.06))
    out["small"] = (sp, S.make_camera((0.5, -1.6, 0.7), (0.5, 0.5, 0.5), (0, 0, 1), 48, 64), torch.tensor([0.3, 0.0, 0.0]))
    # ties: 15 coincident copies of 40 positions -> equal depths inside every tile list (stable order = splat index)
    sp = S.random_splats(600, 302, scale_range=(0.01, 0.05))
    sp["means3D"] = sp["means3D"][torch.arange(600) % 40].contiguous()
    out["ties"] = (sp, S.make_camera((0.5, -1.6, 0.7), (0.5, 0.5, 0.5), (0, 0, 1), 64, 64), torch.zeros(3))
    # opaque: a third of the splats at opacity 0.995 / 0.9995 -> the 0.99 clamp and early termination (T < 1e-4) everywhere
    sp = S.random_splats(500, 303, scale_range=(0.02, 0.09))
    sp["opacities"][0::3] = 0.995
    sp["opacities"][1::6] = 0.9995
    out["opaque"] = (sp, S.make_camera((2.0, 1.4, 1.1), (0.4, 0.5, 0.6), (0, 0, 1), 64, 80), torch.tensor([0.1, 0.0, 0.0]))
    # room: camera INSIDE the cloud -- near culls, screen-filling splats, long lists
    sp = S.random_splats(1500, 304, scale_range=(0.006, 0.08))
    out["room"] = (sp, S.make_camera((0.5, 0.5, 0.5), (0.9, 0.2, 0.5), (0, 0, 1), 96, 128), torch.zeros(3))
    return out


def freeze(name, sp, cam, bg):
    H, W = cam.image_height, cam.image_width
    g = torch.Generator().manual_seed(1000 + len(name))
    dcol, dinv, damap = (torch.randn(1, H, W, generator=g), torch.randn(1, H, W, generator=g), torch.randn(4, H, W, generator=g))
    fw = oracle_forward(sp, cam, bg)
    gr = ORA.backward(fw, dcol.numpy(), dinv.numpy(), damap.numpy())
    gr_train = ORA.backward(fw, dcol.numpy(), None, None)   # the training configuration: only dL/dcolour flows in
    tfx, tfy = tanfov(cam)
    d = dict(
        # inputs
        means3D=sp["means3D"].numpy(), scales=sp["scales"].numpy(), rotations=sp["rotations"].numpy(),
        opacities=sp["opacities"].numpy(), colors=sp["colors"].numpy(), all_map=sp["all_map"].numpy(), bg=bg.numpy(),
        viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
        campos=cam.camera_center.numpy(), tanfov=np.array([tfx, tfy], np.float64), hw=np.array([H, W], np.int64),
        FoV=np.array([cam.FoVx, cam.FoVy], np.float64),
        dL_dcolor=dcol.numpy(), dL_dinvdepth=dinv.numpy(), dL_dout_all_map=damap.numpy(),
        # forward
        color=fw.color, invdepth=fw.invdepth, out_all_map=fw.out_all_map, radii=fw.radii.astype(np.int32),
        final_T=fw.final_T, n_contrib=fw.n_contrib.astype(np.uint32), num_rendered=np.array([fw.num_rendered], np.int64),
        ranges=fw.ranges.astype(np.uint32), point_list=fw.point_list.astype(np.uint32),
        means2D=fw.means2D, conic_opacity=fw.conic_opacity, depths=fw.depths,
    )
    for k, v in gr.items():
        if v is not None and k != "dL_dsh":
            d["g_" + k] = np.asarray(v, np.float32)
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dconic"):
        d["gt_" + k] = np.asarray(gr_train[k], np.float32)
    fw.free()
    path = os.path.join(HERE, f"raster_{name}.npz")
    np.savez_compressed(path, **d)
    print(f"{name}: P={sp['means3D'].shape[0]} {W}x{H} R={int(d['num_rendered'][0])} visible={(d['radii'] > 0).sum()} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def load_scene(name):
    """-> (splat dict of torch tensors, camera, background, npz) for the tests."""
    z = np.load(os.path.join(HERE, f"raster_{name}.npz"))
    sp = {k: torch.from_numpy(z[k].copy()) for k in ("means3D", "scales", "rotations", "opacities", "colors", "all_map")}
    H, W = (int(v) for v in z["hw"])
    cam = S.SynthCamera(H, W, float(z["FoV"][0]), float(z["FoV"][1]), torch.from_numpy(z["viewmatrix"].copy()),
                        torch.from_numpy(z["projmatrix"].copy()), torch.from_numpy(z["campos"].copy()))
    return sp, cam, torch.from_numpy(z["bg"].copy()), z


if __name__ == "__main__":
    ORA.set_num_threads(1)   # one thread: the double-precision per-splat sums are then accumulated in a fixed order
    for name, (sp, cam, bg) in scenes().items():
        freeze(name, sp, cam, bg)
