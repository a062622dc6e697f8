#!/usr/bin/env python
"""Generates tests/golden/*.npz by IMPORTING the reference's own Python (runs only in the authoring container where
/root/reference exists; the reference cannot travel to the GPU box).  Inputs are seeded; outputs are what the
reference's functions return on CPU.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Covers every importable reference function on or next to the hot path:
  utils/general_utils.py   rot_to_quat_batch :33-86, inverse_sigmoid :88-89, get_expon_lr_func :99-132
  utils/loss_utils.py      ssim :56-86 (value + d/dimg1), edge_aware_loss :94-115 (value + grad)
  utils/graphics_utils.py  getWorld2View2 :38-49, getProjectionMatrix :51-71, fov2focal/focal2fov
  utils/sh_utils.py        eval_sh, RGB2SH
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
from utils import general_utils as GU  # noqa: E402
from utils import graphics_utils as GR  # noqa: E402
from utils import loss_utils as LU  # noqa: E402
from utils import sh_utils as SH  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    g = torch.Generator().manual_seed(1234)
    # --- rot_to_quat_batch: true rotations, the curve model's degenerate [v0 | tiny | tiny] matrices, random 3x3
    q = torch.randn(64, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z),
                     1 - 2 * (x * x + z * z), 2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x),
                     1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    v0 = torch.randn(64, 3, generator=g)
    v0 = v0 / v0.norm(dim=1, keepdim=True)
    deg = torch.stack([v0, 1e-3 * torch.randn(64, 3, generator=g), 1e-3 * torch.randn(64, 3, generator=g)], dim=2)
    rnd = torch.randn(64, 3, 3, generator=g)
    mats = torch.cat([R, deg, rnd], 0)
    np.savez(os.path.join(OUT, "rot_to_quat.npz"), mats=mats.numpy(), quats=GU.rot_to_quat_batch(mats).numpy())

    # --- ssim (value, map mean, gradient wrt img1), a few shapes incl. non-multiples of the tile sizes
    ss = {}
    for i, (b, c, h, w) in enumerate([(1, 1, 48, 64), (2, 3, 37, 53), (1, 1, 100, 75)]):
        img1 = torch.rand(b, c, h, w, generator=g).requires_grad_(True)
        img2 = torch.rand(b, c, h, w, generator=g)
        val = LU.ssim(img1, img2)
        val.backward()
        ss[f"img1_{i}"] = img1.detach().numpy()
        ss[f"img2_{i}"] = img2.numpy()
        ss[f"val_{i}"] = val.detach().numpy()
        ss[f"grad_{i}"] = img1.grad.numpy()
    np.savez(os.path.join(OUT, "ssim.npz"), **ss)

    # --- edge_aware_loss
    img = torch.rand(1, 40, 56, generator=g).requires_grad_(True)
    gt = (torch.rand(1, 40, 56, generator=g) > 0.85).float() * torch.rand(1, 40, 56, generator=g)
    val = LU.edge_aware_loss(img, gt)
    val.backward()
    np.savez(os.path.join(OUT, "edge_aware_loss.npz"), image=img.detach().numpy(), gt=gt.numpy(),
             value=val.detach().numpy(), grad=img.grad.numpy())

    # --- camera matrices (scene/cameras.py:59-66 composition)
    Rm = R[:4].numpy().astype(np.float64)
    T = torch.randn(4, 3, generator=g).numpy().astype(np.float64)
    w2v = np.stack([GR.getWorld2View2(Rm[i], T[i]) for i in range(4)])
    proj = GR.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=0.6911, fovY=0.5).numpy()
    wv_t = torch.tensor(w2v).transpose(1, 2)
    full = wv_t.bmm(torch.tensor(proj).t().unsqueeze(0).expand(4, -1, -1))
    np.savez(os.path.join(OUT, "camera.npz"), R=Rm, T=T, world2view=w2v, projection=proj,
             world_view_transform=wv_t.numpy(), full_proj_transform=full.numpy(),
             camera_center=torch.linalg.inv(wv_t)[:, 3, :3].numpy(),
             fov2focal=np.array(GR.fov2focal(0.6911, 1600)), focal2fov=np.array(GR.focal2fov(2222.0, 1600)))

    # --- scalar helpers
    lr = GU.get_expon_lr_func(lr_init=0.0005, lr_final=0.000005, lr_delay_mult=0.01, max_steps=30000)
    steps = np.array([0, 1, 10, 500, 7000, 10000, 30000, 40000])
    xs = torch.linspace(0.01, 0.99, 25)
    np.savez(os.path.join(OUT, "scalars.npz"), lr_steps=steps, lr_values=np.array([lr(int(s)) for s in steps]),
             inv_sigmoid_x=xs.numpy(), inv_sigmoid_y=GU.inverse_sigmoid(xs).numpy())

    # --- SH evaluation (reference python path; 3-channel layout of eval_sh)
    for deg_ in (0, 1, 2, 3):
        pass
    dirs = torch.randn(32, 3, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    shd = {"dirs": dirs.numpy()}
    for deg_ in (0, 1, 2, 3):
        coeffs = torch.randn(32, 1, (deg_ + 1) ** 2, generator=g)
        shd[f"sh_{deg_}"] = coeffs.numpy()
        shd[f"out_{deg_}"] = SH.eval_sh(deg_, coeffs, dirs).numpy()
    np.savez(os.path.join(OUT, "sh.npz"), **shd)

    # --- EMAP frame -> camera (scene/dataset_readers.py:303-322 arithmetic on the reference's own graphics_utils,
    #     then scene/cameras.py:59-66), own generator so the fixtures above keep their random streams
    g2 = torch.Generator().manual_seed(4321)
    q = torch.randn(5, 4, generator=g2)
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q.unbind(1)
    Rc = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z),
                      1 - 2 * (x * x + z * z), 2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x),
                      1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3).double().numpy()
    em = {"H": np.array(600), "W": np.array(800)}
    c2ws, Ks, wvs, fulls, centers, fovs = [], [], [], [], [], []
    for i in range(5):
        c2w = np.eye(4)
        c2w[:3, :3] = Rc[i]
        c2w[:3, 3] = torch.randn(3, generator=g2).double().numpy() * 2.0
        K = np.eye(4)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 700.0 + 50 * i, 650.0 + 40 * i, 400.0, 300.0
        w2c = np.linalg.inv(c2w)
        Rr = np.transpose(w2c[:3, :3])
        Tt = w2c[:3, 3]
        fovy = GR.focal2fov(K[1, 1], 600)
        fovx = GR.focal2fov(K[0, 0], 800)
        wv = torch.tensor(GR.getWorld2View2(Rr, Tt)).transpose(0, 1)
        pr = GR.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = (wv.unsqueeze(0).bmm(pr.unsqueeze(0))).squeeze(0)
        c2ws.append(c2w); Ks.append(K); wvs.append(wv.numpy()); fulls.append(full.numpy())
        centers.append(wv.inverse()[3, :3].numpy()); fovs.append([fovx, fovy])
    np.savez(os.path.join(OUT, "emap_camera.npz"), camtoworld=np.stack(c2ws), intrinsics=np.stack(Ks),
             world_view_transform=np.stack(wvs), full_proj_transform=np.stack(fulls), camera_center=np.stack(centers),
             fov=np.array(fovs), **em)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
