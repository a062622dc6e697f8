"""Tile-list length distribution after exact tile culling (CPU replay of the oracle's lists; no GPU needed).
Which share of the tiles / instances a single-batch (n <= 256) fast path of the forward compositor would cover."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from util import oracle_forward          # noqa
from test_raster_gpu import _curve_splats  # noqa

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sp, cam = _curve_splats(cfg, view)
H, W = cam.image_height, cam.image_width
fw = oracle_forward(sp, cam, torch.zeros(3))
m2d = fw.means2D; co = fw.conic_opacity; ranges = fw.ranges; pl = fw.point_list
gx = (W + 15) // 16
yy, xx = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
lens = []
for t in range(len(ranges)):
    a, b = ranges[t]
    if b <= a:
        lens.append(0); continue
    ids = pl[a:b]
    tx, ty = t % gx, t // gx
    px = (tx * 16 + xx).reshape(1, -1).astype(np.float32)
    py = (ty * 16 + yy).reshape(1, -1).astype(np.float32)
    dx = m2d[ids, 0:1] - px
    dy = m2d[ids, 1:2] - py
    c = co[ids]
    power = -0.5 * (c[:, 0:1] * dx * dx + c[:, 2:3] * dy * dy) - c[:, 1:2] * dx * dy
    alpha = c[:, 3:4] * np.exp(power)
    lens.append(int((alpha >= 1 / 255.).any(1).sum()))
lens = np.array(lens)
print(cfg, "view", view, "tiles", len(lens), "R(culled, approx)", lens.sum(), "mean", lens.mean(), "max", lens.max())
for thr in (64, 128, 192, 256, 320, 384, 512, 768, 1024):
    m = lens <= thr
    print(f"  n <= {thr:5d}: {m.mean()*100:6.2f} % of tiles, {lens[m].sum()/lens.sum()*100:6.2f} % of instances")
print("  percentiles 50/90/99:", np.percentile(lens, [50, 90, 99]))
