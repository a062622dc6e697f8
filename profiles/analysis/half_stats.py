"""Would per-HALF-quadrant lists pay?  (CPU, numpy; no GPU needed.)

The matrix-core exponent tile (csrc/p2_mfma.h) gives lanes 0-31 (rows 0-3 of the 8x8 quadrant) and lanes 32-63 (rows
4-7) independent splat rows, so a wave could walk TWO lists at once, one per half, in max(n0, n1) trips instead of the
|union| trips of the quadrant list.  Also: quadrant split into left / right 4-column halves, and the 16-lane quarters.
Counts use the exact per-pixel acceptance test (alpha >= 1/255), like hit_stats.py."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from util import oracle_forward          # noqa
from test_raster_gpu import _curve_splats  # noqa

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
max_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
sp, cam = _curve_splats(cfg, 0)
H, W = cam.image_height, cam.image_width
fw = oracle_forward(sp, cam, torch.zeros(3))
m2d = fw.means2D; co = fw.conic_opacity; ranges = fw.ranges; pl = fw.point_list
gx = (W + 15) // 16
rng = np.random.default_rng(0)
tiles = np.arange(len(ranges))
if len(tiles) > max_tiles:
    tiles = rng.choice(tiles, max_tiles, replace=False)
yy, xx = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
hits = union = rows_max = cols_max = quarter_max = 0
for t in tiles:
    a, b = ranges[t]
    if b <= a:
        continue
    ids = pl[a:b]
    tx, ty = t % gx, t // gx
    px = (tx * 16 + xx).reshape(1, -1).astype(np.float32)
    py = (ty * 16 + yy).reshape(1, -1).astype(np.float32)
    dx = m2d[ids, 0:1] - px
    dy = m2d[ids, 1:2] - py
    c = co[ids]
    power = -0.5 * (c[:, 0:1] * dx * dx + c[:, 2:3] * dy * dy) - c[:, 1:2] * dx * dy
    alpha = np.minimum(0.99, c[:, 3:4] * np.exp(power))
    hit = (power <= 0) & (alpha >= 1.0 / 255.0) & ((px < W) & (py < H))
    n = hit.shape[0]
    h = hit.reshape(n, 2, 8, 2, 8)                      # [n, qy, y, qx, x]
    hits += int(hit.sum())
    union += int(h.any(axis=(2, 4)).sum())
    hr = h.reshape(n, 2, 2, 4, 2, 8).any(axis=(3, 5))   # [n, qy, half(rows), qx]
    rows_max += int(hr.sum(0).max(axis=1).sum())
    hc = h.reshape(n, 2, 8, 2, 2, 4).any(axis=(2, 5))   # [n, qy, qx, half(cols)]
    cols_max += int(hc.sum(0).max(axis=2).sum())
    hq = h.reshape(n, 2, 4, 2, 2, 8).any(axis=(3, 5))   # [n, qy, quarter(2 rows), qx]
    quarter_max += int(hq.sum(0).max(axis=1).sum())
print(f"{cfg}: {len(tiles)} tiles; true hits {hits}")
for name, trips in (("quadrant lists (now)", union), ("two row-halves, max(n0,n1)", rows_max),
                    ("two column-halves", cols_max), ("four 2-row quarters", quarter_max)):
    print(f"  {name:32s} wave trips {trips:9d}  lane utilisation {hits / (trips * 64):.3f}  trips vs now x{trips / union:.3f}")
