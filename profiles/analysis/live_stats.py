"""How many (quadrant, splat) pairs of the forward walk touch no LIVE pixel?  (CPU, numpy; no GPU needed.)

Replays the reference forward per tile (transmittance, early termination) and counts, per 8x8 quadrant, the list entries
that pass the quadrant's alpha >= 1/255 reach test (>= 1 pixel of the quadrant hit) up to the point where the whole
quadrant has terminated, split into: pairs that hit at least one pixel that is still alive, and pairs whose hits are all
on pixels that terminated earlier (work a wave-level "any live lane hit?" test could skip)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from util import oracle_forward          # noqa
from test_raster_gpu import _curve_splats  # noqa

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
max_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 600
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
qid = ((yy // 8) * 2 + (xx // 8)).reshape(-1)
walked = live_hit = dead_only = lane_evals = live_lane_hits = 0
for t in tiles:
    a, b = ranges[t]
    if b <= a:
        continue
    ids = pl[a:b]
    tx, ty = t % gx, t // gx
    px = (tx * 16 + xx).reshape(-1).astype(np.float32)
    py = (ty * 16 + yy).reshape(-1).astype(np.float32)
    inside = (px < W) & (py < H)
    T = np.ones(256, np.float32)
    alive = inside.copy()
    qdone = np.zeros(4, bool)
    for s in ids:
        dx = m2d[s, 0] - px; dy = m2d[s, 1] - py
        c = co[s]
        power = -0.5 * (c[0] * dx * dx + c[2] * dy * dy) - c[1] * dx * dy
        alpha = np.minimum(0.99, c[3] * np.exp(power))
        hit = (power <= 0) & (alpha >= 1.0 / 255.0) & inside
        for q in range(4):
            if qdone[q]:
                continue
            hq = hit & (qid == q)
            if not hq.any():
                continue
            walked += 1
            lane_evals += 64
            if (hq & alive).any():
                live_hit += 1
                live_lane_hits += int((hq & alive).sum())
            else:
                dead_only += 1
        test_T = T * (1 - alpha)
        stop = hit & alive & (test_T < 1e-4)
        blend = hit & alive & ~stop
        T = np.where(blend, test_T, T)
        alive &= ~stop
        for q in range(4):
            if not qdone[q] and not (alive & (qid == q)).any():
                qdone[q] = True
        if qdone.all():
            break
print(f"{cfg}: {len(tiles)} tiles; walked pairs {walked}; with a live hit {live_hit} ({live_hit / walked:.3f}); hits on dead pixels only {dead_only} ({dead_only / walked:.3f}); live-lane utilisation {live_lane_hits / lane_evals:.3f}")
