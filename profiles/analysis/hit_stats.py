"""Lane-utilisation analysis for the tile compositors (CPU, numpy; no GPU needed).

For one view of a BASELINE config it runs the oracle forward, then for every 16x16 tile and every list entry
evaluates the reference's per-pixel acceptance test (power <= 0 and alpha >= 1/255, forward.cu:362-377) on all 256
pixels and reports, per granularity, how many lane evaluations a compositor that walks culled lists would do:

  tile      every list entry x 256 pixels                                  (the reference)
  quadrant  (8x8 quadrant, splat) pairs x 64 lanes                         (round-1 kernels)
  block4    (4x4 block, splat) pairs x 16 lanes, 4 blocks per wave, trips = max over the wave's 4 blocks
  pixel     per-pixel hit lists; a wave of 64 pixels (8x8) walks them in chunks of `chunk` list entries,
            trips per chunk = max over the 64 lanes of the lane's hits in the chunk
"""
import sys, os, math
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from util import oracle_forward          # noqa
from test_raster_gpu import _curve_splats  # noqa

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
sp, cam = _curve_splats(cfg, view)
H, W = cam.image_height, cam.image_width
fw = oracle_forward(sp, cam, torch.zeros(3))
m2d = fw.means2D; co = fw.conic_opacity; ranges = fw.ranges; pl = fw.point_list
gx = (W + 15) // 16
rng = np.random.default_rng(0)
tiles = np.arange(len(ranges))
if len(tiles) > max_tiles:
    tiles = rng.choice(tiles, max_tiles, replace=False)
yy, xx = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
tot = dict(entries=0, kept=0, hits=0, quad_pairs=0, b4_pairs=0, b4_trips=0)
pix_trips = {32: 0, 64: 0, 256: 0}
maxhits_hist = []
per_splat_hits = []
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
    hit = (power <= 0) & (alpha >= 1.0 / 255.0)          # [n,256]
    inside = ((px < W) & (py < H))
    hit &= inside
    tot["entries"] += len(ids)
    keep = hit.any(1)
    hit = hit[keep]
    n = hit.shape[0]
    tot["kept"] += n
    tot["hits"] += int(hit.sum())
    per_splat_hits.append(hit.sum(1))
    h = hit.reshape(n, 16, 16)
    # quadrants
    q = h.reshape(n, 2, 8, 2, 8).any(axis=(2, 4))         # [n,2,2]
    tot["quad_pairs"] += int(q.sum())
    b4 = h.reshape(n, 4, 4, 4, 4).any(axis=(2, 4))        # [n,4(y),4(x)] blocks
    tot["b4_pairs"] += int(b4.sum())
    # wave = quadrant: its 4 blocks
    cnt = b4.reshape(n, 2, 2, 2, 2).sum(0)                # [qy, by, qx, bx]
    tot["b4_trips"] += int(cnt.transpose(0, 2, 1, 3).reshape(4, 4).max(1).sum())
    # per-pixel lists, wave = 8x8 quadrant
    hq = h.reshape(n, 2, 8, 2, 8).transpose(0, 1, 3, 2, 4).reshape(n, 4, 64)   # [n, quadrant, lane]
    for ch in pix_trips:
        for s in range(0, n, ch):
            pix_trips[ch] += int(hq[s:s + ch].sum(0).max(1).sum())
    maxhits_hist.append(hq.sum(0).max())
P = len(tiles)
psh = np.concatenate(per_splat_hits)
print(f"{cfg} view {view}: {len(tiles)} tiles sampled, list entries {tot['entries']} (reference R), kept by exact tile test {tot['kept']}")
print(f"true (pixel, splat) hits: {tot['hits']}  = {tot['hits']/max(tot['kept'],1):.1f} per kept instance; per-instance hits: median {np.median(psh):.0f} p90 {np.percentile(psh,90):.0f} p99 {np.percentile(psh,99):.0f} max {psh.max()}")
le_tile = tot["kept"] * 256
le_quad = tot["quad_pairs"] * 64
print(f"lane-evals: tile {le_tile/1e6:.1f}M (util {tot['hits']/le_tile:.3f}); quadrant {le_quad/1e6:.1f}M, {tot['quad_pairs']} pairs = wave trips (util {tot['hits']/le_quad:.3f})")
print(f"block4: {tot['b4_pairs']} pairs, wave trips {tot['b4_trips']} (util {tot['hits']/(tot['b4_trips']*64):.3f})")
for ch, v in pix_trips.items():
    print(f"pixel lists, chunk {ch}: wave trips {v} (util {tot['hits']/(v*64):.3f}); vs quadrant trips x{tot['quad_pairs']/v:.2f}")
print(f"max hits per pixel in a tile: mean {np.mean(maxhits_hist):.1f} p99 {np.percentile(maxhits_hist,99):.0f}")
