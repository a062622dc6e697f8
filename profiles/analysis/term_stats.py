"""How much of each tile list the compositors really need (CPU, numpy): early termination statistics.

Per 8x8 quadrant (= one wave) of sampled tiles: accepted (quadrant, splat) pairs in the whole list, pairs before the
quadrant's last blended position (what the backward must visit), pairs up to the point where the forward may stop when
it checks "all 64 pixels terminated" every `g` list entries."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from util import oracle_forward
from test_raster_gpu import _curve_splats
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
sp, cam = _curve_splats(cfg, view)
H, W = cam.image_height, cam.image_width
fw = oracle_forward(sp, cam, torch.zeros(3))
m2d = fw.means2D; co = fw.conic_opacity; ranges = fw.ranges; pl = fw.point_list
ncon = fw.n_contrib; fT = fw.final_T
gx = (W + 15) // 16
rng = np.random.default_rng(0)
tiles = np.arange(len(ranges))
if len(tiles) > max_tiles:
    tiles = rng.choice(tiles, max_tiles, replace=False)
yy, xx = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
S = dict(pairs=0, bwd_pairs=0, bwd_lane_active=0, fwd_ideal=0, entries=0)
fwd_g = {1: 0, 8: 0, 16: 0, 32: 0, 64: 0}
fwd_lane_live = 0
for t in tiles:
    a, b = ranges[t]
    if b <= a: continue
    ids = pl[a:b]; n = len(ids)
    tx, ty = t % gx, t // gx
    px = (tx * 16 + xx).reshape(1, -1).astype(np.float32); py = (ty * 16 + yy).reshape(1, -1).astype(np.float32)
    dx = m2d[ids, 0:1] - px; dy = m2d[ids, 1:2] - py; c = co[ids]
    power = -0.5 * (c[:, 0:1] * dx * dx + c[:, 2:3] * dy * dy) - c[:, 1:2] * dx * dy
    alpha = np.minimum(0.99, c[:, 3:4] * np.exp(power))
    hit = (power <= 0) & (alpha >= 1.0 / 255.0) & (px < W) & (py < H)
    S["entries"] += n
    h = hit.reshape(n, 2, 8, 2, 8).transpose(0, 1, 3, 2, 4).reshape(n, 4, 64)
    qacc = h.any(2)                                  # [n,4] accepted pairs
    S["pairs"] += int(qacc.sum())
    nc = ncon[ty*16:ty*16+16, tx*16:tx*16+16]
    ncp = np.zeros((16, 16), np.int64); ncp[:nc.shape[0], :nc.shape[1]] = nc
    ncq = ncp.reshape(2, 8, 2, 8).transpose(0, 2, 1, 3).reshape(4, 64)    # per lane last blended (1-based)
    wave_last = ncq.max(1)
    pos = np.arange(n)[:, None]
    need_b = qacc & (pos < wave_last[None, :])
    S["bwd_pairs"] += int(need_b.sum())
    lane_act = h & (pos[:, :, None] < ncq[None, :, :])
    S["bwd_lane_active"] += int(lane_act.sum())
    S["bwd_pairs_live"] = S.get("bwd_pairs_live", 0) + int(lane_act.any(2).sum())
    after = h & (pos[:, :, None] >= ncq[None, :, :])
    anyafter = after.any(0)
    death = np.where(anyafter, after.argmax(0), n)            # [4,64] entry index at which the lane dies (n = never)
    inside = ((px < W) & (py < H)).reshape(16, 16).reshape(2, 8, 2, 8).transpose(0, 2, 1, 3).reshape(4, 64)
    death = np.where(inside, death, -1)
    wave_death = death.max(1)                                 # all lanes dead after this entry index
    for q in range(4):
        stop_ideal = min(n, wave_death[q] + 1)
        S["fwd_ideal"] += int(qacc[:stop_ideal, q].sum())
        for g in fwd_g:
            stop = min(n, ((wave_death[q] + 1 + g - 1) // g) * g) if wave_death[q] < n else n
            fwd_g[g] += int(qacc[:stop, q].sum())
        live = h[:stop_ideal, q, :] & (np.arange(stop_ideal)[:, None] <= death[q][None, :])
        fwd_lane_live += int(live.sum())
print(f"{cfg} view {view}, {len(tiles)} tiles: list entries {S['entries']}, accepted (quadrant,splat) pairs {S['pairs']}")
print(f"backward: pairs before wave_last {S['bwd_pairs']} ({S['bwd_pairs']/S['pairs']:.3f} of all); active lanes per visited pair {S['bwd_lane_active']/max(S['bwd_pairs'],1):.1f}/64")
print(f"backward: pairs with at least one active lane {S['bwd_pairs_live']} ({S['bwd_pairs_live']/S['pairs']:.3f} of all accepted)")
print(f"forward: ideal stop {S['fwd_ideal']} ({S['fwd_ideal']/S['pairs']:.3f}); live-hit lanes per visited pair {fwd_lane_live/max(S['fwd_ideal'],1):.1f}/64")
for g, v in fwd_g.items():
    print(f"   check every {g:3d} entries: {v} pairs ({v/S['pairs']:.3f})")
