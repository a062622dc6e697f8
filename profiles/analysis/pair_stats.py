"""Work distribution of the pair-major unit-colour backward (csrc/render_unit_bwd.hip).  (CPU, numpy; no GPU needed.)

Replays the oracle's forward state of one view: per tile, the (splat, quadrant) pairs the kernel lists -- quadrant reached
(here: some pixel of the quadrant passes the exact alpha >= 1/255 test; the kernel's box test is conservative, so it lists a
few more) and list position in front of the deepest cut of that quadrant -- pooled per batch of 256 staged entries into
chunks of 32, against the per-quadrant lists in groups of 16 (walked eight at a time) of the pixel-major kernel.  Also: how
many chunks contain a pair behind the shallowest cut of its quadrant and therefore take the walk with the per-pixel position
test."""
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
ncon = np.asarray(fw.n_contrib).reshape(H, W)
gx = (W + 15) // 16
rng = np.random.default_rng(0)
tiles = np.arange(len(ranges))
if len(tiles) > max_tiles:
    tiles = rng.choice(tiles, max_tiles, replace=False)
yy, xx = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
pairs = padded_new = chunks = slow_chunks = padded_old = hits = inst = 0
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
    inside = ((px < W) & (py < H))
    hit = (power <= 0) & (alpha >= 1.0 / 255.0) & inside
    n = hit.shape[0]
    q_hit = hit.reshape(n, 2, 8, 2, 8).any(axis=(2, 4)).reshape(n, 4)          # [entry, quadrant (qy, qx)]
    # per-pixel cut: list position of the last blended entry (1-based) -> entries at 0-based positions < cut matter
    ys = np.clip(ty * 16 + yy, 0, H - 1); xs = np.clip(tx * 16 + xx, 0, W - 1)
    cut = np.where(inside.reshape(16, 16), ncon[ys, xs], 0).reshape(2, 8, 2, 8)
    qmax = cut.max(axis=(1, 3)).reshape(4)
    cin = np.where(inside.reshape(2, 8, 2, 8), cut, 1 << 30)
    qmin = cin.min(axis=(1, 3)).reshape(4)
    pos = np.arange(n)[:, None]
    listed = q_hit & (pos < qmax[None, :])
    behind = listed & (pos >= qmin[None, :])
    hits += int((hit & (pos < cut.reshape(1, -1))).sum())
    inst += int(listed.any(axis=1).sum())
    nb = int(min(n, qmax.max()))
    for b0 in range(0, nb, 256):
        l = listed[b0:b0 + 256]
        k = int(l.sum())
        pairs += k
        nch = (k + 31) // 32
        chunks += nch
        padded_new += nch * 32
        flat_slow = behind[b0:b0 + 256][l]                                    # pooled order: (entry, quadrant)
        for cidx in range(nch):
            slow_chunks += int(flat_slow[cidx * 32:(cidx + 1) * 32].any())
    for b0 in range(0, nb, 128):                                              # pixel-major kernel: 128 staged per round,
        l = listed[b0:b0 + 128]                                               # per-quadrant lists walked 8 at a time
        padded_old += int(((l.sum(0) + 7) // 8 * 8).sum())
print(f"{cfg}: {len(tiles)} tiles, {inst} instances with a listed pair, {pairs} pairs ({pairs / max(inst, 1):.2f} per instance), "
      f"{hits} (pixel, pair) hits = {hits / max(pairs * 64, 1):.3f} of the evaluated lanes")
print(f"  pair-major : {chunks} chunks of 32, {padded_new} evaluated pairs (x{padded_new / max(pairs, 1):.3f}); "
      f"{slow_chunks / max(chunks, 1):.3f} of the chunks take the walk with the per-pixel position test")
print(f"  pixel-major: {padded_old} evaluated pairs (x{padded_old / max(pairs, 1):.3f})")
