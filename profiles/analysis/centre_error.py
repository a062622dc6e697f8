"""Error of the matrix-core exponent log2(alpha) = c0 + cu u + cv v + A2 u^2 + B2 uv + C2 v^2 (three-way bf16 split of the
coefficients, f32 accumulation: csrc/p2_mfma.h) as a function of the EXPANSION CENTRE: half-quadrant centre (the forward),
quadrant centre (the unit backward), tile centre (one coefficient set per splat and tile -- would remove the per-group operand
build from both compositors).  CPU emulation over the culled tile lists of one cfg3 view against float64; no GPU needed."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from util import oracle_forward          # noqa
from test_raster_gpu import _curve_splats  # noqa

f32 = np.float32
LOG2E = 1.4426950408889634


def split3(x):
    x = x.astype(f32)
    hi = (x.view(np.uint32) & 0xffff0000).view(f32)
    r1 = (x - hi).astype(f32)
    mid = (r1.view(np.uint32) & 0xffff0000).view(f32)
    r2 = (r1 - mid).astype(f32)
    lo = (r2.view(np.uint32) & 0xffff0000).view(f32)
    return hi, mid, lo


def mfma_eval(c0, cu, cv, A2, B2, C2, u, v):
    """coefficients [n,1] f32, monomials [1,p]; products exact (f64), accumulated in f32 in K-slot order."""
    acc = np.zeros((c0.shape[0], u.shape[1]), f32)
    terms = [(c0, np.ones_like(u)), (cv, v), (B2, u * v), (cu, u), (A2, u * u), (C2, v * v)]
    for coef, mono in terms:
        for part in split3(coef):
            acc = (acc.astype(np.float64) + part.astype(np.float64) * mono.astype(np.float64)).astype(f32)
    return acc


cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
sp, cam = _curve_splats(cfg, 0)
fw = oracle_forward(sp, cam, torch.zeros(3))
m2d, co, ranges, pl = fw.means2D, fw.conic_opacity, fw.ranges, fw.point_list
gx = (cam.image_width + 15) // 16
rng = np.random.default_rng(0)
tiles = rng.choice(len(ranges), 400, replace=False)
yy, xx = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
res = {k: [] for k in ("half", "quadrant", "tile", "direct_f32")}
for t in tiles:
    a, b = ranges[t]
    if b <= a:
        continue
    ids = pl[a:b]
    tx, ty = t % gx, t // gx
    px = (tx * 16 + xx).reshape(1, -1).astype(np.float64)
    py = (ty * 16 + yy).reshape(1, -1).astype(np.float64)
    cx, cy = m2d[ids, 0:1].astype(np.float64), m2d[ids, 1:2].astype(np.float64)
    A, B, C, op = (co[ids, k:k + 1].astype(np.float64) for k in range(4))
    A2, B2, C2 = (f32(-0.5 * LOG2E) * A.astype(f32)), (f32(-LOG2E) * B.astype(f32)), (f32(-0.5 * LOG2E) * C.astype(f32))
    l2op = np.log2(op.astype(f32)).astype(f32)
    dx, dy = cx - px, cy - py
    exact = A2.astype(np.float64) * dx * dx + B2.astype(np.float64) * dx * dy + C2.astype(np.float64) * dy * dy + l2op.astype(np.float64)
    keep = exact > -14.0
    # direct f32 evaluation of the same form
    dxf, dyf = (cx.astype(f32) - px.astype(f32)), (cy.astype(f32) - py.astype(f32))
    direct = (dxf * (A2 * dxf + B2 * dyf) + C2 * dyf * dyf + l2op).astype(f32)
    res["direct_f32"].append(np.abs(direct - exact)[keep])
    for name, (sx, sy, ox, oy) in {"half": (8, 4, 3.5, 1.5), "quadrant": (8, 8, 3.5, 3.5), "tile": (16, 16, 7.5, 7.5)}.items():
        out = np.zeros_like(exact, dtype=f32)
        for by in range(0, 16, sy):
            for bx in range(0, 16, sx):
                hx, hy = f32(tx * 16 + bx + ox), f32(ty * 16 + by + oy)
                dxc, dyc = (cx.astype(f32) - hx), (cy.astype(f32) - hy)
                c0 = (dxc * (A2 * dxc + B2 * dyc) + C2 * dyc * dyc + l2op).astype(f32)
                cu = (-(f32(2) * A2 * dxc + B2 * dyc)).astype(f32)
                cv = (-(B2 * dxc + f32(2) * C2 * dyc)).astype(f32)
                sel = ((yy >= by) & (yy < by + sy) & (xx >= bx) & (xx < bx + sx)).reshape(-1)
                u = (px[:, sel] - float(hx)).astype(f32)
                v = (py[:, sel] - float(hy)).astype(f32)
                out[:, sel] = mfma_eval(c0, cu, cv, A2, B2, C2, u, v)
        res[name].append(np.abs(out - exact)[keep])
for k, v in res.items():
    e = np.concatenate(v)
    print(f"{k:11s} max |err| {e.max():.2e}  99.99 % {np.quantile(e, 0.9999):.2e}  mean {e.mean():.2e}   (log2 units; x ln 2 = relative error of alpha)  pairs {len(e)}")
