"""Shared helpers for the parity tests: run the CPU oracle and the HIP path on identical inputs and compare."""
import math

import numpy as np
import torch

from curve_gaussian_amd import synthetic as S
from oracle import raster as ORA

REL_TOL = 1e-4          # north-star tolerance (BASELINE.json: "within 1e-4 rel")
OUTLIER_FRAC = 1e-4     # budget for alpha<1/255 / T<1e-4 threshold flips under different expf rounding


def tanfov(cam):
    return math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)


def oracle_forward(sp, cam, bg, render_geo=True, antialiasing=False, scale_modifier=1.0, cov3D=None, sh=None, degree=0):
    tfx, tfy = tanfov(cam)
    n = lambda t: None if t is None else t.detach().cpu().numpy()
    use_cov = cov3D is not None
    return ORA.forward(n(bg), n(sp["means3D"]), None if sh is not None else n(sp["colors"]), n(sp["opacities"]),
                       None if use_cov else n(sp["scales"]), None if use_cov else n(sp["rotations"]), scale_modifier,
                       n(cov3D), n(sp["all_map"]), n(cam.world_view_transform), n(cam.full_proj_transform), tfx, tfy,
                       cam.image_height, cam.image_width, n(sh), degree, n(cam.camera_center),
                       antialiasing=antialiasing, render_geo=render_geo)


# per-call option bits (curve_gaussian_amd.diff_cur_rasterization.OPT_*) the helpers below put into every settings object they
# build: fixtures set them for one test (the library has no process-wide switches)
OPTIONS = [0]


def hip_settings(cam, bg, dev, render_geo=True, antialiasing=False, scale_modifier=1.0, degree=0, debug=False):
    from curve_gaussian_amd.diff_cur_rasterization import GaussianRasterizationSettings
    tfx, tfy = tanfov(cam)
    c = cam.to(dev)
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=tfx, tanfovy=tfy, bg=bg.to(dev),
        scale_modifier=scale_modifier, viewmatrix=c.world_view_transform, projmatrix=c.full_proj_transform,
        sh_degree=degree, campos=c.camera_center, prefiltered=False, debug=debug, antialiasing=antialiasing,
        render_geo=render_geo, options=OPTIONS[0])


def close_frac(a, b, rel=REL_TOL, abs_floor=None):
    """fraction of elements with |a-b| > rel * max|b| (+ floor), and the worst normalised error."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0, 0.0
    scale = np.abs(b).max()
    tol = rel * scale + (abs_floor if abs_floor is not None else 1e-7)
    err = np.abs(a - b)
    return float((err > tol).mean()), float(err.max() / (scale + 1e-30))


MAX_OUTLIER = {"image": 5e-3}   # one alpha >= 1/255 flip moves a pixel by at most 1/255 = 3.9e-3 of the (unit) maximum


REL_BIG = 1e-3          # relative tolerance on the elements that matter ...
BIG_FRAC = 1e-2         # ... those above this fraction of the tensor's maximum


def assert_close(name, a, b, rel=REL_TOL, outlier_frac=OUTLIER_FRAC, abs_floor=None, max_outlier=None, tile_cluster=8,
                 rel_big=REL_BIG, big_frac=BIG_FRAC, outlier_frac_big=None, min_outliers=0, max_outlier_abs=None):
    """|a - b| <= rel * max|b| on all but `outlier_frac` of the elements -- the budget for alpha >= 1/255 and T < 1e-4 decisions
    that flip under a different rounding of the exponent -- AND the budget is capped in magnitude and in space:

    * max_outlier: no element may be off by more than this fraction of max|b|.  Default 5e-3 for image-shaped tensors
      ([C,H,W] with H, W >= 16: one threshold flip moves a pixel by <= 1/255 of the maximum, so 5e-3 admits a single flip and
      nothing systematic); per-splat / per-curve gradient tensors have no universal bound (a flipped pair changes ONE splat's
      sums by a large fraction while the tensor's maximum sits elsewhere) -- callers pass the cap they measured.
    * tile_cluster: for image-shaped tensors, at most this many outliers inside one 16x16 tile -- flips are isolated
      pixels; a fault in a rare branch (a border quadrant, the long-list sort path, a chunk's padding lane) shows up as a
      cluster and must not hide in the fraction.
    * min_outliers: the fractional budgets round to zero on small tensors (a 47 x 150 image: 0.7 pixels), where a single
      flipped decision is still a legal outcome -- campaigns over tiny random scenes pass the element count of one or two
      flips here; the suite's fixed cases keep 0."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    frac, worst = close_frac(a, b, rel, abs_floor)
    assert frac * a.size <= max(outlier_frac * a.size, min_outliers) + 1e-9, \
        f"{name}: {frac:.2e} of elements exceed rel tol {rel} (worst normalised err {worst:.3e})"
    # `rel * max|b|` is an ABSOLUTE tolerance: an element at 1 % of the maximum may be off by 1 % of itself and pass.  So, on top:
    # every element above big_frac of the maximum agrees to rel_big RELATIVE to itself, within the same outlier budget
    # (counted against the whole tensor, like the flips above; `outlier_frac_big` when the caller measured another budget for
    # this criterion).  Elements below big_frac of the maximum are held by the
    # absolute criterion only: their relative error is dominated by cancellation in sums the two sides order differently.
    if outlier_frac_big is None:
        outlier_frac_big = outlier_frac
    if rel_big is not None and a.size:
        scale = np.abs(b).max()
        big = np.abs(b) > big_frac * scale
        bad_rel = big & (np.abs(a - b) > rel_big * np.abs(b))
        n_bad = int(bad_rel.sum())
        assert n_bad <= max(outlier_frac_big * a.size, min_outliers), (
            f"{name}: {n_bad} of {int(big.sum())} elements above {big_frac:g} of the maximum differ by more than {rel_big:g} "
            f"relative (budget {outlier_frac_big * a.size:.1f}); worst {float((np.abs(a - b)[big] / np.abs(b)[big]).max()):.3e}")
    image_like = a.ndim == 3 and a.shape[1] >= 16 and a.shape[2] >= 16
    if max_outlier_abs is not None:   # the cap in the tensor's own units (a flip moves a pixel by <= 1/255 of the SPLAT's value:
        # on a scene of nine splats the image maximum is a fraction of that, and a cap relative to it means nothing)
        assert worst * np.abs(b).max() <= max_outlier_abs, f"{name}: worst element off by {worst * np.abs(b).max():.3e} (cap {max_outlier_abs:.1e})"
    elif max_outlier is None and image_like:
        max_outlier = MAX_OUTLIER["image"]
    if max_outlier is not None:
        assert worst <= max_outlier, f"{name}: worst element off by {worst:.3e} of max (cap {max_outlier:.1e})"
    if image_like and tile_cluster is not None and a.size:
        scale = np.abs(b).max()
        bad = (np.abs(a - b) > rel * scale + (abs_floor if abs_floor is not None else 1e-7)).any(0)
        if bad.any():
            Hh, Ww = bad.shape
            pad = np.zeros(((Hh + 15) // 16 * 16, (Ww + 15) // 16 * 16), bool)
            pad[:Hh, :Ww] = bad
            per_tile = pad.reshape(pad.shape[0] // 16, 16, pad.shape[1] // 16, 16).sum((1, 3))
            assert per_tile.max() <= tile_cluster, (f"{name}: {int(per_tile.max())} outliers inside one 16x16 tile "
                                                    f"(tile {np.unravel_index(per_tile.argmax(), per_tile.shape)}): clustered")
    return worst


def near_threshold_pairs(fw, window=1e-5):
    """Decisions of an oracle forward that may legally fall the other way under another rounding of the exponent (the HIP
    path's differs by ~2e-6): (pixel, splat) pairs whose alpha lies within `window` (relative) of the 1/255 cut
    (forward.cu:369), and pairs whose transmittance test T (1 - alpha) < 1e-4 (forward.cu:374) is decided within `window`.
    Plain numpy over the oracle's tile lists, float32 in list order like the kernel; for campaign-sized scenes only."""
    H, W = fw.H, fw.W
    tiles_x = (W + 15) // 16
    m2, co, pl, ranges = fw.means2D.astype(np.float32), fw.conic_opacity.astype(np.float32), fw.point_list, fw.ranges
    near = 0
    for t, (r0, r1) in enumerate(ranges):
        if r1 <= r0:
            continue
        ty, tx = divmod(t, tiles_x)
        ys, xs = np.arange(ty * 16, min(H, ty * 16 + 16), dtype=np.float32), np.arange(tx * 16, min(W, tx * 16 + 16), dtype=np.float32)
        px, py = np.meshgrid(xs, ys)
        ids = pl[r0:r1]
        dx = m2[ids, 0][:, None] - px.reshape(1, -1)
        dy = m2[ids, 1][:, None] - py.reshape(1, -1)
        c = co[ids]
        power = np.float32(-0.5) * (c[:, 0:1] * dx * dx + c[:, 2:3] * dy * dy) - c[:, 1:2] * dx * dy
        alpha = np.minimum(np.float32(0.99), c[:, 3:4] * np.exp(power))
        blends = (power <= 0) & (alpha >= np.float32(1.0 / 255.0))
        Tn = np.cumprod(np.where(blends, np.float32(1.0) - alpha, np.float32(1.0)), axis=0, dtype=np.float32)   # T after pair j
        alive = np.vstack([np.ones((1, Tn.shape[1]), bool), Tn[:-1] >= np.float32(1e-4)])                       # pair j is reached
        alive = np.logical_and.accumulate(alive, axis=0)
        near += int((alive & (power <= 0) & (np.abs(alpha * np.float32(255.0) - 1.0) < window)).sum())
        near += int((alive & blends & (np.abs(Tn * np.float32(1e4) - 1.0) < window)).sum())
    return near


def carve_offsets(base_ptr_mod, counts_and_sizes):
    """Replicates csrc/common.h::carve (128-byte aligned carve-outs) for tests that decode the scratch buffers."""
    offs = []
    cur = base_ptr_mod
    for count, size in counts_and_sizes:
        cur = (cur + 127) & ~127
        offs.append(cur - base_ptr_mod)
        cur += count * size
    return offs


# ---------------------------------------------------------------------------------------------------------------------------------
# tests/golden/topology.npz (make_topology_golden.py: the REFERENCE's own topology edits run on the CPU) replayed on a model that
# offers the reference's method names -- oracle/topology_ref.RefCurveModel on the CPU, the product's GaussianCurveModel on the GPU
TOPOLOGY_GROUPS = (("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"), ("width", "_width"),
                   ("curve_points", "_curve_points"), ("mask", "_mask"))
TOPOLOGY_TAGS = ("setup", "prune_curves", "reset_opacity", "adam_after_reset", "only_prune", "mask_trim_split", "adam_after_trim",
                 "fix_opacity")


def replay_topology_fixture(model, z, dev, check):
    """Runs the fixture's sequence of Adam steps and edits on `model`; check(tag) is called after every recorded state."""
    import torch
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    step = [0]

    def adam_step():
        k = step[0]
        step[0] += 1
        for name, attr in TOPOLOGY_GROUPS:
            p = getattr(model, attr)
            g = t(f"grad{k}_{name}")
            assert tuple(g.shape) == tuple(p.shape), (k, name, tuple(g.shape), tuple(p.shape))
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
        model.optimizer.step()
        model.prepare_scaling_rot()
    model.xyz_gradient_accum = t("in_xyz_gradient_accum").clone()
    model.denom = t("in_denom").clone()
    model.tmp_radii = t("in_tmp_radii").clone()
    for _ in range(3):
        adam_step()
    check("setup")
    model.prune_curves(t("prune_mask"))
    check("prune_curves")
    model.reset_opacity()
    check("reset_opacity")
    adam_step()
    check("adam_after_reset")
    with torch.no_grad():
        model._opacity.add_(t("opacity_bump"))
    model.only_prune(0.12, 0.3)
    check("only_prune")
    model.mask_trim_split(0.4)
    check("mask_trim_split")
    adam_step()
    check("adam_after_trim")
    model.fix_opacity()
    check("fix_opacity")
    assert model._opacity.requires_grad == bool(z["fix_opacity.requires_grad"]) == False
    lr = model.optimizer.lr_of("opacity") if hasattr(model.optimizer, "lr_of") else \
        [grp["lr"] for grp in model.optimizer.param_groups if grp["name"] == "opacity"][0]
    assert float(lr) == float(z["fix_opacity.lr_opacity"]) == 0.0
