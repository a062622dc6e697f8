#!/usr/bin/env python
"""Generates tests/golden/prepare_scaling_rot.npz and tests/golden/curve_fitting.npz by IMPORTING the reference's own Python
and calling its functions unmodified, on CPU (runs only in the authoring container where /root/reference exists; the
reference cannot travel to the GPU box -- the fixtures, numeric arrays only, do).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_model_golden.py

What is called (SURVEY.md section 8a rows 1 and 2, and the helpers of the topology step next to the path):
  scene/gaussian_curve_model.py   GaussianCurveModel.prepare_scaling_rot :180-198 (with get_curve_gaussians :70-79,
                                  get_curve_tangent :81-91, utils/general_utils.rot_to_quat_batch), get_opacity :110-113,
                                  get_curve_width :119-121 -- values and, through torch autograd over the reference's own
                                  code, the gradients of a seeded linear functional w.r.t. the control points and log-widths
  edge_extraction/fitting.py      line_fitting :28-50, bezier_fit :52-71, fit_straight_line :74-99
  edge_extraction/merging.py      compute_pairwise_cosine_similarity :58-61, compute_pairwise_distances :84-108

How it is imported.  Those three modules import, at module level and for OTHER functions (point-cloud I/O, plotting,
RANSAC, the CUDA extensions), packages this image does not have: open3d, plyfile, seaborn, simple_knn, pytorch3d,
skimage, cv2 (the script prints the list).  The imports are satisfied by EMPTY placeholder modules, and the placeholders are poisoned
before anything is called: any attribute looked up on them from then on raises.  So every number written below was
computed by the reference's unmodified code on top of torch / numpy / scipy / scikit-learn / einops, which ARE here, and
this script fails loudly if a called function reaches for anything else.  The model object is created with
`object.__new__` (its __init__ places `sample_t` on 'cuda'): `sample_t` is built on the CPU by the expression of
gaussian_curve_model.py:58-60, the activations are taken from GaussianModel.setup_functions."""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
OUT = os.path.dirname(os.path.abspath(__file__))

_ARMED = [False]
_PLACEHOLDERS = []


def _placeholder(name):
    m = types.ModuleType(name)
    m.__path__ = []

    def _getattr(attr, name=name):
        if attr.startswith("__"):
            raise AttributeError(attr)      # (the interpreter's own probes: inspect.getmodule walks sys.modules asking for __file__)
        if _ARMED[0]:
            raise RuntimeError(f"reference code reached for {name}.{attr}: a package this image lacks -- nothing may be pinned through it")
        return type(attr, (), {})      # `from skimage.measure import ransac` at module level needs a name to bind

    m.__getattr__ = _getattr
    _PLACEHOLDERS.append(name)
    return m


def import_reference(modname):
    """import `modname` from /root/reference, adding a placeholder for every absent third-party package it names."""
    for _ in range(64):
        try:
            return __import__(modname, fromlist=["_"])
        except ModuleNotFoundError as e:
            parts = e.name.split(".")
            if os.path.exists(os.path.join(REF, parts[0])) or os.path.exists(os.path.join(REF, parts[0] + ".py")):
                raise          # a module of the reference itself: never replaced
            for i in range(1, len(parts) + 1):
                n = ".".join(parts[:i])
                if n not in sys.modules:
                    sys.modules[n] = _placeholder(n)
    raise RuntimeError("too many missing modules")


def main():
    GCM = import_reference("scene.gaussian_curve_model")
    FIT = import_reference("edge_extraction.fitting")
    MRG = import_reference("edge_extraction.merging")
    _ARMED[0] = True       # from here on a placeholder that is touched raises
    print("placeholders:", sorted(set(_PLACEHOLDERS)))
    torch.set_num_threads(1)
    torch.manual_seed(0)

    # ------------------------------------------------------------------------- prepare_scaling_rot and the per-splat activations
    def model(curve_points, width, opacity, is_bezier, m=12):
        g = object.__new__(GCM.GaussianCurveModel)
        g.n_gaussians = m
        g.max_sh_degree = 0
        g.setup_functions()                                                   # scene/gaussian_model.py: the activations
        t = torch.linspace(0.5 / m, 1 - 0.5 / m, m)                           # gaussian_curve_model.py:58-60, on the CPU
        g.sample_t = t[:, None, None]
        g._curve_points, g._width, g._opacity, g.is_bezier = curve_points, width, opacity, is_bezier
        return g

    out = {}
    gen = torch.Generator().manual_seed(20260930)
    for case, B, frac_lines in (("mixed", 37, 0.35), ("bezier", 23, 0.0), ("lines", 11, 1.0)):
        # curves of a unit-cube scene: a random anchor, three random increments (bent, never degenerate), log-widths around
        # log(5e-3) and opacity logits around 0.6's (create_from_pcd :154-156)
        p0 = torch.rand(B, 1, 3, generator=gen)
        steps = 0.08 * torch.randn(B, 3, 3, generator=gen) + torch.tensor([0.0, 0.05, 0.0])
        cp = torch.cat([p0, p0 + torch.cumsum(steps, dim=1)], dim=1).contiguous()
        width = (torch.log(torch.tensor(5e-3)) + 0.4 * torch.randn(B, 1, generator=gen)).contiguous()
        opacity = (0.405 + torch.randn(B, 1, generator=gen)).contiguous()
        is_bezier = torch.rand(B, generator=gen) >= frac_lines
        if frac_lines == 0.0:
            assert bool(is_bezier.all())
        cpg, wg, og = cp.clone().requires_grad_(True), width.clone().requires_grad_(True), opacity.clone().requires_grad_(True)
        g = model(cpg, wg, og, is_bezier)
        g.prepare_scaling_rot()
        xyz, rot, scl = g._xyz, g._rotation, g._scaling
        op, cw = g.get_opacity, g.get_curve_width
        P = B * 12
        cx, cr, cs = torch.randn(P, 3, generator=gen), torch.randn(P, 4, generator=gen), torch.randn(P, 3, generator=gen)
        co = torch.randn(P, 1, generator=gen)
        ((xyz * cx).sum() + (rot * cr).sum() + (scl * cs).sum() + (op * co).sum()).backward()
        n = lambda t: t.detach().numpy().copy()
        out.update({f"{case}_curve_points": n(cp), f"{case}_width": n(width), f"{case}_opacity": n(opacity),
                    f"{case}_is_bezier": n(is_bezier), f"{case}_xyz": n(xyz), f"{case}_rotation": n(rot),
                    f"{case}_scaling": n(scl), f"{case}_splat_opacity": n(op), f"{case}_curve_width": n(cw),
                    f"{case}_cot_xyz": n(cx), f"{case}_cot_rotation": n(cr), f"{case}_cot_scaling": n(cs),
                    f"{case}_cot_opacity": n(co), f"{case}_grad_curve_points": n(cpg.grad), f"{case}_grad_width": n(wg.grad),
                    f"{case}_grad_opacity": n(og.grad)})
        print(case, "B", B, "lines", int((~is_bezier).sum()), "|xyz|", float(xyz.norm()), "|grad cp|", float(cpg.grad.norm()))
    np.savez_compressed(os.path.join(OUT, "prepare_scaling_rot.npz"), **out)

    # ------------------------------------------------------------------------- fitting / merging helpers of the topology step
    rng = np.random.default_rng(7)
    fit = {}
    # noisy samples of straight segments and of cubic Bezier curves (what merge_curves / fit_curve_to_line feed them)
    for i in range(4):
        a, b = rng.random(3), rng.random(3)
        t = np.sort(rng.random(24))[:, None]
        pts = a + t * (b - a) + 1e-3 * rng.standard_normal((24, 3))
        line, lam = FIT.line_fitting(pts)
        s, e, d, mean, tmin, tmax = FIT.fit_straight_line(pts.copy())
        fit.update({f"line{i}_points": pts, f"line{i}_line_fitting": line, f"line{i}_lambda": np.float64(lam),
                    f"line{i}_start": s, f"line{i}_end": e, f"line{i}_direction": d, f"line{i}_mean": mean,
                    f"line{i}_tmin": np.float64(tmin), f"line{i}_tmax": np.float64(tmax)})
    for i in range(4):
        ctrl = rng.random((4, 3))
        tt = np.linspace(0, 1, 40)[:, None]
        pts = ((1 - tt) ** 3 * ctrl[0] + 3 * (1 - tt) ** 2 * tt * ctrl[1] + 3 * (1 - tt) * tt ** 2 * ctrl[2] + tt ** 3 * ctrl[3]
               + (2e-3 if i < 3 else 8e-2) * rng.standard_normal((40, 3)))
        popt = FIT.bezier_fit(pts.copy(), error_threshold=0.02)
        fit[f"bezier{i}_points"] = pts
        fit[f"bezier{i}_accepted"] = np.bool_(popt is not None)
        fit[f"bezier{i}_popt"] = np.zeros(12) if popt is None else np.asarray(popt, np.float64)
    segs = np.concatenate([rng.random((9, 3)), rng.random((9, 3))], axis=1)
    segs[3, 3:] = segs[3, :3] + 2.0 * (segs[1, 3:] - segs[1, :3])      # one pair of parallel segments
    fit["segments"] = segs
    fit["pairwise_cosine_similarity"] = MRG.compute_pairwise_cosine_similarity(segs)
    fit["pairwise_distances"] = MRG.compute_pairwise_distances(segs)
    np.savez_compressed(os.path.join(OUT, "curve_fitting.npz"), **fit)
    print("wrote prepare_scaling_rot.npz, curve_fitting.npz")


if __name__ == "__main__":
    main()
