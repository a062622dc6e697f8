#!/usr/bin/env python
"""Generates tests/golden/topology.npz by running the REFERENCE's own topology edits on the CPU (authoring container only;
same import method as make_model_golden.py: the reference module is imported unmodified, packages this image lacks are
poisoned placeholders that raise if any called function touches them).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_topology_golden.py

Called, all from scene/gaussian_curve_model.py (the methods without a hard-coded device='cuda'; densification_postfix :306-327
and densify_and_split_curve :329-347 allocate on 'cuda' and cannot run here):
    _prune_optimizer :246-262, reset_opacity :264-268, fix_opacity :270-279, prune_curves :282-304, de_casteljau_trim :366-369,
    de_casteljau_split :388-421, only_prune :424-431, mask_trim_split :433-457, is_curve_straight :623-631,
    prepare_scaling_rot :180-198 (inside the edits), and scene/gaussian_model.py replace_tensor_to_optimizer :460-473
over a real torch.optim.Adam built from the group list of training_setup :203-213 (training_setup itself allocates its two
statistics buffers on 'cuda'; its list is repeated here, the buffers are made on the CPU).

The file holds the inputs (curves, the gradients of every Adam step, masks, thresholds) and, after every edit, the state the
reference is left in: the six parameter tensors, both Adam moments of every group, is_bezier, the statistics buffers and the
derived splat tensors.  tests/test_model_golden_cpu.py replays the sequence on oracle/topology_ref.py, tests/test_topology_oracle_gpu.py
on the product's scene/topology.py."""
import os
import sys

import numpy as np
import torch
from torch import nn

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_model_golden as MG  # noqa: E402  (import machinery + placeholders)

OUT = os.path.dirname(os.path.abspath(__file__))
GROUPS = (("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"), ("width", "_width"),
          ("curve_points", "_curve_points"), ("mask", "_mask"))
LR = dict(f_dc=0.0025, f_rest=0.0025 / 20.0, opacity=0.025, width=0.005, curve_points=0.0005, mask=0.01)   # arguments/__init__.py defaults


def main():
    GCM = MG.import_reference("scene.gaussian_curve_model")
    MG._ARMED[0] = True
    torch.set_num_threads(1)
    gen = torch.Generator().manual_seed(606)
    out = {}
    n = lambda t: t.detach().numpy().copy()

    B, m = 60, 12
    p0 = torch.rand(B, 1, 3, generator=gen)
    steps = 0.06 * torch.randn(B, 3, 3, generator=gen) + torch.tensor([0.0, 0.05, 0.0])
    cp = torch.cat([p0, p0 + torch.cumsum(steps, dim=1)], dim=1).contiguous()
    cp[5] = cp[5, 0] + torch.tensor([[0.0, 0, 0], [0.0, 1e-3, 0], [0.0, 2e-3, 0], [0.0, 3e-3, 0]])      # a curve only_prune calls too small
    width = (torch.log(torch.tensor(5e-3)) + 0.3 * torch.randn(B, 1, generator=gen)).contiguous()
    opacity = (2.0 * torch.randn(B, 1, generator=gen)).contiguous()
    mask = (3.0 * torch.randn(B, m, 1, generator=gen)).contiguous()
    fdc = torch.randn(B, m, 1, 1, generator=gen).contiguous()
    frest = torch.zeros(B, m, 0, 1)
    is_bezier = torch.ones(B, dtype=torch.bool)
    is_bezier[::4] = False
    out.update(in_curve_points=n(cp), in_width=n(width), in_opacity=n(opacity), in_mask=n(mask), in_f_dc=n(fdc),
               in_is_bezier=n(is_bezier))

    g = object.__new__(GCM.GaussianCurveModel)
    g.n_gaussians, g.max_sh_degree, g.optimizer_type = m, 0, "default"
    g.setup_functions()
    g.sample_t = torch.linspace(0.5 / m, 1 - 0.5 / m, m)[:, None, None]           # :58-60 on the CPU
    par = lambda t: nn.Parameter(t.clone().requires_grad_(True))
    g._curve_points, g._width, g._opacity, g._mask = par(cp), par(width), par(opacity), par(mask)
    g._features_dc, g._features_rest = par(fdc), par(frest)
    g.is_bezier = is_bezier.clone()
    P = B * m
    g.max_radii2D = torch.zeros(P)
    g.xyz_gradient_accum = torch.rand(P, 1, generator=gen) * 1e-3
    g.denom = torch.randint(0, 3, (P, 1), generator=gen).float()
    g.tmp_radii = (torch.rand(P, generator=gen) * 9).int()
    out.update(in_xyz_gradient_accum=n(g.xyz_gradient_accum), in_denom=n(g.denom), in_tmp_radii=n(g.tmp_radii))
    g.prepare_scaling_rot()
    l = [{'params': [g._features_dc], 'lr': LR["f_dc"], "name": "f_dc"},                 # training_setup :203-210
         {'params': [g._features_rest], 'lr': LR["f_rest"], "name": "f_rest"},
         {'params': [g._opacity], 'lr': LR["opacity"], "name": "opacity"},
         {'params': [g._width], 'lr': LR["width"], "name": "width"},
         {'params': [g._curve_points], 'lr': LR["curve_points"], "name": "curve_points"},
         {'params': [g._mask], 'lr': LR["mask"], "name": "mask"}]
    g.optimizer = torch.optim.Adam(l, lr=0.0, eps=1e-15)                                 # :213

    step_no = [0]

    def adam_step():
        k = step_no[0]
        step_no[0] += 1
        for name, attr in GROUPS:
            p = getattr(g, attr)
            gr = torch.randn(p.shape, generator=gen) * 1e-2
            p.grad = gr.clone()
            out[f"grad{k}_{name}"] = n(gr)
        g.optimizer.step()
        g.prepare_scaling_rot()

    def snap(tag):
        for name, attr in GROUPS:
            p = getattr(g, attr)
            out[f"{tag}.{name}"] = n(p)
            st = g.optimizer.state.get(p, None)
            assert st is not None, (tag, name)
            out[f"{tag}.exp_avg.{name}"] = n(st["exp_avg"])
            out[f"{tag}.exp_avg_sq.{name}"] = n(st["exp_avg_sq"])
        out[f"{tag}.is_bezier"] = n(g.is_bezier)
        out[f"{tag}.xyz_gradient_accum"] = n(g.xyz_gradient_accum)
        out[f"{tag}.denom"] = n(g.denom)
        out[f"{tag}.max_radii2D"] = n(g.max_radii2D)
        out[f"{tag}.xyz"], out[f"{tag}.rotation"], out[f"{tag}.scaling"] = n(g._xyz), n(g._rotation), n(g._scaling)
        print(f"{tag:18s} curves {g._curve_points.shape[0]:3d}  straight {int((~g.is_bezier).sum()):2d}")

    for _ in range(3):
        adam_step()
    snap("setup")
    prune = torch.rand(B, generator=gen) < 0.2
    out["prune_mask"] = n(prune)
    g.prune_curves(prune)                                   # :282-304
    snap("prune_curves")
    g.reset_opacity()                                       # :264-268
    snap("reset_opacity")
    adam_step()
    snap("adam_after_reset")
    with torch.no_grad():                                   # (reset_opacity caps every opacity at 0.1: give only_prune something to keep)
        bump = 2.5 * torch.rand(g._opacity.shape, generator=gen)
        g._opacity.add_(bump)
    out["opacity_bump"] = n(bump)
    before = g._curve_points.shape[0]
    g.only_prune(0.12, 0.3)                                 # :424-431
    assert 0 < g._curve_points.shape[0] < before
    snap("only_prune")
    g.mask_trim_split(0.4)                                  # :433-457
    snap("mask_trim_split")
    adam_step()
    snap("adam_after_trim")
    g.fix_opacity()                                         # :270-279 (train.py:199)
    snap("fix_opacity")
    out["fix_opacity.requires_grad"] = np.bool_(g._opacity.requires_grad)
    out["fix_opacity.lr_opacity"] = np.float64([grp["lr"] for grp in g.optimizer.param_groups if grp["name"] == "opacity"][0])

    # ---- the pure functions
    Bc = g._curve_points.shape[0]
    curves = g._curve_points.detach().clone()
    t = 0.1 + 0.8 * torch.rand(Bc, 1, generator=gen)
    left, right = g.de_casteljau_split(curves, t, g.is_bezier)                            # :388-421
    from_t = 0.3 * torch.rand(Bc, 1, generator=gen)
    end_t = 0.6 + 0.4 * torch.rand(Bc, 1, generator=gen)
    trimmed = g.de_casteljau_trim(curves, from_t, end_t, g.is_bezier)                     # :366-369
    out.update(dc_curves=n(curves), dc_is_bezier=n(g.is_bezier), dc_t=n(t), dc_left=n(left), dc_right=n(right),
               dc_from_t=n(from_t), dc_end_t=n(end_t), dc_trimmed=n(trimmed))
    rng = np.random.default_rng(3)
    for i, noise in enumerate((2e-4, 1.5e-3, 8e-3)):                                       # straight, borderline, bent
        a, b = rng.random(3), rng.random(3)
        tt = np.linspace(0, 1, 100)[:, None]
        pts = (a + tt * (b - a) + noise * np.sin(6.0 * tt) * np.array([0.3, -0.5, 0.8])).astype(np.float32)
        ok, s, e = g.is_curve_straight(torch.from_numpy(pts))                             # :623-631
        out[f"straight{i}_points"], out[f"straight{i}_ok"], out[f"straight{i}_start"], out[f"straight{i}_end"] = pts, np.bool_(ok), s, e
        print("is_curve_straight", i, bool(ok))
    np.savez_compressed(os.path.join(OUT, "topology.npz"), **out)
    print("placeholders:", sorted(set(MG._PLACEHOLDERS)))
    print("wrote topology.npz")


if __name__ == "__main__":
    main()
