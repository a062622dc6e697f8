"""scene/topology.py against the CPU restatement of the reference's topology edits (oracle/topology_ref.py, a literal
statement-by-statement restatement of gaussian_curve_model.py:246-463 + gaussian_model.py:460-533 over a real
torch.optim.Adam): parameters, is_bezier, statistics buffers, derived splat tensors and the Adam moments of every group
after prune / split / densify_and_prune / reset_opacity / curvature split / only_prune / mask_trim_split, for both
optimizer back ends of the product (torch.optim.Adam, and the flat one-launch Adam of the hot path)."""
import numpy as np
import pytest
import torch

from oracle.topology_ref import RefCurveModel
from util import S

pytestmark = pytest.mark.gpu
DEV = "cuda"
GROUPS = ("curve_points", "f_dc", "f_rest", "opacity", "width", "mask")
ATTR = {"curve_points": "_curve_points", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
        "width": "_width", "mask": "_mask"}


def _pair(B, seed, flat, lines=True):
    """The same curves in the product model (GPU) and in the reference restatement (CPU), both with optimizers that have
    taken three Adam steps on identical synthetic gradients (non-trivial moments), identical statistics buffers."""
    from curve_gaussian_amd.ops.optim import FlatAdam
    from curve_gaussian_amd.scene import GaussianCurveModel
    from curve_gaussian_amd.view_parallel import FlatGrads
    g = torch.Generator().manual_seed(seed)
    c = S.make_curves(B, seed)
    c["curve_points"][:, 1:3] += 0.02 * torch.randn(B, 2, 3, generator=g)            # some real curvature
    c["opacity"] = torch.randn(B, 1, generator=g) * 2.0                             # opacities from 0.02 to 0.98
    c["mask"] = torch.randn(B, 12, 1, generator=g) * 3.0
    c["width"] = c["width"] + 0.3 * torch.randn(B, 1, generator=g)
    isb = torch.ones(B, dtype=torch.bool)
    if lines:
        isb[::4] = False
    fdc = torch.randn(B, 12, 1, 1, generator=g)
    gm = GaussianCurveModel(0, 12, device=DEV).create_from_curves(c["curve_points"], c["width"], c["opacity"], c["mask"], isb)
    with torch.no_grad():
        gm._features_dc.copy_(fdc.to(DEV))
    ref = RefCurveModel(c["curve_points"], c["width"], c["opacity"], c["mask"], fdc, torch.zeros(B, 12, 0, 1), isb)
    gm.training_setup()
    ref.training_setup()
    if flat:
        named = {"curve_points": gm._curve_points, "width": gm._width, "opacity": gm._opacity, "mask": gm._mask,
                 "f_dc": gm._features_dc, "f_rest": gm._features_rest}
        fg = FlatGrads(named)
        gm.optimizer = FlatAdam(named, {grp["name"]: grp["lr"] for grp in gm.optimizer.param_groups}, fg, eps=1e-15)
        gm.prepare_scaling_rot()
    for grp in ref.optimizer.param_groups:       # (lr 0.0 default of the constructor is overridden per group)
        assert grp["lr"] > 0
    for _ in range(3):
        _adam_step(gm, ref, g)
    P = B * 12
    vs = torch.randn(P, 3, generator=g) * 3e-4
    filt = torch.rand(P, generator=g) > 0.3
    for _ in range(2):
        class VS:
            grad = vs.to(DEV)
        gm.add_densification_stats(VS, filt.to(DEV))
        ref.add_densification_stats(vs, filt)
    radii = (torch.rand(P, generator=g) * 9).int()
    return gm, ref, g, radii


def _adam_step(gm, ref, g):
    """One optimizer step on both sides with the same gradients."""
    for name in GROUPS:
        p = getattr(ref, ATTR[name])
        gr = torch.randn(p.shape, generator=g) * 1e-2
        p.grad = gr.clone()
        q = getattr(gm, ATTR[name])
        if q.grad is None:
            q.grad = gr.to(DEV).clone()
        else:
            q.grad.copy_(gr.to(DEV))
    ref.optimizer.step()
    gm.optimizer.step()
    ref.prepare_scaling_rot()
    gm.prepare_scaling_rot()


def _moments(gm, name):
    from curve_gaussian_amd.ops.optim import FlatAdam
    opt = gm.optimizer
    if isinstance(opt, FlatAdam):
        st = opt.state_of(name)
        return None if st is None else (st[0], st[1])
    for grp in opt.param_groups:
        if grp["name"] == name:
            st = opt.state.get(grp["params"][0], None)
            return None if st is None else (st["exp_avg"], st["exp_avg_sq"])
    raise KeyError(name)


def _check(gm, ref, what):
    snap = ref.snapshot()
    n = lambda t: t.detach().cpu().numpy()
    for name in GROUPS:
        got, want = getattr(gm, ATTR[name]), snap[name]
        assert tuple(got.shape) == tuple(want.shape), f"{what}: shape of {name}"
        np.testing.assert_allclose(n(got), n(want), rtol=2e-5, atol=2e-6, err_msg=f"{what}: {name}")
        assert got.requires_grad
        mom = _moments(gm, name)
        assert mom is not None, f"{what}: {name} lost its Adam state"
        if want.numel():
            np.testing.assert_allclose(n(mom[0]), n(snap["exp_avg." + name]), rtol=2e-5, atol=1e-9, err_msg=f"{what}: exp_avg {name}")
            np.testing.assert_allclose(n(mom[1]), n(snap["exp_avg_sq." + name]), rtol=2e-5, atol=1e-12, err_msg=f"{what}: exp_avg_sq {name}")
    assert torch.equal(gm.is_bezier.cpu(), snap["is_bezier"]), what
    for name in ("xyz_gradient_accum", "denom", "max_radii2D"):
        np.testing.assert_allclose(n(getattr(gm, name)), n(snap[name]), rtol=1e-6, atol=0, err_msg=f"{what}: {name}")
    np.testing.assert_allclose(n(gm._xyz), n(snap["xyz"]), rtol=2e-5, atol=2e-6, err_msg=f"{what}: xyz")
    np.testing.assert_allclose(n(gm._scaling), n(snap["scaling"]), rtol=1e-4, atol=2e-6, err_msg=f"{what}: scaling")


def _refill_stats(gm, ref, g):
    P = ref._xyz.shape[0]
    vs = torch.randn(P, 3, generator=g) * 3e-4
    filt = torch.rand(P, generator=g) > 0.3

    class VS:
        grad = vs.to(DEV)
    gm.add_densification_stats(VS, filt.to(DEV))
    ref.add_densification_stats(vs, filt)


@pytest.mark.parametrize("flat", [False, True])
def test_topology_edits_match_the_reference_restatement(flat):
    gm, ref, g, radii = _pair(90, 11, flat)
    _check(gm, ref, "setup")
    B = ref._curve_points.shape[0]
    # prune_curves (GCM:283-304)
    mask = torch.rand(B, generator=g) < 0.2
    gm.tmp_radii, ref.tmp_radii = radii.to(DEV), radii.clone()
    gm.prune_curves(mask.to(DEV)); ref.prune_curves(mask)
    _check(gm, ref, "prune_curves")
    # densify_and_split_curve (GCM:330-349) at per-curve parameters
    B = ref._curve_points.shape[0]
    sel = torch.rand(B, generator=g) < 0.3
    t = 0.15 + 0.7 * torch.rand(int(sel.sum()), 1, generator=g)
    gm.densify_and_split_curve(sel.to(DEV), t.to(DEV)); ref.densify_and_split_curve(sel, t)
    _check(gm, ref, "densify_and_split_curve")
    _adam_step(gm, ref, g)
    _check(gm, ref, "Adam step after split")
    # densify_and_prune (GCM:351-365) on accumulated statistics
    _refill_stats(gm, ref, g)
    _refill_stats(gm, ref, g)
    P = ref._xyz.shape[0]
    radii2 = (torch.rand(P, generator=g) * 9).int()
    grads = (ref.xyz_gradient_accum / ref.denom).nan_to_num(0.0).reshape(-1, 12)
    thr = float(grads.max(1).values.median())
    gm.densify_and_prune(thr, 0.05, 1.0, 20, radii2.to(DEV)); ref.densify_and_prune(thr, 0.05, 1.0, 20, radii2.clone())
    assert ref._curve_points.shape[0] != P // 12
    _check(gm, ref, "densify_and_prune")
    # reset_opacity (GCM:264-268): moments of the opacity group restart from zero
    gm.reset_opacity(); ref.reset_opacity()
    _check(gm, ref, "reset_opacity")
    assert float(gm.get_curve_opacity.max()) <= 0.1 + 1e-6
    _adam_step(gm, ref, g)
    _check(gm, ref, "Adam step after reset_opacity")
    # curve_split_curvature (GCM:373-390)
    before = ref._curve_points.shape[0]
    gm.curve_split_curvature(3, 5); ref.curve_split_curvature(3, 5)
    assert ref._curve_points.shape[0] > before
    _check(gm, ref, "curve_split_curvature")
    # only_prune (GCM:428-435)
    before = ref._curve_points.shape[0]
    gm.only_prune(0.02, 0.3); ref.only_prune(0.02, 0.3)
    assert 0 < ref._curve_points.shape[0] < before
    _check(gm, ref, "only_prune")
    # mask_trim_split (GCM:437-463)
    gm.mask_trim_split(0.4); ref.mask_trim_split(0.4)
    _check(gm, ref, "mask_trim_split")
    _adam_step(gm, ref, g)
    _check(gm, ref, "Adam step after mask_trim_split")


def test_all_bezier_model_takes_the_short_path():
    """`if self.is_bezier.all()` (GCM:410): a model without straight segments never builds the chord variants."""
    gm, ref, g, radii = _pair(40, 3, False, lines=False)
    sel = torch.rand(40, generator=g) < 0.5
    t = 0.2 + 0.6 * torch.rand(int(sel.sum()), 1, generator=g)
    gm.densify_and_split_curve(sel.to(DEV), t.to(DEV)); ref.densify_and_split_curve(sel, t)
    _check(gm, ref, "split, all Bezier")


@pytest.mark.parametrize("flat", [False, True])
def test_topology_edits_match_the_reference_generated_fixture(flat):
    """scene/topology.py against tests/golden/topology.npz -- the states the REFERENCE ITSELF went through on the CPU
    (tests/golden/make_topology_golden.py imports scene/gaussian_curve_model.py and calls prune_curves, reset_opacity, only_prune and
    mask_trim_split unmodified over a real torch.optim.Adam, Adam steps in between): after every recorded step the six parameter
    tensors, both Adam moments of every group, is_bezier, the statistics buffers and the derived splat tensors, for both optimizer
    back ends of the product.  (The restatement oracle/topology_ref.py is held to the same file by tests/test_model_golden_cpu.py.)"""
    import os
    from curve_gaussian_amd.ops.optim import FlatAdam
    from curve_gaussian_amd.scene import GaussianCurveModel
    from curve_gaussian_amd.view_parallel import FlatGrads
    from util import replay_topology_fixture
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "topology.npz"))
    t = lambda k: torch.from_numpy(z[k])
    gm = GaussianCurveModel(0, 12, device=DEV).create_from_curves(t("in_curve_points"), t("in_width"), t("in_opacity"), t("in_mask"),
                                                                   t("in_is_bezier"))
    with torch.no_grad():
        gm._features_dc.copy_(t("in_f_dc").to(DEV))
    gm.training_setup()
    if flat:
        named = {"curve_points": gm._curve_points, "width": gm._width, "opacity": gm._opacity, "mask": gm._mask,
                 "f_dc": gm._features_dc, "f_rest": gm._features_rest}
        fg = FlatGrads(named)
        gm.optimizer = FlatAdam(named, {grp["name"]: grp["lr"] for grp in gm.optimizer.param_groups}, fg, eps=1e-15)
        gm.prepare_scaling_rot()
    n = lambda x: x.detach().cpu().numpy()
    seen = []

    def check(tag):
        seen.append(tag)
        for name in GROUPS:
            got, want = getattr(gm, ATTR[name]), z[f"{tag}.{name}"]
            assert tuple(got.shape) == want.shape, f"{tag}: shape of {name}"
            np.testing.assert_allclose(n(got), want, rtol=2e-5, atol=2e-6, err_msg=f"{tag}: {name}")
            mom = _moments(gm, name)
            assert mom is not None, f"{tag}: {name} lost its Adam state"
            if want.size:
                np.testing.assert_allclose(n(mom[0]), z[f"{tag}.exp_avg.{name}"], rtol=2e-5, atol=1e-9, err_msg=f"{tag}: exp_avg {name}")
                np.testing.assert_allclose(n(mom[1]), z[f"{tag}.exp_avg_sq.{name}"], rtol=2e-5, atol=1e-12, err_msg=f"{tag}: exp_avg_sq {name}")
        assert np.array_equal(n(gm.is_bezier), z[f"{tag}.is_bezier"]), tag
        for name in ("xyz_gradient_accum", "denom", "max_radii2D"):
            np.testing.assert_allclose(n(getattr(gm, name)), z[f"{tag}.{name}"], rtol=1e-6, atol=0, err_msg=f"{tag}: {name}")
        np.testing.assert_allclose(n(gm._xyz), z[f"{tag}.xyz"], rtol=2e-5, atol=2e-6, err_msg=f"{tag}: xyz")
        np.testing.assert_allclose(n(gm._scaling), z[f"{tag}.scaling"], rtol=1e-4, atol=2e-6, err_msg=f"{tag}: scaling")
    replay_topology_fixture(gm, z, DEV, check)
    assert len(seen) == 8 and gm._curve_points.shape[0] == z["adam_after_trim.curve_points"].shape[0] == 41
    # de Casteljau at per-curve parameters, mixed Bezier / straight curves (gaussian_curve_model.py:388-421, :366-369)
    from curve_gaussian_amd.scene import topology as T
    gm.is_bezier = t("dc_is_bezier").to(DEV)
    left, right = T.de_casteljau_split(gm, t("dc_curves").to(DEV), t("dc_t").to(DEV), t("dc_is_bezier").to(DEV))
    np.testing.assert_allclose(n(left), z["dc_left"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(n(right), z["dc_right"], rtol=0, atol=2e-7)
    trimmed = T.de_casteljau_trim(gm, t("dc_curves").to(DEV), t("dc_from_t").to(DEV), t("dc_end_t").to(DEV), t("dc_is_bezier").to(DEV))
    np.testing.assert_allclose(n(trimmed), z["dc_trimmed"], rtol=0, atol=4e-7)
