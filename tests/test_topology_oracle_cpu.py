"""The topology oracle (oracle/topology_ref.py) on its own, CPU only: the restated edits keep parameters, Adam state
and statistics buffers aligned, and the de Casteljau halves reproduce the curve they split."""
import numpy as np
import torch

from curve_gaussian_amd import synthetic as S
from oracle.topology_ref import RefCurveModel


def _ref(B=30, seed=2):
    g = torch.Generator().manual_seed(seed)
    c = S.make_curves(B, seed)
    isb = torch.ones(B, dtype=torch.bool)
    isb[::5] = False
    r = RefCurveModel(c["curve_points"], c["width"], torch.randn(B, 1, generator=g), torch.randn(B, 12, 1, generator=g) * 3,
                      torch.randn(B, 12, 1, 1, generator=g), torch.zeros(B, 12, 0, 1), isb)
    r.training_setup()
    for grp in r.optimizer.param_groups:
        grp["params"][0].grad = torch.randn(grp["params"][0].shape, generator=g) * 1e-2
    r.optimizer.step()
    return r, g


def _bez(cp, t):
    t = t.reshape(-1, 1)
    return (1 - t) ** 3 * cp[:, 0] + 3 * (1 - t) ** 2 * t * cp[:, 1] + 3 * (1 - t) * t ** 2 * cp[:, 2] + t ** 3 * cp[:, 3]


def test_split_halves_reproduce_the_curve_and_state_stays_aligned():
    r, g = _ref()
    cp = r.get_curve_points.detach()
    t = 0.2 + 0.6 * torch.rand(30, 1, generator=g)
    left, right = r.de_casteljau_split(cp, t, r.is_bezier)
    s = torch.full((30,), 0.37)
    isb = r.is_bezier
    np.testing.assert_allclose(_bez(left, s)[isb].numpy(), _bez(cp, t[:, 0] * s)[isb].numpy(), atol=2e-6)
    np.testing.assert_allclose(_bez(right, s)[isb].numpy(), _bez(cp, t[:, 0] + (1 - t[:, 0]) * s)[isb].numpy(), atol=2e-6)
    chord = lambda u: cp[:, 0] + u.reshape(-1, 1) * (cp[:, 3] - cp[:, 0])
    np.testing.assert_allclose(_bez(left, s)[~isb].numpy(), chord(t[:, 0] * s)[~isb].numpy(), atol=2e-6)
    sel = torch.zeros(30, dtype=torch.bool)
    sel[[1, 5, 7]] = True
    op_before = r._opacity.detach().clone()
    m_before = r.optimizer.state[r._opacity]["exp_avg"].clone()
    r.densify_and_split_curve(sel, t[sel])
    assert r._curve_points.shape[0] == 33 and r.is_bezier.shape[0] == 33 and r.denom.shape == (33 * 12, 1)
    # survivors keep values and moments; the six halves inherit the value of their parent and start with zero moments
    np.testing.assert_array_equal(r._opacity.detach()[:27].numpy(), op_before[~sel].numpy())
    np.testing.assert_array_equal(r._opacity.detach()[27:30].numpy(), op_before[sel].numpy())
    st = r.optimizer.state[r._opacity]
    np.testing.assert_array_equal(st["exp_avg"][:27].numpy(), m_before[~sel].numpy())
    assert not st["exp_avg"][27:].any() and not st["exp_avg_sq"][27:].any()
    assert not r.is_bezier[28] and r.is_bezier[27]            # curve 5 was a straight segment, 1 and 7 Bezier
    for grp in r.optimizer.param_groups:                     # the optimizer still steps
        grp["params"][0].grad = torch.ones_like(grp["params"][0])
    r.optimizer.step()


def test_prune_trim_and_reset():
    r, g = _ref(24, 4)
    mask = torch.zeros(24, dtype=torch.bool)
    mask[::3] = True
    r.tmp_radii = torch.arange(24 * 12)
    r.prune_curves(mask)
    assert r._curve_points.shape[0] == 16 and r.tmp_radii.shape[0] == 16 * 12 and r._xyz.shape[0] == 16 * 12
    r.reset_opacity()
    assert float(torch.sigmoid(r._opacity).max()) <= 0.1 + 1e-6 and not r.optimizer.state[r._opacity]["exp_avg"].any()
    with torch.no_grad():
        r._mask[:, :3] = -5.0           # first three samples masked out everywhere
    cp = r.get_curve_points.detach().clone()
    valid = torch.sigmoid(r._mask.detach())[:, :, 0] > 0.5
    first = torch.argmax(valid.int(), dim=1)
    assert (first >= 3).all()
    r.mask_trim_split(0.5)
    isb = r.is_bezier
    # new start point = B(sample_t[first] - 0.5/12) = B(first / 12) on Bezier curves
    start = _bez(cp, first.float() / 12)
    np.testing.assert_allclose(r._curve_points.detach()[isb][:, 0].numpy(), start[isb].numpy(), atol=2e-6)
    assert not r.optimizer.state[r._curve_points]["exp_avg"].any() and not r.optimizer.state[r._mask]["exp_avg_sq"].any()
