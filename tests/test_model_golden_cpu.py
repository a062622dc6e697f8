"""Fixtures the REFERENCE ITSELF produced (tests/golden/make_model_golden.py imports scene/gaussian_curve_model.py and
edge_extraction/{fitting,merging}.py from /root/reference and calls their functions unmodified on the CPU):

  prepare_scaling_rot.npz   GaussianCurveModel.prepare_scaling_rot / get_opacity / get_curve_width on three curve sets (mixed Bezier
                            and straight curves, all Bezier, all straight): values, and the gradients of a seeded linear functional
                            through the reference's own autograd graph
  curve_fitting.npz         line_fitting, fit_straight_line, bezier_fit, compute_pairwise_distances, compute_pairwise_cosine_similarity
  topology.npz              (make_topology_golden.py) prune_curves, reset_opacity, only_prune, mask_trim_split over a real
                            torch.optim.Adam with Adam steps in between; de_casteljau_split / _trim, is_curve_straight

Here (no GPU): the oracle's restatement of the sampling chain (oracle/torch_ref.py) and the product's host-side restatements of the
fitting helpers (curve_gaussian_amd/scene/topology.py) are held to those files, and the topology restatement
(oracle/topology_ref.py) to the third.  tests/test_sampling_gpu.py holds the HIP kernels to the first one,
tests/test_topology_oracle_gpu.py the product's topology edits to the third."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as TR

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ("mixed", "bezier", "lines")


def _load(name):
    return np.load(os.path.join(GOLD, name))


@pytest.mark.parametrize("case", CASES)
def test_oracle_sampling_chain_matches_the_reference_outputs(case):
    """oracle/torch_ref.prepare_scaling_rot against the reference's own prepare_scaling_rot (gaussian_curve_model.py:180-198), same
    dtype (float32), same torch: positions and scalings to the last bit or two, rotations (rot_to_quat_batch: a chain of sqrt / max /
    divisions) within 1e-6; gradients of the reference's autograd within 1e-5 of the largest component."""
    z = _load("prepare_scaling_rot.npz")
    t = lambda k: torch.from_numpy(z[f"{case}_{k}"])
    cp, w, op = t("curve_points").requires_grad_(True), t("width").requires_grad_(True), t("opacity").requires_grad_(True)
    is_b = t("is_bezier")
    xyz, rot, scl = TR.prepare_scaling_rot(cp, w, is_b, 12)
    splat_op = torch.sigmoid(op).unsqueeze(1).expand(-1, 12, -1).reshape(-1, 1)              # get_opacity :110-113
    np.testing.assert_allclose(xyz.detach().numpy(), z[f"{case}_xyz"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(scl.detach().numpy(), z[f"{case}_scaling"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(rot.detach().numpy(), z[f"{case}_rotation"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(splat_op.detach().numpy(), z[f"{case}_splat_opacity"], rtol=3e-7, atol=0)   # (one ulp: torch picks a vectorised or a scalar sigmoid by size)
    np.testing.assert_allclose(torch.exp(w).detach().numpy(), z[f"{case}_curve_width"], rtol=3e-7, atol=0)
    ((xyz * t("cot_xyz")).sum() + (rot * t("cot_rotation")).sum() + (scl * t("cot_scaling")).sum()
     + (splat_op * t("cot_opacity")).sum()).backward()
    for name, got in (("curve_points", cp.grad), ("width", w.grad), ("opacity", op.grad)):
        want = z[f"{case}_grad_{name}"]
        assert np.abs(got.numpy() - want).max() <= 1e-5 * np.abs(want).max(), name
    # the straight rows of the mixed set really took the line branch: their four "control points" only enter through P0 and P3
    if case != "bezier":
        lines = ~z[f"{case}_is_bezier"]
        assert lines.any() and np.abs(z[f"{case}_grad_curve_points"][lines][:, 1:3]).max() == 0.0


def test_fitting_helpers_match_the_reference_outputs():
    """The host-side helpers of fit_curve_to_line / merge_curves (curve_gaussian_amd/scene/topology.py) against the reference's
    edge_extraction functions on the same points.  fit_straight_line and line_fitting return a principal direction whose SIGN is the
    eigen / SVD routine's choice: compared as a segment (either orientation).  bezier_fit: the reference calls scipy's curve_fit on a
    model that is linear in its parameters, the restatement solves the least-squares problem directly -- same minimum to ~1e-6."""
    from curve_gaussian_amd.scene import topology as T
    z = _load("curve_fitting.npz")

    def same_segment(a0, a1, b0, b1, tol):
        d_same = max(np.abs(a0 - b0).max(), np.abs(a1 - b1).max())
        d_flip = max(np.abs(a0 - b1).max(), np.abs(a1 - b0).max())
        assert min(d_same, d_flip) <= tol, (d_same, d_flip)

    for i in range(4):
        pts = z[f"line{i}_points"]
        s, e, d, mean, tmin, tmax = T.fit_straight_line(pts.copy())
        same_segment(s, e, z[f"line{i}_start"], z[f"line{i}_end"], 1e-12)
        np.testing.assert_allclose(mean, z[f"line{i}_mean"], rtol=0, atol=1e-15)
        assert abs(abs(float(np.dot(d, z[f"line{i}_direction"]))) - 1.0) < 1e-12
        assert abs((tmax - tmin) - (float(z[f"line{i}_tmax"]) - float(z[f"line{i}_tmin"]))) < 1e-12
        line = T._line_fitting(pts.copy())
        same_segment(line[:3], line[3:], z[f"line{i}_line_fitting"][:3], z[f"line{i}_line_fitting"][3:], 1e-12)
    for i in range(4):
        popt = T._bezier_fit(z[f"bezier{i}_points"].copy(), error_threshold=0.02)
        assert (popt is not None) == bool(z[f"bezier{i}_accepted"]), i
        if popt is not None:
            np.testing.assert_allclose(popt, z[f"bezier{i}_popt"], rtol=0, atol=2e-6)
    assert not bool(z["bezier3_accepted"]) and bool(z["bezier0_accepted"])      # both outcomes are in the file
    seg = z["segments"]
    np.testing.assert_allclose(T._pairwise_segment_distances(seg), z["pairwise_distances"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(T._pairwise_cosine_similarity(seg), z["pairwise_cosine_similarity"], rtol=0, atol=1e-14)


def test_oracle_topology_edits_match_the_reference_run():
    """oracle/topology_ref.RefCurveModel replays the sequence the reference itself ran for tests/golden/topology.npz (three Adam steps,
    prune_curves, reset_opacity, an Adam step, only_prune, mask_trim_split, an Adam step): after every step the six parameter tensors,
    both Adam moments of every group, is_bezier, the statistics buffers and the derived splat tensors equal the reference's."""
    from oracle.topology_ref import RefCurveModel
    from util import TOPOLOGY_GROUPS, replay_topology_fixture
    z = _load("topology.npz")
    t = lambda k: torch.from_numpy(z[k])
    B = z["in_curve_points"].shape[0]
    ref = RefCurveModel(t("in_curve_points"), t("in_width"), t("in_opacity"), t("in_mask"), t("in_f_dc"), torch.zeros(B, 12, 0, 1),
                        t("in_is_bezier"))
    ref.training_setup()
    seen = []

    def check(tag):
        seen.append(tag)
        snap = ref.snapshot()
        for name, _ in TOPOLOGY_GROUPS:
            assert tuple(snap[name].shape) == z[f"{tag}.{name}"].shape, (tag, name)
            np.testing.assert_allclose(snap[name].numpy(), z[f"{tag}.{name}"], rtol=1e-6, atol=1e-7, err_msg=f"{tag}: {name}")
            np.testing.assert_allclose(snap["exp_avg." + name].numpy(), z[f"{tag}.exp_avg.{name}"], rtol=1e-6, atol=1e-10,
                                       err_msg=f"{tag}: exp_avg {name}")
            np.testing.assert_allclose(snap["exp_avg_sq." + name].numpy(), z[f"{tag}.exp_avg_sq.{name}"], rtol=1e-6, atol=1e-14,
                                       err_msg=f"{tag}: exp_avg_sq {name}")
        assert np.array_equal(snap["is_bezier"].numpy(), z[f"{tag}.is_bezier"]), tag
        for name in ("xyz_gradient_accum", "denom", "max_radii2D"):
            np.testing.assert_array_equal(snap[name].numpy(), z[f"{tag}.{name}"], err_msg=f"{tag}: {name}")
        np.testing.assert_allclose(snap["xyz"].numpy(), z[f"{tag}.xyz"], rtol=0, atol=3e-7, err_msg=f"{tag}: xyz")
        np.testing.assert_allclose(snap["scaling"].numpy(), z[f"{tag}.scaling"], rtol=3e-6, atol=1e-9, err_msg=f"{tag}: scaling")
        np.testing.assert_allclose(snap["rotation"].numpy(), z[f"{tag}.rotation"], rtol=0, atol=2e-6, err_msg=f"{tag}: rotation")
    replay_topology_fixture(ref, z, "cpu", check)
    assert len(seen) == 8 and z["only_prune.curve_points"].shape[0] < z["adam_after_reset.curve_points"].shape[0]
    # ---- the pure functions
    ref.is_bezier = t("dc_is_bezier")
    left, right = ref.de_casteljau_split(t("dc_curves"), t("dc_t"), t("dc_is_bezier"))
    np.testing.assert_allclose(left.numpy(), z["dc_left"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(right.numpy(), z["dc_right"], rtol=0, atol=1e-7)
    trimmed = ref.de_casteljau_trim(t("dc_curves"), t("dc_from_t"), t("dc_end_t"), t("dc_is_bezier"))
    np.testing.assert_allclose(trimmed.numpy(), z["dc_trimmed"], rtol=0, atol=2e-7)
    from curve_gaussian_amd.scene import topology as T
    outcomes = []
    for i in range(3):
        ok, s, e = T.is_curve_straight(None, torch.from_numpy(z[f"straight{i}_points"]))
        assert bool(ok) == bool(z[f"straight{i}_ok"]), i
        d_same = max(np.abs(s - z[f"straight{i}_start"]).max(), np.abs(e - z[f"straight{i}_end"]).max())
        d_flip = max(np.abs(s - z[f"straight{i}_end"]).max(), np.abs(e - z[f"straight{i}_start"]).max())
        assert min(d_same, d_flip) < 1e-6, i
        outcomes.append(bool(ok))
    assert outcomes == [True, True, False]
