"""CPU: today's oracle chain of the per-view path (oracle/torch_ref.py around oracle/raster_ref.c) against its FROZEN outputs in
tests/golden/view_*.npz (tests/golden/make_view_golden.py): curve tensors -> image -> curve-parameter gradients, incl. straight
lines (is_bezier = False) and use_mask.  The GPU suite holds cgs_view_forward / cgs_view_backward to the same files
(tests/test_pipeline_gpu.py::test_view_path_matches_the_frozen_oracle_chain).  Freezes the oracle, pins nothing to the reference."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_view_golden import chain, load_scene, scenes, upstream  # noqa: E402

NAMES = ["small", "lines", "masked"]


def _tight(name, got, ref, rel=1e-6):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, name
    tol = rel * max(np.abs(ref).max(), 1e-30)
    assert np.abs(got - ref).max() <= tol, f"{name}: drifted from the frozen output by {np.abs(got - ref).max():.3e} (tol {tol:.1e})"


@pytest.mark.parametrize("name", NAMES)
def test_fixture_inputs_are_what_the_generator_makes(name):
    curves, mask, cam, bg = scenes()[name]
    fc, fmask, fcam, fbg, z = load_scene(name)
    for k in ("curve_points", "width", "opacity", "is_bezier"):
        assert np.array_equal(curves[k].numpy(), fc[k].numpy()), k
    assert (mask is None) == (fmask is None) and (mask is None or np.array_equal(mask.numpy(), fmask.numpy()))
    assert np.array_equal(cam.world_view_transform.numpy(), fcam.world_view_transform.numpy()) and bg == pytest.approx(fbg)
    assert np.array_equal(upstream(name, cam.image_height, cam.image_width), z["dL_dcolor"])


@pytest.mark.parametrize("name", NAMES)
def test_oracle_chain_reproduces_its_frozen_outputs(name):
    curves, mask, cam, bg, z = load_scene(name)
    res = chain(curves, mask, cam, bg, z["dL_dcolor"])
    assert np.array_equal(res["radii"], z["radii"]) and int(res["num_rendered"][0]) == int(z["num_rendered"][0])
    for k in ("color", "invdepth", "out_all_map", "final_T", "g_means2D"):
        _tight(k, res[k], z[k])
    # the pull-back through the float32 torch graph sums 12 samples per curve: 1e-5 of the maximum covers another BLAS / torch
    # summation order, nothing else
    for k in ("g_curve_points", "g_width", "g_opacity") + (("g_mask",) if mask is not None else ()):
        _tight(k, res[k], z[k], rel=1e-5)


def test_fixtures_exercise_what_they_claim():
    z = {n: load_scene(n)[4] for n in NAMES}
    assert z["small"]["is_bezier"].all() and not z["lines"]["is_bezier"].all() and (~z["lines"]["is_bezier"]).sum() > 50
    m = z["masked"]
    sg = 1.0 / (1.0 + np.exp(-m["mask"].astype(np.float64)))
    off = (sg <= float(m["mask_thr"][0])).mean()
    assert 0.1 < off < 0.6                                   # logits on both sides of the threshold
    assert np.abs(m["g_mask"]).max() > 0 and (m["g_mask"] != 0).mean() > 0.1     # the straight-through gradient reaches them
    assert float(z["lines"]["bg"][0]) > 0 and np.abs(z["lines"]["g_curve_points"]).max() > 0
