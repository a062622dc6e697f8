"""CPU: today's build of the rasterizer oracle (oracle/raster_ref.c, compiled on this box) against the FROZEN copy of its
outputs in tests/golden/raster_*.npz (tests/golden/make_raster_golden.py).  Guards against oracle / kernel co-drift: the GPU
suite holds the HIP path to the same files.  It does not pin the oracle to the reference (nothing can here)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_raster_golden import load_scene, scenes  # noqa: E402

from oracle import raster as ORA  # noqa: E402
from util import oracle_forward  # noqa: E402

NAMES = ["small", "ties", "opaque", "room"]


def _tight(name, got, ref):
    """Same source, same compiler flags (-ffp-contract=off): bit-exact on this image; 1e-6 of the maximum leaves room for a
    different OpenMP summation order of the double-precision per-splat sums only."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, name
    tol = 1e-6 * max(np.abs(ref).max(), 1e-30)
    assert np.abs(got - ref).max() <= tol, f"{name}: oracle drifted from its frozen output by {np.abs(got - ref).max():.3e} (tol {tol:.1e})"


@pytest.mark.parametrize("name", NAMES)
def test_fixture_inputs_are_what_the_generator_makes(name):
    """The committed inputs are the generator's (seeded) scenes: the script and the files belong together."""
    sp, cam, bg = scenes()[name]
    fsp, fcam, fbg, z = load_scene(name)
    for k in sp:
        assert np.array_equal(sp[k].numpy(), fsp[k].numpy()), k
    assert np.array_equal(cam.world_view_transform.numpy(), fcam.world_view_transform.numpy())
    assert np.array_equal(cam.full_proj_transform.numpy(), fcam.full_proj_transform.numpy())
    assert (cam.image_height, cam.image_width) == (fcam.image_height, fcam.image_width) and np.array_equal(bg.numpy(), fbg.numpy())


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_its_frozen_outputs(name):
    sp, cam, bg, z = load_scene(name)
    fw = oracle_forward(sp, cam, bg)
    # integer work: bit-exact
    assert np.array_equal(fw.radii, z["radii"])
    assert fw.num_rendered == int(z["num_rendered"][0])
    assert np.array_equal(fw.ranges, z["ranges"]) and np.array_equal(fw.point_list, z["point_list"])
    assert np.array_equal(fw.n_contrib, z["n_contrib"])
    for k, v in (("color", fw.color), ("invdepth", fw.invdepth), ("out_all_map", fw.out_all_map), ("final_T", fw.final_T),
                 ("means2D", fw.means2D), ("conic_opacity", fw.conic_opacity), ("depths", fw.depths)):
        _tight(k, v, z[k])
    gr = ORA.backward(fw, z["dL_dcolor"], z["dL_dinvdepth"], z["dL_dout_all_map"])
    for k, v in gr.items():
        if v is not None and k != "dL_dsh":
            _tight("g_" + k, v, z["g_" + k])
    gt = ORA.backward(fw, z["dL_dcolor"], None, None)
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dconic"):
        _tight("gt_" + k, gt[k], z["gt_" + k])
    fw.free()


def test_fixtures_exercise_what_they_claim():
    z = {n: load_scene(n)[3] for n in NAMES}
    # ties: some tile list holds two entries of equal depth in ascending splat index
    t = z["ties"]
    d = t["depths"][t["point_list"]]
    same = 0
    for a, b in t["ranges"]:
        seg_d, seg_i = d[a:b], t["point_list"][a:b].astype(np.int64)
        eq = seg_d[1:] == seg_d[:-1]
        same += int(eq.sum())
        assert (seg_i[1:][eq] > seg_i[:-1][eq]).all()
    assert same > 50
    # opaque: terminated pixels (final_T < 1e-4 is impossible: the terminating splat is not blended) and clamped alphas
    o = z["opaque"]
    assert (o["opacities"] > 0.99).mean() > 0.3 and (o["final_T"] < 1e-3).mean() > 0.05
    # room: near-culled splats and at least one splat covering the whole image
    r = z["room"]
    assert (r["radii"] == 0).mean() > 0.5 and r["radii"].max() > max(int(r["hw"][0]), int(r["hw"][1]))
