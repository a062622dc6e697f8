"""CPU: oracle/raster_ref.c against hand-computed expectations for SURVEY App. B quirks 4, 5, 7, 9, 12 (tests/quirk_cases.py;
the same expectations are asserted on the HIP path in tests/test_quirks_gpu.py)."""
import math

import numpy as np
import pytest
import torch

import quirk_cases as Q
from oracle import raster as ORA
from util import oracle_forward


def _run(sp, bg=0.0, **kw):
    return oracle_forward(sp, Q.camera(), torch.full((3,), float(bg)), **kw)


def check_quirk4(fw_plain, fw_aa, s):
    """+0.3 px^2 dilation always; antialiasing only rescales the opacity (forward.cu:219-227)."""
    a = Q.cov_diag(s, 2.0)
    for fw, scale in ((fw_plain, 1.0), (fw_aa, math.sqrt(max(0.000025, (a - 0.3) ** 2 / (a * a))))):
        img = fw.color[0] if hasattr(fw, "color") else fw[0]
        for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (2, 1), (3, 0)):
            al = Q.alpha_at(0.8 * scale, a, dx, dy)
            want = al if al >= 1.0 / 255.0 else 0.0          # forward.cu:366-367: alpha < 1/255 -> skipped
            got = float(img[16 + dy, 16 + dx])
            assert abs(got - want) <= 2e-6 + 1e-5 * want, f"pixel (+{dx},+{dy}): {got} vs closed form {want} (scale {scale})"


def test_quirk4_dilation_and_antialiasing_rescale():
    s = 2.0 * math.sqrt(0.3) / Q.focal()               # (f s / z)^2 = 0.3 at z = 2: cov2D = 0.6 I, AA factor = 0.3 / 0.6
    sp = Q.splats([(16, 16)], 2.0, s, 0.8)
    a, b = _run(sp), _run(sp, antialiasing=True)
    check_quirk4(a, b, s)
    assert a.radii[0] == Q.radius_of(0.6) == 3
    a.free(); b.free()


def test_quirk5_eigenvalue_floor_and_det_zero_drop():
    # a vanishing splat: cov2D = 0.3 I exactly; without the 0.1 floor of the radicand its radius would be ceil(3 sqrt(0.3)) = 2
    sp = Q.splats([(16, 16)], 2.0, 1e-7, 0.8)
    fw = _run(sp)
    assert fw.radii[0] == Q.radius_of(0.3) == 3 and math.ceil(3.0 * math.sqrt(0.3)) == 2
    fw.free()
    # det == 0 -> the splat is dropped (forward.cu:232-233).  Reachable through cov3D_precomp only: Sigma_xx = -0.3 at
    # focal / z = 1 gives cov2D.x = -0.3 + 0.3 = 0 exactly, hence det = 0 * cov2D.z - 0 = 0
    sp = Q.splats([(15.5, 15.5)], Q.focal(), 0.0, 0.8)
    sp["means3D"][0, 2] = torch.tensor(np.float32(Q.focal()))
    z = float(sp["means3D"][0, 2])
    assert np.float32(Q.focal()) / np.float32(z) == 1.0
    cov = torch.tensor([[-0.3, 0.0, 0.0, 0.5, 0.0, 0.0]], dtype=torch.float32)
    fw = _run(sp, cov3D=cov)
    assert fw.radii[0] == 0 and fw.num_rendered == 0 and fw.color.max() == 0
    fw.free()
    cov[0, 0] = 0.25                                       # the same call with a regular covariance renders
    fw = _run(sp, cov3D=cov)
    assert fw.radii[0] > 0 and fw.color.max() > 0.5
    fw.free()


def quirk7_scene():
    # eight identical, nearly point-like splats stacked on pixel (16, 16), alpha = 0.8 there: T = 0.2^k after k of them;
    # the sixth would leave T = 6.4e-5 < 1e-4, so it is NOT blended and ends the pixel (forward.cu:371-376)
    return Q.splats([(16, 16)] * 8, [2.0 + 0.1 * i for i in range(8)], 1e-7, 0.8)


def check_quirk7_forward(color, final_T, n_contrib):
    assert n_contrib[16, 16] == 5                          # 1-based list position of the last blended splat (:353,394,403)
    assert abs(float(final_T[16, 16]) - 0.2 ** 5) <= 1e-4 * 0.2 ** 5      # (1 - alpha)^5: five times alpha's 1e-5-class error
    assert abs(float(color[0, 16, 16]) - (1.0 - 0.2 ** 5)) <= 2e-6
    # a neighbouring pixel never terminates: all eight blended, T = (1 - alpha_1)^8
    a1 = Q.alpha_at(0.8, 0.3, 1, 0)
    assert n_contrib[16, 17] == 8 and abs(float(final_T[16, 17]) - (1 - a1) ** 8) <= 1e-5


def check_quirk7_backward(dL_dopacity):
    # upstream gradient 1 at pixel (16, 16) only.  C = 1 - prod_{blended}(1 - alpha_j): dC/dalpha_j = T_final / (1 - alpha_j)
    # = 0.2^4 for the five blended splats; the three behind the cut get EXACTLY zero (backward.cu:576-578)
    want = 0.2 ** 4
    for j in range(5):
        assert abs(float(dL_dopacity[j]) - want) <= 2e-5 * want, (j, float(dL_dopacity[j]), want)
    assert (np.asarray(dL_dopacity[5:]) == 0).all()


def test_quirk7_transmittance_stop_excludes_the_splat():
    fw = _run(quirk7_scene())
    check_quirk7_forward(fw.color, fw.final_T.reshape(Q.H, Q.W), fw.n_contrib.reshape(Q.H, Q.W))
    d = np.zeros((1, Q.H, Q.W), np.float32)
    d[0, 16, 16] = 1.0
    gr = ORA.backward(fw, d, None, None)
    check_quirk7_backward(gr["dL_dopacity"].reshape(-1))
    fw.free()


def quirk9_expected(op, a):
    # one splat centred on pixel (16, 16); upstream gradient 1 at pixel (15, 16), i.e. d = centre - pixel = (+1, 0):
    # C = op G, G = exp(-d^2 / 2a): dC/dcentre_x = -op G dx / a, times the NDC factor W / 2 (backward.cu:542-543,663-664)
    G = math.exp(-0.5 / a)
    return -op * G * (1.0 / a) * 0.5 * Q.W


def check_quirk9(dL_dmeans2D, a):
    g = np.asarray(dL_dmeans2D).reshape(-1, 3)
    want = quirk9_expected(0.8, a)
    assert abs(float(g[0, 0]) - want) <= 2e-5 * abs(want), (float(g[0, 0]), want)
    assert abs(float(g[0, 1])) <= 1e-6 * abs(want)
    assert (g[:, 2] == 0).all()                            # a [P,3] tensor whose z column is never written


def test_quirk9_means2D_gradient_is_in_ndc_units():
    s = 2.0 * math.sqrt(0.7) / Q.focal()                   # cov2D = 1.0 I
    fw = _run(Q.splats([(16, 16)], 2.0, s, 0.8))
    d = np.zeros((1, Q.H, Q.W), np.float32)
    d[0, 16, 15] = 1.0
    gr = ORA.backward(fw, d, None, None)
    check_quirk9(gr["dL_dmeans2D"], 1.0)
    fw.free()


def test_quirk12_masked_splat_stays_in_the_pipeline():
    # use_mask multiplies scales AND opacity by the binarised mask (gaussian_renderer/__init__.py:72-76): a masked-out splat
    # has zero scale and zero opacity but still projects (cov2D = 0.3 I from the dilation), keeps a radius of 3 px and is
    # binned; it contributes nothing (alpha = 0 < 1/255)
    sp = Q.splats([(16, 16), (8, 8)], 2.0, [0.0, 0.1], [0.0, 0.8])
    fw = _run(sp)
    assert fw.radii[0] == Q.radius_of(0.3) == 3 and fw.radii[1] > 3
    assert fw.num_rendered >= 2 and float(fw.color[0, 16, 16]) == 0.0 and float(fw.color[0, 8, 8]) > 0.7
    assert fw.n_contrib.reshape(Q.H, Q.W)[16, 16] == 0
    fw.free()
