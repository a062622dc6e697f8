"""CPU: the C-ABI library loads, exports every symbol declared in include/curvegs.h, and the Python operator layer
validates arguments like the reference before touching a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "curvegs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cgs_[a-z0-9_]+)\s*\(", src)) - {"cgs_alloc_fn"})


def test_library_exports_every_declared_symbol():
    from curve_gaussian_amd import _lib
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/curvegs.h but not exported by libcurvegs.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in curve_gaussian_amd/_lib.py"
    assert lib.cgs_target_arch() == b"gfx950" and lib.cgs_version() >= 100


def test_workspace_size_queries_are_consistent():
    from curve_gaussian_amd import _lib
    lib = _lib.load()
    assert lib.cgs_geometry_bytes(1000) >= 1000 * (64 + 64 + 4 + 1)  # 64 B record + 64 B accumulators per splat
    assert lib.cgs_image_bytes(1600, 1600) >= 1600 * 1600 * 8 + 10000 * 16
    assert lib.cgs_binning_bytes(10 ** 6) >= 12 * 10 ** 6
    assert lib.cgs_geometry_bytes(2000) > lib.cgs_geometry_bytes(1000)
    assert lib.cgs_knn_workspace_bytes(3375) >= 3375 * 12


def test_invalid_arguments_are_rejected_without_a_gpu():
    from curve_gaussian_amd import _lib
    lib = _lib.load()
    cb = _lib.ALLOC_FN(lambda u, n: None)
    rc = lib.cgs_rasterize_forward(cb, None, cb, None, cb, None, 10, 0, 0, None, 64, 64, *([None] * 5), 1.0,
                                   *([None] * 6), 0.3, 0.3, 0, None, None, None, 0, 1, None, 0, None)
    assert rc == -1 and b"invalid argument" in lib.cgs_last_error()
    assert lib.cgs_mark_visible(-1, None, None, None, None, None) == -1
    assert lib.cgs_sample_curves_forward(5, 12, None, None, None, None, ctypes.c_float(1e-8), None, None, None, None, None) == -1
    assert lib.cgs_knn_mean_dist2(0, None, None, None, None) == 0  # empty input is a no-op


def test_rasterizer_argument_checks_match_reference():
    """GaussianRasterizer.forward raises before any device work (reference __init__.py:189-193)."""
    from curve_gaussian_amd.diff_cur_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    fields = GaussianRasterizationSettings._fields
    # the reference's 14 fields in the reference's order (:153-167), then the optional extensions with defaults
    assert fields[:14] == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                           "projmatrix", "sh_degree", "campos", "prefiltered", "debug", "antialiasing", "render_geo")
    assert fields[14:] == ("static_bucket_cap", "status_sink", "options")
    assert GaussianRasterizationSettings._field_defaults == {"static_bucket_cap": 0, "status_sink": None, "options": 0}
    rs = GaussianRasterizationSettings(8, 8, 0.3, 0.3, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                       torch.zeros(3), False, False, False, True)
    r = GaussianRasterizer(rs)
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=torch.zeros(4, 1), scales=m, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 1), colors_precomp=torch.zeros(4, 1),
          scales=m, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=torch.zeros(4, 1), colors_precomp=torch.zeros(4, 1), scales=m)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=torch.zeros(4, 1), colors_precomp=torch.zeros(4, 1), scales=m,
          rotations=torch.zeros(4, 4), cov3D_precomp=torch.zeros(4, 6))


def test_host_shim_loads_and_mirrors_the_reference_pybind_module(monkeypatch):
    """curve_gaussian_amd._cgs_torch (csrc/torch_shim.cpp) is what `diff_cur_rasterization._C` resolves to: the three entry
    points of the reference's pybind module (ext.cpp:15-19) with its argument counts, plus the extensions; CGS_TORCH_SHIM=0
    selects the ctypes bindings over the same library."""
    from curve_gaussian_amd import _lib
    from curve_gaussian_amd import diff_cur_rasterization as D
    shim = _lib.shim()
    for name in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible", "rasterize_gaussians_static",
                 "forward_status", "view_forward", "view_backward", "view_wait", "view_abandon"):
        assert callable(getattr(shim, name)), name
    assert _lib.use_shim()
    monkeypatch.setattr(D._ExtProxy, "_impl", None)
    assert D._C.rasterize_gaussians is shim.rasterize_gaussians
    # RasterizeGaussiansCUDA takes 22 arguments (rasterize_points.h:18-44): one short is a TypeError before any device work
    with pytest.raises(TypeError):
        shim.rasterize_gaussians(*([torch.zeros(3)] * 21))
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):   # rasterize_points.cu:60-62
        shim.rasterize_gaussians(torch.zeros(3), torch.zeros(5), torch.ones(5, 1), torch.ones(5, 1), torch.ones(5, 3),
                                 torch.ones(5, 4), 1.0, torch.empty(0), torch.ones(5, 4), torch.eye(4), torch.eye(4), 0.5, 0.5, 16,
                                 16, torch.empty(0), 0, torch.zeros(3), False, False, True, False)
    monkeypatch.setenv("CGS_TORCH_SHIM", "0")
    monkeypatch.setattr(D._ExtProxy, "_impl", None)
    assert D._C.mark_visible is not None and isinstance(D._ExtProxy._impl, D._Ext)
    monkeypatch.setattr(D._ExtProxy, "_impl", None)


def test_product_has_no_cpu_fallback():
    from curve_gaussian_amd import _lib
    from curve_gaussian_amd.diff_cur_rasterization import _C
    from curve_gaussian_amd.fused_ssim import fused_ssim
    from curve_gaussian_amd.ops.curve_sampling import sample_curves
    from curve_gaussian_amd.simple_knn import distCUDA2
    with pytest.raises(_lib.CurveGSError, match="GPU tensor"):
        _C.mark_visible(torch.zeros(3, 3), torch.eye(4), torch.eye(4))
    with pytest.raises(_lib.CurveGSError, match="GPU tensor"):
        fused_ssim(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8))
    with pytest.raises(_lib.CurveGSError, match="GPU tensor"):
        sample_curves(torch.zeros(2, 4, 3), torch.zeros(2, 1))
    with pytest.raises(_lib.CurveGSError, match="GPU tensor"):
        distCUDA2(torch.zeros(5, 3))
    e = torch.empty(0)
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        _C.rasterize_gaussians(torch.zeros(3), torch.zeros(4), e, e, e, e, 1.0, e, e, torch.eye(4), torch.eye(4), 0.5, 0.5, 8, 8, e, 0,
                               torch.zeros(3), False, False, True, False)


def test_sampling_coefficients_match_reference_expressions():
    from curve_gaussian_amd.ops.curve_sampling import sample_coefficients
    from oracle import torch_ref as TR
    m = 12
    c = sample_coefficients(m, "cpu")
    t = TR.sample_t(m)[:, 0, 0]
    assert torch.equal(c[:, 0], (1 - t) ** 3) and torch.equal(c[:, 9], 6 * (1 - t) * t)
    assert torch.allclose(c[:, :4].sum(1), torch.ones(m), atol=1e-6)
    assert torch.equal(c[:, 13], 1 - (t - 0.5 / m))
