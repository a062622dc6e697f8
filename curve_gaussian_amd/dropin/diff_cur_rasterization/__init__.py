"""PYTHONPATH shim: lets the reference's ``from diff_cur_rasterization import ...`` resolve to libcurvegs."""
from curve_gaussian_amd.diff_cur_rasterization import *  # noqa: F401,F403
from curve_gaussian_amd.diff_cur_rasterization import _C, _RasterizeGaussians, rasterize_gaussians  # noqa: F401
