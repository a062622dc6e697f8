"""curve_gaussian_amd -- MI355X (gfx950) native hot path of Curve-Gaussian.

HIP kernels + C ABI live in ``csrc/`` (built into ``libcurvegs.so``); the sub-packages mirror the reference's
Python operator API (``diff_cur_rasterization``, ``fused_ssim``, ``simple_knn``, ``gaussian_renderer``,
``scene.gaussian_curve_model``).  There is no CPU fallback: the CPU restatement lives in ``oracle/`` and is
test infrastructure only.
"""
__version__ = "0.1.0"
