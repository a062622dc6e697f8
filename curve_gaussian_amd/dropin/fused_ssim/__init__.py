"""PYTHONPATH shim for the reference's ``from fused_ssim import fused_ssim``."""
from curve_gaussian_amd.fused_ssim import *  # noqa: F401,F403
from curve_gaussian_amd.fused_ssim import FusedSSIMMap, fused_ssim, fusedssim, fusedssim_backward  # noqa: F401
