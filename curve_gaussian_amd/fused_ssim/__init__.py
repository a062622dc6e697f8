"""Drop-in for the reference's ``fused_ssim`` package (/root/reference/submodules/fused-ssim/fused_ssim/__init__.py),
backed by libcurvegs.so.  ``fusedssim`` / ``fusedssim_backward`` mirror the pybind functions of ssim.h:7-26."""
import ctypes as C

import torch

from .. import _lib as L

allowed_padding = ["same", "valid"]


def fusedssim(C1, C2, img1, img2, train=True):
    L.require_gpu_tensor(img1, "img1")
    lib = L.load()
    dev = img1.device
    with L.device_guard(dev):
        img1 = img1.float().contiguous()
        img2 = img2.float().contiguous()
        B, CH, H, W = img1.shape
        target = torch.empty_like(img1)
        if train:
            dm_dmu1, dm_ds1, dm_ds12 = torch.empty_like(img1), torch.empty_like(img1), torch.empty_like(img1)
        else:
            dm_dmu1 = dm_ds1 = dm_ds12 = torch.empty(0)
        rc = lib.cgs_ssim_forward(B, CH, H, W, C.c_float(C1), C.c_float(C2), L.ptr(img1), L.ptr(img2), L.ptr(target),
                                  L.ptr(dm_dmu1), L.ptr(dm_ds1), L.ptr(dm_ds12),
                                  L.raw_stream(dev))
        L.check(rc, "cgs_ssim_forward")
    return target, dm_dmu1, dm_ds1, dm_ds12


def fusedssim_backward(C1, C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12):
    L.require_gpu_tensor(img1, "img1")
    lib = L.load()
    dev = img1.device
    with L.device_guard(dev):
        img1 = img1.float().contiguous()
        img2 = img2.float().contiguous()
        dL_dmap = dL_dmap.float().contiguous()
        B, CH, H, W = img1.shape
        out = torch.empty_like(img1)
        rc = lib.cgs_ssim_backward(B, CH, H, W, C.c_float(C1), C.c_float(C2), L.ptr(img1), L.ptr(img2), L.ptr(dL_dmap),
                                   L.ptr(dm_dmu1), L.ptr(dm_dsigma1_sq), L.ptr(dm_dsigma12), L.ptr(out),
                                   L.raw_stream(dev))
        L.check(rc, "cgs_ssim_backward")
    return out


SSIM_C1, SSIM_C2 = 0.01 ** 2, 0.03 ** 2   # stabilisers of the SSIM ratio for images in [0, 1]
_BORDER = 5                                # half width of the 11-tap window: what "valid" padding crops away


def _crop(t, padding):
    return t if padding == "same" else t[..., _BORDER:-_BORDER, _BORDER:-_BORDER]


class FusedSSIMMap(torch.autograd.Function):
    """SSIM map of (img1, img2) with gradient w.r.t. img1 only; same call signature, saved state and ``None`` pattern
    of the returned gradients as the reference's class of this name (fused_ssim/__init__.py:8-31)."""

    @staticmethod
    def forward(ctx, C1, C2, img1, img2, padding="same", train=True):
        full_map, *partials = fusedssim(C1, C2, img1, img2, train)
        ctx.save_for_backward(img1.detach(), img2, *partials)
        ctx.consts = (C1, C2, padding)
        return _crop(full_map, padding)

    @staticmethod
    def backward(ctx, grad_map):
        img1, img2, d_mu1, d_var1, d_cov = ctx.saved_tensors
        C1, C2, padding = ctx.consts
        if padding != "same":   # the kernels work on the uncropped map: embed the cropped gradient in zeros
            full = torch.zeros_like(img1)
            _crop(full, padding).copy_(grad_map)
            grad_map = full
        g_img1 = fusedssim_backward(C1, C2, img1, img2, grad_map, d_mu1, d_var1, d_cov)
        return (None, None, g_img1, None, None, None)


def fused_ssim(img1, img2, padding="same", train=True):
    """Mean SSIM of two [B, C, H, W] image batches (drop-in for ``fused_ssim.fused_ssim``)."""
    if padding not in allowed_padding:
        raise AssertionError(f"padding must be one of {allowed_padding}")
    return FusedSSIMMap.apply(SSIM_C1, SSIM_C2, img1, img2, padding, train).mean()
