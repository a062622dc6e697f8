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
    with torch.cuda.device(dev):
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
                                  torch.cuda.current_stream(dev).cuda_stream)
        L.check(rc, "cgs_ssim_forward")
    return target, dm_dmu1, dm_ds1, dm_ds12


def fusedssim_backward(C1, C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12):
    L.require_gpu_tensor(img1, "img1")
    lib = L.load()
    dev = img1.device
    with torch.cuda.device(dev):
        img1 = img1.float().contiguous()
        img2 = img2.float().contiguous()
        dL_dmap = dL_dmap.float().contiguous()
        B, CH, H, W = img1.shape
        out = torch.empty_like(img1)
        rc = lib.cgs_ssim_backward(B, CH, H, W, C.c_float(C1), C.c_float(C2), L.ptr(img1), L.ptr(img2), L.ptr(dL_dmap),
                                   L.ptr(dm_dmu1), L.ptr(dm_dsigma1_sq), L.ptr(dm_dsigma12), L.ptr(out),
                                   torch.cuda.current_stream(dev).cuda_stream)
        L.check(rc, "cgs_ssim_backward")
    return out


class FusedSSIMMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, C1, C2, img1, img2, padding="same", train=True):
        ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = fusedssim(C1, C2, img1, img2, train)
        if padding == "valid":
            ssim_map = ssim_map[:, :, 5:-5, 5:-5]
        ctx.save_for_backward(img1.detach(), img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)
        ctx.C1 = C1
        ctx.C2 = C2
        ctx.padding = padding
        return ssim_map

    @staticmethod
    def backward(ctx, opt_grad):
        img1, img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = ctx.saved_tensors
        C1, C2, padding = ctx.C1, ctx.C2, ctx.padding
        dL_dmap = opt_grad
        if padding == "valid":
            dL_dmap = torch.zeros_like(img1)
            dL_dmap[:, :, 5:-5, 5:-5] = opt_grad
        grad = fusedssim_backward(C1, C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)
        return None, None, grad, None, None, None


def fused_ssim(img1, img2, padding="same", train=True):
    C1 = 0.01 ** 2
    C2 = 0.03 ** 2
    assert padding in allowed_padding
    map = FusedSSIMMap.apply(C1, C2, img1, img2, padding, train)
    return map.mean()
