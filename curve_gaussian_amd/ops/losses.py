"""Fused losses of the training step (reference train.py:101-107).

``edge_aware_loss`` == utils/loss_utils.py:94-115 as one HIP op (value + gradient in the same pass);
``fused_ssim`` lives in curve_gaussian_amd.fused_ssim (drop-in for the reference package)."""
import ctypes as C

import torch

from .. import _lib as L


class _EdgeAwareLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt_image, threshold):
        L.require_gpu_tensor(image, "image")
        lib = L.load()
        dev = image.device
        with torch.cuda.device(dev):
            img = image.detach().float().contiguous()
            gt = gt_image.detach().float().contiguous()
            Cn, H, W = img.shape[-3], img.shape[-2], img.shape[-1]
            scratch = torch.empty(2, dtype=torch.float64, device=dev)
            grad = torch.empty_like(img)
            rc = lib.cgs_edge_aware_loss(Cn, H, W, L.ptr(img), L.ptr(gt), C.c_float(threshold), L.ptr(scratch),
                                         L.ptr(grad), torch.cuda.current_stream(dev).cuda_stream)
            L.check(rc, "cgs_edge_aware_loss")
            loss = (scratch[1] / float(Cn * H * W)).float()
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def edge_aware_loss(image, gt_image, threshold=0.1):
    """image, gt_image: [C,H,W] (the reference passes the 1-channel render and gt[:1])."""
    return _EdgeAwareLoss.apply(image, gt_image, threshold)


class _PhotometricLoss(torch.autograd.Function):
    """loss = lambda_mse * ((1 - lambda_dssim) * edge_aware_loss(image, gt) + lambda_dssim * (1 - fused_ssim(image, gt)))
    (train.py:101-107) with value and d loss / d image produced in one forward: two loss kernels + the two SSIM kernels +
    one mean + one add, instead of ~25 scalar / elementwise kernels and their autograd graph."""
    _const_cache = {}

    @staticmethod
    def forward(ctx, image, gt_image, lambda_mse, lambda_dssim, threshold):
        from ..fused_ssim import fusedssim, fusedssim_backward
        L.require_gpu_tensor(image, "image")
        lib = L.load()
        dev = image.device
        with torch.cuda.device(dev):
            img = image.detach().float().contiguous()
            gt = gt_image.detach().float().contiguous()
            Cn, H, W = img.shape
            n = float(Cn * H * W)
            scratch = torch.empty(2, dtype=torch.float64, device=dev)
            g_edge = torch.empty_like(img)
            rc = lib.cgs_edge_aware_loss(Cn, H, W, L.ptr(img), L.ptr(gt), C.c_float(threshold), L.ptr(scratch),
                                         L.ptr(g_edge), torch.cuda.current_stream(dev).cuda_stream)
            L.check(rc, "cgs_edge_aware_loss")
            C1, C2 = 0.01 ** 2, 0.03 ** 2
            i4, g4 = img.unsqueeze(0), gt.unsqueeze(0)
            ssim_map, dm1, dm2, dm3 = fusedssim(C1, C2, i4, g4, True)
            a = lambda_mse * (1.0 - lambda_dssim)
            b = lambda_mse * lambda_dssim
            key = (str(dev), Cn, H, W, b)
            cmap = _PhotometricLoss._const_cache.get(key)
            if cmap is None:    # d(b (1 - mean ssim)) / d ssim_map = -b / N everywhere (constant, cached)
                cmap = torch.full((1, Cn, H, W), -b / n, dtype=torch.float32, device=dev)
                _PhotometricLoss._const_cache = {key: cmap}
            g_ssim = fusedssim_backward(C1, C2, i4, g4, cmap, dm1, dm2, dm3)
            grad = torch.add(g_ssim.squeeze(0), g_edge, alpha=a)
            loss = (scratch[1] * (a / n) + b - b * ssim_map.mean(dtype=torch.float64)).float()
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None


def photometric_loss(image, gt_image, lambda_mse=10.0, lambda_dssim=0.1, threshold=0.1):
    """image, gt_image: [C,H,W].  Same value/gradient as composing edge_aware_loss and fused_ssim (tested)."""
    return _PhotometricLoss.apply(image, gt_image, lambda_mse, lambda_dssim, threshold)
