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
