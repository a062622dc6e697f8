"""Fused losses of the training step (reference train.py:101-107).

``edge_aware_loss`` == utils/loss_utils.py:94-115 as one HIP op (value + gradient in the same pass);
``fused_ssim`` lives in curve_gaussian_amd.fused_ssim (drop-in for the reference package)."""
import ctypes as C

import torch

from .. import _lib as L


_UNIT = {}


def unit_grad(device):
    """A shared 0-dim tensor holding 1.0.  ``loss.backward(gradient=unit_grad(dev))`` (what the train steps do) lets the
    loss ops recognise the usual d loss / d loss = 1 by its storage and hand their pre-computed gradient on without
    the ``grad * g`` kernel; any other upstream gradient is multiplied in as autograd requires."""
    key = str(device)
    t = _UNIT.get(key)
    if t is None:
        t = _UNIT[key] = torch.ones((), dtype=torch.float32, device=device)
    return t


def scale_by_upstream(grad, g):
    u = _UNIT.get(str(grad.device))
    if u is not None and g.data_ptr() == u.data_ptr() and g.numel() == 1:
        return grad
    return grad * g


class _EdgeAwareLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt_image, threshold):
        L.require_gpu_tensor(image, "image")
        lib = L.load()
        dev = image.device
        with L.device_guard(dev):
            img = image.detach().float().contiguous()
            gt = gt_image.detach().float().contiguous()
            Cn, H, W = img.shape[-3], img.shape[-2], img.shape[-1]
            scratch = torch.empty(2, dtype=torch.float64, device=dev)
            grad = torch.empty_like(img)
            rc = lib.cgs_edge_aware_loss(Cn, H, W, L.ptr(img), L.ptr(gt), C.c_float(threshold), L.ptr(scratch),
                                         L.ptr(grad), L.raw_stream(dev))
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


class _EdgeCountCache:
    """#{gt > threshold} depends on the gt edge map only; the reference recomputes it every iteration
    (loss_utils.py:100-103).  One device scalar per (gt storage, version, threshold), computed on first use."""
    _cache = {}

    @classmethod
    def get(cls, gt, threshold, stream):
        key = (gt.data_ptr(), gt._version, tuple(gt.shape), float(threshold), str(gt.device))
        hit = cls._cache.get(key)
        if hit is None:
            if len(cls._cache) > 4096:
                cls._cache.clear()
            n_pos = torch.empty(1, dtype=torch.int32, device=gt.device)
            Cn, H, W = gt.shape
            rc = L.load().cgs_edge_count(Cn, H, W, L.ptr(gt), C.c_float(threshold), L.ptr(n_pos), stream)
            L.check(rc, "cgs_edge_count")
            hit = cls._cache[key] = (n_pos, gt)   # keep gt alive so the data_ptr key cannot be recycled
        return hit[0]


class _PhotometricLoss(torch.autograd.Function):
    """loss = lambda_mse * ((1 - lambda_dssim) * edge_aware_loss(x, gt) + lambda_dssim * (1 - fused_ssim(x, gt))),
    x = clamp(image, 0, 1) if clamp else image (train.py:101-107 + render()'s clamp), value and d loss / d image from
    cgs_photometric_loss: three kernels instead of ~30 elementwise / reduction launches and their autograd graph."""
    _workspaces = {}

    @staticmethod
    def forward(ctx, image, gt_image, lambda_mse, lambda_dssim, threshold, clamp, n_pos):
        L.require_gpu_tensor(image, "image")
        lib = L.load()
        dev = image.device
        with L.device_guard(dev):
            img = image.detach().float().contiguous()
            gt = gt_image.detach().float().contiguous()
            Cn, H, W = img.shape
            if Cn != 1:
                raise L.CurveGSError("photometric_loss: the fused path renders 1 channel (got %d)" % Cn)
            stream = L.raw_stream(dev)
            if n_pos is None:
                n_pos = _EdgeCountCache.get(gt, threshold, stream)
            key = (str(dev), H, W, stream)
            ws = _PhotometricLoss._workspaces.get(key)
            if ws is None:
                # bounded: every hipGraph capture runs on a fresh stream and would otherwise pin one workspace (3 maps)
                # per capture forever.  Dropping an entry is safe: a captured graph keeps using the block inside its
                # own memory pool, eager callers simply allocate a new one.
                while len(_PhotometricLoss._workspaces) >= 6:
                    _PhotometricLoss._workspaces.pop(next(iter(_PhotometricLoss._workspaces)))
                ws = _PhotometricLoss._workspaces[key] = torch.zeros(
                    int(lib.cgs_photometric_workspace_bytes(H, W)), dtype=torch.uint8, device=dev)
            grad = torch.empty_like(img)
            loss = torch.empty((), dtype=torch.float32, device=dev)
            a = lambda_mse * (1.0 - lambda_dssim)
            b = lambda_mse * lambda_dssim
            rc = lib.cgs_photometric_loss(H, W, L.ptr(img), L.ptr(gt), C.c_float(threshold), L.ptr(n_pos), C.c_float(a),
                                          C.c_float(b), 1 if clamp else 0, L.ptr(ws), L.ptr(grad), L.ptr(loss), stream)
            L.check(rc, "cgs_photometric_loss")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return scale_by_upstream(grad, g), None, None, None, None, None, None


def edge_pixel_count(gt_image, threshold=0.1):
    """Device int32 scalar #{mean_c gt > threshold} (loss_utils.py:100-103), cached per gt tensor."""
    gt = gt_image.detach().float().contiguous()
    return _EdgeCountCache.get(gt, threshold, L.raw_stream(gt.device))


def photometric_loss(image, gt_image, lambda_mse=10.0, lambda_dssim=0.1, threshold=0.1, clamp=False, n_pos=None):
    """image, gt_image: [1,H,W].  Same value/gradient as composing (clamp,) edge_aware_loss and fused_ssim (tested).
    clamp=True takes the UNclamped rasterizer output and applies render()'s clamp(0,1) inside the kernels.
    n_pos: optional device int32 scalar from ``edge_pixel_count`` (graph-captured steps feed it through a static buffer)."""
    if L.use_shim() and image.is_cuda and image.dim() == 3 and image.shape[0] == 1:
        # the C++ autograd node of the compiled shim (csrc/torch_shim.cpp::PhotometricLossFn): same kernels, no Python frame in
        # the backward
        lib = L.load()
        dev = image.device
        gt = gt_image.detach().float().contiguous()
        H, W = int(image.shape[1]), int(image.shape[2])
        stream = L.raw_stream(dev)
        if n_pos is None:
            n_pos = _EdgeCountCache.get(gt, threshold, stream)
        key = (str(dev), H, W, stream)
        ws = _PhotometricLoss._workspaces.get(key)
        if ws is None:
            while len(_PhotometricLoss._workspaces) >= 6:
                _PhotometricLoss._workspaces.pop(next(iter(_PhotometricLoss._workspaces)))
            ws = _PhotometricLoss._workspaces[key] = torch.zeros(int(lib.cgs_photometric_workspace_bytes(H, W)), dtype=torch.uint8,
                                                                  device=dev)
        return L.shim().photometric_loss(image, gt, n_pos, ws, float(threshold), lambda_mse * (1.0 - lambda_dssim),
                                         lambda_mse * lambda_dssim, bool(clamp), unit_grad(dev))
    return _PhotometricLoss.apply(image, gt_image, lambda_mse, lambda_dssim, threshold, clamp, n_pos)
