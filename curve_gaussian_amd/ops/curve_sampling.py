"""Curve -> Gaussian sampling and per-view splat attributes, backed by the fused HIP kernels of csrc/sampling.hip.

``sample_curves``  == GaussianCurveModel.prepare_scaling_rot (reference scene/gaussian_curve_model.py:180-198 with
                      :70-89 and rot_to_quat_batch, utils/general_utils.py:33-86), forward + hand-written backward.
``splat_attributes`` == get_rotation / get_opacity / straight-through mask / get_main_axis + all_map build
                      (:99-110,121-122; gaussian_renderer/__init__.py:72-76,98-104), forward + backward.

GPU only: there is no CPU path in the product (the PyTorch restatement lives in oracle/torch_ref.py).
"""
import ctypes as C

import torch

from .. import _lib as L

_f = C.c_float
_COEF_CACHE = {}
_ALL_BEZIER_CACHE = {}


def _bezier_mask(is_bezier, dev):
    """u8 mask for the kernel, or None when every curve is a Bezier curve (the common case; the reference branches on
    ``self.is_bezier.all()`` too, :74,84).  The device->host read behind ``.all()`` is cached per tensor version so the
    per-step path stays free of host syncs."""
    if is_bezier is None:
        return None
    key = (is_bezier.data_ptr(), is_bezier._version, is_bezier.numel(), str(is_bezier.device))
    hit = _ALL_BEZIER_CACHE.get(key)
    if hit is None:
        if len(_ALL_BEZIER_CACHE) > 64:
            _ALL_BEZIER_CACHE.clear()
        allb = bool(is_bezier.all())
        # the entry holds the key tensor itself: while it is cached its storage cannot be freed and handed to a NEW
        # is_bezier tensor with the same address and version 0 (topology edits allocate fresh ones), which would alias
        # a stale mask
        hit = (allb, None if allb else is_bezier.to(device=dev, dtype=torch.uint8).contiguous(), is_bezier)
        _ALL_BEZIER_CACHE[key] = hit
    return hit[1]


def sample_coefficients(m: int, device) -> torch.Tensor:
    """[m,16] float32 per-sample weights, evaluated on the host with the same float32 torch expressions the
    reference uses (sample_t = linspace(0.5/m, 1-0.5/m, m); (1-t)**3, 3*(1-t)**2*t, ...), so the kernel reproduces
    torch's rounding of the polynomial weights exactly."""
    key = (m, str(device))
    c = _COEF_CACHE.get(key)
    if c is None:
        t = torch.linspace(0.5 / m, 1 - 0.5 / m, m, dtype=torch.float32)
        tf = t - 0.5 / m
        cols = [(1 - t) ** 3, 3 * (1 - t) ** 2 * t, 3 * (1 - t) * t ** 2, t ** 3,
                (1 - tf) ** 3, 3 * (1 - tf) ** 2 * tf, 3 * (1 - tf) * tf ** 2, tf ** 3,
                3 * (1 - t) ** 2, 6 * (1 - t) * t, 3 * t ** 2,
                (1 - t), t, (1 - tf), tf, torch.zeros_like(t)]
        c = torch.stack(cols, dim=1).contiguous().to(device)
        _COEF_CACHE[key] = c
    return c


def _stream(dev):
    return L.raw_stream(dev)


class _SampleCurves(torch.autograd.Function):
    @staticmethod
    def forward(ctx, curve_points, width, is_bezier, m, eps):
        L.require_gpu_tensor(curve_points, "curve_points")
        lib = L.load()
        dev = curve_points.device
        with L.device_guard(dev):
            cp = curve_points.detach().float().contiguous()
            w = width.detach().float().contiguous()
            B = cp.shape[0]
            P = B * m
            isb = _bezier_mask(is_bezier, dev)
            coef = sample_coefficients(m, dev)
            norms = torch.empty(384, dtype=torch.float64, device=dev)
            xyz = torch.empty((P, 3), dtype=torch.float32, device=dev)
            rot = torch.empty((P, 4), dtype=torch.float32, device=dev)
            scl = torch.empty((P, 3), dtype=torch.float32, device=dev)
            rc = lib.cgs_sample_curves_forward(B, m, L.ptr(cp), L.ptr(w), L.ptr(isb), L.ptr(coef), _f(eps),
                                               L.ptr(norms), L.ptr(xyz), L.ptr(rot), L.ptr(scl), _stream(dev))
            L.check(rc, "cgs_sample_curves_forward")
        ctx.save_for_backward(cp, w, isb if isb is not None else torch.empty(0, device=dev), coef, norms)
        ctx.m, ctx.eps = m, eps
        ctx.set_materialize_grads(False)
        return xyz, rot, scl

    @staticmethod
    def backward(ctx, g_xyz, g_rot, g_scl):
        cp, w, isb, coef, norms = ctx.saved_tensors
        lib = L.load()
        dev = cp.device
        B = cp.shape[0]
        with L.device_guard(dev):
            c = lambda t: None if t is None else t.float().contiguous()
            g_xyz, g_rot, g_scl = c(g_xyz), c(g_rot), c(g_scl)
            g_cp = torch.empty_like(cp)
            g_w = torch.empty_like(w)
            # norms[192:320] are backward scratch (re-zeroed by every call); norms[:192] (forward partial sums) are only read
            scratch = torch.empty((B * ctx.m, 9), dtype=torch.float32, device=dev) if g_rot is not None else None
            rc = lib.cgs_sample_curves_backward(B, ctx.m, L.ptr(cp), L.ptr(w), L.ptr(isb), L.ptr(coef), _f(ctx.eps),
                                                L.ptr(norms), L.ptr(g_xyz), L.ptr(g_rot), L.ptr(g_scl), L.ptr(g_cp),
                                                L.ptr(g_w), L.ptr(scratch), _stream(dev))
            L.check(rc, "cgs_sample_curves_backward")
        return g_cp, g_w, None, None, None


def sample_curves(curve_points, width, is_bezier=None, m: int = 12, eps: float = 1e-8):
    """-> (_xyz [P,3], _rotation [P,4] un-normalised (w,x,y,z), _scaling [P,3]); splat index = b*m + i."""
    return _SampleCurves.apply(curve_points, width, is_bezier, m, eps)


class _SplatAttrs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rot_raw, xyz, opacity_logit, scaling, mask_logit, mask_thr, campos, viewmatrix, m):
        L.require_gpu_tensor(rot_raw, "rotation")
        lib = L.load()
        dev = rot_raw.device
        with L.device_guard(dev):
            c = lambda t: None if t is None else t.detach().float().contiguous()
            rot_raw, xyz, opacity_logit, scaling, mask_logit = c(rot_raw), c(xyz), c(opacity_logit), c(scaling), c(mask_logit)
            campos, viewmatrix = c(campos), c(viewmatrix)
            P = rot_raw.shape[0]
            B = P // m
            rot_n = torch.empty_like(rot_raw)
            opac = torch.empty((P, 1), dtype=torch.float32, device=dev)
            all_map = torch.empty((P, 4), dtype=torch.float32, device=dev)
            scl_out = torch.empty_like(scaling) if mask_logit is not None else None
            rc = lib.cgs_splat_attrs_forward(B, m, L.ptr(rot_raw), L.ptr(xyz), L.ptr(opacity_logit), L.ptr(mask_logit),
                                             _f(mask_thr), L.ptr(scaling), L.ptr(campos), L.ptr(viewmatrix),
                                             L.ptr(rot_n), L.ptr(opac), L.ptr(scl_out), L.ptr(all_map), _stream(dev))
            L.check(rc, "cgs_splat_attrs_forward")
        e = torch.empty(0, device=dev)
        ctx.save_for_backward(rot_raw, xyz, opacity_logit, scaling, mask_logit if mask_logit is not None else e, campos,
                              viewmatrix)
        ctx.m, ctx.mask_thr, ctx.has_mask = m, mask_thr, mask_logit is not None
        ctx.set_materialize_grads(False)
        if scl_out is None:
            scl_out = torch.empty(0, device=dev)  # no mask: the wrapper hands the caller's scaling tensor through
        return rot_n, opac, scl_out, all_map

    @staticmethod
    def backward(ctx, g_rot_n, g_opac, g_scl_out, g_all_map):
        rot_raw, xyz, opacity_logit, scaling, mask_logit, campos, viewmatrix = ctx.saved_tensors
        lib = L.load()
        dev = rot_raw.device
        m = ctx.m
        P = rot_raw.shape[0]
        B = P // m
        with L.device_guard(dev):
            c = lambda t: None if t is None else t.float().contiguous()
            g_rot_n, g_opac, g_scl_out, g_all_map = c(g_rot_n), c(g_opac), c(g_scl_out), c(g_all_map)
            g_rot_raw = torch.empty_like(rot_raw)
            g_logit = torch.empty_like(opacity_logit)
            if ctx.has_mask:
                g_mask = torch.empty_like(mask_logit)
                g_scaling = torch.empty_like(scaling) if g_scl_out is not None else None
            else:
                g_mask = None
                g_scaling = None
            rc = lib.cgs_splat_attrs_backward(
                B, m, L.ptr(rot_raw), L.ptr(xyz), L.ptr(opacity_logit), L.ptr(mask_logit) if ctx.has_mask else None,
                _f(ctx.mask_thr), L.ptr(scaling), L.ptr(campos), L.ptr(viewmatrix), L.ptr(g_rot_n), L.ptr(g_opac),
                L.ptr(g_scl_out) if ctx.has_mask else None, L.ptr(g_all_map), L.ptr(g_rot_raw), L.ptr(g_logit),
                L.ptr(g_mask), L.ptr(g_scaling), _stream(dev))
            L.check(rc, "cgs_splat_attrs_backward")
        # xyz only enters through the (non-differentiable) camera-facing sign test
        return g_rot_raw, None, g_logit, g_scaling, g_mask, None, None, None, None


def splat_attributes(rot_raw, xyz, opacity_logit, scaling, campos, viewmatrix, m: int = 12, mask_logit=None,
                     mask_thr: float = 0.01):
    """-> (rotations [P,4] normalised, opacity [P,1], scales [P,3], all_map [P,4]) for one view."""
    rot_n, opac, scl_out, all_map = _SplatAttrs.apply(rot_raw, xyz, opacity_logit, scaling, mask_logit, mask_thr,
                                                      campos, viewmatrix, m)
    return rot_n, opac, (scl_out if mask_logit is not None else scaling), all_map


def quaternion_to_matrix(q):
    """pytorch3d.transforms.quaternion_to_matrix (used by the reference's get_rotation_matrix, :95-97); torch ops --
    off the per-view hot path (the hot path gets column 0 from the fused splat_attributes kernel)."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))
