"""The per-iteration regularisers of the reference's training loop (train.py:110-131), sync-free.

  mask          lambda_mask * mean(sigmoid(_mask))                                  (:110-111, iteration >= densify_until)
  opacity       opacity_loss_weight * mean(log(1 + o^2 / 0.5)) over visible splats  (:114-117, after an opacity reset)
  curve_smo     lambda_curve_smo * mean(1 - |cos(d_i, d_{i+1})|), d = main axis     (:119-124, if any splat is visible)
  width         lambda_width * mean(width - 0.005 over curves with width >= 0.005)  (:126-131)

The reference decides with host-side conditions (``visibility_filter.sum() > 0``, ``mask.any()``: a device-to-host
sync each); here the same values come out of masked means whose denominators are clamped, so the step stays
stream-ordered and capturable.

  points_conn   lambda_points_conn * mean distance of end points of different curves closer than 0.05   (:133-146,
                iteration > conn_from_iter) -- ``connection_loss_reference`` is the literal torch.cdist form (O(B^2)
                memory, small B only: the test reference), ``connection_loss`` the O(B)-memory HIP op.

The plain torch functions act on per-curve / per-splat tensors once per iteration, outside the per-view hot path; the
fused ops replace them in the training step."""
import torch
import torch.nn.functional as F


def mask_loss(gaussians, lambda_mask=0.0005):
    return lambda_mask * torch.mean(torch.sigmoid(gaussians._mask))


def opacity_loss(gaussians, radii, weight=0.01):
    """train.py:114-117 with visibility_filter = (radii > 0): mean over the visible splats, 0 if none is visible."""
    vis = (radii > 0).to(torch.float32).unsqueeze(-1)
    o = gaussians.get_opacity
    return weight * (torch.log(1 + o ** 2 / 0.5) * vis).sum() / vis.sum().clamp(min=1.0)


def curve_smoothness_loss(gaussians, radii, weight=0.1):
    """train.py:119-124: applied only when at least one splat is visible (the factor below is 0 or 1)."""
    m = gaussians.n_gaussians
    d = gaussians.get_rotation_matrix[..., 0].reshape(-1, m, 3)
    cos_sim = 1 - F.cosine_similarity(d[:, :-1, :], d[:, 1:, :], dim=-1).abs()
    any_visible = (radii > 0).any().to(torch.float32)
    return weight * cos_sim.mean() * any_visible


def width_loss(gaussians, weight=0.01, width_thr=0.005):
    """train.py:126-131: mean excess over the curves at or above the threshold, 0 if there is none."""
    w = gaussians.get_curve_width
    sel = (w >= width_thr).to(w.dtype)
    return weight * ((w - width_thr) * sel).sum() / sel.sum().clamp(min=1.0)


# ---------------------------------------------------------------------------------------------- fused HIP version
import ctypes as _C

from .. import _lib as _L


class _CurveRegularizers(torch.autograd.Function):
    """opacity + smoothness + width terms above from cgs_curve_regularizers: value and gradients in three launches
    (the torch-op versions cost ~60 launches and 0.8 ms at P = 200 k)."""
    _workspaces = {}

    @staticmethod
    def forward(ctx, rotation_raw, opacity_logit, width_log, radii, m, w_opacity, opacity_gate, w_smooth, w_width, width_thr):
        _L.require_gpu_tensor(rotation_raw, "rotation")
        lib = _L.load()
        dev = rotation_raw.device
        rot = rotation_raw.detach().float().contiguous()
        op = opacity_logit.detach().float().contiguous()
        wl = width_log.detach().float().contiguous()
        rad = radii.detach().to(torch.int32).contiguous()
        P = rot.shape[0]
        B = P // m
        stream = _L.raw_stream(dev)
        key = (str(dev), stream)
        ws = _CurveRegularizers._workspaces.get(key)
        if ws is None:
            while len(_CurveRegularizers._workspaces) >= 8:
                _CurveRegularizers._workspaces.pop(next(iter(_CurveRegularizers._workspaces)))
            ws = _CurveRegularizers._workspaces[key] = torch.zeros(
                int(lib.cgs_curve_regularizers_workspace_bytes()), dtype=torch.uint8, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        g_rot = torch.empty_like(rot)
        g_op = torch.empty_like(op)
        g_w = torch.empty_like(wl)
        gate = None
        if torch.is_tensor(opacity_gate):
            gate = opacity_gate.detach().float().contiguous()
            w_op = float(w_opacity)
        else:
            w_op = float(w_opacity) * float(opacity_gate)
        rc = lib.cgs_curve_regularizers(B, int(m), _L.ptr(rot), _L.ptr(op), _L.ptr(wl), _L.ptr(rad), _C.c_float(w_op),
                                        _L.ptr(gate) if gate is not None else None, _C.c_float(w_smooth),
                                        _C.c_float(w_width), _C.c_float(width_thr), _L.ptr(ws), _L.ptr(loss),
                                        _L.ptr(g_rot), _L.ptr(g_op), _L.ptr(g_w), stream)
        _L.check(rc, "cgs_curve_regularizers")
        ctx.save_for_backward(g_rot, g_op, g_w)
        ctx.shapes = (rotation_raw.shape, opacity_logit.shape, width_log.shape)
        return loss

    @staticmethod
    def backward(ctx, g):
        g_rot, g_op, g_w = ctx.saved_tensors
        s = ctx.shapes
        from .losses import scale_by_upstream as sc
        return (sc(g_rot, g).view(s[0]), sc(g_op, g).view(s[1]), sc(g_w, g).view(s[2]), None, None, None, None, None,
                None, None)


def curve_regularizers(gaussians, radii, w_opacity=0.01, opacity_gate=1.0, w_smooth=0.1, w_width=0.01, width_thr=0.005):
    """opacity_loss * gate + curve_smoothness_loss + width_loss (the three functions above), fused.
    opacity_gate: float or 0-dim device tensor (train.py:114's ``reset_timestep > 0``; the counter is incremented at the
    top of every iteration, train.py:76, so the gate is 1 from the first iteration on)."""
    return _CurveRegularizers.apply(gaussians._rotation, gaussians._opacity, gaussians._width, radii, gaussians.n_gaussians,
                                    w_opacity, opacity_gate, w_smooth, w_width, width_thr)


def connection_loss_reference(gaussians, weight=0.1, dis_thr=0.05):
    """train.py:133-146 as written (torch.cdist over all 2B end points; exact differences instead of the matmul
    expansion cdist may pick for large inputs).  O(B^2) memory: the test reference of ``connection_loss``."""
    curve_points = gaussians.get_curve_points
    start_points, end_points = curve_points[:, 0], curve_points[:, -1]
    all_points = torch.cat([start_points, end_points], dim=0)
    mask = torch.eye(len(start_points), dtype=torch.bool, device=start_points.device)
    mask = torch.cat([torch.cat([mask, mask], dim=1), torch.cat([mask, mask], dim=1)], dim=0)
    dist = torch.cdist(all_points, all_points, p=2, compute_mode="donot_use_mm_for_euclid_dist")
    with torch.no_grad():
        valid_mask = (dist < dis_thr) & (~mask)
    if valid_mask.any():
        return weight * dist[valid_mask].mean()
    return dist.sum() * 0.0


class _ConnectionLoss(torch.autograd.Function):
    """cgs_endpoint_connection_loss: value and gradient in one sweep, O(B) memory."""

    @staticmethod
    def forward(ctx, curve_points, weight, dis_thr):
        _L.require_gpu_tensor(curve_points, "curve_points")
        lib = _L.load()
        dev = curve_points.device
        cp = curve_points.detach().float().contiguous()
        B = cp.shape[0]
        with _L.device_guard(dev):
            ws = torch.empty(int(lib.cgs_endpoint_connection_workspace_bytes(B)), dtype=torch.uint8, device=dev)
            loss = torch.empty((), dtype=torch.float32, device=dev)
            grad = torch.empty_like(cp)
            rc = lib.cgs_endpoint_connection_loss(B, _L.ptr(cp), _C.c_float(dis_thr), _C.c_float(weight), _L.ptr(ws),
                                                  _L.ptr(loss), _L.ptr(grad), 0, _L.raw_stream(dev))
            _L.check(rc, "cgs_endpoint_connection_loss")
        ctx.save_for_backward(grad)
        ctx.shape = curve_points.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        from .losses import scale_by_upstream as sc
        return sc(grad, g).view(ctx.shape), None, None


def connection_loss(gaussians, weight=0.1, dis_thr=0.05):
    """lambda_points_conn * mean distance between end points of different curves closer than `dis_thr` (train.py:133-146)."""
    return _ConnectionLoss.apply(gaussians._curve_points, float(weight), float(dis_thr))
