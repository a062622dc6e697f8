"""The per-iteration regularisers of the reference's training loop (train.py:110-131), sync-free.

  mask          lambda_mask * mean(sigmoid(_mask))                                  (:110-111, iteration >= densify_until)
  opacity       opacity_loss_weight * mean(log(1 + o^2 / 0.5)) over visible splats  (:114-117, after an opacity reset)
  curve_smo     lambda_curve_smo * mean(1 - |cos(d_i, d_{i+1})|), d = main axis     (:119-124, if any splat is visible)
  width         lambda_width * mean(width - 0.005 over curves with width >= 0.005)  (:126-131)

The reference decides with host-side conditions (``visibility_filter.sum() > 0``, ``mask.any()``: a device-to-host
sync each); here the same values come out of masked means whose denominators are clamped, so the step stays
stream-ordered and capturable.  The O(B^2) end-point connection loss (:133-146, torch.cdist over all end points) is not
reproduced.  These are plain torch ops: they act on per-curve / per-splat tensors once per
iteration, outside the per-view hot path."""
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
