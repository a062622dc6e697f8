"""Curve-topology edits of the reference's ``GaussianCurveModel`` kept on the GPU:
densification statistics -> split (de Casteljau) / prune, opacity reset, trim, with the optimizer-state surgery they
need.  They run every 500-2000 iterations (train.py:150-216), resize every per-curve tensor and are plain tensor
programs in the reference too -- no kernels here, but they must cooperate with the hot path's flat parameter /
gradient / Adam-state buffers (ops/optim.FlatAdam, view_parallel.FlatGrads) as well as with torch.optim.Adam.

Reference: /root/reference/scene/gaussian_curve_model.py
  _prune_optimizer :246-262 (gaussian_model.py:475-492)   cat_tensors_to_optimizer gaussian_model.py:513-533
  replace_tensor_to_optimizer gaussian_model.py:460-473   reset_opacity :264-268   fix_opacity :270-279
  prune_curves :283-304   densification_postfix :306-326   densify_and_split_curve :330-349
  densify_and_prune :351-365   de_casteljau_trim :368-371   curve_split_curvature :373-390
  de_casteljau_split :392-425   only_prune :428-435   mask_trim_split :437-463
``merge_curves`` (:466ff, RANSAC line fitting on the host) is not reproduced.

The functions take the model as first argument and are installed as methods of
``curve_gaussian_amd.scene.GaussianCurveModel`` under the reference's names.  Parity: the reference's scene package
cannot be imported here (open3d / pytorch3d / simple_knn missing); the checker is oracle/topology_ref.py, a
statement-by-statement CPU restatement of the cited lines over a real torch.optim.Adam -- parameters, statistics
buffers, derived splat tensors and the Adam moments of every group are compared after each edit for both optimizer back
ends (tests/test_topology_oracle_gpu.py), next to the property tests of tests/test_topology_gpu.py."""
import torch
from torch import nn

GROUPS = ("curve_points", "f_dc", "f_rest", "opacity", "width", "mask")
ATTR = {"curve_points": "_curve_points", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
        "width": "_width", "mask": "_mask"}


# ------------------------------------------------------------------------------------------------ optimizer surgery
def _is_flat(opt):
    from ..ops.optim import FlatAdam
    return isinstance(opt, FlatAdam)


def _install(g, tensors):
    for name, t in tensors.items():
        setattr(g, ATTR[name], t)
    for cb in getattr(g, "_topology_listeners", []):
        cb()


def _rebuild(g, new_params, new_state):
    """Install new per-curve parameter values (dict name -> tensor) with their Adam moments (dict name ->
    (exp_avg, exp_avg_sq) or None = zeros) in whatever optimizer the model uses; returns name -> nn.Parameter."""
    opt = g.optimizer
    if opt is None:
        return {n: nn.Parameter(t.detach().clone().requires_grad_(True)) for n, t in new_params.items()}
    if _is_flat(opt):
        return opt.rebuild(new_params, new_state)
    out = {}
    for group in opt.param_groups:                       # torch.optim.Adam: the reference's own bookkeeping
        name = group["name"]
        if name not in new_params:
            continue
        old = group["params"][0]
        stored = opt.state.get(old, None)
        p = nn.Parameter(new_params[name].detach().clone().requires_grad_(True))
        if stored is not None:
            st = new_state.get(name)
            stored["exp_avg"] = torch.zeros_like(p) if st is None else st[0]
            stored["exp_avg_sq"] = torch.zeros_like(p) if st is None else st[1]
            del opt.state[old]
            opt.state[p] = stored
        group["params"][0] = p
        out[name] = p
    return out


def _state_of(g, name):
    """(exp_avg, exp_avg_sq) of a group, or None when the optimizer has not stepped yet."""
    opt = g.optimizer
    if opt is None:
        return None
    if _is_flat(opt):
        return opt.state_of(name)
    for group in opt.param_groups:
        if group["name"] == name:
            st = opt.state.get(group["params"][0], None)
            return None if st is None else (st["exp_avg"], st["exp_avg_sq"])
    return None


def _prune_optimizer(g, mask):
    params, state = {}, {}
    for name in GROUPS:
        params[name] = getattr(g, ATTR[name]).detach()[mask]
        st = _state_of(g, name)
        state[name] = None if st is None else (st[0][mask], st[1][mask])
    return _rebuild(g, params, state)


def cat_tensors_to_optimizer(g, tensors_dict):
    params, state = {}, {}
    for name in GROUPS:
        ext = tensors_dict[name].detach()
        params[name] = torch.cat((getattr(g, ATTR[name]).detach(), ext), dim=0)
        st = _state_of(g, name)
        state[name] = None if st is None else (torch.cat((st[0], torch.zeros_like(ext)), dim=0),
                                                torch.cat((st[1], torch.zeros_like(ext)), dim=0))
    return _rebuild(g, params, state)


def replace_tensor_to_optimizer(g, tensor, name):
    """New values for one group, its Adam moments reset to zero (gaussian_model.py:460-473)."""
    params = {n: getattr(g, ATTR[n]).detach() for n in GROUPS}
    state = {n: _state_of(g, n) for n in GROUPS}
    params[name] = tensor.detach()
    state[name] = (torch.zeros_like(tensor), torch.zeros_like(tensor)) if state[name] is not None else None
    out = _rebuild(g, params, state)
    return {name: out[name]}, out


# ------------------------------------------------------------------------------------------------ geometry
def _lerp(a, b, t):
    return a + t * (b - a)


def de_casteljau_split(g, curves, t, is_bezier):
    """:392-425 -- split every curve at its own parameter t ([n] or [n,1]) into (left, right) control polygons.
    Bezier curves: the three levels of de Casteljau interpolation; straight segments (is_bezier False) are cut on
    their chord P0-P3 and get evenly spaced inner control points, as the reference does."""
    t = t.reshape(-1, 1)
    p0, p1, p2, p3 = curves.unbind(dim=1)
    a0, a1, a2 = _lerp(p0, p1, t), _lerp(p1, p2, t), _lerp(p2, p3, t)     # level 1
    b0, b1 = _lerp(a0, a1, t), _lerp(a1, a2, t)                           # level 2
    split = _lerp(b0, b1, t)                                              # the point B(t)
    left = torch.stack([p0, a0, b0, split], dim=1)
    right = torch.stack([split, b1, a2, p3], dim=1)
    if bool(g.is_bezier.all()):
        return left, right
    cut = _lerp(p0, p3, t)
    third = 1.0 / 3.0
    left_line = torch.stack([p0, _lerp(p0, cut, third), _lerp(p0, cut, 2 * third), cut], dim=1)
    right_line = torch.stack([cut, _lerp(cut, p3, third), _lerp(cut, p3, 2 * third), p3], dim=1)
    sel = is_bezier[:, None, None]
    return torch.where(sel, left, left_line), torch.where(sel, right, right_line)


def de_casteljau_trim(g, curves, from_t, end_t, is_bezier):
    """:368-371 (the second split is applied to the right part at end_t as written in the reference)."""
    _, right_curves = de_casteljau_split(g, curves, from_t, is_bezier)
    left_curves, _ = de_casteljau_split(g, right_curves, end_t, is_bezier)
    return left_curves


# ------------------------------------------------------------------------------------------------ topology edits
def _sample_t(g):
    m = g.n_gaussians
    return torch.linspace(0.5 / m, 1 - 0.5 / m, m, device=g._curve_points.device)[:, None, None]


def _stats_buffers(g):
    P = g._curve_points.shape[0] * g.n_gaussians
    dev = g._curve_points.device
    g.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
    g.denom = torch.zeros((P, 1), device=dev)
    g.max_radii2D = torch.zeros((P,), device=dev)


def _check_ranks(g):
    """View-parallel runs: every rank must come out of a topology edit with the same number of curves (one small
    all-reduce per public edit; no-op on a single rank)."""
    from ..view_parallel import assert_same_topology
    assert_same_topology(int(g._curve_points.shape[0]), g._curve_points.device)


def prune_curves(g, mask):
    """:283-304 -- remove the curves where mask is True."""
    valid = ~mask
    opt_t = _prune_optimizer(g, valid)
    m = g.n_gaussians
    valid_points = valid.unsqueeze(1).repeat(1, m).flatten()
    for name in ("xyz_gradient_accum", "denom", "max_radii2D", "tmp_radii"):
        v = getattr(g, name, None)
        if v is not None and v.shape[0] == valid_points.shape[0]:
            setattr(g, name, v[valid_points])
    g.is_bezier = g.is_bezier[valid]
    _install(g, opt_t)
    g.prepare_scaling_rot()


def densification_postfix(g, new_curve_points, new_features_dc, new_features_rest, new_opacities, new_widths, new_masks,
                          new_is_bezier):
    """:306-326 -- append curves; statistics buffers restart from zero."""
    d = {"curve_points": new_curve_points, "f_dc": new_features_dc, "f_rest": new_features_rest,
         "opacity": new_opacities, "width": new_widths, "mask": new_masks}
    opt_t = cat_tensors_to_optimizer(g, d)
    g.is_bezier = torch.cat((g.is_bezier, new_is_bezier))
    _install(g, opt_t)
    _stats_buffers(g)


def densify_and_split_curve(g, selected_pts_mask, t, N=2):
    """:330-349 -- replace every selected curve by its two de Casteljau halves at t."""
    k = int(selected_pts_mask.sum())
    cp = g.get_curve_points.detach()
    new_curve_points = cp[selected_pts_mask].repeat(N, 1, 1)
    new_features_dc = g._features_dc.detach()[selected_pts_mask].repeat(N, 1, 1, 1)
    new_features_rest = g._features_rest.detach()[selected_pts_mask].repeat(N, 1, 1, 1)
    new_opacities = g._opacity.detach()[selected_pts_mask].repeat(N, 1)
    new_widths = g._width.detach()[selected_pts_mask].repeat(N, 1)
    new_masks = g._mask.detach()[selected_pts_mask].repeat(N, 1, 1)
    new_is_bezier = g.is_bezier[selected_pts_mask].repeat(N)
    left, right = de_casteljau_split(g, cp[selected_pts_mask], t, g.is_bezier[selected_pts_mask])
    new_curve_points[0:k, ...] = left
    new_curve_points[k:, ...] = right
    densification_postfix(g, new_curve_points, new_features_dc, new_features_rest, new_opacities, new_widths, new_masks,
                          new_is_bezier)
    prune_filter = torch.cat((selected_pts_mask, torch.zeros(N * k, device=cp.device, dtype=torch.bool)))
    prune_curves(g, prune_filter)


def densify_and_prune(g, max_grad, min_opacity, extent=None, max_screen_size=None, radii=None):
    """:351-365 -- split the curves whose largest per-splat mean screen-space gradient reaches max_grad at the sample
    where it is largest, then prune curves below min_opacity."""
    from ..view_parallel import global_densification_stats
    accum, denom = global_densification_stats(g)   # view-parallel runs: the sums over all ranks' views (copies)
    grads = accum / denom
    grads[grads.isnan()] = 0.0
    g.tmp_radii = radii
    m = g.n_gaussians
    grads = grads.reshape(-1, m, grads.shape[-1])
    max_values, max_indices = torch.max(torch.norm(grads, dim=-1), dim=1)
    selected = max_values >= max_grad
    if int(selected.sum()) > 0:
        t = _sample_t(g)[max_indices[selected]]
        densify_and_split_curve(g, selected, t.squeeze(-1))
    prune_mask = (g.get_curve_opacity < min_opacity).squeeze(-1)
    prune_curves(g, prune_mask)
    _check_ranks(g)


def curve_split_curvature(g, threshold_angle=20, threshold_radian_skip=30):
    """:373-390 -- split curves that bend by more than threshold_angle between neighbouring samples (or
    threshold_radian_skip between samples two apart) right after the sharpest bend."""
    m = g.n_gaussians
    th = torch.tensor(threshold_angle * (torch.pi / 180))
    th_skip = torch.tensor(threshold_radian_skip * (torch.pi / 180))
    axis = g.get_rotation_matrix[..., 0].detach().reshape(-1, m, 3)
    cos_theta = torch.einsum('bij,bij->bi', axis[:, :-1, :], axis[:, 1:, :])
    angles = torch.acos(cos_theta.clamp(-1, 1))
    cos_skip = torch.einsum('bij,bij->bi', axis[:, :-2, :], axis[:, 2:, :])
    angles_skip = torch.acos(cos_skip.clamp(-1, 1))
    mask_split = torch.max(angles, dim=-1).values > th
    mask_split |= torch.max(angles_skip, dim=-1).values > th_skip
    _, t = torch.max(angles, dim=-1)
    end_t = _sample_t(g)[t] + 0.5 / m
    if int(mask_split.sum()) > 0:
        densify_and_split_curve(g, mask_split, end_t[mask_split].squeeze(-1))
    g.prepare_scaling_rot()
    _check_ranks(g)


def only_prune(g, min_opacity, mask_threshold):
    """:428-435"""
    m = g.n_gaussians
    prune_mask = torch.logical_or((torch.sigmoid(g._mask.detach()) <= mask_threshold).all(dim=1).squeeze(-1),
                                  (g.get_curve_opacity.detach() < min_opacity).squeeze(-1))
    small = g._scaling[:, 0].clone().detach().reshape(-1, m).sum(-1) < 1e-2
    prune_curves(g, torch.logical_or(small, prune_mask))
    _check_ranks(g)


def reset_opacity(g):
    """:264-268 -- clamp every curve's opacity to at most 0.1; the opacity group's Adam moments restart from zero."""
    op = g.get_curve_opacity.detach()
    new = torch.logit(torch.min(op, torch.ones_like(op) * 0.1))   # inverse_sigmoid
    _, allp = replace_tensor_to_optimizer(g, new, "opacity")
    _install(g, allp)


def fix_opacity(g):
    """:270-279 -- lift opacities to at least 0.6 and freeze them."""
    op = g.get_curve_opacity.detach()
    new = torch.logit(torch.max(op, 0.6 * torch.ones_like(op)))
    _, allp = replace_tensor_to_optimizer(g, new, "opacity")
    _install(g, allp)
    g._opacity.requires_grad = False
    for group in g.optimizer.param_groups:
        if group["name"] == "opacity":
            group["lr"] = 0.


def mask_trim_split(g, mask_threshold):
    """:437-463 -- trim both ends of every curve to its first / last valid sample (de Casteljau) and resample the mask
    logits of the kept stretch back to m samples (bilinear, as the reference's F.interpolate call)."""
    m = g.n_gaussians
    st = _sample_t(g)
    valid_mask = (torch.sigmoid(g._mask.detach()) > mask_threshold).squeeze(-1)
    start_idx = torch.argmax(valid_mask.int(), dim=1)
    end_idx = m - 1 - torch.argmax(torch.flip(valid_mask, [1]).int(), dim=1)
    from_t = st[start_idx, :, :].squeeze(-1) - 0.5 / m
    end_t = st[end_idx, :, :].squeeze(-1) + 0.5 / m
    trim_curve_points = de_casteljau_trim(g, g.get_curve_points.detach(), from_t, end_t, g.is_bezier)
    trim_curve_mask = g._mask.clone().detach()
    changed = (start_idx != 0) | (end_idx != m - 1)
    for i in torch.nonzero(changed).squeeze(-1).tolist():
        seg = trim_curve_mask[i][int(start_idx[i]):int(end_idx[i]) + 1]
        inter = torch.nn.functional.interpolate(seg.unsqueeze(0).unsqueeze(0), size=(m, 1), mode='bilinear')
        trim_curve_mask[i] = inter[0, 0]
    _, allp = replace_tensor_to_optimizer(g, trim_curve_mask, "mask")
    _install(g, allp)
    _, allp = replace_tensor_to_optimizer(g, trim_curve_points, "curve_points")
    _install(g, allp)
    g.prepare_scaling_rot()


METHODS = dict(prune_curves=prune_curves, densification_postfix=densification_postfix,
               densify_and_split_curve=densify_and_split_curve, densify_and_prune=densify_and_prune,
               curve_split_curvature=curve_split_curvature, only_prune=only_prune, reset_opacity=reset_opacity,
               fix_opacity=fix_opacity, mask_trim_split=mask_trim_split, de_casteljau_split=de_casteljau_split,
               de_casteljau_trim=de_casteljau_trim, cat_tensors_to_optimizer=cat_tensors_to_optimizer,
               _prune_optimizer=_prune_optimizer)
