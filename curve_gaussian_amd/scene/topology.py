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
``fit_curve_to_line`` :597-621 / ``merge_curves`` :459-595 (host-side numpy there and here; skimage's RANSAC and scipy's curve_fit
restated, see below) -- what ever sets ``is_bezier = False`` during training (train.py:209-211).

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


# ------------------------------------------------------------------------------------------------ line fitting / merging
# train.py:209-211 (every merge interval): fit_curve_to_line turns Bezier curves that are straight into line segments --
# what makes the line branch of prepare_scaling_rot live -- and merge_curves fuses end-to-end neighbours.  Host-side numpy in
# the reference (scene/gaussian_curve_model.py:459-632 over edge_extraction/fitting.py, merging.py), host-side numpy here.
# Two third-party pieces of the reference are absent from this image and restated: skimage's ransac(LineModelND) (the reference
# calls it unseeded: its result is random; here a SEEDED sampler with the same model, residual and trial count) and
# scipy.optimize.curve_fit on a model that is linear in its 12 parameters (= its linear least-squares solution).
def get_curve_gaussians(g, t):
    """:70-79 -- curve points at parameters t ([n,1,1]) -> [n,B,3]; straight segments on their chord."""
    cp = g._curve_points
    bez = (1 - t) ** 3 * cp[:, 0, :] + 3 * (1 - t) ** 2 * t * cp[:, 1, :] + 3 * (1 - t) * t ** 2 * cp[:, 2, :] + t ** 3 * cp[:, 3, :]
    if bool(g.is_bezier.all()):
        return bez
    line = (1 - t) * cp[:, 0, :] + t * cp[:, 3, :]
    return torch.where(g.is_bezier.unsqueeze(0).unsqueeze(2), bez, line)


def fit_straight_line(points):
    """edge_extraction/fitting.py:74-97 -- principal axis of the points and the extent of their projections on it."""
    import numpy as np
    mean_point = np.mean(points, axis=0)
    centered = points - mean_point
    cov = np.dot(centered.T, centered) / len(points)
    eigenvalues, eigenvectors = np.linalg.eigh(cov)
    direction = eigenvectors[:, np.argmax(eigenvalues)]
    direction = direction / np.linalg.norm(direction)
    projections = np.dot(points - mean_point, direction)
    t_min, t_max = np.min(projections), np.max(projections)
    return mean_point + t_min * direction, mean_point + t_max * direction, direction, mean_point, t_min, t_max


def is_curve_straight(g, sample_points, threshold=0.002, threshold_max=0.004):
    """:624-631 -- mean and maximum distance of the samples to their fitted segment below the two thresholds."""
    import numpy as np
    pts = sample_points.detach().cpu().numpy() if torch.is_tensor(sample_points) else np.asarray(sample_points)
    start, end, direction, mean_point, t_min, t_max = fit_straight_line(pts)
    t = np.dot(pts - mean_point, direction)
    closest = mean_point + np.clip(t, t_min, t_max).reshape(-1, 1) * direction
    d = np.linalg.norm(pts - closest, axis=1)
    return bool((np.mean(d) < threshold) & (d.max() < threshold_max)), start, end


def fit_curve_to_line(g, threshold=0.002, threshold_max=0.004, sample_num=100):
    """:597-621 -- Bezier curves whose 100 samples lie on a segment become straight segments (is_bezier = False).  Like the
    reference the control points themselves are NOT moved (its `new_curve_points[selected_mask][:, 0] = ...` assigns into a
    copy made by the boolean index): the segment is the chord P0-P3 from then on, and the curve-point group's Adam moments
    restart from zero (replace_tensor_to_optimizer)."""
    dev = g._curve_points.device
    t = torch.linspace(0, 1, sample_num, device=dev)[:, None, None]
    with torch.no_grad():
        samples = get_curve_gaussians(g, t).permute(1, 0, 2).contiguous().cpu()     # 'm b c -> b m c'
        is_bez = g.is_bezier.cpu()
    selected = torch.zeros(samples.shape[0], dtype=torch.bool)
    for i in range(samples.shape[0]):
        if not bool(is_bez[i]):
            continue
        ok, _start, _end = is_curve_straight(g, samples[i], threshold, threshold_max)
        selected[i] = ok
    if bool(selected.any()):
        new_is_bezier = g.is_bezier.clone()
        new_is_bezier[selected.to(dev)] = False
        g.is_bezier = new_is_bezier
        _, allp = replace_tensor_to_optimizer(g, g._curve_points.clone().detach(), "curve_points")
        _install(g, allp)
        g.prepare_scaling_rot()      # (the reference leaves that to the next statement, merge_curves / train.py:242-243)
    _check_ranks(g)
    return int(selected.sum())


def _ransac_line(pts, residual_threshold, max_trials, rng):
    """skimage.measure.ransac(pts, LineModelND, min_samples=2, residual_threshold, max_trials): the inlier mask of the best
    two-point line (most inliers, then smallest residual sum), restated; `rng` seeded by the caller."""
    import numpy as np
    n = len(pts)
    best, best_count, best_res = None, 0, np.inf
    for _ in range(max_trials):
        i, j = rng.choice(n, 2, replace=False)
        d = pts[j] - pts[i]
        nd = np.linalg.norm(d)
        if nd == 0:
            continue
        d = d / nd
        r = pts - pts[i]
        res = np.linalg.norm(r - np.outer(r @ d, d), axis=1)
        inl = res < residual_threshold
        cnt, rs = int(inl.sum()), float((res ** 2).sum())
        if cnt > best_count or (cnt == best_count and rs < best_res):
            best, best_count, best_res = inl, cnt, rs
        if best_count == n:
            break
    if best is None or best_count < 2:
        raise ValueError("ransac: no line found")
    return best


def _line_fitting(endpoints):
    """edge_extraction/fitting.py:27-50 -- SVD line through the points -> [start(3), end(3)]."""
    import numpy as np
    center = np.mean(endpoints, axis=0)
    c = endpoints - center
    _u, _s, vh = np.linalg.svd(c, full_matrices=False)
    main = vh[0] / np.linalg.norm(vh[0])
    proj = c @ main
    out = np.zeros(6)
    out[:3] = center + main * proj.min()
    out[3:] = center + main * proj.max()
    return out


def _pairwise_segment_distances(seg):
    """edge_extraction/merging.py:63-108 -- [n,6] segments -> symmetric [n,n]: for a < b the smaller of the distances of b's two end
    points to segment a (point-to-segment, the foot clipped to the segment)."""
    import numpy as np
    n = len(seg)
    dmat = np.zeros((n, n))

    def seg_point(s6, q):   # :63-82
        p1, p2 = s6[:3], s6[3:]
        d = p2 - p1
        u = np.clip(np.dot(q - p1, d) / np.dot(d, d), 0, 1)
        return np.linalg.norm(p1 + u * d - q)
    for a in range(n):      # :84-106 (upper triangle, mirrored)
        for b in range(a + 1, n):
            dmat[a, b] = min(seg_point(seg[a], seg[b][:3]), seg_point(seg[a], seg[b][3:]))
    return dmat + dmat.T


def _pairwise_cosine_similarity(seg):
    """edge_extraction/merging.py:58-61 (sklearn's cosine_similarity of the direction vectors, signed)."""
    import numpy as np
    dv = seg[:, 3:] - seg[:, :3]
    dvn = dv / np.maximum(np.linalg.norm(dv, axis=1, keepdims=True), 1e-30)
    return dvn @ dvn.T




def _bezier_fit(xyz, error_threshold=0.02):
    """edge_extraction/fitting.py:52-72 -- cubic Bezier through n ordered points at t = linspace(0, 1, n); the model is linear in
    its control points, so curve_fit's minimum is the least-squares solution.  -> 12 control-point coordinates, or None when the
    RMSE exceeds error_threshold."""
    import numpy as np
    n = len(xyz)
    t = np.linspace(0, 1, n)
    T = np.stack([t ** 3, t ** 2, t, np.ones(n)], axis=1)
    Wm = np.array([[-1, 3, -3, 1], [3, -6, 3, 0], [-3, 3, 0, 0], [1, 0, 0, 0]], float)
    A = T @ Wm                                   # [n,4]: Bernstein weights of the four control points
    P, *_ = np.linalg.lstsq(A, xyz.astype(float), rcond=None)
    rmse = np.sqrt(np.mean(np.sum((xyz - A @ P) ** 2, axis=1)))
    return None if rmse > error_threshold else P.reshape(-1)


def merge_curves(g, distance_threshold=0.02, similarity_threshold=0.97, sample_num=100, ransac_thresh=0.005, seed=0):
    """:459-595 -- (1) Bezier curves whose end points lie within 2 * distance_threshold and whose end tangents are parallel
    (|cos| > similarity_threshold) are paired greedily (most parallel partner first), their 200 samples ordered along the RANSAC
    line of the pair and refitted by ONE cubic Bezier if its RMSE stays below distance_threshold; (2) straight segments that are
    close and parallel are merged per connected component into the segment spanning their samples.  Merged curves are pruned, the
    new ones appended with the mean opacity / width of their sources.  -> number of curves removed."""
    import numpy as np
    dev = g._curve_points.device
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        t = torch.linspace(0, 1, sample_num, device=dev)[:, None, None]
        samples = get_curve_gaussians(g, t).permute(1, 0, 2).contiguous()            # [B,n,3]
        cp = g._curve_points.detach()
        B = cp.shape[0]
        is_bez = g.is_bezier.cpu().numpy().astype(bool)
        all_points = torch.cat([cp[:, 0], cp[:, -1]], dim=0)
        all_tangs = torch.cat([cp[:, 1] - cp[:, 0], cp[:, 2] - cp[:, -1]], dim=0)
        all_tangs = all_tangs / (torch.norm(all_tangs, dim=-1, keepdim=True) + 1e-6)
        similarity = torch.abs(all_tangs @ all_tangs.T)
        dist = torch.cdist(all_points, all_points, p=2)
        mm = (dist < 2 * distance_threshold) & (similarity > similarity_threshold)
        adjacency = (mm[:B, :B] | mm[:B, B:] | mm[B:, :B] | mm[B:, B:]).cpu().numpy()
        confidence = torch.max(torch.max(similarity[:B, :B], similarity[:B, B:]),
                               torch.max(similarity[B:, :B], similarity[B:, B:])).cpu().numpy()
        samples_h = samples.cpu().numpy()
        merge_mask = np.zeros(B, bool)
        new = dict(cp=[], op=[], w=[], bez=[])
        merged, pairs = set(), []
        for i in range(B):
            if i in merged or not is_bez[i]:
                continue
            nb = [j for j in np.nonzero(adjacency[i])[0].tolist() if j not in merged and j != i and is_bez[j]]
            if not nb:
                continue
            best_j = max(nb, key=lambda j: confidence[i, j])
            merged.update((i, best_j))
            pairs.append([i, best_j])
        for comp in pairs:
            pts = np.concatenate([samples_h[i] for i in comp], axis=0)
            try:
                inl = _ransac_line(pts, ransac_thresh, 1000, rng)
                line_eps = _line_fitting(pts[inl])
            except Exception:   # noqa: BLE001 -- like the reference: a pair without a line is left alone
                continue
            main = line_eps[3:] - line_eps[:3]
            main = main / np.linalg.norm(main)
            mean_pt = (line_eps[3:] + line_eps[:3]) / 2
            pts = pts[np.argsort((pts - mean_pt) @ main)]
            out = _bezier_fit(pts, error_threshold=distance_threshold)
            if out is not None:
                merge_mask[comp] = True
                new["cp"].append(torch.from_numpy(out.reshape(4, 3)).float())
                new["op"].append(g._opacity.detach()[comp].mean(dim=0, keepdim=True))
                new["w"].append(g._width.detach()[comp].mean(dim=0, keepdim=True))
                new["bez"].append(True)
        line_idx = np.nonzero(~is_bez)[0]
        if len(line_idx) > 0:
            from scipy.sparse.csgraph import connected_components
            seg = cp.cpu().numpy()[line_idx][:, [0, -1], :].reshape(len(line_idx), 6)
            dmat = _pairwise_segment_distances(seg)
            sim = np.abs(_pairwise_cosine_similarity(seg))     # (:557)
            ncomp, labels = connected_components((dmat <= distance_threshold) & (sim >= similarity_threshold))
            for c in range(ncomp):
                members = np.nonzero(labels == c)[0]
                if len(members) == 1:
                    continue
                comp = line_idx[members]
                merge_mask[comp] = True
                start, end, *_ = fit_straight_line(samples_h[comp].reshape(-1, 3))
                out = np.zeros((4, 3), np.float32)
                out[0], out[-1] = start, end
                new["cp"].append(torch.from_numpy(out).float())
                new["op"].append(g._opacity.detach()[comp.tolist()].mean(dim=0, keepdim=True))
                new["w"].append(g._width.detach()[comp.tolist()].mean(dim=0, keepdim=True))
                new["bez"].append(False)
    removed = int(merge_mask.sum())
    if removed:
        k = len(new["cp"])
        fdc, frest, msk = g._features_dc.detach()[0:1], g._features_rest.detach()[0:1], torch.ones_like(g._mask.detach()[0:1])
        prune_curves(g, torch.from_numpy(merge_mask).to(dev))
        densification_postfix(g, torch.stack(new["cp"]).to(dev), fdc.repeat(k, 1, 1, 1), frest.repeat(k, 1, 1, 1),
                              torch.cat(new["op"]).to(dev), torch.cat(new["w"]).to(dev), msk.repeat(k, 1, 1),
                              torch.tensor(new["bez"], dtype=torch.bool, device=dev))
        g.prepare_scaling_rot()
    _check_ranks(g)
    return removed


METHODS = dict(fit_curve_to_line=fit_curve_to_line, is_curve_straight=is_curve_straight, merge_curves=merge_curves,
               get_curve_gaussians=get_curve_gaussians, prune_curves=prune_curves, densification_postfix=densification_postfix,
               densify_and_split_curve=densify_and_split_curve, densify_and_prune=densify_and_prune,
               curve_split_curvature=curve_split_curvature, only_prune=only_prune, reset_opacity=reset_opacity,
               fix_opacity=fix_opacity, mask_trim_split=mask_trim_split, de_casteljau_split=de_casteljau_split,
               de_casteljau_trim=de_casteljau_trim, cat_tensors_to_optimizer=cat_tensors_to_optimizer,
               _prune_optimizer=_prune_optimizer)
