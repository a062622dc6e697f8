"""Independent pure-PyTorch restatements used to CHECK the C oracle and the HIP kernels.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

* ``dense_render``  -- differentiable dense rasterizer (every pixel x every splat), float32 or float64.  Its
  *autograd* gradients are an independent derivation of the hand-written backward of
  /root/reference/submodules/diff-cur-rasterization/cuda_rasterizer/backward.cu (restated in raster_ref.c).
  Semantics follow forward.cu:155-274 (preprocess) and :279-417 (render): near cull z<=0.2, un-normalised
  quaternion, 1.3*tanfov clamp, +0.3 dilation, radius = ceil(3 sqrt(lambda_max)), tile-rect membership,
  power>0 / alpha<1/255 skips, T<1e-4 stop *before* blending, background on colour only.
  Where the reference's backward is NOT the exact derivative (alpha clamp at 0.99, the 1.3*tanfov clamp)
  the tests keep inputs out of those regimes.
* ``prepare_scaling_rot`` / ``get_main_axis`` / ``quaternion_to_matrix`` -- restatement of
  /root/reference/scene/gaussian_curve_model.py:70-105,180-198 (``rot_to_quat_batch`` restated from
  /root/reference/utils/general_utils.py:9-86 and pinned by tests/golden/rot_to_quat.npz).
* ``ssim`` -- restatement of /root/reference/utils/loss_utils.py:56-86 (pinned by tests/golden/ssim.npz).
* ``knn_mean_dist2`` -- brute-force oracle for simple-knn (mean of 3 smallest squared distances).
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- rasterizer (dense)
def _quat_to_R(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    return R


def dense_render(means3D, opacities, scales, rotations, colors, all_map, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                 H, W, bg, scale_modifier=1.0, means2D_ndc_offset=None, antialiasing=False):
    """Returns (color[1,H,W], radii[P], invdepth[1,H,W], out_all_map[4,H,W]).  All float inputs share one dtype."""
    dt = means3D.dtype
    P = means3D.shape[0]
    vm = viewmatrix.to(dt).reshape(16)
    pm = projmatrix.to(dt).reshape(16)
    V = vm.reshape(4, 4).t()      # math matrices
    PM = pm.reshape(4, 4).t()
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    p_view = (ph @ V.t())[:, :3]
    p_hom = ph @ PM.t()
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    p_proj = p_hom[:, :3] * p_w[:, None]
    if means2D_ndc_offset is not None:
        p_proj = torch.cat([p_proj[:, :2] + means2D_ndc_offset, p_proj[:, 2:]], 1)
    fx = W / (2.0 * tan_fovx)
    fy = H / (2.0 * tan_fovy)
    Rq = _quat_to_R(rotations)
    S2 = (scale_modifier * scales) ** 2
    Sigma = Rq @ torch.diag_embed(S2) @ Rq.transpose(1, 2)
    tz = p_view[:, 2]
    limx, limy = 1.3 * tan_fovx, 1.3 * tan_fovy
    tx = torch.clamp(p_view[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(p_view[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    Jm = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], 1).reshape(-1, 2, 3)
    Rwc = V[:3, :3]
    Mt = Jm @ Rwc
    cov = Mt @ Sigma @ Mt.transpose(1, 2)
    a0, b0, c0 = cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]
    a, b, c = a0 + 0.3, b0, c0 + 0.3
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], 1)
    opac = opacities.reshape(-1)
    if antialiasing:
        opac = opac * torch.sqrt(torch.clamp((a0 * c0 - b0 * b0) / det, min=0.000025))
    with torch.no_grad():
        mid = 0.5 * (a + c)
        lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        radius = torch.ceil(3.0 * torch.sqrt(lam))
    px = ((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16
    with torch.no_grad():
        rminx = torch.clamp(((px - radius) / 16).to(torch.int64), 0, gx)   # trunc toward zero == C (int) cast
        rminy = torch.clamp(((py - radius) / 16).to(torch.int64), 0, gy)
        rmaxx = torch.clamp(((px + radius + 15) / 16).to(torch.int64), 0, gx)
        rmaxy = torch.clamp(((py + radius + 15) / 16).to(torch.int64), 0, gy)
        visible = (tz > 0.2) & (det != 0) & ((rmaxx - rminx) * (rmaxy - rminy) > 0)
        radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)
        # stable (depth, idx) order == stable sort on (tile | depth_bits) restricted to any one tile
        order = torch.argsort(tz.to(torch.float32), stable=True)
        order = order[visible[order]]
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pixx = xs.reshape(-1).to(dt)
    pixy = ys.reshape(-1).to(dt)
    tix = (xs.reshape(-1) // 16)
    tiy = (ys.reshape(-1) // 16)
    npix = H * W
    T = torch.ones(npix, dtype=dt)
    done = torch.zeros(npix, dtype=torch.bool)
    C = torch.zeros(npix, dtype=dt)
    D = torch.zeros(npix, dtype=dt)
    A = torch.zeros(4, npix, dtype=dt)
    for s in order.tolist():
        member = (tix >= rminx[s]) & (tix < rmaxx[s]) & (tiy >= rminy[s]) & (tiy < rmaxy[s])
        dx = px[s] - pixx
        dy = py[s] - pixy
        power = -0.5 * (conic[s, 0] * dx * dx + conic[s, 2] * dy * dy) - conic[s, 1] * dx * dy
        alpha = torch.clamp(opac[s] * torch.exp(power), max=0.99)
        with torch.no_grad():
            ok = member & ~done & (power <= 0) & (alpha >= 1.0 / 255.0)
            test_T = T * (1 - alpha)
            stop = ok & (test_T < 0.0001)
            done = done | stop
            ok = ok & ~stop
        alpha = torch.where(ok, alpha, torch.zeros_like(alpha))
        w = alpha * T
        C = C + colors[s, 0] * w
        D = D + (1.0 / tz[s]) * w
        A = A + all_map[s][:, None] * w[None, :]
        T = T * (1 - alpha)
    color = (C + T * bg[0]).reshape(1, H, W)
    return color, radii, D.reshape(1, H, W), A.reshape(4, H, W)


# ----------------------------------------------------------------------------- curve sampling
def standardize_quaternion(q):
    return torch.where(q[..., 0:1] < 0, -q, q)


def _sqrt_positive_part(x):
    ret = torch.zeros_like(x)
    positive_mask = x > 0
    ret[positive_mask] = torch.sqrt(x[positive_mask])
    return ret


def rot_to_quat_batch(rot):
    """utils/general_utils.py:33-86 restated (pytorch3d matrix_to_quaternion variant)."""
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(rot.reshape(-1, 9), dim=-1)
    q_abs = _sqrt_positive_part(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                                             1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1))
    quat_by_rijk = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    flr = torch.tensor(0.1).to(dtype=q_abs.dtype, device=q_abs.device)
    quat_candidates = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    out = quat_candidates[F.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5, :].reshape(-1, 4)
    return standardize_quaternion(out)


def sample_t(m, dtype=torch.float32):
    """gaussian_curve_model.py:58-60"""
    return torch.linspace(0.5 / m, 1 - 0.5 / m, m, dtype=dtype)[:, None, None]


def curve_gaussians(cp, is_bezier, t):
    """gaussian_curve_model.py:70-78"""
    bez = (1 - t) ** 3 * cp[:, 0, :] + 3 * (1 - t) ** 2 * t * cp[:, 1, :] + 3 * (1 - t) * t ** 2 * cp[:, 2, :] + t ** 3 * cp[:, 3, :]
    if bool(is_bezier.all()):
        return bez
    line = (1 - t) * cp[:, 0, :] + t * cp[:, 3, :]
    return torch.where(is_bezier.unsqueeze(0).unsqueeze(2), bez, line)


def curve_tangent(cp, is_bezier, t):
    """gaussian_curve_model.py:80-89"""
    bez = 3 * (1 - t) ** 2 * (cp[:, 1, :] - cp[:, 0, :]) + 6 * (1 - t) * t * (cp[:, 2, :] - cp[:, 1, :]) + 3 * t ** 2 * (cp[:, 3, :] - cp[:, 2, :])
    if bool(is_bezier.all()):
        return bez
    line = (cp[:, 3, :] - cp[:, 0, :]).unsqueeze(0).expand_as(bez)
    return torch.where(is_bezier.unsqueeze(0).unsqueeze(2), bez, line)


def prepare_scaling_rot(curve_points, width, is_bezier, m=12, eps=1e-8):
    """gaussian_curve_model.py:180-198 -> (_xyz [P,3], _rotation [P,4] (un-normalised), _scaling [P,3]);
    splat index = b*m + i ('m b c -> (b m) c')."""
    t = sample_t(m, curve_points.dtype)
    B = curve_points.shape[0]
    xyz_mb = curve_gaussians(curve_points, is_bezier, t)
    front = curve_gaussians(curve_points, is_bezier, t - 0.5 / m)
    dist = torch.norm(xyz_mb - front, dim=-1)                       # [m,B]
    tangent = curve_tangent(curve_points, is_bezier, t)
    xyz = xyz_mb.permute(1, 0, 2).reshape(B * m, 3)
    tangent = tangent.permute(1, 0, 2).reshape(B * m, 3)
    v0 = tangent / (torch.linalg.vector_norm(tangent, dim=-1, keepdim=True) + eps)
    world_up = torch.tensor([[0.0, 0.0, 1.0]], dtype=curve_points.dtype)
    v1 = torch.linalg.cross(tangent, world_up.expand_as(tangent), dim=-1)  # torch.cross w/o dim picks dim -1 unless P == 3
    v1 = v1 / torch.norm(v1)                                        # GLOBAL Frobenius norm (quirk 2)
    v2 = torch.linalg.cross(tangent, v1, dim=-1)
    v2 = v2 / torch.norm(v2)
    rotation = torch.stack((v0, v1, v2), dim=1).transpose(-2, -1)   # columns v0 v1 v2
    rot = rot_to_quat_batch(rotation)
    s0 = dist.permute(1, 0).reshape(B * m)
    s1 = torch.exp(width).repeat(1, m).reshape(B * m)
    scaling = torch.stack((s0, s1, s1), dim=1)
    return xyz, rot, scaling


def quaternion_to_matrix(q):
    """pytorch3d.transforms.quaternion_to_matrix (third-party, version unpinned by the reference's environment.yml):
    two_s = 2 / |q|^2, standard formula."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def get_main_axis(rotation_raw, xyz, camera_center):
    """gaussian_curve_model.py:99-105 with get_rotation = F.normalize (:121-122)."""
    Rm = quaternion_to_matrix(F.normalize(rotation_raw))
    d = Rm[..., 0]
    to_cam = camera_center - xyz
    neg = (d * to_cam).sum(-1) < 0.0
    return torch.where(neg[:, None], -d, d)


def build_all_map(rotation_raw, xyz, camera_center, world_view_transform):
    """gaussian_renderer/__init__.py:98-104"""
    gn = get_main_axis(rotation_raw, xyz, camera_center)
    local = gn @ world_view_transform[:3, :3]
    return torch.cat([local, torch.ones_like(local[:, :1])], dim=1)


# ----------------------------------------------------------------------------- losses / satellites
def _gauss_window(window_size=11, sigma=1.5):
    g = torch.Tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def ssim_map(img1, img2, window_size=11):
    """utils/loss_utils.py:56-86 (per-pixel map; .mean() gives the reference's scalar)."""
    ch = img1.size(-3)
    w1 = _gauss_window(window_size).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    window = w2.expand(ch, 1, window_size, window_size).contiguous().type_as(img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=ch)
    mu2 = F.conv2d(img2, window, padding=pad, groups=ch)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, window, padding=pad, groups=ch) - mu1_sq
    s2 = F.conv2d(img2 * img2, window, padding=pad, groups=ch) - mu2_sq
    s12 = F.conv2d(img1 * img2, window, padding=pad, groups=ch) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))


def edge_aware_loss(image, gt_image, threshold=0.1):
    """utils/loss_utils.py:94-115"""
    edge_map = gt_image.mean(dim=0, keepdim=True)
    num_positive = (torch.sum(edge_map > threshold)).float()
    num_negative = (torch.sum(edge_map <= threshold)).float()
    mask = torch.zeros_like(edge_map)
    mask[edge_map > threshold] = 5. * (num_negative + 1) / (num_positive + num_negative)
    mask[edge_map <= threshold] = 1.0 * (num_positive + 1) / (num_positive + num_negative)
    loss = (image - gt_image) ** 2
    return (loss * mask).mean()


def knn_mean_dist2(points):
    """Brute-force oracle for simple_knn distCUDA2 (simple_knn.cu:148-184): mean of the 3 smallest squared
    distances to OTHER points (index-distinct; duplicates at distance 0 count)."""
    P = points.shape[0]
    d2 = torch.cdist(points.double(), points.double()) ** 2
    d2[torch.arange(P), torch.arange(P)] = float("inf")
    k = min(3, P - 1)
    best, _ = torch.topk(d2, k, dim=1, largest=False)
    if k < 3:
        fmax = torch.full((P, 3 - k), 3.4028234663852886e38, dtype=torch.float64)
        best = torch.cat([best, fmax], 1)
    return (best.sum(1) / 3.0).float()


# ----------------------------------------------------------------------------- model initialisation
def initialize_bezier_curves(points, bound, n_control_points=4):
    """/root/reference/scene/gaussian_curve_model.py:27-51 restated."""
    assert n_control_points == 4
    direction = torch.cat([torch.zeros_like(bound), bound, torch.zeros_like(bound)], dim=1)
    P0 = points - direction
    P3 = points + direction
    P1 = points - 0.5 * direction
    P2 = points + 0.5 * direction
    return torch.stack([P0, P1, P2, P3], dim=1)


def create_from_pcd(points, colors, n_gaussians=12, max_sh_degree=0, init_size=0.5):
    """/root/reference/scene/gaussian_curve_model.py:142-178 restated on the CPU (distCUDA2 -> brute-force 3-NN):
    returns the tensors create_from_pcd installs on the model, plus the derived per-splat tensors."""
    C0 = 0.28209479177387814
    pts = torch.as_tensor(points).float()
    B = pts.shape[0]
    dist2 = torch.clamp_min(knn_mean_dist2(pts), 0.0000001)
    bound = init_size * torch.sqrt(dist2).unsqueeze(1)
    cp = initialize_bezier_curves(pts, bound)
    opac = torch.full((B, 1), 0.6)
    opacities = torch.log(opac / (1 - opac))                         # inverse_sigmoid, utils/general_utils.py:88-89
    widths = torch.log(5e-3 * torch.ones(B, 1))
    pcd_colors = torch.as_tensor(colors).float()[:, None, :].repeat(1, n_gaussians, 1)
    fused_color = (pcd_colors[..., 0:1] - 0.5) / C0                  # RGB2SH
    features = torch.zeros(B, n_gaussians, 1, (max_sh_degree + 1) ** 2)
    features[:, :, :1, 0] = fused_color
    out = dict(curve_points=cp, opacity=opacities, width=widths,
               features_dc=features[:, :, :, 0:1].transpose(2, 3).contiguous(),
               features_rest=features[:, :, :, 1:].transpose(2, 3).contiguous(),
               mask=torch.ones(B, n_gaussians, 1), is_bezier=torch.ones(B, dtype=torch.bool),
               dist=torch.sqrt(dist2).mean())
    out["xyz"], out["rotation"], out["scaling"] = prepare_scaling_rot(cp, widths, out["is_bezier"], n_gaussians)
    return out
