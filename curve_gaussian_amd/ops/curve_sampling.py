"""Curve -> Gaussian sampling (reference scene/gaussian_curve_model.py:70-89,180-198 + utils/general_utils.py:33-86).

INTERIM: expressed with PyTorch device ops exactly like the reference (runs wherever the parameters live); the fused
HIP kernel replaces it (see csrc/sampling.hip once landed).
"""
import torch
import torch.nn.functional as F


def _sqrt_positive_part(x):
    ret = torch.zeros_like(x)
    positive_mask = x > 0
    ret[positive_mask] = torch.sqrt(x[positive_mask])
    return ret


def rot_to_quat_batch(rot):
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
    return torch.where(out[..., 0:1] < 0, -out, out)


def sample_curves(cp, width, is_bezier, m=12, eps=1e-8):
    dev, dt = cp.device, cp.dtype
    B = cp.shape[0]
    t = torch.linspace(0.5 / m, 1 - 0.5 / m, m, device=dev, dtype=dt)[:, None, None]
    allb = bool(is_bezier.all())

    def pts(tt):
        bez = (1 - tt) ** 3 * cp[:, 0, :] + 3 * (1 - tt) ** 2 * tt * cp[:, 1, :] + 3 * (1 - tt) * tt ** 2 * cp[:, 2, :] + tt ** 3 * cp[:, 3, :]
        if allb:
            return bez
        line = (1 - tt) * cp[:, 0, :] + tt * cp[:, 3, :]
        return torch.where(is_bezier.unsqueeze(0).unsqueeze(2), bez, line)

    xyz_mb = pts(t)
    dist = torch.norm(xyz_mb - pts(t - 0.5 / m), dim=-1)
    tan = 3 * (1 - t) ** 2 * (cp[:, 1, :] - cp[:, 0, :]) + 6 * (1 - t) * t * (cp[:, 2, :] - cp[:, 1, :]) + 3 * t ** 2 * (cp[:, 3, :] - cp[:, 2, :])
    if not allb:
        tan = torch.where(is_bezier.unsqueeze(0).unsqueeze(2), tan, (cp[:, 3, :] - cp[:, 0, :]).unsqueeze(0).expand_as(tan))
    xyz = xyz_mb.permute(1, 0, 2).reshape(B * m, 3)
    tan = tan.permute(1, 0, 2).reshape(B * m, 3)
    v0 = tan / (torch.linalg.vector_norm(tan, dim=-1, keepdim=True) + eps)
    up = torch.tensor([[0.0, 0.0, 1.0]], device=dev, dtype=dt).expand_as(tan)
    v1 = torch.linalg.cross(tan, up, dim=-1)
    v1 = v1 / torch.norm(v1)
    v2 = torch.linalg.cross(tan, v1, dim=-1)
    v2 = v2 / torch.norm(v2)
    rot = rot_to_quat_batch(torch.stack((v0, v1, v2), dim=1).transpose(-2, -1))
    s0 = dist.permute(1, 0).reshape(B * m)
    s1 = torch.exp(width).repeat(1, m).reshape(B * m)
    return xyz, rot, torch.stack((s0, s1, s1), dim=1)


def quaternion_to_matrix(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))
