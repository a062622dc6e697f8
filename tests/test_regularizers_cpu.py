"""CPU: the torch twins of the training regularisers that serve as test references for the fused HIP ops
(curve_gaussian_amd/ops/regularizers.py) against independent float64 restatements of train.py:110-146."""
import numpy as np
import torch

from curve_gaussian_amd.ops import regularizers as RG


class _G:
    pass


def test_connection_loss_reference_against_a_double_loop():
    """train.py:133-146: mean distance over ordered pairs of end points of DIFFERENT curves closer than 0.05, and its
    gradient (2/count * sum of unit directions; coincident points contribute no gradient)."""
    g = torch.Generator().manual_seed(5)
    B = 60
    cp = torch.rand(B, 4, 3, generator=g, dtype=torch.float64) * 0.2
    cp[3, 0] = cp[8, 3]            # coincident end points of different curves
    cp[5, 3] = cp[5, 0]            # closed curve: its own end points never pair up
    m = _G()
    m._curve_points = cp.clone().requires_grad_(True)
    m.get_curve_points = m._curve_points
    loss = RG.connection_loss_reference(m, weight=0.1, dis_thr=0.05)
    loss.backward()
    pts = np.concatenate([cp[:, 0].numpy(), cp[:, 3].numpy()])
    N = 2 * B
    total, count = 0.0, 0
    grad = np.zeros_like(pts)
    for i in range(N):
        for j in range(N):
            if i % B == j % B:
                continue
            d = np.linalg.norm(pts[i] - pts[j])
            if d < 0.05:
                total += d
                count += 1
                if d > 0:
                    grad[i] += (pts[i] - pts[j]) / d
                    grad[j] -= (pts[i] - pts[j]) / d
    assert count > 0
    np.testing.assert_allclose(float(loss), 0.1 * total / count, rtol=1e-12)
    got = m._curve_points.grad.numpy()
    np.testing.assert_allclose(got[:, 0], 0.1 * grad[:B] / count, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(got[:, 3], 0.1 * grad[B:] / count, rtol=1e-10, atol=1e-14)
    assert not got[:, 1:3].any()
    # no pair within the threshold: the term vanishes (train.py: `if valid_mask.any()`)
    far = _G()
    far._curve_points = (torch.arange(B * 12, dtype=torch.float64).reshape(B, 4, 3)).requires_grad_(True)
    far.get_curve_points = far._curve_points
    z = RG.connection_loss_reference(far, 0.1)
    assert float(z) == 0.0


def test_width_and_mask_twins():
    """train.py:110-111 and :126-131 on small tensors."""
    m = _G()
    m._mask = torch.tensor([[[0.0], [2.0]], [[-1.0], [0.5]]])
    np.testing.assert_allclose(float(RG.mask_loss(m, 0.0005)), 0.0005 * torch.sigmoid(m._mask).mean().item(), rtol=1e-6)
    m._width = torch.log(torch.tensor([[0.004], [0.006], [0.010]]))
    m.get_curve_width = torch.exp(m._width)
    expect = 0.01 * ((0.006 - 0.005) + (0.010 - 0.005)) / 2
    np.testing.assert_allclose(float(RG.width_loss(m, 0.01)), expect, rtol=1e-4)
