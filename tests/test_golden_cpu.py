"""CPU: the oracle restatements are pinned against golden vectors produced by the reference's own Python
(tests/golden/make_golden.py imports /root/reference in the authoring container)."""
import os

import numpy as np
import torch

from curve_gaussian_amd import synthetic as S
from oracle import raster as ORA
from oracle import torch_ref as TR

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_rot_to_quat_batch_matches_reference():
    d = np.load(os.path.join(G, "rot_to_quat.npz"))
    out = TR.rot_to_quat_batch(torch.tensor(d["mats"]))
    assert torch.equal(out, torch.tensor(d["quats"]))  # same torch ops, same machine: bit-exact


def test_ssim_matches_reference_value_and_grad():
    d = np.load(os.path.join(G, "ssim.npz"))
    for i in range(3):
        img1 = torch.tensor(d[f"img1_{i}"]).requires_grad_(True)
        img2 = torch.tensor(d[f"img2_{i}"])
        val = TR.ssim_map(img1, img2).mean()
        val.backward()
        assert torch.isclose(val.detach(), torch.tensor(d[f"val_{i}"]))  # the reference test's own criterion
        assert torch.isclose(img1.grad, torch.tensor(d[f"grad_{i}"])).all()


def test_edge_aware_loss_matches_reference():
    d = np.load(os.path.join(G, "edge_aware_loss.npz"))
    img = torch.tensor(d["image"]).requires_grad_(True)
    val = TR.edge_aware_loss(img, torch.tensor(d["gt"]))
    val.backward()
    np.testing.assert_allclose(val.item(), d["value"], rtol=1e-6)
    np.testing.assert_allclose(img.grad.numpy(), d["grad"], rtol=1e-5, atol=1e-9)


def test_camera_conventions_match_reference():
    d = np.load(os.path.join(G, "camera.npz"))
    for i in range(4):
        np.testing.assert_array_equal(S.world2view(d["R"][i], d["T"][i]), d["world2view"][i])
    np.testing.assert_array_equal(S.projection_matrix(0.01, 100.0, 0.6911, 0.5).numpy(), d["projection"])
    # make_camera composes them exactly like scene/cameras.py:59-66
    cam = S.make_camera((0.5, -1.6, 0.7), (0.5, 0.5, 0.5), (0, 0, 1), 64, 64)
    wv = cam.world_view_transform
    np.testing.assert_allclose((wv[:3, :3].T @ wv[:3, :3]).numpy(), np.eye(3), atol=1e-6)   # rotation block
    np.testing.assert_allclose(cam.camera_center.numpy(), [0.5, -1.6, 0.7], atol=1e-5)
    centre_view = torch.tensor([0.5, 0.5, 0.5, 1.0]) @ wv
    assert centre_view[2] > 0 and abs(centre_view[0]) < 1e-5 and abs(centre_view[1]) < 1e-5  # +z looks at the target


def test_sh_forward_of_c_oracle_matches_reference_eval_sh():
    """forward.cu:20-75 (single channel) == clamp(eval_sh + 0.5, 0) of utils/sh_utils.py for direction pos - campos."""
    d = np.load(os.path.join(G, "sh.npz"))
    dirs = d["dirs"]
    P = dirs.shape[0]
    campos = np.array([0.3, -0.2, 0.1], np.float32)
    means = (campos[None] + 2.0 * dirs + np.array([0.0, 0.0, 3.0], np.float32)).astype(np.float32)
    cam = S.make_camera((0.3, -0.2, 0.1), (0.3, -0.2, 3.1), (0, 1, 0), 64, 64)
    dnorm = means - campos[None]
    dnorm = dnorm / np.linalg.norm(dnorm, axis=1, keepdims=True)
    for deg in (0, 1, 2, 3):
        sh = d[f"sh_{deg}"][:, 0, :].astype(np.float32)  # [P, M]
        import sys
        sys.path.insert(0, os.path.dirname(G))
        fw = ORA.forward(np.zeros(3, np.float32), means, None, np.full((P, 1), 0.5, np.float32),
                         np.full((P, 3), 0.01, np.float32), np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1)), 1.0,
                         None, np.zeros((P, 4), np.float32), cam.world_view_transform.numpy(),
                         cam.full_proj_transform.numpy(), 0.36, 0.36, 64, 64, sh, deg, campos)
        vis = fw.radii > 0
        assert vis.sum() > 0
        # independent evaluation with the reference's formula on the oracle's own directions
        from oracle.torch_ref import torch as _t  # noqa: F401
        ref = _eval_sh_np(deg, sh, dnorm) + 0.5
        got = np.array([fw._view("ora_rgb", P, np.float32)])[0]
        np.testing.assert_allclose(got[vis], np.maximum(ref, 0)[vis], rtol=1e-5, atol=1e-6)
        fw.free()


def _eval_sh_np(deg, sh, dirs):
    """utils/sh_utils.py eval_sh restated for one channel (checked against tests/golden/sh.npz below)."""
    C0 = 0.28209479177387814
    C1 = 0.4886025119029199
    C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
    C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435]
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    r = C0 * sh[:, 0]
    if deg > 0:
        r = r - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = r + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6] + \
            C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8]
    if deg > 2:
        r = r + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10] + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + \
            C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12] + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + \
            C3[5] * z * (xx - yy) * sh[:, 14] + C3[6] * x * (xx - 3 * yy) * sh[:, 15]
    return r


def test_eval_sh_restatement_matches_reference_golden():
    d = np.load(os.path.join(G, "sh.npz"))
    for deg in (0, 1, 2, 3):
        got = _eval_sh_np(deg, d[f"sh_{deg}"][:, 0, :], d["dirs"])
        np.testing.assert_allclose(got, d[f"out_{deg}"][:, 0], rtol=1e-5, atol=1e-6)


def test_scalar_helpers_golden_shapes():
    d = np.load(os.path.join(G, "scalars.npz"))
    # inverse_sigmoid is the logit used for _opacity init (gaussian_curve_model.py:153-154)
    x = torch.tensor(d["inv_sigmoid_x"])
    np.testing.assert_allclose(torch.log(x / (1 - x)).numpy(), d["inv_sigmoid_y"], rtol=1e-6)
    # curve-point lr schedule (gaussian_curve_model.py:223-244 -> utils/general_utils.py:99-132)
    from curve_gaussian_amd.scene.gaussian_curve_model import get_expon_lr_func
    lr = get_expon_lr_func(lr_init=0.0005, lr_final=0.000005, lr_delay_mult=0.01, max_steps=30000)
    np.testing.assert_allclose([lr(int(s)) for s in d["lr_steps"]], d["lr_values"], rtol=1e-12)
