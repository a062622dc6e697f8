"""On-disk formats either side of the hot path: EMAP scan reader/writer, parametric_edges.json."""
import json
import os

import numpy as np
import torch

from curve_gaussian_amd import synthetic as S
from curve_gaussian_amd.scene import dataset_io as IO

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_emap_frame_to_camera_matches_reference_golden():
    """camtoworld + intrinsics -> R, T, FoV, world_view / full_proj / centre: the reference's own graphics_utils
    (imported by tests/golden/make_golden.py) composed as dataset_readers.py:303-322 and cameras.py:59-66."""
    d = np.load(os.path.join(GOLD, "emap_camera.npz"))
    H, W = int(d["H"]), int(d["W"])
    for i in range(d["camtoworld"].shape[0]):
        cam = IO.camera_from_emap_frame(i, str(i), d["camtoworld"][i], d["intrinsics"][i], torch.zeros(3, H, W))
        np.testing.assert_allclose([cam.FoVx, cam.FoVy], d["fov"][i], rtol=1e-12)
        np.testing.assert_allclose(cam.world_view_transform.numpy(), d["world_view_transform"][i], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(cam.full_proj_transform.numpy(), d["full_proj_transform"][i], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(cam.camera_center.numpy(), d["camera_center"][i], rtol=1e-5, atol=1e-5)


def test_emap_scan_round_trip(tmp_path):
    """write_emap -> read_emap reproduces the cameras (pose, field of view, size) and the edge maps up to 8-bit
    quantisation; RGBA opening replicates the single-channel map into three equal channels (dataset_readers.py:319)."""
    cams = S.fibonacci_cameras(5, 48, 64)
    g = torch.Generator().manual_seed(0)
    maps = [torch.rand(1, 48, 64, generator=g) for _ in cams]
    IO.write_emap(str(tmp_path), cams, maps)
    meta = json.load(open(tmp_path / "meta_data.json"))
    assert meta["height"] == 48 and meta["width"] == 64 and len(meta["frames"]) == 5
    assert set(meta["frames"][0]) == {"rgb_path", "camtoworld", "intrinsics"}
    back = IO.read_emap(str(tmp_path))
    assert len(back) == 5
    for a, b, m in zip(cams, back, maps):
        assert (b.image_height, b.image_width) == (48, 64)
        np.testing.assert_allclose([b.FoVx, b.FoVy], [a.FoVx, a.FoVy], rtol=1e-9)
        np.testing.assert_allclose(b.world_view_transform.numpy(), a.world_view_transform.numpy(), atol=2e-6)
        np.testing.assert_allclose(b.full_proj_transform.numpy(), a.full_proj_transform.numpy(), atol=2e-5)
        np.testing.assert_allclose(b.camera_center.numpy(), a.camera_center.numpy(), atol=2e-5)
        assert b.original_image.shape == (3, 48, 64)
        assert torch.equal(b.original_image[0], b.original_image[1]) and torch.equal(b.original_image[0], b.original_image[2])
        assert float((b.original_image[:1] - m).abs().max()) <= 0.5 / 255 + 1e-6
    import pytest
    with pytest.raises(ValueError, match="not supported"):
        IO.read_emap(str(tmp_path), detector="Canny")


def test_parametric_edges_writer(tmp_path):
    """parametric_edges.json layout (train.py:266-293 through process_geometry_data): Bezier curves as 4x3 control
    points, line segments as 6 floats (first and last control point); edge points every 5 mm of arc length."""
    class G:
        pass
    g = G()
    cp = torch.tensor([[[0, 0, 0], [0.1, 0, 0], [0.2, 0, 0], [0.3, 0, 0]],          # straight Bezier, length 0.3
                       [[0, 0, 0], [0, 0.5, 0], [0, 0.5, 0], [0, 1.0, 0]],           # becomes a line segment
                       [[0, 0, 0], [0.0, 0.1, 0], [0.1, 0.1, 0], [0.1, 0.0, 0]]], dtype=torch.float32)
    g.get_curve_points = cp
    g.is_bezier = torch.tensor([True, False, True])
    edge_dict, pts = IO.write_parametric_edges(g, str(tmp_path))
    saved = json.load(open(tmp_path / "parametric_edges.json"))
    assert saved == edge_dict
    assert np.array(saved["curves_ctl_pts"]).shape == (2, 4, 3) and np.array(saved["lines_end_pts"]).shape == (1, 6)
    np.testing.assert_allclose(saved["lines_end_pts"][0], [0, 0, 0, 0, 1.0, 0])
    np.testing.assert_allclose(IO.bezier_curve_length(cp[0].numpy()), 0.3, rtol=1e-6)   # float32 control points
    # arc length of the third curve against a fine polyline
    t = np.linspace(0, 1, 20001)[:, None]
    P = cp[2].numpy().astype(np.float64)
    poly = (1 - t) ** 3 * P[0] + 3 * (1 - t) ** 2 * t * P[1] + 3 * (1 - t) * t ** 2 * P[2] + t ** 3 * P[3]
    L = np.linalg.norm(np.diff(poly, axis=0), axis=1).sum()
    np.testing.assert_allclose(IO.bezier_curve_length(P), L, rtol=1e-6)
    n_expected = (int(IO.bezier_curve_length(cp[0].numpy()) // 0.005) + int(IO.bezier_curve_length(P) // 0.005) +
                  int(np.linalg.norm(cp[1, 0].numpy().astype(np.float64) - cp[1, 3].numpy()) // 0.005))
    assert abs(n_expected - (60 + int(L // 0.005) + 200)) <= 2
    assert len(pts) == n_expected
    head = open(tmp_path / "edge_points.ply").read().split("end_header")[0]
    assert "format ascii 1.0" in head and f"element vertex {n_expected}" in head


def test_splat_snapshot_ply_round_trip(tmp_path):
    """save_ply: the reference's point_cloud.ply vertex layout (gaussian_model.py:267-280, 383-400) -- attribute names
    and order, float32 little-endian, opacity stored as its logit -- read back value for value."""
    class G:
        pass
    g = G()
    P = 24
    gen = torch.Generator().manual_seed(3)
    g._xyz = torch.randn(P, 3, generator=gen)
    g._scaling = torch.rand(P, 3, generator=gen) * 0.01
    g._rotation = torch.randn(P, 4, generator=gen)
    g.get_features = torch.randn(P, 4, 1, generator=gen)          # sh degree 1: 1 dc + 3 rest coefficients, one channel
    g.get_opacity = torch.rand(P, 1, generator=gen) * 0.98 + 0.01
    path = tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply"
    names = IO.save_ply(g, str(path))
    assert names == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_rest_0", "f_rest_1", "f_rest_2", "opacity",
                     "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    head = open(path, "rb").read(400).decode("ascii", "ignore")
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 24\nproperty float x\n")
    v = IO.read_ply_vertices(str(path))
    np.testing.assert_array_equal(np.stack([v["x"], v["y"], v["z"]], 1), g._xyz.numpy())
    assert not np.any(v["nx"]) and not np.any(v["nz"])
    np.testing.assert_array_equal(v["f_dc_0"], g.get_features[:, 0, 0].numpy())
    np.testing.assert_array_equal(v["f_rest_2"], g.get_features[:, 3, 0].numpy())
    np.testing.assert_allclose(1.0 / (1.0 + np.exp(-v["opacity"].astype(np.float64))), g.get_opacity[:, 0].numpy(), rtol=1e-5)
    np.testing.assert_array_equal(np.stack([v[f"scale_{i}"] for i in range(3)], 1), g._scaling.numpy())
    np.testing.assert_array_equal(np.stack([v[f"rot_{i}"] for i in range(4)], 1), g._rotation.numpy())


def test_emap_wide_images_keep_their_field_of_view(tmp_path):
    """Images wider than 1600 px are rescaled by loadCam (camera_utils.py:28-42) AFTER readEMAP computed the field of view
    from the original size (dataset_readers.py:320-321): the pixels shrink, the projection does not change."""
    from PIL import Image
    ow, oh = 2000, 1000
    cam = S.make_camera((1.5, 0.2, 0.4), (0.5, 0.5, 0.5), (0, 0, 1), oh, ow, fovx=0.9, fovy=0.5)
    os.makedirs(tmp_path / "edge_DexiNed")
    Image.fromarray((np.random.default_rng(0).random((oh, ow)) * 255).astype(np.uint8)).save(tmp_path / "edge_DexiNed" / "0_colors.png")
    K = np.eye(4)
    K[0, 0], K[1, 1] = IO.fov2focal(cam.FoVx, ow), IO.fov2focal(cam.FoVy, oh)
    K[0, 2], K[1, 2] = ow / 2, oh / 2
    c2w = np.linalg.inv(cam.world_view_transform.numpy().T.astype(np.float64))
    json.dump({"height": oh, "width": ow, "frames": [{"rgb_path": "0_colors.png", "camtoworld": c2w.tolist(),
                                                      "intrinsics": K.tolist()}]}, open(tmp_path / "meta_data.json", "w"))
    back = IO.read_emap(str(tmp_path))[0]
    assert (back.image_width, back.image_height) == (1600, 800)
    np.testing.assert_allclose([back.FoVx, back.FoVy], [IO.focal2fov(K[0, 0], ow), IO.focal2fov(K[1, 1], oh)], rtol=1e-12)
    np.testing.assert_allclose([back.FoVx, back.FoVy], [cam.FoVx, cam.FoVy], rtol=1e-9)
    np.testing.assert_allclose(back.full_proj_transform.numpy(), cam.full_proj_transform.numpy(), atol=2e-5)


def test_grid_point_cloud_is_the_reference_seed_grid():
    """dataset_readers.py:404-412: 15^3 points, np.meshgrid's default 'xy' ordering, colours SH2RGB(U[0,1)/255)."""
    pcd = IO.grid_point_cloud(15, np.random.default_rng(1))
    assert pcd.points.shape == (3375, 3) and pcd.colors.shape == (3375, 3) and not pcd.normals.any()
    x = np.linspace(-0.05, 1.05, 15)
    np.testing.assert_array_equal(pcd.points[0], [x[0], x[0], x[0]])
    np.testing.assert_array_equal(pcd.points[1], [x[0], x[0], x[1]])      # z runs fastest
    np.testing.assert_array_equal(pcd.points[15], [x[1], x[0], x[0]])     # then x ('xy' meshgrid), then y
    np.testing.assert_array_equal(pcd.points[225], [x[0], x[1], x[0]])
    assert 0.5 <= pcd.colors.min() and pcd.colors.max() <= 0.5 + IO.SH_C0 / 255 + 1e-12
    np.testing.assert_allclose(IO.RGB2SH(IO.SH2RGB(np.array([0.1, 0.7]))), [0.1, 0.7])
