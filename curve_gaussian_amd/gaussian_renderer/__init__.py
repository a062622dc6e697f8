"""Drop-in for the reference's ``gaussian_renderer.render`` (/root/reference/gaussian_renderer/__init__.py:18-157)."""
import math

import torch

from ..diff_cur_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from ..ops.curve_sampling import splat_attributes


class PipelineParams:
    """Defaults of arguments/__init__.py:68-75."""
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False
    antialiasing = False
    render_geo = True


_ones_cache = {}


def _ones(n, dev):
    """[n,1] unit colours (never written, no grad): one allocation per (n, device) instead of a fill per view."""
    key = (n, str(dev))
    t = _ones_cache.get(key)
    if t is None:
        _ones_cache.clear()
        t = _ones_cache[key] = torch.ones(n, 1, device=dev)
    return t


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, separate_sh=False, override_color=None,
           use_trained_exp=False, use_mask=False, mask_thr=0.01, compute_visibility=True, clamp=True,
           compute_rend_dir=True, static_bucket_cap=0, status_sink=None):
    """Render the scene.  Background tensor (bg_color) must be on the GPU.  Returns the reference's dict
    {render, viewspace_points, visibility_filter, radii, depth, rend_dir, rend_alpha} (:147-155)."""
    dev = pc.get_xyz.device
    # (:30-34) a zero tensor whose .grad receives the screen-space gradients; a leaf needs no retain_grad()
    screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True, device=dev)
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug,
        antialiasing=pipe.antialiasing, render_geo=pipe.render_geo,
        # extensions: sync-free forward for stream-ordered / graph-captured steps (train_step.GraphedTrainStep)
        static_bucket_cap=static_bucket_cap, status_sink=status_sink)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    means3D = pc.get_xyz
    means2D = screenspace_points
    # :57-76, :96-104 fused into one HIP kernel each way (rotation normalisation, opacity sigmoid + expansion,
    # straight-through mask on scales/opacity, camera-facing main axis -> view-space direction map):
    rotations, opacity, scales, input_all_map = splat_attributes(
        pc._rotation, pc._xyz, pc._opacity, pc.get_scaling, viewpoint_camera.camera_center,
        viewpoint_camera.world_view_transform, pc.n_gaussians, pc._mask if use_mask else None, mask_thr)
    # SH path is dead in the reference (:96-97): single-channel unit colour
    colors_precomp = _ones(means3D.shape[0], dev)
    rendered_image, radii, depth_image, out_all_map = rasterizer(
        means3D=means3D, means2D=means2D, shs=None, colors_precomp=colors_precomp, opacities=opacity, scales=scales,
        rotations=rotations, all_map=input_all_map, cov3D_precomp=None)
    # clamp=False (extension): hand the raw composite to ops.losses.photometric_loss(clamp=True), which applies the
    # clamp and its gradient mask inside the loss kernels
    if clamp:
        rendered_image = rendered_image.clamp(0, 1)
    rendered_alpha = out_all_map[3:4, ]
    rendered_dir = out_all_map[0:3]
    # view space -> world space (:143-145).  The reference does this with a [H*W,3] x [3,3] matmul; the same
    # contraction as three broadcast FMAs avoids a 150 us GEMM launch on a 1600^2 image.  compute_rend_dir=False
    # (extension) leaves the view-space map in place for callers that do not use it (the photometric train step).
    if compute_rend_dir:
        wv = viewpoint_camera.world_view_transform[:3, :3]
        rendered_dir = (rendered_dir[0:1] * wv[:, 0].view(3, 1, 1) + rendered_dir[1:2] * wv[:, 1].view(3, 1, 1) +
                        rendered_dir[2:3] * wv[:, 2].view(3, 1, 1))
    else:
        rendered_dir = None
    # compute_visibility=False (extension) skips the nonzero(), which is a host sync (train.py only needs it for the
    # densification statistics and the opacity regulariser)
    return {"render": rendered_image, "viewspace_points": screenspace_points,
            "visibility_filter": (radii > 0).nonzero() if compute_visibility else None, "radii": radii,
            "depth": depth_image,
            "rend_dir": rendered_dir, "rend_alpha": rendered_alpha}
