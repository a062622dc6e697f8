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


def _fused_route_ok(pc, pipe, scaling_modifier, override_color):
    """The fused per-view path (ops/view_render.py) computes exactly what the default configuration of the reference's
    render() computes -- scales + rotations from the curve model, unit colours, scale modifier 1, no antialiasing, geometry
    maps on -- and nothing else; the derived splat tensors must be the ones prepare_scaling_rot made from the CURRENT curve
    parameters (the reference renders `pc.get_xyz` as it finds it)."""
    if not (hasattr(pc, "_curve_points") and hasattr(pc, "_width") and hasattr(pc, "is_bezier")):
        return False
    if (pipe.compute_cov3D_python or pipe.convert_SHs_python or pipe.antialiasing or pipe.debug or not pipe.render_geo or
            scaling_modifier != 1.0 or override_color is not None):
        return False
    if pc._curve_points.dim() != 3 or pc._curve_points.shape[0] == 0 or pc._curve_points.shape[0] * pc.n_gaussians >= (1 << 28):
        return False
    return getattr(pc, "_derived_from", None) == _param_stamp(pc)


def _param_stamp(pc):
    """What prepare_scaling_rot derived `pc.get_xyz` ... from: parameter storage + versions, the curve count and the is_bezier
    tensor (a changed curve type re-samples differently); the eps it used travels beside it (`pc._derived_eps`) and is handed
    to the fused route, which re-samples the curves itself."""
    return (pc._curve_points.data_ptr(), pc._curve_points._version, pc._width.data_ptr(), pc._width._version,
            tuple(pc._curve_points.shape), pc.is_bezier.data_ptr(), pc.is_bezier._version)


def _grad_sinks(pc, use_mask):
    """The parameters' `.grad` tensors as sinks of the fused backward, or None when any of them cannot serve (absent, not
    float32, not contiguous, or gradients switched off): the ordinary autograd accumulation then applies."""
    ps = [pc._curve_points, pc._width, pc._opacity] + ([pc._mask] if use_mask else [])
    if not torch.is_grad_enabled():
        return None
    for p in ps:
        g = p.grad
        if (not p.requires_grad or g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape
                or g.device != p.device or g.data_ptr() % 16):   # (the kernels write 16-byte rows: a misaligned view cannot serve)
            return None
    return [p.grad for p in ps]


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, separate_sh=False, override_color=None,
           use_trained_exp=False, use_mask=False, mask_thr=0.01, compute_visibility=True, clamp=True,
           compute_rend_dir=True, static_bucket_cap=0, status_sink=None, fused=None, grad_sinks=False):
    """Render the scene.  Background tensor (bg_color) must be on the GPU.  Returns the reference's dict
    {render, viewspace_points, visibility_filter, radii, depth, rend_dir, rend_alpha} (:147-155).

    Two routes, same results: for a ``GaussianCurveModel`` under the reference's default pipeline flags the whole view is ONE
    autograd node over ``cgs_view_forward_checked`` / ``cgs_view_backward`` (curve sampling, splat attributes, projection,
    binning, unit-colour compositors -- the kernels bench.py times); anything else goes through ``GaussianRasterizer`` like
    the reference (:96-129).  ``fused``: None = choose automatically, False = always the general route, True = insist.
    Flags of the reference's signature: ``pipe.compute_cov3D_python`` feeds ``pc.get_covariance`` as cov3D_precomp (:67-68);
    ``use_trained_exp`` applies the exposure with the reference's own expression (:131-135); ``separate_sh=True`` raises the
    TypeError the reference raises (its rasterizer has no ``dc`` argument: SURVEY quirk 20); ``override_color`` and
    ``pipe.convert_SHs_python`` are overwritten by the unit colours upstream (:96-97) and change nothing here either.
    ``grad_sinks=True`` (fused route, training loops that own their ``.grad`` buffers): the backward kernels add the gradients
    of curve points / width / opacity / mask straight into the parameters' existing ``.grad`` tensors -- the same values
    autograd's AccumulateGrad would have added, three to four kernel launches fewer per iteration; gradient hooks on those
    parameters do not see this contribution."""
    if separate_sh:   # (:108-119) GaussianRasterizer.forward() got an unexpected keyword argument 'dc'
        raise TypeError("GaussianRasterizer.forward() got an unexpected keyword argument 'dc' (render(separate_sh=True): the "
                        "reference's rasterizer has no separate-SH entry point, gaussian_renderer/__init__.py:108-119)")
    use_fused = _fused_route_ok(pc, pipe, scaling_modifier, override_color) if fused is None else bool(fused)
    if use_fused and hasattr(pc, "n_splats"):   # (the fused route never reads the derived tensors: a lazy model keeps them deferred)
        dev = pc._curve_points.device
        new_points = lambda: torch.zeros(pc.n_splats, 3, dtype=pc._curve_points.dtype, requires_grad=True, device=dev)
    else:
        dev = pc.get_xyz.device
        new_points = lambda: torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True, device=dev)
    # (:30-34) a zero tensor whose .grad receives the screen-space gradients; a leaf needs no retain_grad()
    screenspace_points = new_points()
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    if fused and not _fused_route_ok(pc, pipe, scaling_modifier, override_color):
        raise ValueError("render(fused=True): the fused view path needs a GaussianCurveModel whose derived tensors are current "
                         "and the reference's default pipeline flags")
    if use_fused:
        from ..ops import view_render as VR
        pend = []
        sinks = _grad_sinks(pc, use_mask) if grad_sinks else None
        while True:
            # clamp and direction map (:138-145) come out of the same autograd node (one epilogue launch); the exposure of
            # use_trained_exp sits between compositor and clamp upstream, so that combination keeps the separate ops
            fold = not use_trained_exp
            rendered_image, depth_image, out_all_map, radii, rend_dir = VR.view_render(
                pc._curve_points, pc._width, pc._opacity, pc._mask if use_mask else None, screenspace_points, pc.is_bezier,
                pc.n_gaussians, mask_thr, bg_color, viewpoint_camera, tanfovx, tanfovy, static_bucket_cap, status_sink,
                clamp and fold, compute_rend_dir and fold, pend, getattr(pc, "_derived_eps", 1e-8), sinks)
            try:
                pkg = _package(viewpoint_camera, pc, rendered_image, radii, depth_image, out_all_map, screenspace_points,
                               use_trained_exp, clamp and not fold, compute_rend_dir and not fold, False)
            except Exception:
                # (use_trained_exp on a one-channel image raises like the reference: no forward left outstanding)
                VR.finish(pend.pop() if pend else None)
                raise
            if compute_rend_dir and fold:
                pkg["rend_dir"] = rend_dir
            this = pend.pop() if pend else None
            ok, n_visible = VR.finish(this)
            if ok:
                break
            # a tile list outgrew its bucket (first view of a new scene, or a much denser one): the capacity has been raised
            screenspace_points = new_points()
        if compute_visibility:
            pkg["visibility_filter"] = VR.visible_indices(radii, n_visible, this)
        return pkg
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
    cov3D_precomp = None
    if pipe.compute_cov3D_python:   # (:67-68; the scales / rotations pair is then not passed, :69-71)
        cov3D_precomp = pc.get_covariance(scaling_modifier)
        # (:72-76) with use_mask the reference still sets `scales`, and its rasterizer then refuses the call (:196-197)
        scales, rotations = (scales if use_mask else None), None
    rendered_image, radii, depth_image, out_all_map = rasterizer(
        means3D=means3D, means2D=means2D, shs=None, colors_precomp=colors_precomp, opacities=opacity, scales=scales,
        rotations=rotations, all_map=input_all_map, cov3D_precomp=cov3D_precomp)
    # the blocking forward's status readback carries the radii > 0 count whenever it took the bucket path (-1 otherwise):
    # visibility_filter then needs no device-wide sync
    from .. import _lib as L
    n_visible = int(L.load().cgs_last_forward_visible()) if (compute_visibility and not static_bucket_cap) else None
    return _package(viewpoint_camera, pc, rendered_image, radii, depth_image, out_all_map, screenspace_points,
                    use_trained_exp, clamp, compute_rend_dir, compute_visibility, n_visible)


class _Epilogue(torch.autograd.Function):
    """(:138-145) clamp of the image and view -> world transform of the direction map in ONE launch (cgs_render_epilogue)
    instead of a clamp kernel and three broadcast multiply-adds; backward: torch.clamp's gradient mask in one launch
    (cgs_clamp_backward), the direction map's (rare) gradient with a 3x3 contraction."""

    @staticmethod
    def forward(ctx, image, all_map, view, clamp, want_dir):
        from .. import _lib as L
        lib = L.load()
        dev = image.device
        raw = image.detach().contiguous()
        am = all_map.detach().contiguous()
        vw = view.detach().float().contiguous()
        H, W = raw.shape[-2], raw.shape[-1]
        out = torch.empty_like(raw) if clamp else raw
        rend_dir = torch.empty((3, H, W), dtype=torch.float32, device=dev) if want_dir else torch.empty(0, device=dev)
        with L.device_guard(dev):
            L.check(lib.cgs_render_epilogue(H, W, L.ptr(raw), L.ptr(am), L.ptr(vw), 1, L.ptr(out) if clamp else None,
                                            L.ptr(rend_dir), L.raw_stream(dev)), "cgs_render_epilogue")
        ctx.raw = raw if clamp else None
        ctx.view = vw
        ctx.set_materialize_grads(False)
        return out, rend_dir

    @staticmethod
    def backward(ctx, g_img, g_dir):
        from .. import _lib as L
        g_raw = g_map = None
        if g_img is not None:
            g_img = g_img.float().contiguous()
            if ctx.raw is not None:
                g_raw = torch.empty_like(g_img)
                with L.device_guard(g_img.device):
                    L.check(L.load().cgs_clamp_backward(g_img.numel(), L.ptr(ctx.raw), L.ptr(g_img), L.ptr(g_raw),
                                                        L.raw_stream(g_img.device)), "cgs_clamp_backward")
            else:
                g_raw = g_img
        if g_dir is not None and g_dir.numel():
            g3 = torch.einsum("ik,ihw->khw", ctx.view[:3, :3], g_dir)      # out_i = sum_k d_k wv[i][k]
            g_map = torch.cat([g3, torch.zeros_like(g3[:1])], 0)
        return g_raw, g_map, None, None, None


def _visible(radii, n_visible):
    """(:150) `(radii > 0).nonzero()`; the fused route knows the count from its status readback and skips the host sync."""
    from ..ops.view_render import visible_indices
    return visible_indices(radii, n_visible)


def _package(viewpoint_camera, pc, rendered_image, radii, depth_image, out_all_map, screenspace_points, use_trained_exp,
             clamp, compute_rend_dir, compute_visibility, n_visible=None):
    """:131-155 -- exposure, clamp, world-space direction map, the result dict."""
    if use_trained_exp:   # (:131-135) the reference's expression, verbatim in meaning: a [H,W,C] x [3,3] product, i.e. it only
        # type-checks for a 3-channel image -- with this rasterizer's single channel it raises torch's shape error upstream too
        exposure = pc.get_exposure_from_name(viewpoint_camera.image_name)
        rendered_image = (torch.matmul(rendered_image.permute(1, 2, 0), exposure[:3, :3]).permute(2, 0, 1)
                          + exposure[:3, 3, None, None])
    # clamp=False (extension): hand the raw composite to ops.losses.photometric_loss(clamp=True), which applies the
    # clamp and its gradient mask inside the loss kernels
    rendered_alpha = out_all_map[3:4, ]
    # view space -> world space (:143-145).  The reference does this with a [H*W,3] x [3,3] matmul (a 150 us GEMM launch on
    # a 1600^2 image) after a clamp kernel; both in one launch here.  compute_rend_dir=False (extension) skips the map for
    # callers that do not use it (the photometric train step).
    rendered_dir = None
    if (clamp or compute_rend_dir) and rendered_image.is_cuda and rendered_image.shape[0] == 1 and rendered_image.dtype == torch.float32:
        rendered_image, rd = _Epilogue.apply(rendered_image, out_all_map, viewpoint_camera.world_view_transform, bool(clamp),
                                             bool(compute_rend_dir))
        if compute_rend_dir:
            rendered_dir = rd
    else:
        if clamp:
            rendered_image = rendered_image.clamp(0, 1)
        if compute_rend_dir:
            wv = viewpoint_camera.world_view_transform[:3, :3]
            d = out_all_map[0:3]
            rendered_dir = torch.addcmul(torch.addcmul(d[0:1] * wv[:, 0].view(3, 1, 1), d[1:2], wv[:, 1].view(3, 1, 1)),
                                         d[2:3], wv[:, 2].view(3, 1, 1))
    # compute_visibility=False (extension) skips the nonzero(), which is a host sync (train.py only needs it for the
    # densification statistics and the opacity regulariser)
    return {"render": rendered_image, "viewspace_points": screenspace_points,
            "visibility_filter": _visible(radii, n_visible) if compute_visibility else None, "radii": radii,
            "depth": depth_image,
            "rend_dir": rendered_dir, "rend_alpha": rendered_alpha}
