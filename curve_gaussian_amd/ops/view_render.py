"""One view of the training configuration as ONE autograd node over the fused per-view entry points of libcurvegs
(``cgs_view_forward_checked`` / ``cgs_view_backward``, csrc/view.hip + the unit-colour compositors): curve parameters in,
image / inverse depth / all_map / radii out; image gradient in, curve-parameter gradients and the screen-space gradient of
``add_densification_stats`` out.  This is what ``gaussian_renderer.render`` runs for a ``GaussianCurveModel`` under the
reference's default pipeline flags, i.e. the call of /root/reference/train.py:95-97 through
/root/reference/gaussian_renderer/__init__.py:18-157 -- the same kernels ``bench.py`` and ``GraphedTrainStep`` launch,
reached through the reference's own call sequence.

GPU only (no CPU path in the product)."""
import ctypes as C

import torch

from .. import _lib as L
from .curve_sampling import _bezier_mask, sample_coefficients

import threading

_f = C.c_float
_caps = {}   # (device index, P, W, H) -> bucket capacity that held the last forward of this shape
_caps_mu = threading.Lock()


class Pending:
    """One eager forward between cgs_view_forward_begin and cgs_view_forward_wait: the library's handle and what finish()
    needs to judge the readback.  Carried by the caller (render() gets it back from view_render), so forwards of different
    threads, devices, streams or models never see each other's."""
    __slots__ = ("handle", "key", "cap", "img")

    def __init__(self, handle, key, cap, img=None):
        self.handle, self.key, self.cap = handle, key, cap
        self.img = img     # the forward's image buffer: holds the per-chunk visible counts cgs_visible_indices reads

    def __del__(self):   # dropped without finish() (an exception in between): give the slot back
        if self.handle is not None and self.handle >= 0:
            try:
                L.load().cgs_view_forward_abandon(self.handle)
            except Exception:
                pass
            self.handle = None


_NORMS_RANGE = []


def _norms_backward_range():
    """(first, count) of the f64 words of `norms` the sampling backward accumulates into -- asked of the library once."""
    if not _NORMS_RANGE:
        a, b = C.c_int(0), C.c_int(0)
        L.load().cgs_view_norms_backward_range(C.byref(a), C.byref(b))
        _NORMS_RANGE.extend((int(a.value), int(b.value)))
    return _NORMS_RANGE[0], _NORMS_RANGE[1]


def _capacity(lib, dev, P, W, H):
    with _caps_mu:
        cap = _caps.get((dev.index, P, W, H), 0)
    hint = int(lib.cgs_bucket_capacity_hint(P, W, H))
    return max(cap, hint, 128)


class _ViewRender(torch.autograd.Function):
    """forward(curve_points [B,4,3], width [B,1], opacity_logit [B,1], mask_logit [B,m,1] | None, means2D [P,3] zeros, ...)
    -> (color [1,H,W], invdepth [1,H,W], all_map [4,H,W], radii [P] int32, rend_dir).  The fast backward handles d/dcolor -- the
    loss of train.py:98-107 reads `render` alone.  A gradient arriving at inverse depth, all_map or the direction map (a depth
    / normal loss: the reference's rasterizer backward takes grad_out_depth and grad_out_all_map,
    diff_cur_rasterization/__init__.py:117-151) is served too: the backward then re-renders the view through the general
    operator route under autograd and pulls all upstream gradients through it (_general_backward).  Tolerance of that route:
    the re-render bins with exact-size lists instead of the forward's fixed-capacity buckets and evaluates alpha on the general
    compositor; its image equals the one the forward returned within the parity criterion of tests/util.py (1e-4 of the
    tensor's maximum: test_default_render_takes_the_fused_route_and_equals_the_general_one), so a mixed colour + depth loss gets
    gradients that are exact for the re-rendered image and within 2e-4 relative of the all-general computation
    (test_fused_route_serves_depth_and_normal_losses)."""

    @staticmethod
    def forward(ctx, curve_points, width, opacity_logit, mask_logit, means2D, is_bezier, m, mask_thr, bg, cam, tanx, tany,
                static_cap=0, status_sink=None, clamp=False, want_dir=False, pending_out=None, eps=1e-8, grad_sinks=None):
        L.require_gpu_tensor(curve_points, "curve_points")
        ctx.sinks = grad_sinks
        ctx.sink_owners = ((curve_points, width, opacity_logit) + ((mask_logit,) if mask_logit is not None else ())) if grad_sinks else None
        L.require_gpu_tensor(bg, "bg_color")                       # "Background tensor (bg_color) must be on GPU!" (:23)
        L.require_gpu_tensor(cam.world_view_transform, "viewpoint_camera.world_view_transform")
        lib = L.load()
        dev = curve_points.device
        if L.use_shim():
            return _ViewRender._forward_shim(ctx, lib, curve_points, width, opacity_logit, mask_logit, is_bezier, m, mask_thr, bg,
                                             cam, tanx, tany, static_cap, status_sink, clamp, want_dir, pending_out, eps)
        with L.device_guard(dev):
            c = lambda t: None if t is None else t.detach().float().contiguous()
            cp, w, ol, mk = c(curve_points), c(width), c(opacity_logit), c(mask_logit)
            B = cp.shape[0]
            P = B * m
            H, W = int(cam.image_height), int(cam.image_width)
            tiles = ((W + 15) // 16) * ((H + 15) // 16)
            isb = _bezier_mask(is_bezier, dev)
            coef = sample_coefficients(m, dev)
            u8 = lambda n: torch.empty(int(n), dtype=torch.uint8, device=dev)
            f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            norms = torch.empty(384, dtype=torch.float64, device=dev)
            geom, img = u8(lib.cgs_geometry_bytes(P)), u8(lib.cgs_image_bytes(W, H))
            color, invd, amap = f32(1, H, W), f32(1, H, W), f32(4, H, W)
            radii = torch.empty(P, dtype=torch.int32, device=dev)
            view, proj, campos = c(cam.world_view_transform), c(cam.full_proj_transform), c(cam.camera_center)
            bgc = c(bg)
            st = L.raw_stream(dev)
            cap = int(static_cap) if static_cap else _capacity(lib, dev, P, W, H)
            limit = int(lib.cgs_bucket_capacity_limit())
            while static_cap:   # sync-free (stream-ordered / graph-captured callers): nothing is read back, the caller checks
                nbin = int(lib.cgs_binning_bytes(cap * tiles))   # the status words it is handed through status_sink
                binb = u8(nbin)
                L.check(lib.cgs_view_forward(
                    B, m, L.ptr(cp), L.ptr(w), L.ptr(isb), L.ptr(coef), _f(eps), L.ptr(norms), L.ptr(ol), L.ptr(mk),
                    _f(mask_thr), None, L.ptr(geom), L.ptr(binb), nbin, L.ptr(img), cap, L.ptr(bgc), W, H, L.ptr(view),
                    L.ptr(proj), L.ptr(campos), _f(tanx), _f(tany), L.ptr(color), L.ptr(invd), L.ptr(amap), L.ptr(radii),
                    None, None, None, st), "cgs_view_forward")
                if status_sink is not None:
                    off, nw = int(lib.cgs_image_status_offset(W, H)), int(lib.cgs_status_words())
                    status_sink.append(img[off:off + 4 * nw].view(torch.int32))
                break
            if not static_cap:
                # eager callers: everything is enqueued, including a 16-byte status readback right behind the scatter; the
                # caller queues what else it has (render(): clamp, direction map) and then calls finish(), which blocks on
                # that readback only -- the compositor is still running -- and says whether the buckets held
                nbin = int(lib.cgs_binning_bytes(cap * tiles))
                binb = u8(nbin)
                handle = L.check(lib.cgs_view_forward_begin(
                    B, m, L.ptr(cp), L.ptr(w), L.ptr(isb), L.ptr(coef), _f(eps), L.ptr(norms), L.ptr(ol), L.ptr(mk),
                    _f(mask_thr), None, L.ptr(geom), L.ptr(binb), nbin, L.ptr(img), cap, L.ptr(bgc), W, H, L.ptr(view),
                    L.ptr(proj), L.ptr(campos), _f(tanx), _f(tany), L.ptr(color), L.ptr(invd), L.ptr(amap), L.ptr(radii),
                    None, None, None, st), "cgs_view_forward_begin")
                pend = Pending(handle, (dev.index, P, W, H), cap, img)
                if pending_out is not None:
                    pending_out.append(pend)
                else:   # nobody will call finish(): wait here, like cgs_view_forward_checked
                    ok, _ = finish(pend)
                    if not ok:
                        raise L.CurveGSError("view_render: a tile list outgrew its bucket; call again (the capacity for this "
                                             "shape has been raised) or pass pending_out and retry on finish() == False")
            # render()'s epilogue (:138-145) in the same stream, one launch: clamp of the image, view -> world direction map
            color_raw, rend_dir = color, None
            if clamp or want_dir:
                if clamp:
                    color = f32(1, H, W)
                if want_dir:
                    rend_dir = f32(3, H, W)
                L.check(lib.cgs_render_epilogue(H, W, L.ptr(color_raw), L.ptr(amap), L.ptr(view), 1, L.ptr(color) if clamp else None,
                                                L.ptr(rend_dir), st), "cgs_render_epilogue")
        ctx.save_for_backward(cp, w, ol, mk if mk is not None else torch.empty(0, device=dev), geom, binb, img, radii, norms,
                              bgc, view, proj, campos)
        ctx.isb, ctx.coef = isb, coef
        ctx.is_bezier, ctx.clamped = is_bezier, bool(clamp)
        ctx.dims = (B, m, H, W, float(mask_thr), float(tanx), float(tany), float(eps))
        ctx.has_mask = mk is not None
        ctx.raw = color_raw if clamp else None          # the clamp's gradient mask needs the unclamped image
        if rend_dir is None:
            rend_dir = torch.empty(0, device=dev)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return color, invd, amap, radii, rend_dir

    @staticmethod
    def _forward_shim(ctx, lib, curve_points, width, opacity_logit, mask_logit, is_bezier, m, mask_thr, bg, cam, tanx, tany,
                      static_cap, status_sink, clamp, want_dir, pending_out, eps):
        """The same forward as ONE call into the compiled host shim (csrc/torch_shim.cpp::view_forward)."""
        dev = curve_points.device
        H, W = int(cam.image_height), int(cam.image_width)
        P = curve_points.shape[0] * m
        isb = _bezier_mask(is_bezier, dev)
        coef = sample_coefficients(m, dev)
        cap = int(static_cap) if static_cap else _capacity(lib, dev, P, W, H)
        color, invd, amap, radii, rend_dir, color_raw, saved, handle = L.shim().view_forward(
            curve_points, width, opacity_logit, mask_logit, isb, coef, m, mask_thr, bg, cam.world_view_transform,
            cam.full_proj_transform, cam.camera_center, tanx, tany, H, W, cap, bool(static_cap), bool(clamp), bool(want_dir), eps)
        cp, w, ol, mk, geom, binb, img, _radii, norms, bgc, view, proj, campos = saved
        if static_cap:
            if status_sink is not None:
                off, nw = int(lib.cgs_image_status_offset(W, H)), int(lib.cgs_status_words())
                status_sink.append(img[off:off + 4 * nw].view(torch.int32))
        else:
            pend = Pending(handle, (dev.index, P, W, H), cap, img)
            if pending_out is not None:
                pending_out.append(pend)
            else:
                ok, _ = finish(pend)
                if not ok:
                    raise L.CurveGSError("view_render: a tile list outgrew its bucket; call again (the capacity for this "
                                         "shape has been raised) or pass pending_out and retry on finish() == False")
        ctx.save_for_backward(cp, w, ol, mk if mk is not None else torch.empty(0, device=dev), geom, binb, img, radii, norms,
                              bgc, view, proj, campos)
        ctx.isb, ctx.coef = isb, coef
        ctx.is_bezier, ctx.clamped = is_bezier, bool(clamp)
        ctx.dims = (curve_points.shape[0], m, H, W, float(mask_thr), float(tanx), float(tany), float(eps))
        ctx.has_mask = mk is not None
        ctx.raw = color_raw
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return color, invd, amap, radii, rend_dir

    @staticmethod
    def backward(ctx, g_color, g_invd, g_amap, _g_radii, g_dir):
        if g_invd is not None or g_amap is not None or g_dir is not None:
            return _general_backward(ctx, g_color, g_invd, g_amap, g_dir) + (None,) * 14
        cp, w, ol, mk, geom, binb, img, radii, norms, bgc, view, proj, campos = ctx.saved_tensors
        B, m, H, W, mask_thr, tanx, tany, eps = ctx.dims
        lib = L.load()
        dev = cp.device
        P = B * m
        mkp = mk if ctx.has_mask else None
        if getattr(ctx, "ran_backward", False):
            # a second backward over this forward (retain_graph): the two grid-wide sums of the sampling backward were cleared
            # by the forward's norm pass once (include/curvegs.h: one view backward per view forward) -- clear them again
            first, count = _norms_backward_range()
            norms[first:first + count].zero_()
        sinks = ctx.sinks
        if sinks is not None:
            # the sinks are the `.grad` tensors the parameters had at FORWARD time: if one was replaced since (zero_grad(
            # set_to_none=True), a rebound flat buffer, a topology edit) the kernels would add into an orphan -- hand the
            # gradients to autograd the ordinary way instead
            owners = getattr(ctx, "sink_owners", None)
            if owners is None or any(p.grad is not s for p, s in zip(owners, sinks)):
                sinks = None
        if L.use_shim():   # clamp gradient + cgs_view_backward in one call (csrc/torch_shim.cpp::view_backward)
            # (sinks: the kernels add the curve-level gradients to the caller's buffers; None comes back for those inputs)
            out = tuple(L.shim().view_backward(cp, w, ol, mkp, ctx.isb, ctx.coef, geom, binb, img, radii, norms, bgc, view, proj,
                                               campos, m, mask_thr, tanx, tany, H, W, eps, g_color, ctx.raw, sinks)) + (None,) * 14
            ctx.ran_backward = True
            return out
        ctx.ran_backward = True   # (ctypes bindings: the kernels below are queued before anything can raise)
        with L.device_guard(dev):
            f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            g_cp, g_w, g_ol, g_m2d = f32(B, 4, 3), f32(B, 1), f32(B, 1), f32(P, 3)
            g_mk = torch.empty_like(mkp) if mkp is not None else None
            if g_color is None:
                g_cp.zero_(); g_w.zero_(); g_ol.zero_(); g_m2d.zero_()
                if g_mk is not None:
                    g_mk.zero_()
                return (g_cp, g_w, g_ol, g_mk, g_m2d) + (None,) * 14
            g_color = g_color.float().contiguous()
            if ctx.raw is not None:   # torch.clamp's gradient rule on the unclamped image
                g_raw = torch.empty_like(g_color)
                L.check(lib.cgs_clamp_backward(g_color.numel(), L.ptr(ctx.raw), L.ptr(g_color), L.ptr(g_raw), L.raw_stream(dev)),
                        "cgs_clamp_backward")
                g_color = g_raw
            scratch = f32(int(lib.cgs_view_backward_scratch_floats(B, m)))
            rc = lib.cgs_view_backward(
                B, m, L.ptr(cp), L.ptr(w), L.ptr(ctx.isb), L.ptr(ctx.coef), _f(eps), L.ptr(norms), L.ptr(ol), L.ptr(mkp),
                _f(mask_thr), None, L.ptr(geom), L.ptr(binb), L.ptr(img), L.ptr(bgc), W, H, L.ptr(view), L.ptr(proj),
                L.ptr(campos), _f(tanx), _f(tany), L.ptr(radii), L.ptr(g_color), None, L.ptr(g_m2d), L.ptr(g_cp), L.ptr(g_w),
                L.ptr(g_ol), L.ptr(g_mk), L.ptr(scratch), 0, L.raw_stream(dev))
            L.check(rc, "cgs_view_backward")
        return (g_cp, g_w, g_ol, g_mk, g_m2d) + (None,) * 14


def _general_backward(ctx, g_color, g_invd, g_amap, g_dir):
    """Backward of the fused node when a gradient reaches inverse depth / all_map / the direction map: the view is rendered
    again from the saved parameters through the differentiable general route (sample_curves -> splat_attributes ->
    GaussianRasterizer, the reference's own call sequence, gaussian_renderer/__init__.py:57-129) and every upstream gradient
    is pulled through that graph.  Same splats, same image to rounding; costs one extra forward, only for such losses."""
    from ..diff_cur_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from .curve_sampling import sample_curves, splat_attributes
    cp, w, ol, mk, _geom, _binb, _img, _radii, _norms, bgc, view, proj, campos = ctx.saved_tensors
    B, m, H, W, mask_thr, tanx, tany, eps = ctx.dims
    dev = cp.device
    P = B * m
    with torch.enable_grad():
        leaves = [t.detach().requires_grad_(True) for t in (cp, w, ol)]
        mk_l = mk.detach().requires_grad_(True) if ctx.has_mask else None
        xyz, rot, scl = sample_curves(leaves[0], leaves[1], ctx.is_bezier, m, eps)
        rotn, opac, scales, amap_in = splat_attributes(rot, xyz, leaves[2], scl, campos, view, m, mk_l, mask_thr)
        m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
        rs = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=tanx, tanfovy=tany, bg=bgc, scale_modifier=1.0, viewmatrix=view.view(4, 4),
            projmatrix=proj.view(4, 4), sh_degree=0, campos=campos, prefiltered=False, debug=False, antialiasing=False,
            render_geo=True)
        color, _r, invd, amap = GaussianRasterizer(rs)(
            means3D=xyz, means2D=m2d, shs=None, colors_precomp=torch.ones(P, 1, device=dev), opacities=opac, scales=scales,
            rotations=rotn, all_map=amap_in, cov3D_precomp=None)
        outs, gos = [], []
        if g_color is not None:
            outs.append(color.clamp(0, 1) if ctx.clamped else color)
            gos.append(g_color.reshape(color.shape))
        if g_invd is not None:
            outs.append(invd)
            gos.append(g_invd)
        if g_amap is not None:
            outs.append(amap)
            gos.append(g_amap)
        if g_dir is not None:   # (:143-145) rendered_dir.permute(1, 2, 0) @ world_view_transform[:3, :3].T
            wv = view.view(4, 4)[:3, :3]
            outs.append(torch.einsum("ik,khw->ihw", wv, amap[0:3]))
            gos.append(g_dir)
        wrt = leaves + ([mk_l] if mk_l is not None else []) + [m2d]
        gr = torch.autograd.grad(outs, wrt, gos, allow_unused=True)
    gr = [torch.zeros_like(t) if g is None else g for g, t in zip(gr, wrt)]
    g_mk = gr[3] if mk_l is not None else None
    return gr[0], gr[1], gr[2], g_mk, gr[-1]


def view_render(curve_points, width, opacity_logit, mask_logit, means2D, is_bezier, m, mask_thr, bg, cam, tanx, tany,
                static_cap=0, status_sink=None, clamp=False, want_dir=False, pending_out=None, eps=1e-8, grad_sinks=None):
    """-> (image [1,H,W] (clamped to [0,1] when `clamp`), inverse depth [1,H,W], all_map [4,H,W], radii [P], world-space
    direction map [3,H,W] or an empty tensor).  Eager callers (static_cap == 0) pass a list as `pending_out`: it receives the
    forward's `Pending`, to be handed to finish() once the caller has queued whatever else it has; without it the call waits
    for the status readback itself.

    `grad_sinks` (compiled shim only; ignored otherwise): float32 tensors shaped like (curve_points, width, opacity_logit[,
    mask_logit]) -- normally the parameters' `.grad` -- that the backward kernels ADD their gradients to; the node then hands
    autograd no gradient for those inputs (no AccumulateGrad kernels).  A loss that reaches depth / all_map / the direction
    map takes the general backward, which returns its gradients the ordinary way."""
    if L.use_shim():   # the C++ autograd node of the compiled shim (csrc/torch_shim.cpp::ViewRenderFn)
        return _view_render_cpp(curve_points, width, opacity_logit, mask_logit, means2D, is_bezier, m, mask_thr, bg, cam, tanx, tany,
                                static_cap, status_sink, clamp, want_dir, pending_out, eps, grad_sinks)
    grad_sinks = None
    return _ViewRender.apply(curve_points, width, opacity_logit, mask_logit, means2D, is_bezier, m, mask_thr, bg, cam, tanx,
                             tany, static_cap, status_sink, clamp, want_dir, pending_out, eps, grad_sinks)


_cpp_ready = []


def _general_backward_cpp(cp, w, ol, mk, bgc, view, proj, campos, is_bezier, m, H, W, mask_thr, tanx, tany, eps, clamped, g_color, g_invd,
                          g_amap, g_dir):
    """Called by the C++ node's backward (with the GIL) when a gradient reaches inverse depth / all_map / the direction map."""
    import types
    empty = torch.empty(0, device=cp.device)
    ctx = types.SimpleNamespace(
        saved_tensors=(cp, w, ol, mk if mk is not None else empty, None, None, None, None, None, bgc, view, proj, campos),
        dims=(cp.shape[0], int(m), int(H), int(W), float(mask_thr), float(tanx), float(tany), float(eps)), is_bezier=is_bezier,
        has_mask=mk is not None, clamped=bool(clamped))
    return _general_backward(ctx, g_color, g_invd, g_amap, g_dir)


def _view_render_cpp(curve_points, width, opacity_logit, mask_logit, means2D, is_bezier, m, mask_thr, bg, cam, tanx, tany, static_cap,
                     status_sink, clamp, want_dir, pending_out, eps, grad_sinks):
    """view_render() on the compiled shim's autograd node: one pybind call builds the node and runs the forward; its backward
    runs on the autograd engine's device thread without the GIL (the general backward of depth / normal losses calls back)."""
    L.require_gpu_tensor(curve_points, "curve_points")
    shim = L.shim()
    if not _cpp_ready:
        shim.set_general_backward(_general_backward_cpp)
        _cpp_ready.append(True)
    lib = L.load()
    dev = curve_points.device
    H, W = int(cam.image_height), int(cam.image_width)
    P = curve_points.shape[0] * m
    isb = _bezier_mask(is_bezier, dev)
    coef = sample_coefficients(m, dev)
    cap = int(static_cap) if static_cap else _capacity(lib, dev, P, W, H)
    color, invd, amap, radii, rend_dir, handle, img = shim.view_render(
        curve_points, width, opacity_logit, mask_logit, means2D, isb, is_bezier, coef, m, mask_thr, bg, cam.world_view_transform,
        cam.full_proj_transform, cam.camera_center, tanx, tany, H, W, cap, bool(static_cap), bool(clamp), bool(want_dir), eps, grad_sinks)
    if static_cap:
        if status_sink is not None:
            off, nw = int(lib.cgs_image_status_offset(W, H)), int(lib.cgs_status_words())
            status_sink.append(img[off:off + 4 * nw].view(torch.int32))
    else:
        pend = Pending(handle, (dev.index, P, W, H), cap, img)
        if pending_out is not None:
            pending_out.append(pend)
        else:
            ok, _ = finish(pend)
            if not ok:
                raise L.CurveGSError("view_render: a tile list outgrew its bucket; call again (the capacity for this "
                                     "shape has been raised) or pass pending_out and retry on finish() == False")
    return color, invd, amap, radii, rend_dir


def finish(pend):
    """Second half of an eager view_render(): wait for the status readback of `pend` (None: a sync-free forward, nothing to
    wait for).  -> (ok, n_visible); ok = False means a tile list outgrew its bucket -- the outputs of that forward are INVALID,
    the capacity for this shape has been raised, render again."""
    if pend is None or pend.handle is None:
        return True, -1
    lib = L.load()
    handle, pend.handle = pend.handle, None   # (the wait releases the library's slot whatever it returns)
    nvis = C.c_int64(-1)
    longest = int(L.check(lib.cgs_view_forward_wait(handle, C.byref(nvis)), "cgs_view_forward_wait"))
    if longest <= pend.cap:
        with _caps_mu:
            _caps[pend.key] = pend.cap
        return True, int(nvis.value)
    limit = int(lib.cgs_bucket_capacity_limit())
    if longest > limit:   # a tile list the bucket layout cannot hold
        raise L.CurveGSError(f"render: a tile list of {longest} entries exceeds the bucket limit {limit}; use fused=False")
    with _caps_mu:
        _caps[pend.key] = min(limit, (longest * 5 // 4 + 64 + 63) & ~63)
    return False, -1


def visible_indices(radii, n_visible, pend=None):
    """(radii > 0).nonzero() (gaussian_renderer/__init__.py:150).  With the count known (the checked forward's status readback
    carries it) nothing waits for the device; with the forward's `Pending` as well it is ONE launch of the library
    (cgs_visible_indices: the forward left per-chunk counts in its image buffer) instead of a compare + nonzero_static."""
    if n_visible is not None and n_visible >= 0:
        if pend is not None and pend.img is not None and radii.is_cuda and radii.dtype == torch.int32 and radii.numel() > 0:
            out = torch.empty((int(n_visible), 1), dtype=torch.int64, device=radii.device)
            if n_visible > 0:
                _dev, _P, W, H = pend.key
                with L.device_guard(radii.device):
                    L.check(L.load().cgs_visible_indices(radii.numel(), L.ptr(radii), L.ptr(pend.img), W, H, L.ptr(out),
                                                         L.raw_stream(radii.device)), "cgs_visible_indices")
            return out
        return torch.nonzero_static(radii > 0, size=int(n_visible))
    return (radii > 0).nonzero()
