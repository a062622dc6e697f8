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

_f = C.c_float
_caps = {}   # (device index, P, W, H) -> bucket capacity that held the last forward of this shape
_last_visible = [-1]   # radii > 0 count of the last checked forward (read back with its status words), -1: unknown
_pending = []          # checked forwards begun and not yet finished (at most one: finish() follows every eager forward)


def _capacity(lib, dev, P, W, H):
    cap = _caps.get((dev.index, P, W, H), 0)
    hint = int(lib.cgs_bucket_capacity_hint(P, W, H))
    return max(cap, hint, 128)


class _ViewRender(torch.autograd.Function):
    """forward(curve_points [B,4,3], width [B,1], opacity_logit [B,1], mask_logit [B,m,1] | None, means2D [P,3] zeros, ...)
    -> (color [1,H,W], invdepth [1,H,W], all_map [4,H,W], radii [P] int32, rend_dir).  Only d/dcolor is supported upstream -- the loss of
    train.py:98-107 reads `render` alone; a gradient arriving at inverse depth or all_map raises."""

    @staticmethod
    def forward(ctx, curve_points, width, opacity_logit, mask_logit, means2D, is_bezier, m, mask_thr, bg, cam, tanx, tany,
                static_cap=0, status_sink=None, clamp=False, want_dir=False):
        L.require_gpu_tensor(curve_points, "curve_points")
        L.require_gpu_tensor(bg, "bg_color")                       # "Background tensor (bg_color) must be on GPU!" (:23)
        L.require_gpu_tensor(cam.world_view_transform, "viewpoint_camera.world_view_transform")
        lib = L.load()
        dev = curve_points.device
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
                    B, m, L.ptr(cp), L.ptr(w), L.ptr(isb), L.ptr(coef), _f(1e-8), L.ptr(norms), L.ptr(ol), L.ptr(mk),
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
                L.check(lib.cgs_view_forward_begin(
                    B, m, L.ptr(cp), L.ptr(w), L.ptr(isb), L.ptr(coef), _f(1e-8), L.ptr(norms), L.ptr(ol), L.ptr(mk),
                    _f(mask_thr), None, L.ptr(geom), L.ptr(binb), nbin, L.ptr(img), cap, L.ptr(bgc), W, H, L.ptr(view),
                    L.ptr(proj), L.ptr(campos), _f(tanx), _f(tany), L.ptr(color), L.ptr(invd), L.ptr(amap), L.ptr(radii),
                    None, None, None, st), "cgs_view_forward_begin")
                del _pending[:]   # one forward outstanding per thread (the library keeps one readback slot): a forward whose
                _pending.append((dev.index, P, W, H, cap))   # finish() never ran (exception in between) is superseded
            _last_visible[0] = -1
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
        ctx.dims = (B, m, H, W, float(mask_thr), float(tanx), float(tany))
        ctx.has_mask = mk is not None
        ctx.raw = color_raw if clamp else None          # the clamp's gradient mask needs the unclamped image
        if rend_dir is None:
            rend_dir = torch.empty(0, device=dev)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return color, invd, amap, radii, rend_dir

    @staticmethod
    def backward(ctx, g_color, g_invd, g_amap, _g_radii, g_dir):
        if g_invd is not None or g_amap is not None or g_dir is not None:
            raise L.CurveGSError(
                "render (fused view path): only `render` may carry a gradient; a loss on depth / rend_dir / rend_alpha needs "
                "the general rasterizer -- call render(..., fused=False)")
        cp, w, ol, mk, geom, binb, img, radii, norms, bgc, view, proj, campos = ctx.saved_tensors
        B, m, H, W, mask_thr, tanx, tany = ctx.dims
        lib = L.load()
        dev = cp.device
        P = B * m
        mkp = mk if ctx.has_mask else None
        with L.device_guard(dev):
            f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            g_cp, g_w, g_ol, g_m2d = f32(B, 4, 3), f32(B, 1), f32(B, 1), f32(P, 3)
            g_mk = torch.empty_like(mkp) if mkp is not None else None
            if g_color is None:
                g_cp.zero_(); g_w.zero_(); g_ol.zero_(); g_m2d.zero_()
                if g_mk is not None:
                    g_mk.zero_()
                return g_cp, g_w, g_ol, g_mk, g_m2d, None, None, None, None, None, None, None, None, None, None, None
            g_color = g_color.float().contiguous()
            if ctx.raw is not None:   # torch.clamp's gradient rule on the unclamped image
                g_raw = torch.empty_like(g_color)
                L.check(lib.cgs_clamp_backward(g_color.numel(), L.ptr(ctx.raw), L.ptr(g_color), L.ptr(g_raw), L.raw_stream(dev)),
                        "cgs_clamp_backward")
                g_color = g_raw
            scratch = f32(int(lib.cgs_view_backward_scratch_floats(B, m)))
            rc = lib.cgs_view_backward(
                B, m, L.ptr(cp), L.ptr(w), L.ptr(ctx.isb), L.ptr(ctx.coef), _f(1e-8), L.ptr(norms), L.ptr(ol), L.ptr(mkp),
                _f(mask_thr), None, L.ptr(geom), L.ptr(binb), L.ptr(img), L.ptr(bgc), W, H, L.ptr(view), L.ptr(proj),
                L.ptr(campos), _f(tanx), _f(tany), L.ptr(radii), L.ptr(g_color), None, L.ptr(g_m2d), L.ptr(g_cp), L.ptr(g_w),
                L.ptr(g_ol), L.ptr(g_mk), L.ptr(scratch), 0, L.raw_stream(dev))
            L.check(rc, "cgs_view_backward")
        return g_cp, g_w, g_ol, g_mk, g_m2d, None, None, None, None, None, None, None, None, None, None, None


def view_render(curve_points, width, opacity_logit, mask_logit, means2D, is_bezier, m, mask_thr, bg, cam, tanx, tany,
                static_cap=0, status_sink=None, clamp=False, want_dir=False):
    """-> (image [1,H,W] (clamped to [0,1] when `clamp`), inverse depth [1,H,W], all_map [4,H,W], radii [P], world-space
    direction map [3,H,W] or an empty tensor)."""
    return _ViewRender.apply(curve_points, width, opacity_logit, mask_logit, means2D, is_bezier, m, mask_thr, bg, cam, tanx,
                             tany, static_cap, status_sink, clamp, want_dir)


def finish():
    """Second half of an eager view_render(): wait for the status readback.  -> (ok, n_visible); ok = False means a tile list
    outgrew its bucket -- the outputs of that forward are INVALID, the capacity for this shape has been raised, render again."""
    if not _pending:
        return True, -1
    dev_index, P, W, H, cap = _pending.pop()
    lib = L.load()
    longest = int(L.check(lib.cgs_view_forward_wait(), "cgs_view_forward_wait"))
    if longest <= cap:
        _caps[(dev_index, P, W, H)] = cap
        return True, int(lib.cgs_last_forward_visible())
    limit = int(lib.cgs_bucket_capacity_limit())
    if longest > limit:   # a tile list the bucket layout cannot hold
        raise L.CurveGSError(f"render: a tile list of {longest} entries exceeds the bucket limit {limit}; use fused=False")
    _caps[(dev_index, P, W, H)] = min(limit, (longest * 5 // 4 + 64 + 63) & ~63)
    return False, -1


def visible_indices(radii, n_visible):
    """(radii > 0).nonzero() (gaussian_renderer/__init__.py:150) -- without the device-wide sync when the count is known."""
    if n_visible is not None and n_visible >= 0:
        return torch.nonzero_static(radii > 0, size=int(n_visible))
    return (radii > 0).nonzero()
