"""Drop-in for the reference's ``diff_cur_rasterization`` package, backed by libcurvegs.so (HIP, gfx950).

Public surface mirrors /root/reference/submodules/diff-cur-rasterization/diff_cur_rasterization/__init__.py:
``GaussianRasterizationSettings`` (:153-167), ``GaussianRasterizer`` (:169-222), ``rasterize_gaussians`` (:21-44),
``_RasterizeGaussians`` (:46-151) and the extension namespace ``_C`` with ``rasterize_gaussians``,
``rasterize_gaussians_backward`` and ``mark_visible`` (signatures of rasterize_points.h:18-77; argument order of
the *definition* rasterize_points.cu:132-161).
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from .. import _lib as L

NUM_CHANNELS = 1  # reference config.h:15
NUM_ALL_MAP = 4   # reference config.h:16


def _f32c(t, name):
    """contiguous float32 GPU tensor (or the empty-tensor placeholder untouched)."""
    if t is None or t.numel() == 0:
        return t
    L.require_gpu_tensor(t, name)
    if t.dtype != torch.float32:
        t = t.float()
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


# Per-call options of the operator API (include/curvegs.h, CGS_OPT_*): they ride in the `debug` argument of the reference's
# signatures -- bit 0 is the reference's bool -- via GaussianRasterizationSettings.options.  Nothing process-wide.
OPT_NO_TILE_CULLING = 0x100    # forward: bin every tile of the 3-sigma rect like the reference (num_rendered, lists bit-identical)
OPT_GENERAL_BACKWARD = 0x200   # backward: never the unit-colour kernel


def _dbg(debug):
    """bool (the reference's flag) or an int bit set -> the C ABI's int."""
    return int(debug) if isinstance(debug, int) and not isinstance(debug, bool) else int(bool(debug))


class _Ext:
    """Stand-in for the pybind module ``diff_cur_rasterization._C`` (reference ext.cpp:15-19)."""

    @staticmethod
    def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                            all_map, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                            campos, prefiltered, antialiasing, render_geo, debug):
        if means3D.ndimension() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:60-62
        lib = L.load()
        L.require_gpu_tensor(means3D, "means3D")
        dev = means3D.device
        P, H, W = means3D.size(0), int(image_height), int(image_width)
        with L.device_guard(dev):
            means3D = _f32c(means3D, "means3D")
            colors, opacity, scales, rotations = _f32c(colors, "colors"), _f32c(opacity, "opacity"), _f32c(scales, "scales"), _f32c(rotations, "rotations")
            cov3D_precomp, all_map, sh = _f32c(cov3D_precomp, "cov3D_precomp"), _f32c(all_map, "all_map"), _f32c(sh, "sh")
            background, viewmatrix = _f32c(background, "background"), _f32c(viewmatrix, "viewmatrix")
            projmatrix, campos = _f32c(projmatrix, "projmatrix"), _f32c(campos, "campos")
            fopt = dict(dtype=torch.float32, device=dev)
            out_color = torch.empty((NUM_CHANNELS, H, W), **fopt)
            out_invdepth = torch.empty((1, H, W), **fopt)
            out_all_map = torch.empty((NUM_ALL_MAP, H, W), **fopt)
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
            bufs = {}

            def make_alloc(key):
                def alloc(_user, nbytes):
                    t = torch.empty((int(nbytes),), dtype=torch.uint8, device=dev)
                    bufs[key] = t
                    return t.data_ptr()
                return L.ALLOC_FN(alloc)

            a_geom, a_bin, a_img = make_alloc("geom"), make_alloc("bin"), make_alloc("img")
            M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0
            stream = L.raw_stream(dev)
            rendered = lib.cgs_rasterize_forward(
                a_geom, None, a_bin, None, a_img, None, P, int(degree), int(M), L.ptr(background), W, H,
                L.ptr(means3D), L.ptr(sh), L.ptr(colors), L.ptr(opacity), L.ptr(scales), float(scale_modifier),
                L.ptr(rotations), L.ptr(cov3D_precomp), L.ptr(all_map), L.ptr(viewmatrix), L.ptr(projmatrix),
                L.ptr(campos), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), L.ptr(out_color),
                L.ptr(out_invdepth), L.ptr(out_all_map), int(bool(antialiasing)), int(bool(render_geo)),
                L.ptr(radii) if P > 0 else None, _dbg(debug), stream)
            L.check(rendered, "cgs_rasterize_forward")
            empty = torch.empty((0,), dtype=torch.uint8, device=dev)
        return (int(rendered), out_color, radii, bufs.get("geom", empty), bufs.get("bin", empty),
                bufs.get("img", empty), out_invdepth, out_all_map)

    @staticmethod
    def rasterize_gaussians_static(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                                   all_map, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh,
                                   degree, campos, prefiltered, antialiasing, render_geo, debug, bucket_capacity):
        """Extension (no reference counterpart): the sync-free forward, cgs_rasterize_forward_static.  Same returns as
        ``rasterize_gaussians`` except that num_rendered is the constant 1 (nothing is read back; the real count and
        the overflow flag are the status words of the image buffer, see ``forward_status``).  Capturable in a HIP
        graph: every buffer is a plain torch allocation made on the current stream."""
        lib = L.load()
        L.require_gpu_tensor(means3D, "means3D")
        dev = means3D.device
        P, H, W = means3D.size(0), int(image_height), int(image_width)
        if P == 0:
            raise L.CurveGSError("rasterize_gaussians_static: P == 0 (use rasterize_gaussians)")
        means3D = _f32c(means3D, "means3D")
        colors, opacity, scales, rotations = _f32c(colors, "colors"), _f32c(opacity, "opacity"), _f32c(scales, "scales"), _f32c(rotations, "rotations")
        cov3D_precomp, all_map, sh = _f32c(cov3D_precomp, "cov3D_precomp"), _f32c(all_map, "all_map"), _f32c(sh, "sh")
        fopt = dict(dtype=torch.float32, device=dev)
        out_color = torch.empty((NUM_CHANNELS, H, W), **fopt)
        out_invdepth = torch.empty((1, H, W), **fopt)
        out_all_map = torch.empty((NUM_ALL_MAP, H, W), **fopt)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        cap = int(bucket_capacity)
        nbin = int(lib.cgs_binning_bytes(cap * tiles))
        geom = torch.empty((int(lib.cgs_geometry_bytes(P)),), dtype=torch.uint8, device=dev)
        binb = torch.empty((nbin,), dtype=torch.uint8, device=dev)
        img = torch.empty((int(lib.cgs_image_bytes(W, H)),), dtype=torch.uint8, device=dev)
        M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0
        rc = lib.cgs_rasterize_forward_static(
            L.ptr(geom), L.ptr(binb), nbin, L.ptr(img), cap, P, int(degree), int(M), L.ptr(background), W, H,
            L.ptr(means3D), L.ptr(sh), L.ptr(colors), L.ptr(opacity), L.ptr(scales), float(scale_modifier),
            L.ptr(rotations), L.ptr(cov3D_precomp), L.ptr(all_map), L.ptr(viewmatrix), L.ptr(projmatrix),
            L.ptr(campos), float(tan_fovx), float(tan_fovy), L.ptr(out_color), L.ptr(out_invdepth), L.ptr(out_all_map),
            int(bool(antialiasing)), int(bool(render_geo)), L.ptr(radii), L.raw_stream(dev))
        L.check(rc, "cgs_rasterize_forward_static")
        return (1, out_color, radii, geom, binb, img, out_invdepth, out_all_map)

    @staticmethod
    def forward_status(imageBuffer, image_height, image_width):
        """int32 view of the status words of an image buffer (device tensor, valid in stream order):
        [2] = bucket-overflow flag, [4 + 2k] / [5 + 2k] = partial sums / maxima of the tile list lengths."""
        lib = L.load()
        off = int(lib.cgs_image_status_offset(int(image_width), int(image_height)))
        n = int(lib.cgs_status_words())
        return imageBuffer[off:off + 4 * n].view(torch.int32)

    @staticmethod
    def rasterize_gaussians_backward(background, all_map_pixels, means3D, radii, colors, all_maps, opacities, scales,
                                     rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                                     tan_fovy, dL_dout_color, dL_dout_invdepth, dL_dout_all_map, sh, degree, campos,
                                     geomBuffer, R, binningBuffer, imageBuffer, antialiasing, render_geo, debug,
                                     need_color_grad=True):
        lib = L.load()
        L.require_gpu_tensor(means3D, "means3D")
        dev = means3D.device
        P = means3D.size(0)
        H, W = dL_dout_color.size(1), dL_dout_color.size(2)
        with L.device_guard(dev):
            means3D = _f32c(means3D, "means3D")
            colors, opacities, scales, rotations = _f32c(colors, "colors"), _f32c(opacities, "opacities"), _f32c(scales, "scales"), _f32c(rotations, "rotations")
            cov3D_precomp, all_maps, sh = _f32c(cov3D_precomp, "cov3D_precomp"), _f32c(all_maps, "all_maps"), _f32c(sh, "sh")
            dL_dout_color = _f32c(dL_dout_color, "dL_dout_color")
            dL_dout_invdepth = _f32c(dL_dout_invdepth, "dL_dout_invdepth")
            dL_dout_all_map = _f32c(dL_dout_all_map, "dL_dout_all_map")
            M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0
            has_invd = dL_dout_invdepth is not None and dL_dout_invdepth.numel() != 0
            fopt = dict(dtype=torch.float32, device=dev)
            # The reference zero-fills 11 gradient tensors (rasterize_points.cu:173-183); libcurvegs writes every
            # output for all P splats, so one uninitialised allocation is carved into views.
            acc = torch.empty((P, 14), **fopt)
            flat = acc.view(-1)
            o = 0
            def take(n, shape):
                nonlocal o
                v = flat[o:o + n].view(shape)
                o += n
                return v
            dL_dconic = take(4 * P, (P, 2, 2))
            dL_dall_map = take(4 * P, (P, NUM_ALL_MAP))
            dL_dmeans2D = take(3 * P, (P, 3))
            dL_dcolors = take(P, (P, NUM_CHANNELS))
            dL_dopacity = take(P, (P, 1))
            dL_dinvdepths = take(P, (P, 1))
            wr = torch.empty((P, 16), **fopt)
            wflat = wr.view(-1)
            dL_drotations = wflat[0:4 * P].view(P, 4)
            dL_dmeans3D = wflat[4 * P:7 * P].view(P, 3)
            dL_dcov3D = wflat[7 * P:13 * P].view(P, 6)
            dL_dscales = wflat[13 * P:16 * P].view(P, 3)
            has_scales = scales is not None and scales.numel() != 0
            has_amap_g = dL_dout_all_map is not None and dL_dout_all_map.numel() != 0
            # extension over the reference: skip the colour-gradient accumulation when nobody consumes it
            want_col = bool(need_color_grad) or M > 0 or has_invd or (bool(render_geo) and has_amap_g)
            if not has_scales:
                dL_dscales.zero_()
                dL_drotations.zero_()
            # reference shape [P,M,3]; only the first P*M floats are written (quirk 16)
            dL_dsh = torch.zeros((P, M, 3), **fopt) if M > 0 else torch.empty((P, 0, 3), **fopt)
            stream = L.raw_stream(dev)
            if P != 0:
                rc = lib.cgs_rasterize_backward(
                    P, int(degree), int(M), int(R), L.ptr(background), W, H, L.ptr(means3D), L.ptr(sh), L.ptr(colors),
                    L.ptr(all_maps), L.ptr(opacities), L.ptr(scales), float(scale_modifier), L.ptr(rotations),
                    L.ptr(cov3D_precomp), L.ptr(viewmatrix), L.ptr(projmatrix), L.ptr(campos), float(tan_fovx),
                    float(tan_fovy), L.ptr(radii), L.ptr(geomBuffer), L.ptr(binningBuffer), L.ptr(imageBuffer),
                    L.ptr(dL_dout_color), L.ptr(dL_dout_invdepth) if has_invd else None, L.ptr(dL_dout_all_map),
                    L.ptr(dL_dmeans2D), L.ptr(dL_dconic), L.ptr(dL_dopacity), L.ptr(dL_dcolors) if want_col else None,
                    L.ptr(dL_dinvdepths) if has_invd else None, L.ptr(dL_dmeans3D), L.ptr(dL_dcov3D),
                    L.ptr(dL_dsh) if M > 0 else None, L.ptr(dL_dscales) if has_scales else None,
                    L.ptr(dL_drotations) if has_scales else None, L.ptr(dL_dall_map), int(bool(antialiasing)),
                    int(bool(render_geo)), _dbg(debug), stream)
                L.check(rc, "cgs_rasterize_backward")
        # need_color_grad=False (extension, training configuration): the colour gradient is not computed -> None
        return (dL_dmeans2D, dL_dcolors if want_col else None, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
                dL_drotations, dL_dall_map)

    @staticmethod
    def mark_visible(means3D, viewmatrix, projmatrix):
        lib = L.load()
        L.require_gpu_tensor(means3D, "means3D")
        dev = means3D.device
        P = means3D.size(0)
        with L.device_guard(dev):
            means3D = _f32c(means3D, "means3D")
            present = torch.zeros((P,), dtype=torch.bool, device=dev)
            if P != 0:
                rc = lib.cgs_mark_visible(P, L.ptr(means3D), L.ptr(_f32c(viewmatrix, "viewmatrix")),
                                          L.ptr(_f32c(projmatrix, "projmatrix")), L.ptr(present),
                                          L.raw_stream(dev))
                L.check(rc, "cgs_mark_visible")
        return present


class _ExtProxy:
    """``diff_cur_rasterization._C``: the compiled host shim (csrc/torch_shim.cpp, the counterpart of the reference's pybind
    module, ext.cpp:15-19) by default, the ctypes bindings above with CGS_TORCH_SHIM=0.  Resolved on first use."""
    _impl = None

    def __getattr__(self, name):
        impl = _ExtProxy._impl
        if impl is None:
            impl = _ExtProxy._impl = L.shim() if L.use_shim() else _Ext()
        return getattr(impl, name)


_C = _ExtProxy()


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, all_map,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, all_map, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, all_maps,
                raster_settings):
        rs = raster_settings
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                all_maps, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                rs.sh_degree, rs.campos, rs.prefiltered, rs.antialiasing, rs.render_geo,
                _dbg(rs.debug) | int(getattr(rs, "options", 0)))
        cap = getattr(rs, "static_bucket_cap", 0)
        if cap:   # extension: sync-free forward with caller-chosen bucket capacity (stream-ordered / graph capture)
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, invdepths, out_all_map = \
                _C.rasterize_gaussians_static(*args, cap)
            if rs.status_sink is not None:
                rs.status_sink.append(_C.forward_status(imgBuffer, rs.image_height, rs.image_width))
        else:
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, invdepths, out_all_map = \
                _C.rasterize_gaussians(*args)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        # out_all_map is NOT saved: the reference passes it to the backward kernel which never reads it (quirk 21)
        ctx.save_for_backward(colors_precomp, all_maps, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
                              opacities, geomBuffer, binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        # unused upstream grads arrive as None instead of materialised zero images (results are identical: the
        # reference adds exact zeros for them)
        ctx.set_materialize_grads(False)
        return color, radii, invdepths, out_all_map

    @staticmethod
    def backward(ctx, grad_out_color, _, grad_out_depth, grad_out_all_map):
        rs = ctx.raster_settings
        (colors_precomp, all_maps, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities, geomBuffer,
         binningBuffer, imgBuffer) = ctx.saved_tensors
        if grad_out_color is None:
            grad_out_color = torch.zeros((NUM_CHANNELS, rs.image_height, rs.image_width), dtype=torch.float32,
                                         device=means3D.device)
        empty = torch.empty((0,), dtype=torch.float32, device=means3D.device)
        args = (rs.bg, empty, means3D, radii, colors_precomp, all_maps, opacities, scales, rotations,
                rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
                grad_out_color, grad_out_depth if grad_out_depth is not None else empty,
                grad_out_all_map if grad_out_all_map is not None else empty, sh, rs.sh_degree, rs.campos, geomBuffer,
                ctx.num_rendered, binningBuffer, imgBuffer, rs.antialiasing, rs.render_geo,
                _dbg(rs.debug) | int(getattr(rs, "options", 0)))
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations, grad_all_map) = _C.rasterize_gaussians_backward(*args, need_color_grad=ctx.needs_input_grad[3])
        grads = (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                 grad_cov3Ds_precomp, grad_all_map)
        # the reference returns all nine unconditionally (:138-149); autograd rejects gradients for non-tensor inputs
        # (e.g. means2D=None), so inputs that do not require grad get None
        return tuple(g if need else None for g, need in zip(grads, ctx.needs_input_grad[:9])) + (None,)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    antialiasing: bool
    render_geo: bool
    # extensions (defaults = the reference's behaviour): static_bucket_cap > 0 selects the sync-free forward
    # (cgs_rasterize_forward_static) with that many slots per tile; its status-word tensor is appended to status_sink
    static_bucket_cap: int = 0
    status_sink: object = None
    # per-call option bits (OPT_NO_TILE_CULLING | OPT_GENERAL_BACKWARD): parity tests and A/B measurements
    options: int = 0


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, all_map=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])
        if all_map is None:
            all_map = torch.Tensor([])
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   all_map, rs)
