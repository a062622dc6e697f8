// Host shim between PyTorch and the C ABI of libcurvegs.so (include/curvegs.h): the compiled counterpart of the reference's
// torch extension (submodules/diff-cur-rasterization/ext.cpp:15-19, rasterize_points.cu:35-260).  It owns no kernels and no
// algorithm: tensor checks, output / scratch allocation through the caching allocator (the reference's resize callbacks,
// rasterize_points.cu:27-33), the current HIP stream, and ONE call into the C ABI per operation -- so an eager caller pays a
// pybind call instead of ~40 ctypes conversions and a dozen Python-level torch.empty per view.
//
//   rasterize_gaussians / rasterize_gaussians_backward / mark_visible      same argument lists as the reference's pybind module
//   rasterize_gaussians_static, forward_status                             the sync-free forward (extension)
//   view_forward / view_wait / view_abandon / view_backward                the fused per-view path behind render()
//
// Plain C++17, built with g++ against libtorch + libcurvegs.so (csrc/Makefile, target torch_shim).
#include <torch/extension.h>
// (PyTorch-ROCm presents its HIP devices under the device type "cuda": the guard / stream accessors that accept it)
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <string>
#include <tuple>
#include <vector>

#include "../../include/curvegs.h"

namespace {

using at::Tensor;
constexpr int NUM_CHANNELS = 1, NUM_ALL_MAP = 4;   // config.h of the reference

[[noreturn]] void raise_cgs(const std::string& msg) {
    // curve_gaussian_amd._lib.CurveGSError (a RuntimeError subclass): the exception type of every other binding
    py::object cls = py::module_::import("curve_gaussian_amd._lib").attr("CurveGSError");
    PyErr_SetString(cls.ptr(), msg.c_str());
    throw py::error_already_set();
}
template <typename T>
T check(T rc, const char* what) {
    if (rc < 0) raise_cgs(std::string(what) + " failed (status " + std::to_string((long long)rc) + "): " + cgs_last_error());
    return rc;
}
void require_gpu(const Tensor& t, const char* name) {
    if (!t.is_cuda())
        raise_cgs(std::string(name) + " must be a GPU tensor (got device " + t.device().str() + "); libcurvegs has no CPU path");
}
bool has(const Tensor& t) { return t.defined() && t.numel() != 0; }
bool has(const c10::optional<Tensor>& t) { return t.has_value() && has(*t); }
// contiguous float32 GPU tensor, 16-byte aligned (or the empty placeholder untouched)
Tensor f32c(const Tensor& t, const char* name) {
    if (!has(t)) return t;
    require_gpu(t, name);
    Tensor r = t.scalar_type() == at::kFloat ? t : t.to(at::kFloat);
    r = r.contiguous();
    if (reinterpret_cast<uintptr_t>(r.data_ptr()) % 16) r = r.clone();
    return r;
}
Tensor f32c(const c10::optional<Tensor>& t, const char* name) { return t.has_value() ? f32c(*t, name) : Tensor(); }
const float* fp(const Tensor& t) { return has(t) ? t.data_ptr<float>() : nullptr; }
float* fpm(const Tensor& t) { return has(t) ? t.data_ptr<float>() : nullptr; }
void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

struct AllocSlot { Tensor t; c10::TensorOptions opt; };
void* alloc_cb(void* user, size_t bytes) {   // the reference's resizeFunctional (rasterize_points.cu:27-33)
    auto* s = static_cast<AllocSlot*>(user);
    s->t = at::empty({(int64_t)bytes}, s->opt);
    return s->t.data_ptr();
}

// ------------------------------------------------------------------------------------------------ operator API
// RasterizeGaussiansCUDA (rasterize_points.cu:35-130): same argument order, same 8-tuple
std::tuple<int64_t, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_gaussians(
    const Tensor& background, const Tensor& means3D, const Tensor& colors, const Tensor& opacity, const Tensor& scales,
    const Tensor& rotations, double scale_modifier, const Tensor& cov3D_precomp, const Tensor& all_map, const Tensor& viewmatrix,
    const Tensor& projmatrix, double tan_fovx, double tan_fovy, int64_t image_height, int64_t image_width, const Tensor& sh,
    int64_t degree, const Tensor& campos, bool prefiltered, bool antialiasing, bool render_geo, int64_t debug) {   // debug: the
    // reference's bool, or a CGS_OPT_* bit set (include/curvegs.h) -- per-call options ride in the same argument
    if (means3D.dim() != 2 || means3D.size(1) != 3) throw std::runtime_error("means3D must have dimensions (num_points, 3)");   // :60-62
    require_gpu(means3D, "means3D");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());
    const int P = (int)means3D.size(0), H = (int)image_height, W = (int)image_width;
    const Tensor m3 = f32c(means3D, "means3D"), col = f32c(colors, "colors"), op = f32c(opacity, "opacity"), sc = f32c(scales, "scales"),
                 rot = f32c(rotations, "rotations"), cov = f32c(cov3D_precomp, "cov3D_precomp"), am = f32c(all_map, "all_map"),
                 shs = f32c(sh, "sh"), bg = f32c(background, "background"), vm = f32c(viewmatrix, "viewmatrix"),
                 pm = f32c(projmatrix, "projmatrix"), cp = f32c(campos, "campos");
    const auto fopt = m3.options().dtype(at::kFloat);
    Tensor outs = at::empty({NUM_CHANNELS + 1 + NUM_ALL_MAP, H, W}, fopt);   // one allocation, three views
    Tensor out_color = outs.narrow(0, 0, NUM_CHANNELS), out_invdepth = outs.narrow(0, NUM_CHANNELS, 1),
           out_all_map = outs.narrow(0, NUM_CHANNELS + 1, NUM_ALL_MAP);
    Tensor radii = at::empty({P}, m3.options().dtype(at::kInt));
    AllocSlot geom{Tensor(), m3.options().dtype(at::kByte)}, bin = geom, img = geom;
    const int M = has(shs) ? (int)shs.size(1) : 0;
    const int64_t rendered = check(cgs_rasterize_forward(
        alloc_cb, &geom, alloc_cb, &bin, alloc_cb, &img, P, (int)degree, M, fp(bg), W, H, fp(m3), fp(shs), fp(col), fp(op), fp(sc),
        (float)scale_modifier, fp(rot), fp(cov), fp(am), fp(vm), fp(pm), fp(cp), (float)tan_fovx, (float)tan_fovy, prefiltered ? 1 : 0,
        out_color.data_ptr<float>(), out_invdepth.data_ptr<float>(), out_all_map.data_ptr<float>(), antialiasing ? 1 : 0,
        render_geo ? 1 : 0, P > 0 ? radii.data_ptr<int>() : nullptr, (int)debug, stream_of(m3)), "cgs_rasterize_forward");
    const Tensor empty = at::empty({0}, m3.options().dtype(at::kByte));
    return {rendered, out_color, radii, geom.t.defined() ? geom.t : empty, bin.t.defined() ? bin.t : empty,
            img.t.defined() ? img.t : empty, out_invdepth, out_all_map};
}

std::tuple<int64_t, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_gaussians_static(
    const Tensor& background, const Tensor& means3D, const Tensor& colors, const Tensor& opacity, const Tensor& scales,
    const Tensor& rotations, double scale_modifier, const Tensor& cov3D_precomp, const Tensor& all_map, const Tensor& viewmatrix,
    const Tensor& projmatrix, double tan_fovx, double tan_fovy, int64_t image_height, int64_t image_width, const Tensor& sh,
    int64_t degree, const Tensor& campos, bool /*prefiltered*/, bool antialiasing, bool render_geo, bool /*debug*/,
    int64_t bucket_capacity) {
    require_gpu(means3D, "means3D");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());
    const int P = (int)means3D.size(0), H = (int)image_height, W = (int)image_width;
    if (P == 0) raise_cgs("rasterize_gaussians_static: P == 0 (use rasterize_gaussians)");
    const Tensor m3 = f32c(means3D, "means3D"), col = f32c(colors, "colors"), op = f32c(opacity, "opacity"), sc = f32c(scales, "scales"),
                 rot = f32c(rotations, "rotations"), cov = f32c(cov3D_precomp, "cov3D_precomp"), am = f32c(all_map, "all_map"),
                 shs = f32c(sh, "sh"), bg = f32c(background, "background"), vm = f32c(viewmatrix, "viewmatrix"),
                 pm = f32c(projmatrix, "projmatrix"), cp = f32c(campos, "campos");
    const auto fopt = m3.options().dtype(at::kFloat);
    const auto bopt = m3.options().dtype(at::kByte);
    Tensor outs = at::empty({NUM_CHANNELS + 1 + NUM_ALL_MAP, H, W}, fopt);
    Tensor out_color = outs.narrow(0, 0, NUM_CHANNELS), out_invdepth = outs.narrow(0, NUM_CHANNELS, 1),
           out_all_map = outs.narrow(0, NUM_CHANNELS + 1, NUM_ALL_MAP);
    Tensor radii = at::empty({P}, m3.options().dtype(at::kInt));
    const int64_t tiles = (int64_t)((W + 15) / 16) * ((H + 15) / 16);
    const size_t nbin = cgs_binning_bytes((int64_t)bucket_capacity * tiles);
    Tensor geom = at::empty({(int64_t)cgs_geometry_bytes(P)}, bopt), binb = at::empty({(int64_t)nbin}, bopt),
           img = at::empty({(int64_t)cgs_image_bytes(W, H)}, bopt);
    const int M = has(shs) ? (int)shs.size(1) : 0;
    check(cgs_rasterize_forward_static(geom.data_ptr(), binb.data_ptr(), nbin, img.data_ptr(), (uint32_t)bucket_capacity, P,
                                       (int)degree, M, fp(bg), W, H, fp(m3), fp(shs), fp(col), fp(op), fp(sc),
                                       (float)scale_modifier, fp(rot), fp(cov), fp(am), fp(vm), fp(pm), fp(cp),
                                       (float)tan_fovx, (float)tan_fovy, out_color.data_ptr<float>(), out_invdepth.data_ptr<float>(),
                                       out_all_map.data_ptr<float>(), antialiasing ? 1 : 0, render_geo ? 1 : 0, radii.data_ptr<int>(),
                                       stream_of(m3)), "cgs_rasterize_forward_static");
    return {1, out_color, radii, geom, binb, img, out_invdepth, out_all_map};
}

Tensor forward_status(const Tensor& imageBuffer, int64_t image_height, int64_t image_width) {
    const int64_t off = (int64_t)cgs_image_status_offset((int)image_width, (int)image_height);
    const int64_t n = cgs_status_words();
    return imageBuffer.narrow(0, off, 4 * n).view(at::kInt);
}

// RasterizeGaussiansBackwardCUDA (rasterize_points.cu:132-239): same argument order (+ need_color_grad, extension), same 9-tuple
std::tuple<Tensor, c10::optional<Tensor>, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> rasterize_gaussians_backward(
    const Tensor& background, const Tensor& /*all_map_pixels*/, const Tensor& means3D, const Tensor& radii, const Tensor& colors,
    const Tensor& all_maps, const Tensor& opacities, const Tensor& scales, const Tensor& rotations, double scale_modifier,
    const Tensor& cov3D_precomp, const Tensor& viewmatrix, const Tensor& projmatrix, double tan_fovx, double tan_fovy,
    const Tensor& dL_dout_color, const Tensor& dL_dout_invdepth, const Tensor& dL_dout_all_map, const Tensor& sh, int64_t degree,
    const Tensor& campos, const Tensor& geomBuffer, int64_t R, const Tensor& binningBuffer, const Tensor& imageBuffer,
    bool antialiasing, bool render_geo, int64_t debug, bool need_color_grad) {
    require_gpu(means3D, "means3D");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());
    const int64_t P = means3D.size(0);
    const int H = (int)dL_dout_color.size(1), W = (int)dL_dout_color.size(2);
    const Tensor m3 = f32c(means3D, "means3D"), col = f32c(colors, "colors"), op = f32c(opacities, "opacities"),
                 sc = f32c(scales, "scales"), rot = f32c(rotations, "rotations"), cov = f32c(cov3D_precomp, "cov3D_precomp"),
                 am = f32c(all_maps, "all_maps"), shs = f32c(sh, "sh"), g_col = f32c(dL_dout_color, "dL_dout_color"),
                 g_inv = f32c(dL_dout_invdepth, "dL_dout_invdepth"), g_map = f32c(dL_dout_all_map, "dL_dout_all_map"),
                 bg = f32c(background, "background"), vm = f32c(viewmatrix, "viewmatrix"), pm = f32c(projmatrix, "projmatrix"),
                 cpos = f32c(campos, "campos");
    const int M = has(shs) ? (int)shs.size(1) : 0;
    const bool has_invd = has(g_inv), has_scales = has(sc), has_amap_g = has(g_map);
    const auto fopt = m3.options().dtype(at::kFloat);
    // The reference zero-fills 11 gradient tensors (rasterize_points.cu:173-183); libcurvegs writes every output for all P
    // splats, so two uninitialised allocations are carved into views.
    Tensor acc = at::empty({P * 14}, fopt), wr = at::empty({P * 16}, fopt);
    int64_t o = 0;
    auto take = [&](Tensor& buf, int64_t n, at::IntArrayRef shape) { Tensor v = buf.narrow(0, o, n).view(shape); o += n; return v; };
    Tensor dL_dconic = take(acc, 4 * P, {P, 2, 2}), dL_dall_map = take(acc, 4 * P, {P, NUM_ALL_MAP}), dL_dmeans2D = take(acc, 3 * P, {P, 3}),
           dL_dcolors = take(acc, P, {P, NUM_CHANNELS}), dL_dopacity = take(acc, P, {P, 1}), dL_dinvdepths = take(acc, P, {P, 1});
    o = 0;
    Tensor dL_drotations = take(wr, 4 * P, {P, 4}), dL_dmeans3D = take(wr, 3 * P, {P, 3}), dL_dcov3D = take(wr, 6 * P, {P, 6}),
           dL_dscales = take(wr, 3 * P, {P, 3});
    // extension over the reference: skip the colour-gradient accumulation when nobody consumes it
    const bool want_col = need_color_grad || M > 0 || has_invd || (render_geo && has_amap_g);
    if (!has_scales) {
        dL_dscales.zero_();
        dL_drotations.zero_();
    }
    // reference shape [P,M,3]; only the first P*M floats are written (quirk 16)
    Tensor dL_dsh = M > 0 ? at::zeros({P, M, 3}, fopt) : at::empty({P, 0, 3}, fopt);
    if (P != 0)
        check(cgs_rasterize_backward(
            (int)P, (int)degree, M, R, fp(bg), W, H, fp(m3), fp(shs), fp(col), fp(am), fp(op), fp(sc), (float)scale_modifier,
            fp(rot), fp(cov), fp(vm), fp(pm), fp(cpos), (float)tan_fovx, (float)tan_fovy, radii.data_ptr<int>(),
            geomBuffer.data_ptr(), binningBuffer.numel() ? binningBuffer.data_ptr() : nullptr, imageBuffer.data_ptr(), fp(g_col),
            has_invd ? fp(g_inv) : nullptr, fp(g_map), fpm(dL_dmeans2D), fpm(dL_dconic), fpm(dL_dopacity),
            want_col ? fpm(dL_dcolors) : nullptr, has_invd ? fpm(dL_dinvdepths) : nullptr, fpm(dL_dmeans3D), fpm(dL_dcov3D),
            M > 0 ? fpm(dL_dsh) : nullptr, has_scales ? fpm(dL_dscales) : nullptr, has_scales ? fpm(dL_drotations) : nullptr,
            fpm(dL_dall_map), antialiasing ? 1 : 0, render_geo ? 1 : 0, (int)debug, stream_of(m3)), "cgs_rasterize_backward");
    // need_color_grad=False (extension, training configuration): the colour gradient is not computed -> None
    return {dL_dmeans2D, want_col ? c10::optional<Tensor>(dL_dcolors) : c10::nullopt, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh,
            dL_dscales, dL_drotations, dL_dall_map};
}

Tensor mark_visible(const Tensor& means3D, const Tensor& viewmatrix, const Tensor& projmatrix) {   // rasterize_points.cu:241-260
    require_gpu(means3D, "means3D");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(means3D.device());
    const int P = (int)means3D.size(0);
    const Tensor m3 = f32c(means3D, "means3D"), vm = f32c(viewmatrix, "viewmatrix"), pm = f32c(projmatrix, "projmatrix");
    Tensor present = at::zeros({P}, m3.options().dtype(at::kBool));
    if (P != 0)
        check(cgs_mark_visible(P, fp(m3), fp(vm), fp(pm), (uint8_t*)present.data_ptr(), stream_of(m3)), "cgs_mark_visible");
    return present;
}

// ------------------------------------------------------------------------------------------------ fused per-view path
// Forward half of ops/view_render.py::_ViewRender: detached float32 copies of the parameters, every buffer of the view, the
// begin half of the checked forward (or the sync-free forward when static_cap > 0) and render()'s epilogue, in one call.
// -> (color, invdepth, all_map, radii, rend_dir | empty, color_raw | None, saved tensors..., handle, cap)
struct ViewFwd {
    Tensor color_out, invd, amap, radii, rend_dir, color_raw;                               // color_raw: undefined unless clamp
    Tensor cp, w, ol, mk, geom, binb, img, norms, bgc, view, proj, cpos;                   // what the backward needs (mk: undefined = no mask)
    int handle = -1;
};
ViewFwd view_forward_core(const Tensor& curve_points, const Tensor& width, const Tensor& opacity_logit,
                          const c10::optional<Tensor>& mask_logit, const c10::optional<Tensor>& is_bezier_u8, const Tensor& coef,
                          int64_t m, double mask_thr, const Tensor& bg, const Tensor& viewmatrix, const Tensor& projmatrix,
                          const Tensor& campos, double tanx, double tany, int64_t H, int64_t W, int64_t cap, bool sync_free, bool clamp,
                          bool want_dir, double eps) {
    require_gpu(curve_points, "curve_points");
    require_gpu(bg, "bg_color");   // "Background tensor (bg_color) must be on GPU!" (gaussian_renderer/__init__.py:23)
    require_gpu(viewmatrix, "viewpoint_camera.world_view_transform");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(curve_points.device());
    const Tensor cp = f32c(curve_points.detach(), "curve_points"), w = f32c(width.detach(), "width"),
                 ol = f32c(opacity_logit.detach(), "opacity"), mk = mask_logit.has_value() ? f32c(mask_logit->detach(), "mask") : Tensor(),
                 view = f32c(viewmatrix.detach(), "world_view_transform"), proj = f32c(projmatrix.detach(), "full_proj_transform"),
                 cpos = f32c(campos.detach(), "camera_center"), bgc = f32c(bg.detach(), "bg_color");
    const int B = (int)cp.size(0), P = B * (int)m;
    const auto fopt = cp.options().dtype(at::kFloat);
    const auto bopt = cp.options().dtype(at::kByte);
    const int64_t tiles = ((W + 15) / 16) * ((H + 15) / 16);
    const size_t nbin = cgs_binning_bytes(cap * tiles);
    Tensor norms = at::empty({384}, cp.options().dtype(at::kDouble));
    Tensor geom = at::empty({(int64_t)cgs_geometry_bytes(P)}, bopt), img = at::empty({(int64_t)cgs_image_bytes((int)W, (int)H)}, bopt),
           binb = at::empty({(int64_t)nbin}, bopt);
    Tensor outs = at::empty({6, H, W}, fopt);
    Tensor color = outs.narrow(0, 0, 1), invd = outs.narrow(0, 1, 1), amap = outs.narrow(0, 2, 4);
    Tensor radii = at::empty({P}, cp.options().dtype(at::kInt));
    void* st = stream_of(cp);
    const uint8_t* isb = has(is_bezier_u8) ? (const uint8_t*)is_bezier_u8->data_ptr() : nullptr;
    // render()'s epilogue (gaussian_renderer/__init__.py:138-145) is written by the forward compositor itself
    Tensor color_out = color, rend_dir = at::empty({0}, fopt);
    if (clamp) color_out = at::empty({1, H, W}, fopt);
    if (want_dir) rend_dir = at::empty({3, H, W}, fopt);
    const int handle = check(cgs_view_forward_render(sync_free ? 0 : 1, B, (int)m, fp(cp), fp(w), isb, fp(coef), (float)eps,
                                                     norms.data_ptr<double>(), fp(ol), fp(mk), (float)mask_thr, geom.data_ptr(),
                                                     binb.data_ptr(), nbin, img.data_ptr(), (uint32_t)cap, fp(bgc), (int)W, (int)H,
                                                     fp(view), fp(proj), fp(cpos), (float)tanx, (float)tany, color.data_ptr<float>(),
                                                     invd.data_ptr<float>(), amap.data_ptr<float>(), radii.data_ptr<int>(),
                                                     clamp ? color_out.data_ptr<float>() : nullptr,
                                                     want_dir ? rend_dir.data_ptr<float>() : nullptr, st),
                             "cgs_view_forward_render");
    ViewFwd f;
    f.color_out = color_out; f.invd = invd; f.amap = amap; f.radii = radii; f.rend_dir = rend_dir;
    if (clamp) f.color_raw = color;
    f.cp = cp; f.w = w; f.ol = ol; f.mk = mk; f.geom = geom; f.binb = binb; f.img = img; f.norms = norms; f.bgc = bgc;
    f.view = view; f.proj = proj; f.cpos = cpos;
    f.handle = sync_free ? -1 : handle;
    return f;
}
py::tuple view_forward(const Tensor& curve_points, const Tensor& width, const Tensor& opacity_logit,
                       const c10::optional<Tensor>& mask_logit, const c10::optional<Tensor>& is_bezier_u8, const Tensor& coef,
                       int64_t m, double mask_thr, const Tensor& bg, const Tensor& viewmatrix, const Tensor& projmatrix,
                       const Tensor& campos, double tanx, double tany, int64_t H, int64_t W, int64_t cap, bool sync_free, bool clamp,
                       bool want_dir, double eps) {
    const ViewFwd f = view_forward_core(curve_points, width, opacity_logit, mask_logit, is_bezier_u8, coef, m, mask_thr, bg, viewmatrix,
                                        projmatrix, campos, tanx, tany, H, W, cap, sync_free, clamp, want_dir, eps);
    return py::make_tuple(f.color_out, f.invd, f.amap, f.radii, f.rend_dir, f.color_raw.defined() ? py::cast(f.color_raw) : py::none(),
                          py::make_tuple(f.cp, f.w, f.ol, f.mk.defined() ? py::cast(f.mk) : py::none(), f.geom, f.binb, f.img, f.radii,
                                         f.norms, f.bgc, f.view, f.proj, f.cpos),
                          f.handle);
}

// -> (longest tile list, n_visible); releases the handle
std::pair<int64_t, int64_t> view_wait(int64_t handle) {
    int64_t nvis = -1;
    int64_t longest;
    {
        py::gil_scoped_release nogil;   // the wait blocks on a HIP event
        longest = cgs_view_forward_wait((int)handle, &nvis);
    }
    check(longest, "cgs_view_forward_wait");
    return {longest, nvis};
}
void view_abandon(int64_t handle) { cgs_view_forward_abandon((int)handle); }

// Backward half: clamp gradient + cgs_view_backward.  -> (g_cp, g_w, g_ol, g_mk | None, g_m2d)
struct ViewBwd { Tensor g_cp, g_w, g_ol, g_mk, g_m2d; };   // undefined = None (sinks took it / no mask)
ViewBwd view_backward_core(const Tensor& cp, const Tensor& w, const Tensor& ol, const c10::optional<Tensor>& mk,
                           const c10::optional<Tensor>& is_bezier_u8, const Tensor& coef, const Tensor& geom, const Tensor& binb,
                           const Tensor& img, const Tensor& radii, const Tensor& norms, const Tensor& bgc, const Tensor& view,
                           const Tensor& proj, const Tensor& cpos, int64_t m, double mask_thr, double tanx, double tany, int64_t H,
                           int64_t W, double eps, const c10::optional<Tensor>& g_color_in, const c10::optional<Tensor>& color_raw,
                           const c10::optional<std::vector<Tensor>>& sinks, const c10::optional<Tensor>& rot_extra) {
    c10::hip::HIPGuardMasqueradingAsCUDA guard(cp.device());
    const int B = (int)cp.size(0), P = B * (int)m;
    const auto fopt = cp.options().dtype(at::kFloat);
    const bool has_mk = has(mk);
    if (sinks.has_value()) {
        // Gradient sinks (TrainStep: the optimizer's flat gradient buffer): the four curve-level gradients are ADDED to the
        // caller's tensors by the kernels themselves (CGS_VIEW_ACCUMULATE) and the node returns None for those inputs, so
        // autograd has no AccumulateGrad kernels to run for them.
        const auto& sk = *sinks;
        if (sk.size() != (has_mk ? 4u : 3u)) raise_cgs("view_backward: grad_sinks must hold one tensor per differentiable curve input");
        const int64_t want[4] = {(int64_t)B * 12, B, B, has_mk ? mk->numel() : 0};
        for (size_t i = 0; i < sk.size(); ++i) {
            require_gpu(sk[i], "grad sink");
            if (sk[i].scalar_type() != at::kFloat || !sk[i].is_contiguous() || sk[i].numel() != want[i])
                raise_cgs("view_backward: grad sink " + std::to_string(i) + " must be a contiguous float32 tensor shaped like its parameter");
        }
        Tensor g_m2d = g_color_in.has_value() ? at::empty({P, 3}, fopt) : at::zeros({P, 3}, fopt);
        if (g_color_in.has_value()) {
            void* st = stream_of(cp);
            Tensor g_color = f32c(*g_color_in, "grad of render");
            Tensor scratch = at::empty({(int64_t)cgs_view_backward_scratch_floats(B, (int)m)}, fopt);
            const uint8_t* isb = has(is_bezier_u8) ? (const uint8_t*)is_bezier_u8->data_ptr() : nullptr;
            if (has(rot_extra)) {
                // (+ a gradient with respect to the raw splat rotations, e.g. the curve-smoothness regulariser's: enters the
                // fused chain before it is pulled back to the curves; the plain entry point, no clamp mask)
                if (has(color_raw)) raise_cgs("view_backward: rot_extra and color_raw exclude each other");
                Tensor rx = f32c(*rot_extra, "rot_extra");
                if (rx.numel() != (int64_t)P * 4) raise_cgs("view_backward: rot_extra must be [P,4]");
                check(cgs_view_backward(B, (int)m, fp(cp), fp(w), isb, fp(coef), (float)eps, norms.data_ptr<double>(), fp(ol),
                                        has_mk ? fp(*mk) : nullptr, (float)mask_thr, nullptr, geom.data_ptr(), binb.data_ptr(),
                                        img.data_ptr(), fp(bgc), (int)W, (int)H, fp(view), fp(proj), fp(cpos), (float)tanx, (float)tany,
                                        radii.data_ptr<int>(), fp(g_color), fp(rx), g_m2d.data_ptr<float>(), sk[0].data_ptr<float>(),
                                        sk[1].data_ptr<float>(), sk[2].data_ptr<float>(), has_mk ? sk[3].data_ptr<float>() : nullptr,
                                        scratch.data_ptr<float>(), CGS_VIEW_ACCUMULATE, st),
                      "cgs_view_backward");
            } else
            check(cgs_view_backward_render(B, (int)m, fp(cp), fp(w), isb, fp(coef), (float)eps, norms.data_ptr<double>(), fp(ol),
                                           has_mk ? fp(*mk) : nullptr, (float)mask_thr, geom.data_ptr(), binb.data_ptr(), img.data_ptr(),
                                           fp(bgc), (int)W, (int)H, fp(view), fp(proj), fp(cpos), (float)tanx, (float)tany,
                                           radii.data_ptr<int>(), fp(g_color), has(color_raw) ? fp(*color_raw) : nullptr,
                                           g_m2d.data_ptr<float>(), sk[0].data_ptr<float>(), sk[1].data_ptr<float>(),
                                           sk[2].data_ptr<float>(), has_mk ? sk[3].data_ptr<float>() : nullptr,
                                           scratch.data_ptr<float>(), CGS_VIEW_ACCUMULATE, st),
                  "cgs_view_backward_render");
        }
        ViewBwd r;
        r.g_m2d = g_m2d;
        return r;
    }
    if (has(rot_extra)) raise_cgs("view_backward: rot_extra is served together with grad sinks only");
    // one allocation for the four curve-level gradients + the screen-space gradient, one for the scratch
    const int64_t n_curve = (int64_t)B * 14, n_mk = has_mk ? mk->numel() : 0;
    Tensor buf = g_color_in.has_value() ? at::empty({n_curve + n_mk + (int64_t)P * 3}, fopt) : at::zeros({n_curve + n_mk + (int64_t)P * 3}, fopt);
    Tensor g_cp = buf.narrow(0, 0, (int64_t)B * 12).view({B, 4, 3}), g_w = buf.narrow(0, (int64_t)B * 12, B).view({B, 1}),
           g_ol = buf.narrow(0, (int64_t)B * 13, B).view({B, 1});
    Tensor g_mk = has_mk ? buf.narrow(0, n_curve, n_mk).view(mk->sizes()) : Tensor();
    Tensor g_m2d = buf.narrow(0, n_curve + n_mk, (int64_t)P * 3).view({P, 3});
    if (g_color_in.has_value()) {
        void* st = stream_of(cp);
        Tensor g_color = f32c(*g_color_in, "grad of render");
        Tensor scratch = at::empty({(int64_t)cgs_view_backward_scratch_floats(B, (int)m)}, fopt);
        const uint8_t* isb = has(is_bezier_u8) ? (const uint8_t*)is_bezier_u8->data_ptr() : nullptr;
        // (color_raw: torch.clamp's gradient rule on the unclamped image, folded into the compositor's per-pixel load)
        check(cgs_view_backward_render(B, (int)m, fp(cp), fp(w), isb, fp(coef), (float)eps, norms.data_ptr<double>(), fp(ol),
                                       has_mk ? fp(*mk) : nullptr, (float)mask_thr, geom.data_ptr(), binb.data_ptr(), img.data_ptr(),
                                       fp(bgc), (int)W, (int)H, fp(view), fp(proj), fp(cpos), (float)tanx, (float)tany,
                                       radii.data_ptr<int>(), fp(g_color), has(color_raw) ? fp(*color_raw) : nullptr,
                                       g_m2d.data_ptr<float>(), g_cp.data_ptr<float>(), g_w.data_ptr<float>(), g_ol.data_ptr<float>(),
                                       has_mk ? g_mk.data_ptr<float>() : nullptr, scratch.data_ptr<float>(), 0, st),
              "cgs_view_backward_render");
    }
    ViewBwd r;
    r.g_cp = g_cp; r.g_w = g_w; r.g_ol = g_ol; r.g_mk = g_mk; r.g_m2d = g_m2d;
    return r;
}
py::object opt(const Tensor& t) { return t.defined() ? py::cast(t) : py::none(); }
py::tuple view_backward(const Tensor& cp, const Tensor& w, const Tensor& ol, const c10::optional<Tensor>& mk,
                        const c10::optional<Tensor>& is_bezier_u8, const Tensor& coef, const Tensor& geom, const Tensor& binb,
                        const Tensor& img, const Tensor& radii, const Tensor& norms, const Tensor& bgc, const Tensor& view,
                        const Tensor& proj, const Tensor& cpos, int64_t m, double mask_thr, double tanx, double tany, int64_t H,
                        int64_t W, double eps, const c10::optional<Tensor>& g_color_in, const c10::optional<Tensor>& color_raw,
                        const c10::optional<std::vector<Tensor>>& sinks, const c10::optional<Tensor>& rot_extra) {
    const ViewBwd r = view_backward_core(cp, w, ol, mk, is_bezier_u8, coef, geom, binb, img, radii, norms, bgc, view, proj, cpos, m,
                                         mask_thr, tanx, tany, H, W, eps, g_color_in, color_raw, sinks, rot_extra);
    return py::make_tuple(opt(r.g_cp), opt(r.g_w), opt(r.g_ol), opt(r.g_mk), r.g_m2d);
}

// ------------------------------------------------------------------------------------------------ C++ autograd nodes
// The fused view route and the photometric loss as torch::autograd::Function: their backward runs on the autograd engine's
// device thread WITHOUT the GIL and without the Python custom-Function machinery (ctx object, argument tuple checks, a
// Python frame per node) -- what `render()` + `loss.backward()` of the literal drop-in loop (train.py:95-148) pays per
// iteration beside the kernels themselves.  Same kernels, same saved state as ops/view_render.py::_ViewRender, which stays
// as the ctypes-bindings form (CGS_TORCH_SHIM=0).
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// Python: ops/view_render.py::_general_backward_cpp (re-render through the general route).  A leaked pointer on purpose: a
// static py::object would be destroyed after the interpreter has been finalised.
py::object* g_general_backward = nullptr;

struct ViewRenderFn : public torch::autograd::Function<ViewRenderFn> {
    // thread-local side channel for what is not a tensor output (read by view_render() right behind apply())
    static thread_local int t_handle;
    static thread_local Tensor t_img;

    // (absent tensors travel as c10::nullopt: an UNDEFINED Tensor argument makes Function::apply ask it for its device)
    static variable_list forward(AutogradContext* ctx, const Tensor& curve_points, const Tensor& width, const Tensor& opacity_logit,
                                 const c10::optional<Tensor>& mask_logit, const Tensor& means2D,
                                 const c10::optional<Tensor>& is_bezier_u8, const c10::optional<Tensor>& is_bezier, const Tensor& coef, int64_t m, double mask_thr, const Tensor& bg, const Tensor& viewmatrix,
                                 const Tensor& projmatrix, const Tensor& campos, double tanx, double tany, int64_t H, int64_t W, int64_t cap,
                                 bool sync_free, bool clamp, bool want_dir, double eps, const std::vector<Tensor>& sinks) {
        (void)means2D;
        ViewFwd f = view_forward_core(curve_points, width, opacity_logit, mask_logit, is_bezier_u8, coef, m, mask_thr, bg, viewmatrix, projmatrix,
                                      campos, tanx, tany, H, W, cap, sync_free, clamp, want_dir, eps);
        t_handle = f.handle;
        t_img = f.img;
        ctx->save_for_backward({f.cp, f.w, f.ol, f.mk, f.geom, f.binb, f.img, f.radii, f.norms, f.bgc, f.view, f.proj, f.cpos, coef,
                                is_bezier_u8.value_or(Tensor()), f.color_raw, is_bezier.value_or(Tensor())});
        ctx->saved_data["m"] = m; ctx->saved_data["H"] = H; ctx->saved_data["W"] = W;
        ctx->saved_data["mask_thr"] = mask_thr; ctx->saved_data["tanx"] = tanx; ctx->saved_data["tany"] = tany; ctx->saved_data["eps"] = eps;
        ctx->saved_data["clamp"] = clamp;
        ctx->saved_data["ran"] = false;
        if (!sinks.empty()) {
            // the sinks are the `.grad` tensors of FORWARD time; the owners are kept to see whether they still are at backward
            ctx->saved_data["sinks"] = c10::List<Tensor>(sinks);
            std::vector<Tensor> owners = {curve_points, width, opacity_logit};
            if (has(mask_logit)) owners.push_back(*mask_logit);
            ctx->saved_data["owners"] = c10::List<Tensor>(owners);
        }
        ctx->mark_non_differentiable({f.radii});
        ctx->set_materialize_grads(false);
        return {f.color_out, f.invd, f.amap, f.radii, f.rend_dir};
    }

    static variable_list backward(AutogradContext* ctx, variable_list go) {
        const variable_list sv = ctx->get_saved_variables();
        const Tensor &cp = sv[0], &w = sv[1], &ol = sv[2], &mk = sv[3], &geom = sv[4], &binb = sv[5], &img = sv[6], &radii = sv[7],
                     &norms = sv[8], &bgc = sv[9], &view = sv[10], &proj = sv[11], &cpos = sv[12], &coef = sv[13], &isb = sv[14],
                     &color_raw = sv[15], &is_bezier = sv[16];
        const int64_t m = ctx->saved_data["m"].toInt(), H = ctx->saved_data["H"].toInt(), W = ctx->saved_data["W"].toInt();
        const double mask_thr = ctx->saved_data["mask_thr"].toDouble(), tanx = ctx->saved_data["tanx"].toDouble(),
                     tany = ctx->saved_data["tany"].toDouble(), eps = ctx->saved_data["eps"].toDouble();
        variable_list out(24);   // one slot per forward argument; only the first five can carry a gradient
        const Tensor &g_color = go[0], &g_invd = go[1], &g_amap = go[2], &g_dir = go[4];
        if (g_invd.defined() || g_amap.defined() || g_dir.defined()) {
            // a loss on inverse depth / all_map / the direction map: re-render through the differentiable general route (Python)
            py::gil_scoped_acquire gil;
            if (!g_general_backward) raise_cgs("view_render: the general backward is not registered");
            py::tuple r = (*g_general_backward)(cp, w, ol, opt(mk), bgc, view, proj, cpos, is_bezier.defined() ? py::cast(is_bezier) : py::none(),
                                             m, H, W, mask_thr, tanx, tany, eps, ctx->saved_data["clamp"].toBool(), opt(g_color), opt(g_invd),
                                             opt(g_amap), opt(g_dir));
            for (int i = 0; i < 5; i++)
                if (!r[i].is_none()) out[i] = r[i].cast<Tensor>();
            return out;
        }
        if (ctx->saved_data["ran"].toBool()) {
            // a second backward over this forward (retain_graph): the sampling backward's two grid-wide sums were cleared by the
            // forward's norm pass once -- clear them again
            int first = 0, count = 0;
            cgs_view_norms_backward_range(&first, &count);
            norms.narrow(0, first, count).zero_();
        }
        c10::optional<std::vector<Tensor>> sinks;
        if (ctx->saved_data.count("sinks")) {
            const auto sk = ctx->saved_data["sinks"].toTensorList();
            const auto ow = ctx->saved_data["owners"].toTensorList();
            bool same = sk.size() == ow.size();
            for (size_t i = 0; same && i < sk.size(); i++) {
                const Tensor o = ow.get(i), s = sk.get(i);
                same = o.grad().defined() && o.grad().unsafeGetTensorImpl() == s.unsafeGetTensorImpl();
            }
            if (same) sinks = std::vector<Tensor>(sk.begin(), sk.end());   // (else: replaced since the forward -> ordinary gradients)
        }
        const ViewBwd r = view_backward_core(cp, w, ol, mk.defined() ? c10::optional<Tensor>(mk) : c10::nullopt,
                                             isb.defined() ? c10::optional<Tensor>(isb) : c10::nullopt, coef, geom, binb, img, radii, norms,
                                             bgc, view, proj, cpos, m, mask_thr, tanx, tany, H, W, eps,
                                             g_color.defined() ? c10::optional<Tensor>(g_color) : c10::nullopt,
                                             color_raw.defined() ? c10::optional<Tensor>(color_raw) : c10::nullopt, sinks, c10::nullopt);
        ctx->saved_data["ran"] = true;
        out[0] = r.g_cp; out[1] = r.g_w; out[2] = r.g_ol; out[3] = r.g_mk; out[4] = r.g_m2d;
        return out;
    }
};
thread_local int ViewRenderFn::t_handle = -1;
thread_local Tensor ViewRenderFn::t_img;

// -> (color, invdepth, all_map, radii, rend_dir, handle, image buffer)
py::tuple view_render(const Tensor& curve_points, const Tensor& width, const Tensor& opacity_logit, const c10::optional<Tensor>& mask_logit,
                      const Tensor& means2D, const c10::optional<Tensor>& is_bezier_u8, const c10::optional<Tensor>& is_bezier,
                      const Tensor& coef, int64_t m, double mask_thr, const Tensor& bg, const Tensor& viewmatrix, const Tensor& projmatrix,
                      const Tensor& campos, double tanx, double tany, int64_t H, int64_t W, int64_t cap, bool sync_free, bool clamp,
                      bool want_dir, double eps, const c10::optional<std::vector<Tensor>>& sinks) {
    const variable_list o = ViewRenderFn::apply(curve_points, width, opacity_logit, mask_logit, means2D, is_bezier_u8, is_bezier, coef, m, mask_thr, bg,
                                                viewmatrix, projmatrix, campos, tanx, tany, H, W, cap, sync_free, clamp, want_dir, eps,
                                                sinks.value_or(std::vector<Tensor>()));
    const int handle = ViewRenderFn::t_handle;
    Tensor img = ViewRenderFn::t_img;
    ViewRenderFn::t_img = Tensor();
    return py::make_tuple(o[0], o[1], o[2], o[3], o[4], handle, img);
}
void set_general_backward(py::object fn) {
    if (g_general_backward) *g_general_backward = std::move(fn);
    else g_general_backward = new py::object(std::move(fn));
}

// loss = a * edge_aware_loss(x, gt) + b * (1 - ssim(x, gt)), x = clamp(image) if clamp (ops/losses.py::photometric_loss): value and
// d loss / d image from cgs_photometric_loss in the forward; the backward hands the stored gradient on (times the upstream scalar
// unless that is the shared unit tensor `unit`).
struct PhotometricLossFn : public torch::autograd::Function<PhotometricLossFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& image, const Tensor& gt, const Tensor& n_pos, const Tensor& ws, double thr,
                          double a, double b, bool clamp, const c10::optional<Tensor>& unit) {
        require_gpu(image, "image");
        c10::hip::HIPGuardMasqueradingAsCUDA guard(image.device());
        const Tensor img = f32c(image.detach(), "image"), g = f32c(gt.detach(), "gt_image");
        if (img.dim() != 3 || img.size(0) != 1) raise_cgs("photometric_loss: the fused path renders 1 channel");
        const int H = (int)img.size(1), W = (int)img.size(2);
        Tensor grad = at::empty_like(img);
        Tensor loss = at::empty({}, img.options());
        check(cgs_photometric_loss(H, W, fp(img), fp(g), (float)thr, (const uint32_t*)n_pos.data_ptr(), (float)a, (float)b, clamp ? 1 : 0,
                                   ws.data_ptr(), grad.data_ptr<float>(), loss.data_ptr<float>(), stream_of(img)),
              "cgs_photometric_loss");
        ctx->save_for_backward({grad, unit.value_or(Tensor())});
        return loss;
    }
    static variable_list backward(AutogradContext* ctx, variable_list go) {
        const variable_list sv = ctx->get_saved_variables();
        variable_list out(9);
        const Tensor& g = go[0];
        const bool is_unit = sv[1].defined() && g.defined() && g.numel() == 1 && g.data_ptr() == sv[1].data_ptr();
        out[0] = is_unit ? sv[0] : sv[0] * g;
        return out;
    }
};
Tensor photometric_loss(const Tensor& image, const Tensor& gt, const Tensor& n_pos, const Tensor& ws, double thr, double a, double b,
                        bool clamp, const c10::optional<Tensor>& unit) {
    return PhotometricLossFn::apply(image, gt, n_pos, ws, thr, a, b, clamp, unit);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, mod) {
    mod.doc() = "torch <-> libcurvegs host shim (see csrc/torch_shim.cpp)";
    mod.def("rasterize_gaussians", &rasterize_gaussians);
    mod.def("rasterize_gaussians_static", &rasterize_gaussians_static);
    mod.def("forward_status", &forward_status);
    mod.def("rasterize_gaussians_backward", &rasterize_gaussians_backward, py::arg("background"), py::arg("all_map_pixels"),
            py::arg("means3D"), py::arg("radii"), py::arg("colors"), py::arg("all_maps"), py::arg("opacities"), py::arg("scales"),
            py::arg("rotations"), py::arg("scale_modifier"), py::arg("cov3D_precomp"), py::arg("viewmatrix"), py::arg("projmatrix"),
            py::arg("tan_fovx"), py::arg("tan_fovy"), py::arg("dL_dout_color"), py::arg("dL_dout_invdepth"),
            py::arg("dL_dout_all_map"), py::arg("sh"), py::arg("degree"), py::arg("campos"), py::arg("geomBuffer"), py::arg("R"),
            py::arg("binningBuffer"), py::arg("imageBuffer"), py::arg("antialiasing"), py::arg("render_geo"), py::arg("debug"),
            py::arg("need_color_grad") = true);
    mod.def("mark_visible", &mark_visible);
    mod.def("view_forward", &view_forward);
    mod.def("view_wait", &view_wait);
    mod.def("view_abandon", &view_abandon);
    mod.def("view_render", &view_render);
    mod.def("set_general_backward", &set_general_backward);
    mod.def("photometric_loss", &photometric_loss);
    mod.def("view_backward", &view_backward, py::arg("cp"), py::arg("w"), py::arg("ol"), py::arg("mk"), py::arg("is_bezier_u8"),
            py::arg("coef"), py::arg("geom"), py::arg("binb"), py::arg("img"), py::arg("radii"), py::arg("norms"), py::arg("bgc"),
            py::arg("view"), py::arg("proj"), py::arg("cpos"), py::arg("m"), py::arg("mask_thr"), py::arg("tanx"), py::arg("tany"),
            py::arg("H"), py::arg("W"), py::arg("eps"), py::arg("g_color"), py::arg("color_raw"), py::arg("sinks") = py::none(),
            py::arg("rot_extra") = py::none());
}
