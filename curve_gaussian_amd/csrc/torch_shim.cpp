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
py::tuple view_forward(const Tensor& curve_points, const Tensor& width, const Tensor& opacity_logit,
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
    py::object color_raw = py::none();
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
    if (clamp) color_raw = py::cast(color);
    return py::make_tuple(color_out, invd, amap, radii, rend_dir, color_raw,
                          py::make_tuple(cp, w, ol, mk.defined() ? py::cast(mk) : py::none(), geom, binb, img, radii, norms, bgc, view, proj, cpos),
                          sync_free ? -1 : handle);
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
py::tuple view_backward(const Tensor& cp, const Tensor& w, const Tensor& ol, const c10::optional<Tensor>& mk,
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
        return py::make_tuple(py::none(), py::none(), py::none(), py::none(), g_m2d);
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
    return py::make_tuple(g_cp, g_w, g_ol, has_mk ? py::cast(g_mk) : py::none(), g_m2d);
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
    mod.def("view_backward", &view_backward, py::arg("cp"), py::arg("w"), py::arg("ol"), py::arg("mk"), py::arg("is_bezier_u8"),
            py::arg("coef"), py::arg("geom"), py::arg("binb"), py::arg("img"), py::arg("radii"), py::arg("norms"), py::arg("bgc"),
            py::arg("view"), py::arg("proj"), py::arg("cpos"), py::arg("m"), py::arg("mask_thr"), py::arg("tanx"), py::arg("tany"),
            py::arg("H"), py::arg("W"), py::arg("eps"), py::arg("g_color"), py::arg("color_raw"), py::arg("sinks") = py::none(),
            py::arg("rot_extra") = py::none());
}
