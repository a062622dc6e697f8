// Fused per-view per-splat kernels: the three one-thread-per-splat chains of a training view collapsed into one kernel
// per direction, with the intermediate per-splat tensors kept in registers.
//
//   forward   k_sample_f3 -> k_attrs_fwd -> k_preprocess_fwd          = k_view_fwd
//             (xyz / rot / scaling -> rot_n / opacity / all_map -> SplatRec + radii: 100 B/splat written and re-read twice)
//   backward  k_preprocess_bwd -> k_attrs_bwd -> k_sample_bwd<1>      = k_view_bwd
//             (dL/d{mean3D, scale, rot_n, opacity, all_map} -> dL/d{rot_raw, scaling} -> dL/d{v0,v1,v2} + global sums)
//
// Same device functions as the separate kernels (curve_math.h, splat_math.h), so the results are the same; the global
// Frobenius norms still need their own pass before (k_sample_f12) and the per-curve reduction its own pass after
// (k_sample_bwd<3>).  Training configuration only: scales + rotations (no precomputed covariance), precomputed colours
// (no SH), no antialiasing -- everything else goes through the general kernels.
#include "curve_math.h"
#include "splat_math.h"

namespace cgs {

__global__ void __launch_bounds__(256) k_view_fwd(
    int B, int m, const float* __restrict__ cp, const float* __restrict__ width, const uint8_t* __restrict__ is_bezier,
    const SampleCoef* __restrict__ coef, float eps, const double* __restrict__ norms,
    const float* __restrict__ opacity_logit, const float* __restrict__ mask_logit, float mask_thr,
    const float* __restrict__ colors_precomp, const float* __restrict__ campos, ViewParams vp,
    float* __restrict__ xyz_out, float* __restrict__ rot_out, float* __restrict__ scl_out, int* __restrict__ radii,
    SplatRec* __restrict__ rec, float* __restrict__ grad_acc, uint32_t* __restrict__ clear_words, uint32_t n_clear) {
    __shared__ SampleCoef s_coef[MAX_M];
    __shared__ BlockConst s_bc;
    stage_consts(coef, m, norms, s_coef, &s_bc);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    // folded zero fills, as in k_preprocess_fwd: the tile histogram / cursors / status words of the bucket binning and
    // this splat's gradient accumulator record
    for (uint32_t i = (uint32_t)p; i < n_clear; i += gridDim.x * blockDim.x) clear_words[i] = 0u;
    if (p >= B * m) return;
    {
        float4* accp = reinterpret_cast<float4*>(grad_acc + (size_t)p * ACC_STRIDE_VIEW);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < ACC_STRIDE_VIEW / 4; k++) accp[k] = z;
    }
    const int b = p / m, i = p - b * m;
    const CurveCP c = load_curve(cp, is_bezier, b);
    const float w = expf(width[b]);
    const SampleFwd s = sample_forward(c, s_coef[i], s_bc.N1, s_bc.N2, eps);
    float M[3][3], q[4];
    rot_matrix(s, M);
    quat_forward(M, q);
    if (xyz_out) {   // the model's derived tensors (prepare_scaling_rot), for callers that want them
        xyz_out[3 * p] = s.xyz.x; xyz_out[3 * p + 1] = s.xyz.y; xyz_out[3 * p + 2] = s.xyz.z;
        reinterpret_cast<float4*>(rot_out)[p] = make_float4(q[0], q[1], q[2], q[3]);
        scl_out[3 * p] = s.dist; scl_out[3 * p + 1] = w; scl_out[3 * p + 2] = w;
    }
    const V3 cam = {campos[0], campos[1], campos[2]};
    const AttrsFwd a = attrs_forward(make_float4(q[0], q[1], q[2], q[3]), s.xyz, opacity_logit[b], mask_logit != nullptr,
                                     mask_logit ? mask_logit[p] : 0.f, mask_thr, cam, vp.vm);
    const float3 sc = make_float3(s.dist * a.mk, w * a.mk, w * a.mk);
    float cov3D[6];
    cov3d_from_scale_rot(sc, 1.0f, a.rot_n, cov3D);
    SplatGeom g;
    int out_radius = 0;
    if (splat_geometry(make_float3(s.xyz.x, s.xyz.y, s.xyz.z), cov3D, vp, 0, g)) {
        rec[p] = splat_record(g, a.opac, colors_precomp ? colors_precomp[p] : 1.0f, a.all_map);
        out_radius = (int)g.radius;
    }
    radii[p] = out_radius;
}

// Blocks hold whole curves (curves_per_block * m active threads), as k_attrs_bwd / k_sample_bwd<3> do: the per-curve
// opacity-logit gradient is the sample-ordered sum of its m per-splat terms.
// 6 waves per SIMD = 80 VGPRs, the allocation of the compositors this kernel shares the GPU with when several views are
// in flight: at its natural 115 VGPRs its waves only fit a SIMD once several compositor waves have drained, and the
// kernel (18 us alone) stretched to 102 us inside the overlapped schedule (round 2: 0.406 -> 0.385 ms per view at 72
// VGPRs).  Round 3, with both compositors at 80 VGPRs: 7 / 6 / 5 waves = 23.8 / 20.4 / 18.4 us serial and 792 / 793 / 774
// Msplats/s with three views in flight -- 6 takes the serial gain without the overlapped loss.
// Round 5: the waves-per-SIMD bound is a template parameter chosen per launch.  At a million splats (cfg5) the kernel is a
// sixth of the serial view and its 23 spills at the 80-VGPR cap cost more than the co-residence buys: 6 / 5 / 4 waves =
// 1 476 - 1 483 / 1 505 / 1 532 Msplats/s with three views in flight there, 815 - 817 / 797 / 800 at cfg3 (same box).
#ifndef CGS_VIEW_BWD_WAVES
#define CGS_VIEW_BWD_WAVES 6
#endif
#ifndef CGS_VIEW_BWD_WAVES_LARGE
#define CGS_VIEW_BWD_WAVES_LARGE 4
#endif
constexpr int VIEW_BWD_LARGE_P = 512 * 1024;   // splat count from which the roomier instance is launched
template <int WAVES>
__global__ void __launch_bounds__(SAMPLE_BLOCK, WAVES) k_view_bwd(
    int B, int m, int curves_per_block, const float* __restrict__ cp, const float* __restrict__ width,
    const uint8_t* __restrict__ is_bezier, const SampleCoef* __restrict__ coef, float eps, double* __restrict__ norms,
    const float* __restrict__ opacity_logit, const float* __restrict__ mask_logit, float mask_thr,
    const float* __restrict__ campos, ViewParams vp, const int* __restrict__ radii, const SplatRec* __restrict__ rec,
    float* __restrict__ grad_acc, const float* __restrict__ g_rot_raw_extra, float* __restrict__ dL_dmean2D,
    float* __restrict__ g_opacity_logit, float* __restrict__ g_mask_logit, float* __restrict__ curve_part,
    int accumulate_flags) {
    // bit 0: add to the caller's gradient outputs; bit 1: add to the per-curve partials as well (shared-sampling mode: the
    // sampling backward's closing pass runs once for several views and is linear in them)
    const int accumulate = accumulate_flags & 1;
    const bool acc_scr = (accumulate_flags & 2) != 0;
    __shared__ float s_part[CURVE_PART][SAMPLE_BLOCK + 1];
    __shared__ SampleCoef s_coef[MAX_M];
    __shared__ BlockConst s_bc;
    __shared__ float s_go[SAMPLE_BLOCK];
    stage_consts(coef, m, norms, s_coef, &s_bc);
    const int lc = threadIdx.x / m, i = threadIdx.x - lc * m;
    const int b = blockIdx.x * curves_per_block + lc;
    const bool valid = lc < curves_per_block && b < B;
    float g_op_term = 0.f;
    double acc_d2 = 0, acc_a = 0;
    CurveGrad cg;
    cg.gp0 = cg.gp1 = cg.gp2 = cg.gp3 = V3{0.f, 0.f, 0.f};
    cg.gw = 0.f;
    if (valid) {
        const size_t p = (size_t)b * m + i;
        const float N1 = s_bc.N1, N2 = s_bc.N2;
        // ---- recompute the forward quantities of this sample (cheaper than 100 B/splat of round trips)
        const CurveCP c = load_curve(cp, is_bezier, b);
        const float w = expf(width[b]);
        const SampleFwd s = sample_forward(c, s_coef[i], N1, N2, eps);
        float M[3][3], qv[4];
        rot_matrix(s, M);
        const QuatFwd f = quat_forward(M, qv);
        const float4 q = make_float4(qv[0], qv[1], qv[2], qv[3]);
        const V3 cam = {campos[0], campos[1], campos[2]};
        const bool has_mask = mask_logit != nullptr;
        const float ml = has_mask ? mask_logit[p] : 0.f;
        const AttrsFwd a = attrs_forward(q, s.xyz, opacity_logit[b], has_mask, ml, mask_thr, cam, vp.vm);
        const float3 sc = make_float3(s.dist * a.mk, w * a.mk, w * a.mk);
        // ---- rasterizer backward tail (K9 + K10) on the compositor's sums; the record is handed back zeroed
        float4* accp = reinterpret_cast<float4*>(grad_acc + (size_t)p * ACC_STRIDE_VIEW);
        const float4 acc0 = accp[0], acc1 = accp[1];
        {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            accp[0] = z; accp[1] = z;
        }
        const bool vis = radii[p] > 0;
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
        float cov3D[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (vis) {
            ra = rec[p].a;
            rb = rec[p].b;
            cov3d_from_scale_rot(sc, 1.0f, a.rot_n, cov3D);
        }
        SplatGrads o;
        splat_backward(acc0, acc1, vis, ra, rb, make_float3(s.xyz.x, s.xyz.y, s.xyz.z), cov3D, sc, a.rot_n, true, 1.0f, 0.f,
                       vp, 0, false, o);
        dL_dmean2D[3 * p] = o.g2x; dL_dmean2D[3 * p + 1] = o.g2y; dL_dmean2D[3 * p + 2] = 0.f;
        // ---- splat attributes backward
        const V3 gs = {o.dscale.x, o.dscale.y, o.dscale.z};
        const V3 scl = {s.dist, w, w};
        const AttrsBwd ab = attrs_backward(q, s.xyz, opacity_logit[b], has_mask, ml, mask_thr, cam, vp.vm, o.drot, false /* no gradient reaches all_map on this path */, make_float4(0.f, 0.f, 0.f, 0.f),
                                           o.dopac, has_mask, gs, scl);
        g_op_term = ab.g_op_term;
        if (g_mask_logit) g_mask_logit[p] = accumulate ? g_mask_logit[p] + ab.g_mask_logit : ab.g_mask_logit;
        float4 grr = ab.g_rot_raw;
        if (g_rot_raw_extra) {   // e.g. the curve-smoothness regulariser's gradient on the raw rotation
            const float4 e = reinterpret_cast<const float4*>(g_rot_raw_extra)[p];
            grr = make_float4(grr.x + e.x, grr.y + e.y, grr.z + e.z, grr.w + e.w);
        }
        // ---- sampling backward: the two grid-wide sums, and the part of this sample's dL/d{p0..p3, width} that does not
        // depend on them (sample_backward_tail with D2 = D1 = 0; the closing pass k_sample_bwd_close adds the rest) -- summed
        // over the curve below: 13 floats per CURVE leave the kernel where 15 per SPLAT used to (and came back)
        float gM[3][3];
        const float go[4] = {grr.x, grr.y, grr.z, grr.w};
        quat_backward(f, go, gM);
        const V3 g_v0 = {gM[0][0], gM[1][0], gM[2][0]};
        const V3 g_v1 = {gM[0][1], gM[1][1], gM[2][1]};
        const V3 g_v2 = {gM[0][2], gM[1][2], gM[2][2]};
        acc_d2 = (double)dot(g_v2, s.c2v);
        acc_a = (double)dot(g_v1, s.c1v) + (double)((1.f / N2) * dot(cross(g_v2, s.tan), s.c1v));
        const V3 g_x = {o.dmean.x, o.dmean.y, o.dmean.z}, g_s = {gs.x * ab.mk, gs.y * ab.mk, gs.z * ab.mk};
        cg = sample_backward_tail(c, s_coef[i], s, w, eps, N1, N2, 0.f, 0.f, g_v0, g_v1, g_v2, g_x, true, g_s);
    }
    s_go[threadIdx.x] = g_op_term;
    const double acc2v[2] = {acc_d2, acc_a};
    block_accumulate<2>(acc2v, norms, 3);     // (contains the barrier that publishes s_go)
    curve_reduce(cg, s_part, m, curves_per_block, [&](int c2, int f, float sum) {
        const int bb = blockIdx.x * curves_per_block + c2;
        if (bb >= B) return;
        float* dst = curve_part + (size_t)bb * CURVE_PART + f;
        *dst = acc_scr ? *dst + sum : sum;
    });
    if (valid && i == 0) {
        float sum = 0.f;
        for (int k = 0; k < m; k++) sum += s_go[threadIdx.x + k];
        g_opacity_logit[b] = accumulate ? g_opacity_logit[b] + sum : sum;
    }
}

// ------------------------------------------------------------------------------------------------ launchers
void launch_view_forward(hipStream_t s, int B, int m, const float* cp, const float* width, const uint8_t* is_bezier,
                         const void* coef, float eps, const double* norms, const float* opacity_logit,
                         const float* mask_logit, float mask_thr, const float* colors_precomp, const float* campos,
                         const float* viewmatrix, const float* projmatrix, float tan_fovx, float tan_fovy, float focal_x,
                         float focal_y, int W, int H, int grid_x, int grid_y, float* xyz, float* rot, float* scl, int* radii,
                         SplatRec* rec, float* grad_acc, uint32_t* clear_words, size_t n_clear) {
    ProfScope p("view_fwd", s);
    const ViewParams vp{viewmatrix, projmatrix, tan_fovx, tan_fovy, focal_x, focal_y, W, H, grid_x, grid_y};
    hipLaunchKernelGGL(k_view_fwd, dim3((B * m + 255) / 256), dim3(256), 0, s, B, m, cp, width, is_bezier,
                       reinterpret_cast<const SampleCoef*>(coef), eps, norms, opacity_logit, mask_logit, mask_thr,
                       colors_precomp, campos, vp, xyz, rot, scl, radii, rec, grad_acc, clear_words, (uint32_t)n_clear);
}
void launch_view_backward(hipStream_t s, int B, int m, const float* cp, const float* width, const uint8_t* is_bezier,
                          const void* coef, float eps, double* norms, const float* opacity_logit, const float* mask_logit,
                          float mask_thr, const float* campos, const float* viewmatrix, const float* projmatrix,
                          float tan_fovx, float tan_fovy, float focal_x, float focal_y, int W, int H, const int* radii,
                          const SplatRec* rec, float* grad_acc, const float* g_rot_raw_extra, float* dL_dmean2D,
                          float* g_opacity_logit, float* g_mask_logit, float* curve_part, int accumulate) {
    ProfScope p("view_bwd", s);
    const int cpb = SAMPLE_BLOCK / m;
    const ViewParams vp{viewmatrix, projmatrix, tan_fovx, tan_fovy, focal_x, focal_y, W, H, 0, 0};
    if ((long long)B * m >= VIEW_BWD_LARGE_P)
        hipLaunchKernelGGL(k_view_bwd<CGS_VIEW_BWD_WAVES_LARGE>, dim3((B + cpb - 1) / cpb), dim3(SAMPLE_BLOCK), 0, s, B, m, cpb, cp,
                           width, is_bezier, reinterpret_cast<const SampleCoef*>(coef), eps, norms, opacity_logit, mask_logit,
                           mask_thr, campos, vp, radii, rec, grad_acc, g_rot_raw_extra, dL_dmean2D, g_opacity_logit, g_mask_logit,
                           curve_part, accumulate);
    else
        hipLaunchKernelGGL(k_view_bwd<CGS_VIEW_BWD_WAVES>, dim3((B + cpb - 1) / cpb), dim3(SAMPLE_BLOCK), 0, s, B, m, cpb, cp, width,
                           is_bezier, reinterpret_cast<const SampleCoef*>(coef), eps, norms, opacity_logit, mask_logit, mask_thr,
                           campos, vp, radii, rec, grad_acc, g_rot_raw_extra, dL_dmean2D, g_opacity_logit, g_mask_logit, curve_part,
                           accumulate);
}

}  // namespace cgs
