// Per-splat projection / covariance math shared by the preprocess kernels (preprocess.hip) and the fused per-view
// kernels (view.hip).  Forward: reference K1, forward.cu:155-274 (+ computeCov3D :118-152, computeCov2D :78-113);
// backward: reference K9 + K10, backward.cu:146-325, :329-448.  Values in, values out -- the callers own the memory
// traffic.
#pragma once
#include "kernels.h"

namespace cgs {

struct ViewParams {                  // what one view contributes to the per-splat math
    const float* vm;                 // viewmatrix  [16] (column-major math matrix)
    const float* pm;                 // projmatrix  [16]
    float tan_fovx, tan_fovy, focal_x, focal_y;
    int W, H, grid_x, grid_y;
};

struct SplatGeom {                   // forward result for one visible splat
    float px, py;                    // pixel-space mean
    float3 conic;
    float depth;                     // view-space z
    float radius;                    // ceil(3 sqrt(lambda_max))
    float h_scale;                   // antialiasing opacity factor (1 without antialiasing)
    uint2 rmin, rmax;                // tile rectangle
};

// false: the splat is not processed further (reference radius 0): behind the near plane, singular covariance, or an
// empty tile rectangle.
__device__ __forceinline__ bool splat_geometry(const float3 p_orig, const float cov3D[6], const ViewParams& v,
                                               int antialiasing, SplatGeom& g) {
#pragma clang fp contract(off)
    const float3 p_view = xform4x3(p_orig, v.vm);
    if (p_view.z <= 0.2f) return false;  // near cull only, auxiliary.h:166
    const float4 p_hom = xform4x4(p_orig, v.pm);
    const float p_w = 1.0f / (p_hom.w + 0.0000001f);
    const float3 p_proj = make_float3(p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w);
    float3 t, cov;
    float Mt[2][3], txtz, tytz;
    cov2d_terms(p_orig, v.focal_x, v.focal_y, v.tan_fovx, v.tan_fovy, cov3D, v.vm, t, Mt, cov, txtz, tytz);
    constexpr float h_var = 0.3f;
    const float det_cov = cov.x * cov.z - cov.y * cov.y;
    cov.x += h_var;
    cov.z += h_var;
    const float det_cov_plus_h_cov = cov.x * cov.z - cov.y * cov.y;
    g.h_scale = 1.0f;
    if (antialiasing) g.h_scale = sqrtf(fmaxf(0.000025f, det_cov / det_cov_plus_h_cov));
    const float det = det_cov_plus_h_cov;
    if (det == 0.0f) return false;
    const float det_inv = 1.f / det;
    g.conic = make_float3(cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv);
    const float mid = 0.5f * (cov.x + cov.z);
    const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    g.radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    g.px = ndc2pix(p_proj.x, v.W);
    g.py = ndc2pix(p_proj.y, v.H);
    get_rect(g.px, g.py, (int)g.radius, v.grid_x, v.grid_y, g.rmin, g.rmax);
    if ((g.rmax.x - g.rmin.x) * (g.rmax.y - g.rmin.y) == 0) return false;
    g.depth = p_view.z;
    return true;
}

// The 64-byte record the compositors gather (common.h::SplatRec).
__device__ __forceinline__ SplatRec splat_record(const SplatGeom& g, float opacity, float color, const float4 all_map) {
    SplatRec r;
    r.a = make_float4(g.px, g.py, g.conic.x, g.conic.y);
    const float op_eff = opacity * g.h_scale;
    r.b = make_float4(g.conic.z, op_eff, color, 1.f / g.depth);
    r.c = all_map;
    // tau2 = 2 ln(255 * opacity): alpha >= 1/255  <=>  conic quadratic form <= tau2 (used by the quadrant culling)
    // d.w: 1 when the splat's colour or all_map[3] is not exactly 1 -- the bucket scatter ORs it into the image buffer's
    // "non-unit" word, which decides on the device whether the unit-colour backward may run (api.hip, cgs_rasterize_backward)
    r.d = make_float4(g.depth, g.radius, 2.f * logf(255.f * op_eff), (color != 1.f || all_map.w != 1.f) ? 1.f : 0.f);
    return r;
}

struct SplatGrads {                  // backward result for one splat (all zero / pass-through when it was culled)
    float g2x, g2y;                  // dL/dmean2D, NDC-scaled (backward.cu:542-543, 659-664)
    float dcx, dcy, dcz;             // dL/dconic
    float dopac;
    float3 dmean;                    // dL/dmean3D without the SH colour path
    float dcov[6];
    float3 dscale;
    float4 drot;                     // raw quaternion gradient (no normalisation Jacobian, backward.cu:391)
};

// acc0 = {Sg, Sx, Sy, Sxx}, acc1 = {Sxy, Syy, dL/dcolour, dL/dinvdepth}: the compositor's raw per-splat sums of
// g = opacity G dL/dalpha and its first and second moments
// (common.h, ACC_* layout).  `has_scale`: cov3D came from (sc, q) and their gradients are wanted.
__device__ __forceinline__ void splat_backward(const float4 acc0, const float4 acc1, bool vis, const float4 ra,
                                               const float4 rb, const float3 mean, const float cov3D[6], const float3 sc,
                                               const float4 q, bool has_scale, float scale_modifier, float opacity_in,
                                               const ViewParams& v, int antialiasing, bool invdepth_path, SplatGrads& o) {
    o.g2x = o.g2y = o.dcx = o.dcy = o.dcz = 0.f;
    o.dopac = acc0.x;
    o.dmean = make_float3(0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 6; i++) o.dcov[i] = 0.f;
    o.dscale = make_float3(0.f, 0.f, 0.f);
    o.drot = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!vis) return;
    {
        // the sums carry the opacity already (g = opacity G dL/dalpha): dL/dG = opacity dL/dalpha of backward.cu:655 is
        // folded in, and dL/dopacity = sum G dL/dalpha (backward.cu:672) is the opacity-scaled sum divided back once
        const float cA = ra.z, cB = ra.w, cC = rb.x, op = rb.y;
        o.g2x = -(cA * acc0.y + cB * acc0.z) * (float)(0.5 * v.W);  // backward.cu:542-543, 659-664
        o.g2y = -(cC * acc0.z + cB * acc0.y) * (float)(0.5 * v.H);
        o.dcx = -0.5f * acc0.w;                                     // backward.cu:667-669
        o.dcy = -0.5f * acc1.x;
        o.dcz = -0.5f * acc1.y;
        o.dopac = op > 0.f ? acc0.x / op : 0.f;
    }
    const float dcx = o.dcx, dcy = o.dcy, dcz = o.dcz;
    float3 t, cov;
    float T_[2][3], txtz, tytz;
    cov2d_terms(mean, v.focal_x, v.focal_y, v.tan_fovx, v.tan_fovy, cov3D, v.vm, t, T_, cov, txtz, tytz);
    const float limx = 1.3f * v.tan_fovx, limy = 1.3f * v.tan_fovy;
    const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    float c_xx = cov.x, c_xy = cov.y, c_yy = cov.z;
    constexpr float h_var = 0.3f;
    float d_inside_root = 0.f;
    if (antialiasing) {
        const float det_cov = c_xx * c_yy - c_xy * c_xy;
        c_xx += h_var;
        c_yy += h_var;
        const float det_cov_plus_h_cov = c_xx * c_yy - c_xy * c_xy;
        const float h_convolution_scaling = sqrtf(fmaxf(0.000025f, det_cov / det_cov_plus_h_cov));
        const float dL_dopacity_v = o.dopac;
        const float d_h_convolution_scaling = dL_dopacity_v * opacity_in;
        o.dopac = dL_dopacity_v * h_convolution_scaling;
        d_inside_root = (det_cov / det_cov_plus_h_cov) <= 0.000025f ? 0.f : d_h_convolution_scaling / (2 * h_convolution_scaling);
    } else {
        c_xx += h_var;
        c_yy += h_var;
    }
    float dL_dc_xx = 0, dL_dc_xy = 0, dL_dc_yy = 0;
    if (antialiasing) {
        const float x = c_xx, y = c_yy, z = c_xy, w = h_var;
        const float sqv = (w * w + w * (x + y) + x * y - z * z);
        const float denom_f = d_inside_root / (sqv * sqv);
        dL_dc_xx = w * (w * y + y * y + z * z) * denom_f;
        dL_dc_yy = w * (w * x + x * x + z * z) * denom_f;
        dL_dc_xy = -2.f * w * z * (w + x + y) * denom_f;
    }
    const float denom = c_xx * c_yy - c_xy * c_xy;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float* dcov = o.dcov;
    if (denom2inv != 0) {
        dL_dc_xx += denom2inv * (-c_yy * c_yy * dcx + 2 * c_xy * c_yy * dcy + (denom - c_xx * c_yy) * dcz);
        dL_dc_yy += denom2inv * (-c_xx * c_xx * dcz + 2 * c_xx * c_xy * dcy + (denom - c_xx * c_yy) * dcx);
        dL_dc_xy += denom2inv * 2 * (c_xy * c_yy * dcx - (denom + 2 * c_xy * c_xy) * dcy + c_xx * c_xy * dcz);
        dcov[0] = (T_[0][0] * T_[0][0] * dL_dc_xx + T_[0][0] * T_[1][0] * dL_dc_xy + T_[1][0] * T_[1][0] * dL_dc_yy);
        dcov[3] = (T_[0][1] * T_[0][1] * dL_dc_xx + T_[0][1] * T_[1][1] * dL_dc_xy + T_[1][1] * T_[1][1] * dL_dc_yy);
        dcov[5] = (T_[0][2] * T_[0][2] * dL_dc_xx + T_[0][2] * T_[1][2] * dL_dc_xy + T_[1][2] * T_[1][2] * dL_dc_yy);
        dcov[1] = 2 * T_[0][0] * T_[0][1] * dL_dc_xx + (T_[0][0] * T_[1][1] + T_[0][1] * T_[1][0]) * dL_dc_xy + 2 * T_[1][0] * T_[1][1] * dL_dc_yy;
        dcov[2] = 2 * T_[0][0] * T_[0][2] * dL_dc_xx + (T_[0][0] * T_[1][2] + T_[0][2] * T_[1][0]) * dL_dc_xy + 2 * T_[1][0] * T_[1][2] * dL_dc_yy;
        dcov[4] = 2 * T_[0][2] * T_[0][1] * dL_dc_xx + (T_[0][1] * T_[1][2] + T_[0][2] * T_[1][1]) * dL_dc_xy + 2 * T_[1][1] * T_[1][2] * dL_dc_yy;
    }
    const float V[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}};
    float dT0[3], dT1[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float u0 = T_[0][0] * V[j][0] + T_[0][1] * V[j][1] + T_[0][2] * V[j][2];
        const float u1 = T_[1][0] * V[j][0] + T_[1][1] * V[j][1] + T_[1][2] * V[j][2];
        dT0[j] = 2 * u0 * dL_dc_xx + u1 * dL_dc_xy;
        dT1[j] = 2 * u1 * dL_dc_yy + u0 * dL_dc_xy;
    }
    const float* vm = v.vm;
    const float dL_dJ00 = vm[0] * dT0[0] + vm[4] * dT0[1] + vm[8] * dT0[2];
    const float dL_dJ02 = vm[2] * dT0[0] + vm[6] * dT0[1] + vm[10] * dT0[2];
    const float dL_dJ11 = vm[1] * dT1[0] + vm[5] * dT1[1] + vm[9] * dT1[2];
    const float dL_dJ12 = vm[2] * dT1[0] + vm[6] * dT1[1] + vm[10] * dT1[2];
    const float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dL_dtx = x_grad_mul * -v.focal_x * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -v.focal_y * tz2 * dL_dJ12;
    float dL_dtz = -v.focal_x * tz2 * dL_dJ00 - v.focal_y * tz2 * dL_dJ11 + (2 * v.focal_x * t.x) * tz3 * dL_dJ02 +
                   (2 * v.focal_y * t.y) * tz3 * dL_dJ12;
    if (invdepth_path) dL_dtz -= acc1.w / (t.z * t.z);  // backward.cu:313-314
    // K9 assigns (backward.cu:324) ...
    float3 dmean;
    dmean.x = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
    dmean.y = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
    dmean.z = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;
    // ... K10 adds the 2D-mean path (backward.cu:425-439)
    const float* proj = v.pm;
    const float4 m_hom = xform4x4(mean, proj);
    const float m_w = 1.0f / (m_hom.w + 0.0000001f);
    const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
    dmean.x += (proj[0] * m_w - proj[3] * mul1) * o.g2x + (proj[1] * m_w - proj[3] * mul2) * o.g2y;
    dmean.y += (proj[4] * m_w - proj[7] * mul1) * o.g2x + (proj[5] * m_w - proj[7] * mul2) * o.g2y;
    dmean.z += (proj[8] * m_w - proj[11] * mul1) * o.g2x + (proj[9] * m_w - proj[11] * mul2) * o.g2y;
    o.dmean = dmean;
    if (has_scale) {
        // computeCov3D backward, backward.cu:329-392 (raw quaternion gradient, no normalisation Jacobian)
        const float r = q.x, x = q.y, y = q.z, z = q.w;
        float Rq[3][3];
        quat_rows(q, Rq);
        const float s[3] = {scale_modifier * sc.x, scale_modifier * sc.y, scale_modifier * sc.z};
        float Mm[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int a = 0; a < 3; a++) Mm[k][a] = s[k] * Rq[a][k];
        const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
        float dM[3][3];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) dM[a][b] = 2.0f * (Mm[a][0] * dS[0][b] + Mm[a][1] * dS[1][b] + Mm[a][2] * dS[2][b]);
        o.dscale.x = Rq[0][0] * dM[0][0] + Rq[1][0] * dM[0][1] + Rq[2][0] * dM[0][2];
        o.dscale.y = Rq[0][1] * dM[1][0] + Rq[1][1] * dM[1][1] + Rq[2][1] * dM[1][2];
        o.dscale.z = Rq[0][2] * dM[2][0] + Rq[1][2] * dM[2][1] + Rq[2][2] * dM[2][2];
        float G[3][3];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) G[a][b] = s[a] * dM[a][b];
        o.drot.x = 2 * z * (G[0][1] - G[1][0]) + 2 * y * (G[2][0] - G[0][2]) + 2 * x * (G[1][2] - G[2][1]);
        o.drot.y = 2 * y * (G[1][0] + G[0][1]) + 2 * z * (G[2][0] + G[0][2]) + 2 * r * (G[1][2] - G[2][1]) - 4 * x * (G[2][2] + G[1][1]);
        o.drot.z = 2 * x * (G[1][0] + G[0][1]) + 2 * r * (G[2][0] - G[0][2]) + 2 * z * (G[1][2] + G[2][1]) - 4 * y * (G[2][2] + G[0][0]);
        o.drot.w = 2 * r * (G[0][1] - G[1][0]) + 2 * x * (G[2][0] + G[0][2]) + 2 * y * (G[1][2] + G[2][1]) - 4 * z * (G[1][1] + G[0][0]);
    }
}

}  // namespace cgs
