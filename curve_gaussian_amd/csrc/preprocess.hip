// Per-splat kernels: forward preprocess (reference K1: forward.cu:155-274) fused with per-tile counting, and the
// fused backward of the projection / covariance chain (reference K9 + K10: backward.cu:146-325, :397-448).
// One thread per splat; memory-bound; no MFMA (no dense contraction on this path).
#include "kernels.h"

namespace cgs {

// Spherical-harmonics constants, reference auxiliary.h:21-38
__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

// Single-channel SH -> colour, reference forward.cu:20-75
__device__ float sh_to_color(int idx, int deg, int max_coeffs, const float3 pos, const float3 campos,
                             const float* __restrict__ shs, uint8_t* __restrict__ clamped) {
    float3 dir = make_float3(pos.x - campos.x, pos.y - campos.y, pos.z - campos.z);
    const float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
    dir.x /= len; dir.y /= len; dir.z /= len;
    const float* sh = shs + (size_t)idx * max_coeffs;
    float result = SH_C0 * sh[0];
    if (deg > 0) {
        const float x = dir.x, y = dir.y, z = dir.z;
        result = result - SH_C1 * y * sh[1] + SH_C1 * z * sh[2] - SH_C1 * x * sh[3];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            result = result + SH_C2[0] * xy * sh[4] + SH_C2[1] * yz * sh[5] + SH_C2[2] * (2.0f * zz - xx - yy) * sh[6] +
                     SH_C2[3] * xz * sh[7] + SH_C2[4] * (xx - yy) * sh[8];
            if (deg > 2) {
                result = result + SH_C3[0] * y * (3.0f * xx - yy) * sh[9] + SH_C3[1] * xy * z * sh[10] +
                         SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11] +
                         SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12] +
                         SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13] + SH_C3[5] * z * (xx - yy) * sh[14] +
                         SH_C3[6] * x * (xx - 3.0f * yy) * sh[15];
            }
        }
    }
    result += 0.5f;
    clamped[idx] = (result < 0);
    return fmaxf(result, 0.0f);
}

// Backward of sh_to_color, reference backward.cu:23-141: writes dL_dsh[idx, :], returns dL/dmean contribution.
__device__ float3 sh_to_color_bwd(int idx, int deg, int max_coeffs, const float3 pos, const float3 campos,
                                  const float* __restrict__ shs, const uint8_t* __restrict__ clamped,
                                  float dL_dRGB, float* __restrict__ dL_dshs) {
    const float3 dir_orig = make_float3(pos.x - campos.x, pos.y - campos.y, pos.z - campos.z);
    const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
    const float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
    const float* sh = shs + (size_t)idx * max_coeffs;
    dL_dRGB *= clamped[idx] ? 0.f : 1.f;
    float dRGBdx = 0, dRGBdy = 0, dRGBdz = 0;
    float* dL_dsh = dL_dshs + (size_t)idx * max_coeffs;
    dL_dsh[0] = SH_C0 * dL_dRGB;
    if (deg > 0) {
        dL_dsh[1] = (-SH_C1 * y) * dL_dRGB; dL_dsh[2] = (SH_C1 * z) * dL_dRGB; dL_dsh[3] = (-SH_C1 * x) * dL_dRGB;
        dRGBdx = -SH_C1 * sh[3]; dRGBdy = -SH_C1 * sh[1]; dRGBdz = SH_C1 * sh[2];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            dL_dsh[4] = (SH_C2[0] * xy) * dL_dRGB; dL_dsh[5] = (SH_C2[1] * yz) * dL_dRGB;
            dL_dsh[6] = (SH_C2[2] * (2.f * zz - xx - yy)) * dL_dRGB; dL_dsh[7] = (SH_C2[3] * xz) * dL_dRGB;
            dL_dsh[8] = (SH_C2[4] * (xx - yy)) * dL_dRGB;
            dRGBdx += SH_C2[0] * y * sh[4] + SH_C2[2] * 2.f * -x * sh[6] + SH_C2[3] * z * sh[7] + SH_C2[4] * 2.f * x * sh[8];
            dRGBdy += SH_C2[0] * x * sh[4] + SH_C2[1] * z * sh[5] + SH_C2[2] * 2.f * -y * sh[6] + SH_C2[4] * 2.f * -y * sh[8];
            dRGBdz += SH_C2[1] * y * sh[5] + SH_C2[2] * 2.f * 2.f * z * sh[6] + SH_C2[3] * x * sh[7];
            if (deg > 2) {
                dL_dsh[9] = (SH_C3[0] * y * (3.f * xx - yy)) * dL_dRGB; dL_dsh[10] = (SH_C3[1] * xy * z) * dL_dRGB;
                dL_dsh[11] = (SH_C3[2] * y * (4.f * zz - xx - yy)) * dL_dRGB;
                dL_dsh[12] = (SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * dL_dRGB;
                dL_dsh[13] = (SH_C3[4] * x * (4.f * zz - xx - yy)) * dL_dRGB;
                dL_dsh[14] = (SH_C3[5] * z * (xx - yy)) * dL_dRGB;
                dL_dsh[15] = (SH_C3[6] * x * (xx - 3.f * yy)) * dL_dRGB;
                dRGBdx += (SH_C3[0] * sh[9] * 3.f * 2.f * xy + SH_C3[1] * sh[10] * yz + SH_C3[2] * sh[11] * -2.f * xy +
                           SH_C3[3] * sh[12] * -3.f * 2.f * xz + SH_C3[4] * sh[13] * (-3.f * xx + 4.f * zz - yy) +
                           SH_C3[5] * sh[14] * 2.f * xz + SH_C3[6] * sh[15] * 3.f * (xx - yy));
                dRGBdy += (SH_C3[0] * sh[9] * 3.f * (xx - yy) + SH_C3[1] * sh[10] * xz +
                           SH_C3[2] * sh[11] * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * sh[12] * -3.f * 2.f * yz +
                           SH_C3[4] * sh[13] * -2.f * xy + SH_C3[5] * sh[14] * -2.f * yz +
                           SH_C3[6] * sh[15] * -3.f * 2.f * xy);
                dRGBdz += (SH_C3[1] * sh[10] * xy + SH_C3[2] * sh[11] * 4.f * 2.f * yz +
                           SH_C3[3] * sh[12] * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * sh[13] * 4.f * 2.f * xz +
                           SH_C3[5] * sh[14] * (xx - yy));
            }
        }
    }
    const float3 dd = make_float3(dRGBdx * dL_dRGB, dRGBdy * dL_dRGB, dRGBdz * dL_dRGB);
    // dnormvdv, reference auxiliary.h:119-129
    const float3 v = dir_orig;
    const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    return make_float3(((+sum2 - v.x * v.x) * dd.x - v.y * v.x * dd.y - v.z * v.x * dd.z) * invsum32,
                       (-v.x * v.y * dd.x + (sum2 - v.y * v.y) * dd.y - v.z * v.y * dd.z) * invsum32,
                       (-v.x * v.z * dd.x - v.y * v.z * dd.y + (sum2 - v.z * v.z) * dd.z) * invsum32);
}

// ------------------------------------------------------------------------------------------------ forward
__global__ void __launch_bounds__(256) k_preprocess_fwd(
    int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales, float scale_modifier,
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
    uint8_t* __restrict__ clamped, const float* __restrict__ cov3D_precomp, const float* __restrict__ colors_precomp,
    const float* __restrict__ all_map, const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
    const float* __restrict__ cam_pos, int W, int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y,
    int* __restrict__ radii, SplatRec* __restrict__ rec, float* __restrict__ rgb, int grid_x, int grid_y,
    uint32_t* __restrict__ tile_count, int antialiasing, int cull, float* __restrict__ grad_acc,
    uint32_t* __restrict__ clear_words, uint32_t n_clear) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    // Folded zero fills (each was a separate ~5 us launch): this splat's gradient accumulator record for the backward
    // compositor (k_preprocess_bwd leaves it zero again), and -- bucket binning only, where this kernel does not count --
    // the tile histogram / cursors / status words that the scatter kernel (next launch) accumulates into.
    if (idx < P) {
        float4* accp = reinterpret_cast<float4*>(grad_acc + (size_t)idx * ACC_STRIDE);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < ACC_STRIDE / 4; k++) accp[k] = z;
    }
    for (uint32_t i = (uint32_t)(blockIdx.x * blockDim.x + threadIdx.x); i < n_clear; i += gridDim.x * blockDim.x) clear_words[i] = 0u;
    // radius 0 == "not processed further" (forward.cu:187-190)
    int out_radius = 0;
    uint2 rmin = make_uint2(0, 0), rmax = make_uint2(0, 0);
    float4 ra = make_float4(0.f, 0.f, 1.f, 0.f);  // (px, py, conic.x, conic.y) of this lane's splat
    float conic_z = 1.f, tau2 = -1.f;
    if (idx < P) do {
        const float3 p_orig = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
        const float3 p_view = xform4x3(p_orig, viewmatrix);
        if (p_view.z <= 0.2f) break;  // near cull only, auxiliary.h:166
        const float4 p_hom = xform4x4(p_orig, projmatrix);
        const float p_w = 1.0f / (p_hom.w + 0.0000001f);
        const float3 p_proj = make_float3(p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w);
        float cov3D[6];
        if (cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = cov3D_precomp[6 * (size_t)idx + i];
        } else {
            const float3 s = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
            const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
            cov3d_from_scale_rot(s, scale_modifier, q, cov3D);
        }
        float3 t, cov;
        float Mt[2][3], txtz, tytz;
        cov2d_terms(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, t, Mt, cov, txtz, tytz);
        constexpr float h_var = 0.3f;
        const float det_cov = cov.x * cov.z - cov.y * cov.y;
        cov.x += h_var;
        cov.z += h_var;
        const float det_cov_plus_h_cov = cov.x * cov.z - cov.y * cov.y;
        float h_convolution_scaling = 1.0f;
        if (antialiasing) h_convolution_scaling = sqrtf(fmaxf(0.000025f, det_cov / det_cov_plus_h_cov));
        const float det = det_cov_plus_h_cov;
        if (det == 0.0f) break;
        const float det_inv = 1.f / det;
        const float3 conic = make_float3(cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv);
        const float mid = 0.5f * (cov.x + cov.z);
        const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        const float px = ndc2pix(p_proj.x, W), py = ndc2pix(p_proj.y, H);
        uint2 r0, r1;
        get_rect(px, py, (int)my_radius, grid_x, grid_y, r0, r1);
        if ((r1.x - r0.x) * (r1.y - r0.y) == 0) break;
        float color;
        if (colors_precomp) {
            color = colors_precomp[idx];
        } else {
            color = sh_to_color(idx, D, M, p_orig, make_float3(cam_pos[0], cam_pos[1], cam_pos[2]), shs, clamped);
            rgb[idx] = color;
        }
        SplatRec r;
        r.a = make_float4(px, py, conic.x, conic.y);
        const float op_eff = opacities[idx] * h_convolution_scaling;
        r.b = make_float4(conic.z, op_eff, color, 1.f / p_view.z);
        r.c = all_map ? reinterpret_cast<const float4*>(all_map)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        // tau2 = 2 ln(255 * opacity): alpha >= 1/255  <=>  conic quadratic form <= tau2 (used by the quadrant culling)
        r.d = make_float4(p_view.z, my_radius, 2.f * logf(255.f * op_eff), 0.f);
        rec[idx] = r;
        ra = r.a;
        conic_z = r.b.x;
        tau2 = r.d.z;
        out_radius = (int)my_radius;
        rmin = r0;
        rmax = r1;
    } while (false);
    if (idx < P) radii[idx] = out_radius;
    // per-tile instance counts (replaces the reference's per-splat scan K2 + duplicateWithKeys offsets)
    // (with `cull`, only the tiles the splat can reach with alpha >= 1/255; k_scatter takes the identical decision)
    if (!tile_count) return;  // single-pass bucket binning: k_scatter<true> counts while it scatters
    for_each_rect_tile_coop(out_radius > 0, rmin, rmax, [&](int src, uint32_t tx, uint32_t ty) {
        if (cull && !tile_reach_det(readlane_f(ra.x, src), readlane_f(ra.y, src), readlane_f(ra.z, src),
                                    readlane_f(ra.w, src), readlane_f(conic_z, src), readlane_f(tau2, src),
                                    (float)(tx * TILE), (float)(ty * TILE)))
            return;
        atomicAdd(&tile_count[ty * (uint32_t)grid_x + tx], 1u);
    });
}

// reference checkFrustum, rasterizer_impl.cu:54-66
__global__ void k_mark_visible(int P, const float* __restrict__ means3D, const float* __restrict__ viewmatrix,
                               uint8_t* __restrict__ present) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float3 p = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    present[idx] = xform4x3(p, viewmatrix).z > 0.2f;
}

// ------------------------------------------------------------------------------------------------ backward
// Fused K9 (computeCov2DCUDA) + K10 (preprocessCUDA bwd).  cov3D is recomputed from scale/rotation instead of
// being stored by the forward (legal per SURVEY quirk 22).  Every splat writes dL_dmean3D / dL_dcov3D / dL_dscale /
// dL_drot (zeros when culled), so these four outputs need no zero-fill.
__global__ void __launch_bounds__(256) k_preprocess_bwd(
    int P, int D, int M, const float* __restrict__ means3D, const int* __restrict__ radii,
    const float* __restrict__ shs, const uint8_t* __restrict__ clamped, const float* __restrict__ opacities,
    const float* __restrict__ scales, const float* __restrict__ rotations, float scale_modifier,
    const float* __restrict__ cov3D_precomp, const float* __restrict__ viewmatrix,
    const float* __restrict__ projmatrix, const float* __restrict__ cam_pos, float focal_x, float focal_y,
    float tan_fovx, float tan_fovy, int W, int H, const SplatRec* __restrict__ rec,
    float* __restrict__ grad_acc, float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
    float* __restrict__ dL_dinvdepth, float* __restrict__ dL_dopacity, float* __restrict__ dL_dmean3D,
    float* __restrict__ dL_dcolor, float* __restrict__ dL_dall_map, float* __restrict__ dL_dcov3D,
    float* __restrict__ dL_dsh, float* __restrict__ dL_dscale, float* __restrict__ dL_drot, int antialiasing) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    // ---- finish the compositor's per-splat sums: raw moments -> dL/dmean2D (NDC-scaled), dL/dconic, dL/dopacity
    // The record is handed back zeroed (the forward zeroed it the first time): a second backward over the same forward
    // state accumulates from zero again without a separate fill launch.
    float4* accp = reinterpret_cast<float4*>(grad_acc + (size_t)idx * ACC_STRIDE);
    const float4 acc0 = accp[0], acc1 = accp[1], acc2 = accp[2];  // {Sg,Sx,Sy,Sxx} {Sxy,Syy,col,invd} {all_map}
    {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        accp[0] = z; accp[1] = z; accp[2] = z;
    }
    const bool vis = radii[idx] > 0;
    float g2x = 0.f, g2y = 0.f, dcx = 0.f, dcy = 0.f, dcz = 0.f;
    float dopac = acc0.x;
    if (vis) {
        const float4 ra = rec[idx].a, rb = rec[idx].b;
        const float cA = ra.z, cB = ra.w, cC = rb.x, op = rb.y;
        g2x = -op * (cA * acc0.y + cB * acc0.z) * (float)(0.5 * W);  // backward.cu:542-543, 659-664
        g2y = -op * (cC * acc0.z + cB * acc0.y) * (float)(0.5 * H);
        dcx = -0.5f * op * acc0.w;                                   // backward.cu:667-669
        dcy = -0.5f * op * acc1.x;
        dcz = -0.5f * op * acc1.y;
    }
    dL_dmean2D[3 * idx] = g2x; dL_dmean2D[3 * idx + 1] = g2y; dL_dmean2D[3 * idx + 2] = 0.f;
    if (dL_dconic) reinterpret_cast<float4*>(dL_dconic)[idx] = make_float4(dcx, dcy, 0.f, dcz);
    if (dL_dcolor) dL_dcolor[idx] = acc1.z;
    if (dL_dinvdepth) dL_dinvdepth[idx] = acc1.w;
    if (dL_dall_map) reinterpret_cast<float4*>(dL_dall_map)[idx] = acc2;
    float3 dmean = make_float3(0.f, 0.f, 0.f);
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float3 dscale = make_float3(0.f, 0.f, 0.f);
    float4 drot = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vis) {
        const float3 mean = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
        float cov3D[6];
        float3 sc = make_float3(0.f, 0.f, 0.f);
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = cov3D_precomp[6 * (size_t)idx + i];
        } else {
            sc = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
            q = reinterpret_cast<const float4*>(rotations)[idx];
            cov3d_from_scale_rot(sc, scale_modifier, q, cov3D);
        }
        float3 t, cov;
        float T_[2][3], txtz, tytz;
        cov2d_terms(mean, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, t, T_, cov, txtz, tytz);
        const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
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
            const float dL_dopacity_v = dopac;
            const float d_h_convolution_scaling = dL_dopacity_v * opacities[idx];
            dopac = dL_dopacity_v * h_convolution_scaling;
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
        const float* vm = viewmatrix;
        const float dL_dJ00 = vm[0] * dT0[0] + vm[4] * dT0[1] + vm[8] * dT0[2];
        const float dL_dJ02 = vm[2] * dT0[0] + vm[6] * dT0[1] + vm[10] * dT0[2];
        const float dL_dJ11 = vm[1] * dT1[0] + vm[5] * dT1[1] + vm[9] * dT1[2];
        const float dL_dJ12 = vm[2] * dT1[0] + vm[6] * dT1[1] + vm[10] * dT1[2];
        const float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
        const float dL_dtx = x_grad_mul * -focal_x * tz2 * dL_dJ02;
        const float dL_dty = y_grad_mul * -focal_y * tz2 * dL_dJ12;
        float dL_dtz = -focal_x * tz2 * dL_dJ00 - focal_y * tz2 * dL_dJ11 + (2 * focal_x * t.x) * tz3 * dL_dJ02 +
                       (2 * focal_y * t.y) * tz3 * dL_dJ12;
        if (dL_dinvdepth) dL_dtz -= acc1.w / (t.z * t.z);  // backward.cu:313-314
        // K9 assigns (backward.cu:324) ...
        dmean.x = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
        dmean.y = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
        dmean.z = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;
        // ... K10 adds the 2D-mean path (backward.cu:425-439)
        const float* proj = projmatrix;
        const float4 m_hom = xform4x4(mean, proj);
        const float m_w = 1.0f / (m_hom.w + 0.0000001f);
        const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
        dmean.x += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dmean.y += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dmean.z += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        if (shs) {
            const float3 dm = sh_to_color_bwd(idx, D, M, mean, make_float3(cam_pos[0], cam_pos[1], cam_pos[2]), shs,
                                              clamped, acc1.z, dL_dsh);
            dmean.x += dm.x; dmean.y += dm.y; dmean.z += dm.z;
        }
        if (scales) {
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
            dscale.x = Rq[0][0] * dM[0][0] + Rq[1][0] * dM[0][1] + Rq[2][0] * dM[0][2];
            dscale.y = Rq[0][1] * dM[1][0] + Rq[1][1] * dM[1][1] + Rq[2][1] * dM[1][2];
            dscale.z = Rq[0][2] * dM[2][0] + Rq[1][2] * dM[2][1] + Rq[2][2] * dM[2][2];
            float G[3][3];
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) G[a][b] = s[a] * dM[a][b];
            drot.x = 2 * z * (G[0][1] - G[1][0]) + 2 * y * (G[2][0] - G[0][2]) + 2 * x * (G[1][2] - G[2][1]);
            drot.y = 2 * y * (G[1][0] + G[0][1]) + 2 * z * (G[2][0] + G[0][2]) + 2 * r * (G[1][2] - G[2][1]) - 4 * x * (G[2][2] + G[1][1]);
            drot.z = 2 * x * (G[1][0] + G[0][1]) + 2 * r * (G[2][0] - G[0][2]) + 2 * z * (G[1][2] + G[2][1]) - 4 * y * (G[2][2] + G[0][0]);
            drot.w = 2 * r * (G[0][1] - G[1][0]) + 2 * x * (G[2][0] + G[0][2]) + 2 * y * (G[1][2] + G[2][1]) - 4 * z * (G[1][1] + G[0][0]);
        }
    }
    dL_dopacity[idx] = dopac;
    dL_dmean3D[3 * idx] = dmean.x; dL_dmean3D[3 * idx + 1] = dmean.y; dL_dmean3D[3 * idx + 2] = dmean.z;
#pragma unroll
    for (int i = 0; i < 6; i++) dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
    if (dL_dscale) { dL_dscale[3 * idx] = dscale.x; dL_dscale[3 * idx + 1] = dscale.y; dL_dscale[3 * idx + 2] = dscale.z; }
    if (dL_drot) reinterpret_cast<float4*>(dL_drot)[idx] = drot;
}


// ------------------------------------------------------------------------------------------------ launchers
void launch_preprocess_fwd(hipStream_t s, int P, int D, int M, const float* means3D, const float* scales,
                           float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                           uint8_t* clamped, const float* cov3D_precomp, const float* colors_precomp,
                           const float* all_map, const float* viewmatrix, const float* projmatrix,
                           const float* cam_pos, int W, int H, float tan_fovx, float tan_fovy, float focal_x,
                           float focal_y, int* radii, SplatRec* rec, float* rgb, int grid_x, int grid_y,
                           uint32_t* tile_count, int antialiasing, int cull, float* grad_acc, uint32_t* clear_words,
                           size_t n_clear) {
    ProfScope p("preprocess_fwd", s);
    hipLaunchKernelGGL(k_preprocess_fwd, dim3((P + 255) / 256), dim3(256), 0, s, P, D, M, means3D, scales,
                       scale_modifier, rotations, opacities, shs, clamped, cov3D_precomp, colors_precomp, all_map,
                       viewmatrix, projmatrix, cam_pos, W, H, tan_fovx, tan_fovy, focal_x, focal_y, radii, rec, rgb,
                       grid_x, grid_y, tile_count, antialiasing, cull, grad_acc, clear_words, (uint32_t)n_clear);
}
void launch_mark_visible(hipStream_t s, int P, const float* means3D, const float* viewmatrix, uint8_t* present) {
    ProfScope p("mark_visible", s);
    hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, present);
}
void launch_preprocess_bwd(hipStream_t s, int P, int D, int M, const float* means3D, const int* radii,
                           const float* shs, const uint8_t* clamped, const float* opacities, const float* scales,
                           const float* rotations, float scale_modifier, const float* cov3D_precomp,
                           const float* viewmatrix, const float* projmatrix, const float* cam_pos, float focal_x,
                           float focal_y, float tan_fovx, float tan_fovy, int W, int H, const SplatRec* rec,
                           float* grad_acc, float* dL_dmean2D, float* dL_dconic, float* dL_dinvdepth,
                           float* dL_dopacity, float* dL_dmean3D, float* dL_dcolor, float* dL_dall_map,
                           float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int antialiasing) {
    ProfScope p("preprocess_bwd", s);
    hipLaunchKernelGGL(k_preprocess_bwd, dim3((P + 255) / 256), dim3(256), 0, s, P, D, M, means3D, radii, shs, clamped,
                       opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, cam_pos,
                       focal_x, focal_y, tan_fovx, tan_fovy, W, H, rec, grad_acc, dL_dmean2D, dL_dconic, dL_dinvdepth,
                       dL_dopacity, dL_dmean3D, dL_dcolor, dL_dall_map, dL_dcov3D, dL_dsh, dL_dscale, dL_drot,
                       antialiasing);
}

}  // namespace cgs
