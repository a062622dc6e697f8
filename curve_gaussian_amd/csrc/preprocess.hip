// Per-splat kernels: forward preprocess (reference K1: forward.cu:155-274) fused with per-tile counting, and the
// fused backward of the projection / covariance chain (reference K9 + K10: backward.cu:146-325, :397-448).
// One thread per splat; memory-bound; no MFMA (no dense contraction on this path).
#include "splat_math.h"

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
        float cov3D[6];
        if (cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = cov3D_precomp[6 * (size_t)idx + i];
        } else {
            const float3 s = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
            const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
            cov3d_from_scale_rot(s, scale_modifier, q, cov3D);
        }
        const ViewParams vp{viewmatrix, projmatrix, tan_fovx, tan_fovy, focal_x, focal_y, W, H, grid_x, grid_y};
        SplatGeom g;
        if (!splat_geometry(p_orig, cov3D, vp, antialiasing, g)) break;
        float color;
        if (colors_precomp) {
            color = colors_precomp[idx];
        } else {
            color = sh_to_color(idx, D, M, p_orig, make_float3(cam_pos[0], cam_pos[1], cam_pos[2]), shs, clamped);
            rgb[idx] = color;
        }
        const SplatRec r = splat_record(g, opacities[idx], color,
                                        all_map ? reinterpret_cast<const float4*>(all_map)[idx] : make_float4(0.f, 0.f, 0.f, 0.f));
        rec[idx] = r;
        ra = r.a;
        conic_z = r.b.x;
        tau2 = r.d.z;
        out_radius = (int)g.radius;
        rmin = g.rmin;
        rmax = g.rmax;
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
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
    float3 mean = make_float3(0.f, 0.f, 0.f), sc = mean;
    float4 q = ra;
    float cov3D[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (vis) {
        ra = rec[idx].a;
        rb = rec[idx].b;
        mean = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
        if (cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = cov3D_precomp[6 * (size_t)idx + i];
        } else {
            sc = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
            q = reinterpret_cast<const float4*>(rotations)[idx];
            cov3d_from_scale_rot(sc, scale_modifier, q, cov3D);
        }
    }
    const ViewParams vp{viewmatrix, projmatrix, tan_fovx, tan_fovy, focal_x, focal_y, W, H, 0, 0};
    SplatGrads o;
    splat_backward(acc0, acc1, vis, ra, rb, mean, cov3D, sc, q, scales != nullptr, scale_modifier,
                   (vis && antialiasing) ? opacities[idx] : 0.f, vp, antialiasing, dL_dinvdepth != nullptr, o);
    const float g2x = o.g2x, g2y = o.g2y, dcx = o.dcx, dcy = o.dcy, dcz = o.dcz, dopac = o.dopac;
    float3 dmean = o.dmean;
    const float* dcov = o.dcov;
    const float3 dscale = o.dscale;
    const float4 drot = o.drot;
    dL_dmean2D[3 * idx] = g2x; dL_dmean2D[3 * idx + 1] = g2y; dL_dmean2D[3 * idx + 2] = 0.f;
    if (dL_dconic) reinterpret_cast<float4*>(dL_dconic)[idx] = make_float4(dcx, dcy, 0.f, dcz);
    if (dL_dcolor) dL_dcolor[idx] = acc1.z;
    if (dL_dinvdepth) dL_dinvdepth[idx] = acc1.w;
    if (dL_dall_map) reinterpret_cast<float4*>(dL_dall_map)[idx] = acc2;
    if (vis && shs) {
        const float3 dm = sh_to_color_bwd(idx, D, M, mean, make_float3(cam_pos[0], cam_pos[1], cam_pos[2]), shs, clamped,
                                          acc1.z, dL_dsh);
        dmean.x += dm.x; dmean.y += dm.y; dmean.z += dm.z;
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
