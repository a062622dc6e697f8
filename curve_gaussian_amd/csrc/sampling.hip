// Curve -> Gaussian sampling and per-view splat attributes, forward + hand-written backward.
//
// Replaces ~40 tiny PyTorch kernels per step (plus their autograd graph) of the reference:
//   GaussianCurveModel.prepare_scaling_rot   scene/gaussian_curve_model.py:180-198
//     get_curve_gaussians :70-78, get_curve_tangent :80-89, rot_to_quat_batch utils/general_utils.py:33-86
//   get_rotation (F.normalize) :121-122, get_opacity :108-110, get_main_axis :99-105,
//   straight-through mask + all_map build  gaussian_renderer/__init__.py:72-76,98-104
//
// One thread per SPLAT (sample i of curve b, splat index = b*m + i).  The reference divides v1 and v2 by the GLOBAL
// Frobenius norm of the whole [P,3] tensor, so the forward is three passes
//   F1: S1 = sum |cross(tan, up)|^2      F2: S2 = sum |cross(tan, v1)|^2      F3: outputs
// and the backward three more (the norms couple every sample to every other one)
//   B1: D2 = sum <g_v2, c2v>             B2: D1 = sum <g_v1, c1v>             B3: dL/d{control points, width}
// Grid-wide sums are block-reduced and accumulated with one f64 atomic per block (order effects ~1e-16).
#include <algorithm>

#include "curve_math.h"

namespace cgs {

// ------------------------------------------------------------------------------------------------ forward
constexpr int F12_BLOCK = 1024;   // upper bound; the launch picks the workgroup size by splat count (f12_threads)
__global__ void __launch_bounds__(F12_BLOCK) k_sample_f12(int B, int m, const float* __restrict__ cp,
                                                         const uint8_t* __restrict__ is_bezier,
                                                         const SampleCoef* __restrict__ coef, double* __restrict__ norms) {
    // launched on exactly NORM_SLOTS workgroups: workgroup k owns slot k of the three forward sums (plain stores) and
    // clears slot k of the backward's two
    __shared__ SampleCoef s_coef[MAX_M];
    __shared__ double s_part[3][F12_BLOCK / 64];
    const float* src = reinterpret_cast<const float*>(coef);
    float* dst = reinterpret_cast<float*>(s_coef);
    for (int t = threadIdx.x; t < m * 16; t += blockDim.x) dst[t] = src[t];
    __syncthreads();
    double a1 = 0, a2 = 0, a3 = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < B * m; p += gridDim.x * blockDim.x) {
        const int b = p / m, i = p - b * m;
        const CurveCP c = load_curve(cp, is_bezier, b);
        const V3 t = curve_tangent(c, s_coef[i]);
        const V3 c1 = {t.y, -t.x, 0.f};                 // cross(tan, (0,0,1))
        const V3 x = cross(t, c1);                      // N1 * c2v
        a1 += (double)(t.y * t.y + t.x * t.x);
        a2 += (double)(x.x * x.x + x.y * x.y + x.z * x.z);
        a3 += (double)dot(cross(x, t), c1);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a1 += __shfl_xor(a1, off, 64);
        a2 += __shfl_xor(a2, off, 64);
        a3 += __shfl_xor(a3, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_part[0][threadIdx.x >> 6] = a1; s_part[1][threadIdx.x >> 6] = a2; s_part[2][threadIdx.x >> 6] = a3;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = 0;
        for (int wv = 0; wv < (int)(blockDim.x >> 6); wv++) t += s_part[threadIdx.x][wv];
        norms[threadIdx.x * NORM_SLOTS + blockIdx.x] = t;
    } else if (threadIdx.x < NQ_ALL) {
        norms[threadIdx.x * NORM_SLOTS + blockIdx.x] = 0.0;
    }
}

__global__ void __launch_bounds__(256) k_sample_f3(int B, int m, const float* __restrict__ cp,
                                                   const float* __restrict__ width,
                                                   const uint8_t* __restrict__ is_bezier,
                                                   const SampleCoef* __restrict__ coef, float eps,
                                                   const double* __restrict__ norms, float* __restrict__ xyz,
                                                   float* __restrict__ rot, float* __restrict__ scaling) {
    __shared__ SampleCoef s_coef[MAX_M];
    __shared__ BlockConst s_bc;
    stage_consts(coef, m, norms, s_coef, &s_bc);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B * m) return;
    const int b = p / m, i = p - b * m;
    const float N1 = s_bc.N1, N2 = s_bc.N2;
    const CurveCP c = load_curve(cp, is_bezier, b);
    const float w = expf(width[b]);
    const SampleFwd s = sample_forward(c, s_coef[i], N1, N2, eps);
    float M[3][3], q[4];
    rot_matrix(s, M);
    quat_forward(M, q);
    xyz[3 * p] = s.xyz.x; xyz[3 * p + 1] = s.xyz.y; xyz[3 * p + 2] = s.xyz.z;
    reinterpret_cast<float4*>(rot)[p] = make_float4(q[0], q[1], q[2], q[3]);
    scaling[3 * p] = s.dist; scaling[3 * p + 1] = w; scaling[3 * p + 2] = w;
}

// ------------------------------------------------------------------------------------------------ backward
// PASS 1: the two global sums D2 and A (one pass, see NORM layout);  PASS 3: write dL/dcurve_points, dL/dwidth.
// Blocks hold CURVES_PER_BLOCK whole curves (CURVES_PER_BLOCK * m threads are active); pass 3 reduces the per-sample
// contributions to the 13 per-curve outputs through LDS.
template <int PASS>
__global__ void __launch_bounds__(SAMPLE_BLOCK) k_sample_bwd(int B, int m, int curves_per_block,
                                                             const float* __restrict__ cp,
                                                             const float* __restrict__ width,
                                                             const uint8_t* __restrict__ is_bezier,
                                                             const SampleCoef* __restrict__ coef, float eps,
                                                             double* __restrict__ norms,
                                                             const float* __restrict__ g_xyz,
                                                             const float* __restrict__ g_rot,
                                                             const float* __restrict__ g_scaling,
                                                             float* __restrict__ g_cp, float* __restrict__ g_width,
                                                             float* __restrict__ gv_cache, int accumulate = 0) {
    __shared__ float s_part[PASS == 3 ? 13 : 1][SAMPLE_BLOCK + 1];
    __shared__ SampleCoef s_coef[MAX_M];
    __shared__ BlockConst s_bc;
    stage_consts(coef, m, norms, s_coef, &s_bc);
    const float N1 = s_bc.N1, N2 = s_bc.N2;
    const float D2 = PASS >= 2 ? s_bc.D2 : 0.f, D1 = PASS >= 3 ? s_bc.D1 : 0.f;
    double acc = 0, acc_a = 0;
    V3 gp0 = {0, 0, 0}, gp1 = {0, 0, 0}, gp2 = {0, 0, 0}, gp3 = {0, 0, 0};
    float gw = 0.f;
    // pass 1: grid-stride over splats; pass 3: whole curves per block
    int b, i;
    bool valid;
    int sp = blockIdx.x * blockDim.x + threadIdx.x;
    if (PASS == 3) {
        const int lc = threadIdx.x / m;
        i = threadIdx.x - lc * m;
        b = blockIdx.x * curves_per_block + lc;
        valid = lc < curves_per_block && b < B;
    } else {
        valid = sp < B * m;
        b = valid ? sp / m : 0;
        i = sp - b * m;
    }
    while (valid) {
        const CurveCP c = load_curve(cp, is_bezier, b);
        const float w = expf(width[b]);
        const SampleCoef k = s_coef[i];
        const size_t p = (size_t)b * m + i;
        const SampleFwd s = sample_forward(c, k, N1, N2, eps);
        V3 g_v0 = {0, 0, 0}, g_v1 = {0, 0, 0}, g_v2 = {0, 0, 0};
        if (PASS > 1 && g_rot) {  // pass 1 cached dL/d{v0,v1,v2} (the quaternion backward) for passes 2 and 3
            const float* gv = gv_cache + p;   // planes [9][P]: a wave's accesses are contiguous
            const size_t PS = (size_t)B * m;
            g_v0 = {gv[0], gv[PS], gv[2 * PS]}; g_v1 = {gv[3 * PS], gv[4 * PS], gv[5 * PS]}; g_v2 = {gv[6 * PS], gv[7 * PS], gv[8 * PS]};
        } else if (g_rot) {
            float M[3][3], q[4], gM[3][3];
            rot_matrix(s, M);
            const QuatFwd f = quat_forward(M, q);
            const float4 gq = reinterpret_cast<const float4*>(g_rot)[p];
            const float go[4] = {gq.x, gq.y, gq.z, gq.w};
            quat_backward(f, go, gM);
            g_v0 = {gM[0][0], gM[1][0], gM[2][0]};
            g_v1 = {gM[0][1], gM[1][1], gM[2][1]};
            g_v2 = {gM[0][2], gM[1][2], gM[2][2]};
            float* gv = gv_cache + p;         // planes [9][P]
            const size_t PS = (size_t)B * m;
            gv[0] = g_v0.x; gv[PS] = g_v0.y; gv[2 * PS] = g_v0.z; gv[3 * PS] = g_v1.x; gv[4 * PS] = g_v1.y; gv[5 * PS] = g_v1.z;
            gv[6 * PS] = g_v2.x; gv[7 * PS] = g_v2.y; gv[8 * PS] = g_v2.z;
        }
        if (PASS == 1) {
            acc += (double)dot(g_v2, s.c2v);
            acc_a += (double)dot(g_v1, s.c1v) + (double)((1.f / N2) * dot(cross(g_v2, s.tan), s.c1v));
            sp += gridDim.x * blockDim.x;
            valid = sp < B * m;
            b = valid ? sp / m : 0;
            i = sp - b * m;
            continue;
        }
        if (PASS == 2) {
            const float iN2 = 1.f / N2;
            const V3 g_c2v = iN2 * g_v2 - (D2 * iN2 * iN2 * iN2) * s.c2v;
            const V3 g_v1t = g_v1 + cross(g_c2v, s.tan);
            acc += (double)dot(g_v1t, s.c1v);
            sp += gridDim.x * blockDim.x;
            valid = sp < B * m;
            b = valid ? sp / m : 0;
            i = sp - b * m;
            continue;
        }
        V3 g_x = {0, 0, 0}, g_s = {0, 0, 0};
        if (g_xyz) g_x = {g_xyz[3 * p], g_xyz[3 * p + 1], g_xyz[3 * p + 2]};
        if (g_scaling) g_s = {g_scaling[3 * p], g_scaling[3 * p + 1], g_scaling[3 * p + 2]};
        const CurveGrad cg = sample_backward_tail(c, k, s, w, eps, N1, N2, D2, D1, g_v0, g_v1, g_v2, g_x, g_scaling != nullptr, g_s);
        gp0 = cg.gp0; gp1 = cg.gp1; gp2 = cg.gp2; gp3 = cg.gp3; gw = cg.gw;
        break;
    }
    if (PASS == 1) {
        const double acc2[2] = {acc, acc_a};
        block_accumulate<2>(acc2, norms, 3);
    }
    if (PASS == 3) {
        CurveGrad cg;
        cg.gp0 = gp0; cg.gp1 = gp1; cg.gp2 = gp2; cg.gp3 = gp3; cg.gw = gw;
        // 13 outputs per curve, summed over its m samples in sample order (deterministic)
        curve_reduce(cg, s_part, m, curves_per_block, [&](int c2, int f, float sum) {
            const int bb = blockIdx.x * curves_per_block + c2;
            if (bb >= B) return;
            float* dst = f < 12 ? g_cp + (size_t)bb * 12 + f : g_width + bb;
            *dst = accumulate ? *dst + sum : sum;
        });
    }
}

// Closing pass of the FUSED view backward (k_view_bwd left, per curve, the 13 sums of the tail's gradient-dependent part in
// `part`): the part that carries the two grid-wide sums is evaluated from the curve alone -- tail(0, D2, D1) -- and added.
// Reads 48 + 52 bytes per curve, no per-splat data.
__global__ void __launch_bounds__(SAMPLE_BLOCK) k_sample_bwd_close(int B, int m, int curves_per_block, const float* __restrict__ cp,
                                                                   const float* __restrict__ width,
                                                                   const uint8_t* __restrict__ is_bezier,
                                                                   const SampleCoef* __restrict__ coef, float eps,
                                                                   const double* __restrict__ norms, const float* __restrict__ part,
                                                                   float* __restrict__ g_cp, float* __restrict__ g_width,
                                                                   int accumulate) {
    // thread = sample, blocks hold whole curves, like pass 3.  (Thread = curve with a loop over its samples -- no LDS, no
    // barrier -- was measured: 14.6 against 9.3 us at cfg3, 20.3 against 19.5 at cfg5: 260 waves do not fill 1 024 SIMDs.)
    __shared__ float s_part[CURVE_PART][SAMPLE_BLOCK + 1];
    __shared__ SampleCoef s_coef[MAX_M];
    __shared__ BlockConst s_bc;
    stage_consts(coef, m, norms, s_coef, &s_bc);
    const int lc = threadIdx.x / m, i = threadIdx.x - lc * m;
    const int b = blockIdx.x * curves_per_block + lc;
    CurveGrad cg;
    cg.gp0 = cg.gp1 = cg.gp2 = cg.gp3 = V3{0.f, 0.f, 0.f};
    cg.gw = 0.f;
    if (lc < curves_per_block && b < B) {
        const CurveCP c = load_curve(cp, is_bezier, b);
        const SampleCoef k = s_coef[i];
        const SampleFwd s = sample_forward(c, k, s_bc.N1, s_bc.N2, eps);
        const V3 z = {0.f, 0.f, 0.f};
        cg = sample_backward_tail(c, k, s, expf(width[b]), eps, s_bc.N1, s_bc.N2, s_bc.D2, s_bc.D1, z, z, z, z, false, z);
    }
    curve_reduce(cg, s_part, m, curves_per_block, [&](int c2, int f, float sum) {
        const int bb = blockIdx.x * curves_per_block + c2;
        if (bb >= B) return;
        const float v = part[(size_t)bb * CURVE_PART + f] + sum;
        float* dst = f < 12 ? g_cp + (size_t)bb * 12 + f : g_width + bb;
        *dst = accumulate ? *dst + v : v;
    });
}

// ------------------------------------------------------------------------------------------------ splat attributes
// Per-view derived per-splat inputs of the rasterizer (one thread per curve, m samples each):
//   rot_n   = F.normalize(rot_raw)                       (eps 1e-12)
//   opac    = sigmoid(opacity_logit[b]) [* mask]         mask = straight-through (sigmoid(mask_logit) > thr)
//   scl_out = scaling [* mask]
//   all_map = [ flip_toward_camera( R(rot_n)[:,0] ) @ view[:3,:3], 1 ]

__global__ void __launch_bounds__(256) k_attrs_fwd(int B, int m, const float* __restrict__ rot_raw,
                                                   const float* __restrict__ xyz,
                                                   const float* __restrict__ opacity_logit,
                                                   const float* __restrict__ mask_logit, float mask_thr,
                                                   const float* __restrict__ scaling,
                                                   const float* __restrict__ campos, const float* __restrict__ vm,
                                                   float* __restrict__ rot_n, float* __restrict__ opac,
                                                   float* __restrict__ scl_out, float* __restrict__ all_map) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B * m) return;
    const int b = p / m;
    const V3 cam = {campos[0], campos[1], campos[2]};
    const float4 q = reinterpret_cast<const float4*>(rot_raw)[p];
    const V3 x = {xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]};
    const AttrsFwd a = attrs_forward(q, x, opacity_logit[b], mask_logit != nullptr, mask_logit ? mask_logit[p] : 0.f, mask_thr,
                                     cam, vm);
    reinterpret_cast<float4*>(rot_n)[p] = a.rot_n;
    opac[p] = a.opac;
    if (scl_out) {
        scl_out[3 * p] = scaling[3 * p] * a.mk; scl_out[3 * p + 1] = scaling[3 * p + 1] * a.mk;
        scl_out[3 * p + 2] = scaling[3 * p + 2] * a.mk;
    }
    reinterpret_cast<float4*>(all_map)[p] = a.all_map;
}

// Blocks hold whole curves (curves_per_block * m active threads); the per-curve opacity-logit gradient is the
// sample-ordered sum of its m per-splat terms (LDS).
__global__ void __launch_bounds__(SAMPLE_BLOCK) k_attrs_bwd(
    int B, int m, int curves_per_block, const float* __restrict__ rot_raw, const float* __restrict__ xyz,
    const float* __restrict__ opacity_logit, const float* __restrict__ mask_logit, float mask_thr,
    const float* __restrict__ scaling, const float* __restrict__ campos, const float* __restrict__ vm,
    const float* __restrict__ g_rot_n, const float* __restrict__ g_opac, const float* __restrict__ g_scl_out,
    const float* __restrict__ g_all_map, float* __restrict__ g_rot_raw, float* __restrict__ g_opacity_logit,
    float* __restrict__ g_mask_logit, float* __restrict__ g_scaling) {
    __shared__ float s_go[SAMPLE_BLOCK];
    const int lc = threadIdx.x / m, i = threadIdx.x - lc * m;
    const int b = blockIdx.x * curves_per_block + lc;
    const bool valid = lc < curves_per_block && b < B;
    float g_op_term = 0.f;
    if (valid) {
        const size_t p = (size_t)b * m + i;
        const V3 cam = {campos[0], campos[1], campos[2]};
        const float4 q = reinterpret_cast<const float4*>(rot_raw)[p];
        const V3 x = {xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]};
        const float4 gqn = g_rot_n ? reinterpret_cast<const float4*>(g_rot_n)[p] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 ga = g_all_map ? reinterpret_cast<const float4*>(g_all_map)[p] : make_float4(0.f, 0.f, 0.f, 0.f);
        V3 gs = {0.f, 0.f, 0.f}, sc = {0.f, 0.f, 0.f};
        if (g_scl_out) {
            gs = {g_scl_out[3 * p], g_scl_out[3 * p + 1], g_scl_out[3 * p + 2]};
            sc = {scaling[3 * p], scaling[3 * p + 1], scaling[3 * p + 2]};
        }
        const AttrsBwd o = attrs_backward(q, x, opacity_logit[b], mask_logit != nullptr, mask_logit ? mask_logit[p] : 0.f,
                                          mask_thr, cam, vm, gqn, g_all_map != nullptr, ga, g_opac ? g_opac[p] : 0.f,
                                          g_scl_out != nullptr, gs, sc);
        reinterpret_cast<float4*>(g_rot_raw)[p] = o.g_rot_raw;
        g_op_term = o.g_op_term;
        if (g_scl_out && g_scaling) { g_scaling[3 * p] = gs.x * o.mk; g_scaling[3 * p + 1] = gs.y * o.mk; g_scaling[3 * p + 2] = gs.z * o.mk; }
        if (g_mask_logit) g_mask_logit[p] = o.g_mask_logit;
    }
    s_go[threadIdx.x] = g_op_term;
    __syncthreads();
    if (valid && i == 0) {
        float sum = 0.f;
        for (int q = 0; q < m; q++) sum += s_go[threadIdx.x + q];
        g_opacity_logit[b] = sum;
    }
}

// ------------------------------------------------------------------------------------------------ launchers
// Workgroup size of the norm pass (always NORM_SLOTS workgroups): 256 threads up to 128 k splats, 512 up to 600 k, 1024 beyond.
// Big workgroups only fit a CU once several compositor waves of the neighbouring views have drained; with three views in
// flight at cfg3, 1024 threads cost 0.6 % of the headline against 256 / 512 (same box, twice: 813 - 814 vs 818 - 821 Msplats/s),
// while a million splats want the shorter per-thread loop.
static int f12_threads(long long P) { return P <= 128 * 1024 ? 256 : (P <= 600 * 1024 ? 512 : 1024); }

void launch_sample_forward(hipStream_t s, int B, int m, const float* cp, const float* width, const uint8_t* is_bezier,
                           const void* coef, float eps, double* norms, float* xyz, float* rot, float* scaling) {
    const dim3 grid((B * m + 255) / 256), block(256);
    const SampleCoef* k = reinterpret_cast<const SampleCoef*>(coef);
    { ProfScope p("sample_f12", s); hipLaunchKernelGGL(k_sample_f12, dim3(NORM_SLOTS), dim3(f12_threads((long long)B * m)), 0, s, B, m, cp, is_bezier, k, norms); }
    { ProfScope p("sample_f3", s); hipLaunchKernelGGL(k_sample_f3, grid, block, 0, s, B, m, cp, width, is_bezier, k, eps, norms, xyz, rot, scaling); }
}
void launch_sample_backward(hipStream_t s, int B, int m, const float* cp, const float* width, const uint8_t* is_bezier,
                            const void* coef, float eps, double* norms, const float* g_xyz, const float* g_rot,
                            const float* g_scaling, float* g_cp, float* g_width, float* gv_cache) {
    const int cpb = SAMPLE_BLOCK / m;  // whole curves per block
    const dim3 grid((B + cpb - 1) / cpb), block(SAMPLE_BLOCK);
    const SampleCoef* k = reinterpret_cast<const SampleCoef*>(coef);
    const dim3 rgrid(std::max((B * m + SAMPLE_BLOCK - 1) / SAMPLE_BLOCK, 1));
    { ProfScope p("sample_b1", s); hipLaunchKernelGGL(k_sample_bwd<1>, rgrid, block, 0, s, B, m, cpb, cp, width, is_bezier, k, eps, norms, g_xyz, g_rot, g_scaling, g_cp, g_width, gv_cache); }
    { ProfScope p("sample_b3", s); hipLaunchKernelGGL(k_sample_bwd<3>, grid, block, 0, s, B, m, cpb, cp, width, is_bezier, k, eps, norms, g_xyz, g_rot, g_scaling, g_cp, g_width, gv_cache); }
}
void launch_attrs_forward(hipStream_t s, int B, int m, const float* rot_raw, const float* xyz, const float* opacity_logit,
                          const float* mask_logit, float mask_thr, const float* scaling, const float* campos,
                          const float* vm, float* rot_n, float* opac, float* scl_out, float* all_map) {
    ProfScope p("attrs_fwd", s);
    hipLaunchKernelGGL(k_attrs_fwd, dim3((B * m + 255) / 256), dim3(256), 0, s, B, m, rot_raw, xyz, opacity_logit, mask_logit,
                       mask_thr, scaling, campos, vm, rot_n, opac, scl_out, all_map);
}
void launch_attrs_backward(hipStream_t s, int B, int m, const float* rot_raw, const float* xyz,
                           const float* opacity_logit, const float* mask_logit, float mask_thr, const float* scaling,
                           const float* campos, const float* vm, const float* g_rot_n, const float* g_opac,
                           const float* g_scl_out, const float* g_all_map, float* g_rot_raw, float* g_opacity_logit,
                           float* g_mask_logit, float* g_scaling) {
    ProfScope p("attrs_bwd", s);
    const int cpb = SAMPLE_BLOCK / m;
    hipLaunchKernelGGL(k_attrs_bwd, dim3((B + cpb - 1) / cpb), dim3(SAMPLE_BLOCK), 0, s, B, m, cpb, rot_raw, xyz, opacity_logit, mask_logit,
                       mask_thr, scaling, campos, vm, g_rot_n, g_opac, g_scl_out, g_all_map, g_rot_raw, g_opacity_logit,
                       g_mask_logit, g_scaling);
}

void launch_sample_norms(hipStream_t s, int B, int m, const float* cp, const uint8_t* is_bezier, const void* coef, double* norms) {
    ProfScope p("sample_f12", s);
    hipLaunchKernelGGL(k_sample_f12, dim3(NORM_SLOTS), dim3(f12_threads((long long)B * m)), 0, s, B, m, cp, is_bezier,
                       reinterpret_cast<const SampleCoef*>(coef), norms);
}
void launch_sample_backward_close(hipStream_t s, int B, int m, const float* cp, const float* width, const uint8_t* is_bezier,
                                  const void* coef, float eps, const double* norms, const float* part, float* g_cp, float* g_width,
                                  int accumulate) {
    const int cpb = SAMPLE_BLOCK / m;
    ProfScope p("sample_b3", s);
    hipLaunchKernelGGL(k_sample_bwd_close, dim3((B + cpb - 1) / cpb), dim3(SAMPLE_BLOCK), 0, s, B, m, cpb, cp, width, is_bezier,
                       reinterpret_cast<const SampleCoef*>(coef), eps, norms, part, g_cp, g_width, accumulate);
}

int sample_norm_words() { return NORM_WORDS; }
int sample_norm_fwd_words() { return NQ_FWD * NORM_SLOTS; }

}  // namespace cgs
