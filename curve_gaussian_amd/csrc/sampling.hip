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

#include "kernels.h"

namespace cgs {

// Grid-wide sums: every workgroup adds its partial to one of NORM_SLOTS f64 slots per quantity (same-address f64
// atomics serialise at ~20 ns each: 782 blocks on one address cost more than the kernels themselves, 49 per slot do
// not), and every consumer sums the slots while staging its constants.
constexpr int NORM_SLOTS = 64;
// norms[q * NORM_SLOTS + slot].  Forward (k_sample_f12, ONE pass): q0 = S1 = sum |c1v|^2, q1 = S2 = sum |cross(tan,c1v)|^2,
// q2 = BS = sum dot(cross(cross(tan,c1v), tan), c1v), from which N1 = sqrt(S1), N2 = sqrt(S2) / N1 (c2v = cross(tan,
// c1v / N1)).  Backward (k_sample_bwd<1>, ONE pass): q3 = D2 = sum dot(g_v2, c2v), q4 = A = sum dot(g_v1 +
// cross(g_v2, tan) / N2, c1v); the second global term follows in closed form, D1 = A - D2 / N2^3 * BS / N1 (it is
// linear in D2), so neither direction needs a second grid-wide pass.
constexpr int NQ_FWD = 3, NQ_ALL = 5;
constexpr int NORM_WORDS = NQ_ALL * NORM_SLOTS;

struct SampleCoef {  // per-sample coefficients, computed on the host with the reference's float32 torch expressions
    float c[4];      // Bezier point weights at t_i
    float cf[4];     // Bezier point weights at t_i - 0.5/m
    float d[3];      // tangent weights: 3(1-t)^2, 6(1-t)t, 3t^2
    float l[2];      // line weights (1-t), t
    float lf[2];     // line weights at t_i - 0.5/m
    float pad;
};
static_assert(sizeof(SampleCoef) == 64, "SampleCoef must be 16 floats");

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct CurveCP { V3 p0, p1, p2, p3; bool bez; };
__device__ __forceinline__ CurveCP load_curve(const float* __restrict__ cp, const uint8_t* __restrict__ is_bezier, int b) {
    const float4* q = reinterpret_cast<const float4*>(cp + (size_t)b * 12);
    const float4 a = q[0], c = q[1], d = q[2];
    CurveCP r;
    r.p0 = {a.x, a.y, a.z}; r.p1 = {a.w, c.x, c.y}; r.p2 = {c.z, c.w, d.x}; r.p3 = {d.y, d.z, d.w};
    r.bez = is_bezier ? (is_bezier[b] != 0) : true;
    return r;
}
__device__ __forceinline__ V3 curve_tangent(const CurveCP& c, const SampleCoef& k) {
    if (!c.bez) return c.p3 - c.p0;
    return k.d[0] * (c.p1 - c.p0) + k.d[1] * (c.p2 - c.p1) + k.d[2] * (c.p3 - c.p2);
}

// Per-block constants staged once in LDS: the m x 16 coefficient table and the global norms as f32 (the f64 sqrt
// per thread dominated the small kernels).
struct BlockConst {
    float N1, N2, D2, D1;
};
__device__ __forceinline__ void stage_consts(const SampleCoef* __restrict__ coef, int m, const double* __restrict__ norms,
                                             SampleCoef* s_coef, BlockConst* s_bc) {
    const float* src = reinterpret_cast<const float*>(coef);
    float* dst = reinterpret_cast<float*>(s_coef);
    for (int t = threadIdx.x; t < m * 16; t += blockDim.x) dst[t] = src[t];
    __shared__ double s_q[NQ_ALL];
    for (int base = 0; base < NQ_ALL * NORM_SLOTS; base += blockDim.x) {  // NORM_SLOTS-lane groups: q * NORM_SLOTS + slot
        const int idx = base + threadIdx.x;
        double v = idx < NQ_ALL * NORM_SLOTS ? norms[idx] : 0.0;
#pragma unroll
        for (int off = NORM_SLOTS / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (idx < NQ_ALL * NORM_SLOTS && (idx % NORM_SLOTS) == 0) s_q[idx / NORM_SLOTS] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double N1 = sqrt(s_q[0]), N2 = sqrt(s_q[1]) / N1;
        s_bc->N1 = (float)N1;
        s_bc->N2 = (float)N2;
        s_bc->D2 = (float)s_q[3];
        s_bc->D1 = (float)(s_q[4] - s_q[3] / (N2 * N2 * N2) * (s_q[2] / N1));
    }
    __syncthreads();
}
constexpr int MAX_M = 32;  // samples per curve supported by the LDS table (reference default 12)

// block-wide sums of N quantities -> one f64 atomic each on this block's slot of quantities q0, q0+1, ...
template <int N>
__device__ __forceinline__ void block_accumulate(const double (&v)[N], double* norms, int q0) {
    __shared__ double s_part[N][4];
    double w[N];
#pragma unroll
    for (int k = 0; k < N; k++) w[k] = (double)wave_sum((float)v[k]);  // 64 addends in f32 (DPP), the rest in f64
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < N; k++) s_part[k][wave] = w[k];
    }
    __syncthreads();
    if (threadIdx.x < N) {
        double t = 0;
        for (int wv = 0; wv < (int)(blockDim.x >> 6); wv++) t += s_part[threadIdx.x][wv];
        atomicAdd(norms + (q0 + threadIdx.x) * NORM_SLOTS + (blockIdx.x % NORM_SLOTS), t);
    }
}

// ------------------------------------------------------------------------------------------------ forward
__global__ void __launch_bounds__(256) k_sample_f12(int B, int m, const float* __restrict__ cp,
                                                    const uint8_t* __restrict__ is_bezier,
                                                    const SampleCoef* __restrict__ coef, double* __restrict__ norms) {
    __shared__ SampleCoef s_coef[MAX_M];
    const float* src = reinterpret_cast<const float*>(coef);
    float* dst = reinterpret_cast<float*>(s_coef);
    for (int t = threadIdx.x; t < m * 16; t += blockDim.x) dst[t] = src[t];
    __syncthreads();
    float a1 = 0, a2 = 0, a3 = 0;   // the grid covers every splat once: at most one addend per thread
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < B * m; p += gridDim.x * blockDim.x) {
        const int b = p / m, i = p - b * m;
        const CurveCP c = load_curve(cp, is_bezier, b);
        const V3 t = curve_tangent(c, s_coef[i]);
        const V3 c1 = {t.y, -t.x, 0.f};                 // cross(tan, (0,0,1))
        const V3 x = cross(t, c1);                      // N1 * c2v
        a1 += t.y * t.y + t.x * t.x;
        a2 += x.x * x.x + x.y * x.y + x.z * x.z;
        a3 += dot(cross(x, t), c1);
    }
    const double acc3[3] = {(double)a1, (double)a2, (double)a3};
    block_accumulate<3>(acc3, norms, 0);
}

struct QuatFwd { float a[4], qa[4], N[4], D; int k; bool flip; };
// arr[k] for a runtime k without a runtime-indexed (scratch-resident) array: three selects
__device__ __forceinline__ float sel4(const float (&arr)[4], int k) {
    return k == 0 ? arr[0] : (k == 1 ? arr[1] : (k == 2 ? arr[2] : arr[3]));
}
// rot_to_quat_batch for one 3x3 (rows m0*, m1*, m2*), utils/general_utils.py:33-86
__device__ __forceinline__ QuatFwd quat_forward(const float M[3][3], float q[4]) {
    QuatFwd f;
    const float m00 = M[0][0], m01 = M[0][1], m02 = M[0][2], m10 = M[1][0], m11 = M[1][1], m12 = M[1][2], m20 = M[2][0],
                m21 = M[2][1], m22 = M[2][2];
    f.a[0] = 1.0f + m00 + m11 + m22; f.a[1] = 1.0f + m00 - m11 - m22;
    f.a[2] = 1.0f - m00 + m11 - m22; f.a[3] = 1.0f - m00 - m11 + m22;
    int k = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) f.qa[j] = f.a[j] > 0.f ? sqrtf(f.a[j]) : 0.f;  // _sqrt_positive_part
    float qak = f.qa[0];
#pragma unroll
    for (int j = 1; j < 4; j++) {                                                 // argmax, first maximum wins
        const bool gt = f.qa[j] > qak;
        k = gt ? j : k;
        qak = gt ? f.qa[j] : qak;
    }
    f.k = k;
    const float sq = qak * qak;
    if (k == 0) { f.N[0] = sq; f.N[1] = m21 - m12; f.N[2] = m02 - m20; f.N[3] = m10 - m01; }
    else if (k == 1) { f.N[0] = m21 - m12; f.N[1] = sq; f.N[2] = m10 + m01; f.N[3] = m02 + m20; }
    else if (k == 2) { f.N[0] = m02 - m20; f.N[1] = m10 + m01; f.N[2] = sq; f.N[3] = m12 + m21; }
    else { f.N[0] = m10 - m01; f.N[1] = m20 + m02; f.N[2] = m21 + m12; f.N[3] = sq; }
    f.D = 2.0f * fmaxf(qak, 0.1f);
#pragma unroll
    for (int j = 0; j < 4; j++) q[j] = f.N[j] / f.D;
    f.flip = q[0] < 0.f;  // standardize_quaternion
    if (f.flip) {
#pragma unroll
        for (int j = 0; j < 4; j++) q[j] = -q[j];
    }
    return f;
}
// gradient of the above w.r.t. M
__device__ __forceinline__ void quat_backward(const QuatFwd& f, const float g_out[4], float gM[3][3]) {
    float gc[4], gN[4];
    float gD = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        gc[j] = f.flip ? -g_out[j] : g_out[j];
        gN[j] = gc[j] / f.D;
        gD -= gc[j] * f.N[j] / (f.D * f.D);
    }
    const int k = f.k;
    const float qak = sel4(f.qa, k);
    const float g_qa = (qak > 0.1f ? 2.f * gD : 0.f) + sel4(gN, k) * 2.f * qak;
    const float g_a = sel4(f.a, k) > 0.f ? g_qa / (2.f * qak) : 0.f;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) gM[r][c] = 0.f;
    const float s00 = (k == 0 || k == 1) ? 1.f : -1.f, s11 = (k == 0 || k == 2) ? 1.f : -1.f,
                s22 = (k == 0 || k == 3) ? 1.f : -1.f;
    gM[0][0] = s00 * g_a; gM[1][1] = s11 * g_a; gM[2][2] = s22 * g_a;
    if (k == 0) {
        gM[2][1] += gN[1]; gM[1][2] -= gN[1]; gM[0][2] += gN[2]; gM[2][0] -= gN[2]; gM[1][0] += gN[3]; gM[0][1] -= gN[3];
    } else if (k == 1) {
        gM[2][1] += gN[0]; gM[1][2] -= gN[0]; gM[1][0] += gN[2]; gM[0][1] += gN[2]; gM[0][2] += gN[3]; gM[2][0] += gN[3];
    } else if (k == 2) {
        gM[0][2] += gN[0]; gM[2][0] -= gN[0]; gM[1][0] += gN[1]; gM[0][1] += gN[1]; gM[1][2] += gN[3]; gM[2][1] += gN[3];
    } else {
        gM[1][0] += gN[0]; gM[0][1] -= gN[0]; gM[2][0] += gN[1]; gM[0][2] += gN[1]; gM[2][1] += gN[2]; gM[1][2] += gN[2];
    }
}

struct SampleFwd {
    V3 xyz, dvec, tan, v0, c1v, v1, c2v, v2;
    float dist, n;
};
__device__ __forceinline__ SampleFwd sample_forward(const CurveCP& c, const SampleCoef& k, float N1, float N2, float eps) {
    SampleFwd s;
    V3 front;
    if (c.bez) {
        s.xyz = k.c[0] * c.p0 + k.c[1] * c.p1 + k.c[2] * c.p2 + k.c[3] * c.p3;
        front = k.cf[0] * c.p0 + k.cf[1] * c.p1 + k.cf[2] * c.p2 + k.cf[3] * c.p3;
    } else {
        s.xyz = k.l[0] * c.p0 + k.l[1] * c.p3;
        front = k.lf[0] * c.p0 + k.lf[1] * c.p3;
    }
    s.dvec = s.xyz - front;
    s.dist = sqrtf(dot(s.dvec, s.dvec));
    s.tan = curve_tangent(c, k);
    s.n = sqrtf(dot(s.tan, s.tan));
    s.v0 = {s.tan.x / (s.n + eps), s.tan.y / (s.n + eps), s.tan.z / (s.n + eps)};
    s.c1v = {s.tan.y, -s.tan.x, 0.f};
    s.v1 = {s.c1v.x / N1, s.c1v.y / N1, 0.f};
    s.c2v = cross(s.tan, s.v1);
    s.v2 = {s.c2v.x / N2, s.c2v.y / N2, s.c2v.z / N2};
    return s;
}
__device__ __forceinline__ void rot_matrix(const SampleFwd& s, float M[3][3]) {  // columns v0 v1 v2
    M[0][0] = s.v0.x; M[0][1] = s.v1.x; M[0][2] = s.v2.x;
    M[1][0] = s.v0.y; M[1][1] = s.v1.y; M[1][2] = s.v2.y;
    M[2][0] = s.v0.z; M[2][1] = s.v1.z; M[2][2] = s.v2.z;
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
constexpr int SAMPLE_BLOCK = 256;
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
                                                             float* __restrict__ gv_cache) {
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
            const float* gv = gv_cache + 9 * p;
            g_v0 = {gv[0], gv[1], gv[2]}; g_v1 = {gv[3], gv[4], gv[5]}; g_v2 = {gv[6], gv[7], gv[8]};
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
            float* gv = gv_cache + 9 * p;
            gv[0] = g_v0.x; gv[1] = g_v0.y; gv[2] = g_v0.z; gv[3] = g_v1.x; gv[4] = g_v1.y; gv[5] = g_v1.z;
            gv[6] = g_v2.x; gv[7] = g_v2.y; gv[8] = g_v2.z;
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
        const float iN2 = 1.f / N2;
        const V3 g_c2v = iN2 * g_v2 - (D2 * iN2 * iN2 * iN2) * s.c2v;
        // c2v = cross(tan, v1)
        V3 g_tan = cross(s.v1, g_c2v);
        const V3 g_v1t = g_v1 + cross(g_c2v, s.tan);
        if (PASS == 2) {
            acc += (double)dot(g_v1t, s.c1v);
            sp += gridDim.x * blockDim.x;
            valid = sp < B * m;
            b = valid ? sp / m : 0;
            i = sp - b * m;
            continue;
        }
        const float iN1 = 1.f / N1;
        const V3 g_c1v = iN1 * g_v1t - (D1 * iN1 * iN1 * iN1) * s.c1v;
        g_tan.y += g_c1v.x;  // c1v = (ty, -tx, 0)
        g_tan.x -= g_c1v.y;
        if (s.n > 0.f) {     // v0 = tan / (n + eps)
            const float ne = s.n + eps;
            const float coefv = dot(g_v0, s.tan) / (s.n * ne * ne);
            g_tan = g_tan + (1.f / ne) * g_v0 - coefv * s.tan;
        } else {
            g_tan = g_tan + (1.f / eps) * g_v0;
        }
        // scaling = (dist, exp(w), exp(w))
        V3 g_x = {0, 0, 0};
        if (g_xyz) g_x = {g_xyz[3 * p], g_xyz[3 * p + 1], g_xyz[3 * p + 2]};
        V3 g_front = {0, 0, 0};
        if (g_scaling) {
            const float gd = g_scaling[3 * p];
            gw = (g_scaling[3 * p + 1] + g_scaling[3 * p + 2]) * w;
            if (s.dist > 0.f) {
                const V3 gdv = (gd / s.dist) * s.dvec;
                g_x = g_x + gdv;
                g_front = {-gdv.x, -gdv.y, -gdv.z};
            }
        }
        if (c.bez) {
            gp0 = k.c[0] * g_x + k.cf[0] * g_front - k.d[0] * g_tan;
            gp1 = k.c[1] * g_x + k.cf[1] * g_front + (k.d[0] - k.d[1]) * g_tan;
            gp2 = k.c[2] * g_x + k.cf[2] * g_front + (k.d[1] - k.d[2]) * g_tan;
            gp3 = k.c[3] * g_x + k.cf[3] * g_front + k.d[2] * g_tan;
        } else {
            gp0 = k.l[0] * g_x + k.lf[0] * g_front - g_tan;
            gp3 = k.l[1] * g_x + k.lf[1] * g_front + g_tan;
        }
        break;
    }
    if (PASS == 1) {
        const double acc2[2] = {acc, acc_a};
        block_accumulate<2>(acc2, norms, 3);
    }
    if (PASS == 3) {
        const int t = threadIdx.x;
        s_part[0][t] = gp0.x; s_part[1][t] = gp0.y; s_part[2][t] = gp0.z;
        s_part[3][t] = gp1.x; s_part[4][t] = gp1.y; s_part[5][t] = gp1.z;
        s_part[6][t] = gp2.x; s_part[7][t] = gp2.y; s_part[8][t] = gp2.z;
        s_part[9][t] = gp3.x; s_part[10][t] = gp3.y; s_part[11][t] = gp3.z;
        s_part[12][t] = gw;
        __syncthreads();
        // 13 outputs per curve, summed over its m samples in sample order (deterministic)
        for (int o = threadIdx.x; o < curves_per_block * 13; o += blockDim.x) {
            const int c2 = o / 13, f = o - c2 * 13;
            const int bb = blockIdx.x * curves_per_block + c2;
            if (bb >= B) continue;
            float sum = 0.f;
            for (int q = 0; q < m; q++) sum += s_part[f][c2 * m + q];
            if (f < 12) g_cp[(size_t)bb * 12 + f] = sum;
            else g_width[bb] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------ splat attributes
// Per-view derived per-splat inputs of the rasterizer (one thread per curve, m samples each):
//   rot_n   = F.normalize(rot_raw)                       (eps 1e-12)
//   opac    = sigmoid(opacity_logit[b]) [* mask]         mask = straight-through (sigmoid(mask_logit) > thr)
//   scl_out = scaling [* mask]
//   all_map = [ flip_toward_camera( R(rot_n)[:,0] ) @ view[:3,:3], 1 ]
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

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
    const float op = sigmoidf_(opacity_logit[b]);
    const V3 cam = {campos[0], campos[1], campos[2]};
    const float4 q = reinterpret_cast<const float4*>(rot_raw)[p];
    const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float den = fmaxf(nrm, 1e-12f);
    const float r = q.x / den, qi = q.y / den, qj = q.z / den, qk = q.w / den;
    reinterpret_cast<float4*>(rot_n)[p] = make_float4(r, qi, qj, qk);
    float mk = 1.f;
    if (mask_logit) mk = sigmoidf_(mask_logit[p]) > mask_thr ? 1.f : 0.f;
    opac[p] = op * mk;
    if (scl_out) {
        scl_out[3 * p] = scaling[3 * p] * mk; scl_out[3 * p + 1] = scaling[3 * p + 1] * mk;
        scl_out[3 * p + 2] = scaling[3 * p + 2] * mk;
    }
    // pytorch3d quaternion_to_matrix, column 0
    const float two_s = 2.0f / (r * r + qi * qi + qj * qj + qk * qk);
    V3 d = {1.f - two_s * (qj * qj + qk * qk), two_s * (qi * qj + qk * r), two_s * (qi * qk - qj * r)};
    const V3 x = {xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]};
    if (dot(d, cam - x) < 0.0f) d = {-d.x, -d.y, -d.z};
    reinterpret_cast<float4*>(all_map)[p] = make_float4(d.x * vm[0] + d.y * vm[4] + d.z * vm[8],
                                                        d.x * vm[1] + d.y * vm[5] + d.z * vm[9],
                                                        d.x * vm[2] + d.y * vm[6] + d.z * vm[10], 1.0f);
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
        const float op = sigmoidf_(opacity_logit[b]);
        const V3 cam = {campos[0], campos[1], campos[2]};
        const float4 q = reinterpret_cast<const float4*>(rot_raw)[p];
        const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        const float den = fmaxf(nrm, 1e-12f);
        const float r = q.x / den, qi = q.y / den, qj = q.z / den, qk = q.w / den;
        float gq[4] = {0.f, 0.f, 0.f, 0.f};  // gradient w.r.t. the normalised quaternion
        if (g_rot_n) {
            const float4 g = reinterpret_cast<const float4*>(g_rot_n)[p];
            gq[0] = g.x; gq[1] = g.y; gq[2] = g.z; gq[3] = g.w;
        }
        if (g_all_map) {
            const float4 ga = reinterpret_cast<const float4*>(g_all_map)[p];
            // local = d @ view[:3,:3]  ->  g_d[r] = sum_c vm[4r + c] g_local[c]
            V3 gd = {vm[0] * ga.x + vm[1] * ga.y + vm[2] * ga.z, vm[4] * ga.x + vm[5] * ga.y + vm[6] * ga.z,
                     vm[8] * ga.x + vm[9] * ga.y + vm[10] * ga.z};
            const float s2 = r * r + qi * qi + qj * qj + qk * qk;
            const float two_s = 2.0f / s2;
            const V3 d = {1.f - two_s * (qj * qj + qk * qk), two_s * (qi * qj + qk * r), two_s * (qi * qk - qj * r)};
            const V3 x = {xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]};
            if (dot(d, cam - x) < 0.0f) gd = {-gd.x, -gd.y, -gd.z};
            // d0 = 1 - two_s (j^2+k^2), d1 = two_s (ij + kr), d2 = two_s (ik - jr)
            const float e0 = qj * qj + qk * qk, e1 = qi * qj + qk * r, e2 = qi * qk - qj * r;
            const float g_two_s = -gd.x * e0 + gd.y * e1 + gd.z * e2;
            const float g_s2 = g_two_s * (-2.0f / (s2 * s2));
            gq[0] += two_s * (gd.y * qk - gd.z * qj) + g_s2 * 2.f * r;
            gq[1] += two_s * (gd.y * qj + gd.z * qk) + g_s2 * 2.f * qi;
            gq[2] += two_s * (-2.f * gd.x * qj + gd.y * qi - gd.z * r) + g_s2 * 2.f * qj;
            gq[3] += two_s * (-2.f * gd.x * qk + gd.y * r + gd.z * qi) + g_s2 * 2.f * qk;
        }
        // F.normalize backward: y = x / max(|x|, eps)
        float4 gr;
        if (nrm > 1e-12f) {
            const float dq = gq[0] * r + gq[1] * qi + gq[2] * qj + gq[3] * qk;
            gr = make_float4((gq[0] - r * dq) / nrm, (gq[1] - qi * dq) / nrm, (gq[2] - qj * dq) / nrm, (gq[3] - qk * dq) / nrm);
        } else {
            gr = make_float4(gq[0] / 1e-12f, gq[1] / 1e-12f, gq[2] / 1e-12f, gq[3] / 1e-12f);
        }
        reinterpret_cast<float4*>(g_rot_raw)[p] = gr;
        float mk = 1.f, sg = 0.f;
        if (mask_logit) {
            sg = sigmoidf_(mask_logit[p]);
            mk = sg > mask_thr ? 1.f : 0.f;
        }
        const float go = g_opac ? g_opac[p] : 0.f;
        g_op_term = go * mk * op * (1.f - op);
        float g_mask = go * op;
        if (g_scl_out) {
            const float g0 = g_scl_out[3 * p], g1 = g_scl_out[3 * p + 1], g2 = g_scl_out[3 * p + 2];
            g_mask += g0 * scaling[3 * p] + g1 * scaling[3 * p + 1] + g2 * scaling[3 * p + 2];
            if (g_scaling) { g_scaling[3 * p] = g0 * mk; g_scaling[3 * p + 1] = g1 * mk; g_scaling[3 * p + 2] = g2 * mk; }
        }
        if (g_mask_logit) g_mask_logit[p] = mask_logit ? g_mask * sg * (1.f - sg) : 0.f;  // straight-through estimator
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
void launch_sample_forward(hipStream_t s, int B, int m, const float* cp, const float* width, const uint8_t* is_bezier,
                           const void* coef, float eps, double* norms, float* xyz, float* rot, float* scaling) {
    const dim3 grid((B * m + 255) / 256), block(256);
    const dim3 rgrid(std::max((B * m + 255) / 256, 1));
    const SampleCoef* k = reinterpret_cast<const SampleCoef*>(coef);
    { ProfScope p("sample_f12", s); hipLaunchKernelGGL(k_sample_f12, rgrid, block, 0, s, B, m, cp, is_bezier, k, norms); }
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

int sample_norm_words() { return NORM_WORDS; }
int sample_norm_fwd_words() { return NQ_FWD * NORM_SLOTS; }

}  // namespace cgs
