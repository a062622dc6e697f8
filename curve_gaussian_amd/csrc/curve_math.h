// Device-side curve -> Gaussian sampling math shared by the sampling kernels (sampling.hip) and the fused per-view
// kernels (view.hip): coefficient table, global-norm bookkeeping, rot_to_quat_batch forward / backward, one sample's
// forward quantities.  Reference: scene/gaussian_curve_model.py:70-89,180-198, utils/general_utils.py:33-86.
#pragma once
#include "kernels.h"

namespace cgs {

// Grid-wide sums live in NORM_SLOTS f64 slots per quantity; every consumer sums the slots while staging its constants.
// The forward's three sums are WRITTEN, one slot per workgroup of k_sample_f12 (launched on exactly NORM_SLOTS
// workgroups: no atomics, no zero-fill launch, the same bits on every run), which also clears the slots of the
// backward's two; those are accumulated with one fire-and-forget f64 atomic per workgroup (same-address f64 atomics
// serialise at ~20 ns each: 782 blocks on one address cost more than the kernels themselves, 49 per slot do not).
// (Round 5 tried to let the last workgroup of k_sample_bwd<3> hand the backward's slots back cleared, by an arrival
// ticket: a RETURNING device-scope atomic per workgroup -- one address or 64 -- took that kernel from 8.3 to 12.7 -
// 17.1 us at 794 workgroups; on this multi-XCD part the answer comes from beyond the XCD's L2.  Not kept.)
constexpr int NORM_SLOTS = 64;
// norms[q * NORM_SLOTS + slot].  Forward (k_sample_f12, ONE pass): q0 = S1 = sum |c1v|^2, q1 = S2 = sum |cross(tan,c1v)|^2,
// q2 = BS = sum dot(cross(cross(tan,c1v), tan), c1v), from which N1 = sqrt(S1), N2 = sqrt(S2) / N1 (c2v = cross(tan,
// c1v / N1)).  Backward (k_sample_bwd<1>, ONE pass): q3 = D2 = sum dot(g_v2, c2v), q4 = A = sum dot(g_v1 +
// cross(g_v2, tan) / N2, c1v); the second global term follows in closed form, D1 = A - D2 / N2^3 * BS / N1 (it is
// linear in D2), so neither direction needs a second grid-wide pass.
constexpr int NQ_FWD = 3, NQ_ALL = 5;
constexpr int NORM_WORDS = NQ_ALL * NORM_SLOTS;
static_assert(NORM_WORDS <= 384, "norms buffer layout");

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
constexpr int SAMPLE_BLOCK = 256;   // threads per block of the kernels whose blocks hold whole curves
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

// Tail of the sampling backward for ONE sample: from dL/d{v0, v1, v2} (columns of the rotation), dL/dxyz and dL/dscaling to
// this sample's contribution to dL/d{p0..p3} and dL/dwidth (reference: autograd of scene/gaussian_curve_model.py:180-198).
// D2 = sum_all <g_v2, c2v> and D1 (stage_consts) are the two grid-wide sums the global Frobenius norms bring in.  The map is
// LINEAR in (g_v0, g_v1, g_v2, g_x, g_scaling, D2, D1) with forward-only coefficients, so
//     tail(g, D2, D1) = tail(g, 0, 0) + tail(0, D2, D1):
// the fused view backward evaluates the first term per sample and keeps only its per-CURVE sum (13 floats per curve instead of
// 15 per splat through HBM); the closing pass evaluates the second from the curve alone and adds the two.
struct CurveGrad { V3 gp0, gp1, gp2, gp3; float gw; };
__device__ __forceinline__ CurveGrad sample_backward_tail(const CurveCP& c, const SampleCoef& k, const SampleFwd& s, float w, float eps,
                                                          float N1, float N2, float D2, float D1, V3 g_v0, V3 g_v1, V3 g_v2, V3 g_x,
                                                          bool has_scaling, V3 g_scl) {
    CurveGrad o;
    o.gp0 = o.gp1 = o.gp2 = o.gp3 = V3{0.f, 0.f, 0.f};
    o.gw = 0.f;
    const float iN2 = 1.f / N2;
    const V3 g_c2v = iN2 * g_v2 - (D2 * iN2 * iN2 * iN2) * s.c2v;
    // c2v = cross(tan, v1)
    V3 g_tan = cross(s.v1, g_c2v);
    const V3 g_v1t = g_v1 + cross(g_c2v, s.tan);
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
    V3 g_front = {0, 0, 0};
    if (has_scaling) {
        const float gd = g_scl.x;
        o.gw = (g_scl.y + g_scl.z) * w;
        if (s.dist > 0.f) {
            const V3 gdv = (gd / s.dist) * s.dvec;
            g_x = g_x + gdv;
            g_front = {-gdv.x, -gdv.y, -gdv.z};
        }
    }
    if (c.bez) {
        o.gp0 = k.c[0] * g_x + k.cf[0] * g_front - k.d[0] * g_tan;
        o.gp1 = k.c[1] * g_x + k.cf[1] * g_front + (k.d[0] - k.d[1]) * g_tan;
        o.gp2 = k.c[2] * g_x + k.cf[2] * g_front + (k.d[1] - k.d[2]) * g_tan;
        o.gp3 = k.c[3] * g_x + k.cf[3] * g_front + k.d[2] * g_tan;
    } else {
        o.gp0 = k.l[0] * g_x + k.lf[0] * g_front - g_tan;
        o.gp3 = k.l[1] * g_x + k.lf[1] * g_front + g_tan;
    }
    return o;
}
constexpr int CURVE_PART = 13;   // per-curve partial of the fused view backward: dL/d{p0..p3} (12) + dL/dwidth
// Sum the 13 per-sample values of every curve of the block over its m samples (sample order: deterministic) and hand each
// (curve, field) sum to fn(local curve, field, sum).  s_part: [13][SAMPLE_BLOCK + 1] floats of LDS.
template <typename F>
__device__ __forceinline__ void curve_reduce(const CurveGrad& g, float (*s_part)[256 + 1], int m, int curves_per_block, F fn) {
    const int t = threadIdx.x;
    s_part[0][t] = g.gp0.x; s_part[1][t] = g.gp0.y; s_part[2][t] = g.gp0.z;
    s_part[3][t] = g.gp1.x; s_part[4][t] = g.gp1.y; s_part[5][t] = g.gp1.z;
    s_part[6][t] = g.gp2.x; s_part[7][t] = g.gp2.y; s_part[8][t] = g.gp2.z;
    s_part[9][t] = g.gp3.x; s_part[10][t] = g.gp3.y; s_part[11][t] = g.gp3.z;
    s_part[12][t] = g.gw;
    __syncthreads();
    for (int o = threadIdx.x; o < curves_per_block * CURVE_PART; o += blockDim.x) {
        const int c2 = o / CURVE_PART, f = o - c2 * CURVE_PART;
        float sum = 0.f;
        for (int q = 0; q < m; q++) sum += s_part[f][c2 * m + q];
        fn(c2, f, sum);
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------- per-view splat attributes (one splat)
//   rot_n   = F.normalize(rot_raw) (eps 1e-12)        opac = sigmoid(opacity_logit) [* mask]      scl_out = scaling [* mask]
//   all_map = [ flip_toward_camera( R(rot_n)[:,0] ) @ view[:3,:3], 1 ]
// reference: scene/gaussian_curve_model.py:99-110,121-122; gaussian_renderer/__init__.py:72-76,98-104
struct AttrsFwd { float4 rot_n; float opac, mk; float4 all_map; float nrm; };
__device__ __forceinline__ AttrsFwd attrs_forward(const float4 q, const V3 x, float opacity_logit, bool has_mask,
                                                  float mask_logit, float mask_thr, const V3 cam,
                                                  const float* __restrict__ vm) {
    AttrsFwd o;
    const float op = sigmoidf_(opacity_logit);
    o.nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float den = fmaxf(o.nrm, 1e-12f);
    const float r = q.x / den, qi = q.y / den, qj = q.z / den, qk = q.w / den;
    o.rot_n = make_float4(r, qi, qj, qk);
    o.mk = 1.f;
    if (has_mask) o.mk = sigmoidf_(mask_logit) > mask_thr ? 1.f : 0.f;
    o.opac = op * o.mk;
    // pytorch3d quaternion_to_matrix, column 0
    const float two_s = 2.0f / (r * r + qi * qi + qj * qj + qk * qk);
    V3 d = {1.f - two_s * (qj * qj + qk * qk), two_s * (qi * qj + qk * r), two_s * (qi * qk - qj * r)};
    if (dot(d, cam - x) < 0.0f) d = {-d.x, -d.y, -d.z};
    o.all_map = make_float4(d.x * vm[0] + d.y * vm[4] + d.z * vm[8], d.x * vm[1] + d.y * vm[5] + d.z * vm[9],
                            d.x * vm[2] + d.y * vm[6] + d.z * vm[10], 1.0f);
    return o;
}

struct AttrsBwd { float4 g_rot_raw; float g_op_term, g_mask_logit, mk; };
// gq: gradient w.r.t. the normalised quaternion; ga: gradient w.r.t. all_map (ignored unless has_ga); go: w.r.t. opac;
// gs / scaling: gradient w.r.t. scl_out and the unmasked scaling (only with has_gs)
__device__ __forceinline__ AttrsBwd attrs_backward(const float4 q, const V3 x, float opacity_logit, bool has_mask,
                                                   float mask_logit, float mask_thr, const V3 cam,
                                                   const float* __restrict__ vm, float4 gqn, bool has_ga, const float4 ga,
                                                   float go, bool has_gs, const V3 gs, const V3 scaling) {
    AttrsBwd o;
    const float op = sigmoidf_(opacity_logit);
    const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float den = fmaxf(nrm, 1e-12f);
    const float r = q.x / den, qi = q.y / den, qj = q.z / den, qk = q.w / den;
    float gq[4] = {gqn.x, gqn.y, gqn.z, gqn.w};
    if (has_ga) {
        // local = d @ view[:3,:3]  ->  g_d[r] = sum_c vm[4r + c] g_local[c]
        V3 gd = {vm[0] * ga.x + vm[1] * ga.y + vm[2] * ga.z, vm[4] * ga.x + vm[5] * ga.y + vm[6] * ga.z,
                 vm[8] * ga.x + vm[9] * ga.y + vm[10] * ga.z};
        const float s2 = r * r + qi * qi + qj * qj + qk * qk;
        const float two_s = 2.0f / s2;
        const V3 d = {1.f - two_s * (qj * qj + qk * qk), two_s * (qi * qj + qk * r), two_s * (qi * qk - qj * r)};
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
    if (nrm > 1e-12f) {
        const float dq = gq[0] * r + gq[1] * qi + gq[2] * qj + gq[3] * qk;
        o.g_rot_raw = make_float4((gq[0] - r * dq) / nrm, (gq[1] - qi * dq) / nrm, (gq[2] - qj * dq) / nrm, (gq[3] - qk * dq) / nrm);
    } else {
        o.g_rot_raw = make_float4(gq[0] / 1e-12f, gq[1] / 1e-12f, gq[2] / 1e-12f, gq[3] / 1e-12f);
    }
    float sg = 0.f;
    o.mk = 1.f;
    if (has_mask) {
        sg = sigmoidf_(mask_logit);
        o.mk = sg > mask_thr ? 1.f : 0.f;
    }
    o.g_op_term = go * o.mk * op * (1.f - op);
    float g_mask = go * op;
    if (has_gs) g_mask += gs.x * scaling.x + gs.y * scaling.y + gs.z * scaling.z;
    o.g_mask_logit = has_mask ? g_mask * sg * (1.f - sg) : 0.f;  // straight-through estimator
    return o;
}

// dL/d{v0,v1,v2} of one sample from the gradient of its raw quaternion (rot_to_quat_batch backward)
__device__ __forceinline__ void quat_grad_to_axes(const SampleFwd& s, const float4 gq, V3& g_v0, V3& g_v1, V3& g_v2) {
    float M[3][3], q[4], gM[3][3];
    rot_matrix(s, M);
    const QuatFwd f = quat_forward(M, q);
    const float go[4] = {gq.x, gq.y, gq.z, gq.w};
    quat_backward(f, go, gM);
    g_v0 = {gM[0][0], gM[1][0], gM[2][0]};
    g_v1 = {gM[0][1], gM[1][1], gM[2][1]};
    g_v2 = {gM[0][2], gM[1][2], gM[2][2]};
}

}  // namespace cgs
