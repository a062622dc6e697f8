// Fused class-balanced edge loss (reference utils/loss_utils.py:94-115, used at train.py:101) -- ~15 PyTorch kernels
// and two host-visible reductions in the reference, two kernels here:
//   k_edge_count : n_pos = #{ mean_c gt > thr }
//   k_edge_loss  : loss = mean_{c,y,x} (image - gt)^2 * w(y,x),  w = 5 (n_neg+1)/N if edge else (n_pos+1)/N, N = H W
//                  and d loss / d image in the same pass.
#include <algorithm>
#include <cstring>

#include "kernels.h"

namespace cgs {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ void __launch_bounds__(256) k_edge_count(int C, int HW, const float* __restrict__ gt, float thr,
                                                    unsigned int* __restrict__ n_pos) {
    unsigned int cnt = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        float e = 0.f;
        for (int c = 0; c < C; c++) e += gt[(size_t)c * HW + i];
        e = e / (float)C;
        cnt += e > thr ? 1u : 0u;
    }
    __shared__ unsigned int s_cnt[4];
    unsigned int w = cnt;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) w += __shfl_xor(w, off, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (t) atomicAdd(n_pos, t);
    }
}

__global__ void __launch_bounds__(256) k_edge_loss(int C, int HW, const float* __restrict__ image,
                                                   const float* __restrict__ gt, float thr,
                                                   const unsigned int* __restrict__ n_pos_p, double* __restrict__ loss_sum,
                                                   float* __restrict__ grad) {
    __shared__ double s_part[4];
    const float n_pos = (float)(*n_pos_p), n_neg = (float)HW - n_pos;
    const float w_pos = 5.f * (n_neg + 1.f) / (n_pos + n_neg), w_neg = 1.0f * (n_pos + 1.f) / (n_pos + n_neg);
    const float gscale = 2.f / ((float)C * (float)HW);
    double acc = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        float e = 0.f;
        for (int c = 0; c < C; c++) e += gt[(size_t)c * HW + i];
        e = e / (float)C;
        const float w = e > thr ? w_pos : w_neg;
        for (int c = 0; c < C; c++) {
            const float d = image[(size_t)c * HW + i] - gt[(size_t)c * HW + i];
            acc += (double)(d * d * w);
            if (grad) grad[(size_t)c * HW + i] = gscale * d * w;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

void launch_edge_count(hipStream_t s, int C, int HW, const float* gt, float thr, unsigned int* n_pos) {
    ProfScope p("edge_count", s);
    const int blocks = std::min((HW + 255) / 256, 256);
    hipLaunchKernelGGL(k_edge_count, dim3(blocks), dim3(256), 0, s, C, HW, gt, thr, n_pos);
}
void launch_edge_aware_loss(hipStream_t s, int C, int H, int W, const float* image, const float* gt, float thr,
                            void* scratch16, float* grad) {
    const int HW = H * W;
    unsigned int* n_pos = reinterpret_cast<unsigned int*>(scratch16);
    double* loss_sum = reinterpret_cast<double*>(reinterpret_cast<char*>(scratch16) + 8);
    const int blocks = std::min((HW + 255) / 256, 256);  // one workgroup per CU: <= 256 same-address atomics
    { ProfScope p("edge_count", s); hipLaunchKernelGGL(k_edge_count, dim3(blocks), dim3(256), 0, s, C, HW, gt, thr, n_pos); }
    { ProfScope p("edge_loss", s); hipLaunchKernelGGL(k_edge_loss, dim3(blocks), dim3(256), 0, s, C, HW, image, gt, thr, n_pos, loss_sum, grad); }
}

// ------------------------------------------------------------------------------------------------ curve regularisers
// The per-iteration regularisers of train.py:113-131 in three launches (value + gradients), instead of ~60 PyTorch
// elementwise / reduction kernels over all P splats and their autograd graph (0.8 ms at P = 200 k):
//   opacity   w_op * gate * mean_{visible splats} log(1 + sigmoid(o_b)^2 / 0.5)                     (:114-117)
//   smooth    w_smo * [any splat visible] * mean_{b, i<m-1} (1 - |cos(d_i, d_{i+1})|),             (:119-124)
//             d = column 0 of pytorch3d.quaternion_to_matrix(F.normalize(q))
//   width     w_w * mean_{curves with exp(w_b) >= thr} (exp(w_b) - thr)                             (:126-131)
// Empty selections give 0 (the reference skips those terms with host-side ifs).  Blocks hold whole curves.
constexpr int REG_SLOTS = 32;
struct RegArgs { float w_op, w_smo, w_width, width_thr; };
__device__ __forceinline__ float sigmoid_reg(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void __launch_bounds__(256) k_reg_count(int P, int B, const int* __restrict__ radii,
                                                   const float* __restrict__ width, float width_thr,
                                                   unsigned int* __restrict__ counts) {
    unsigned int nv = 0, nw = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
        nv += radii[i] > 0 ? 1u : 0u;
        if (i < B) nw += expf(width[i]) >= width_thr ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        nv += __shfl_xor(nv, off, 64);
        nw += __shfl_xor(nw, off, 64);
    }
    __shared__ unsigned int s_n[2][4];
    if ((threadIdx.x & 63) == 0) { s_n[0][threadIdx.x >> 6] = nv; s_n[1][threadIdx.x >> 6] = nw; }
    __syncthreads();
    if (threadIdx.x < 2) {   // one atomic per block and counter, spread over REG_SLOTS addresses
        const unsigned int t = s_n[threadIdx.x][0] + s_n[threadIdx.x][1] + s_n[threadIdx.x][2] + s_n[threadIdx.x][3];
        if (t) atomicAdd(&counts[threadIdx.x * REG_SLOTS + blockIdx.x % REG_SLOTS], t);
    }
}

__device__ __forceinline__ void quat_axis0(float4 q, float (&d)[3], float& nrm, float (&qn)[4]) {
    nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float den = fmaxf(nrm, 1e-12f);                      // F.normalize
    qn[0] = q.x / den; qn[1] = q.y / den; qn[2] = q.z / den; qn[3] = q.w / den;
    const float r = qn[0], i = qn[1], j = qn[2], k = qn[3];
    const float two_s = 2.0f / (r * r + i * i + j * j + k * k);  // pytorch3d quaternion_to_matrix
    d[0] = 1.f - two_s * (j * j + k * k);
    d[1] = two_s * (i * j + k * r);
    d[2] = two_s * (i * k - j * r);
}
// d(1 - |cos(x, y)|) / dx with torch's cosine_similarity: x.y / (max(|x|, eps) max(|y|, eps)), eps = 1e-8
__device__ __forceinline__ float smooth_pair(const float* x, const float* y, float (&gx)[3]) {
    const float eps = 1e-8f;
    const float nx = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]), ny = sqrtf(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
    const float cx = fmaxf(nx, eps), cy = fmaxf(ny, eps);
    const float c = (x[0] * y[0] + x[1] * y[1] + x[2] * y[2]) / (cx * cy);
    const float sgn = c > 0.f ? 1.f : (c < 0.f ? -1.f : 0.f);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        // d c / d x = y / (cx cy) - c x / (cx nx)   (the second term vanishes where the norm is clamped)
        const float dc = y[a] / (cx * cy) - (nx > eps ? c * x[a] / (cx * nx) : 0.f);
        gx[a] = -sgn * dc;
    }
    return 1.f - fabsf(c);
}

__global__ void __launch_bounds__(256) k_reg_main(int B, int m, int curves_per_block, const float* __restrict__ rot_raw,
                                                  const float* __restrict__ opacity_logit,
                                                  const float* __restrict__ width, const int* __restrict__ radii,
                                                  const unsigned int* __restrict__ counts, RegArgs ra,
                                                  const float* __restrict__ op_gate, double* __restrict__ sums,
                                                  float* __restrict__ g_rot_raw, float* __restrict__ g_opacity_logit,
                                                  float* __restrict__ g_width) {
    __shared__ float s_d[256][3];
    __shared__ float s_go[256];
    __shared__ float s_cnt[2];
    if (threadIdx.x < 64) {   // totals of the two selections (REG_SLOTS partial counters each)
        unsigned int v = threadIdx.x < 2 * REG_SLOTS ? counts[threadIdx.x] : 0u;
#pragma unroll
        for (int off = REG_SLOTS / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (threadIdx.x == 0) s_cnt[0] = (float)v;
        if (threadIdx.x == REG_SLOTS) s_cnt[1] = (float)v;
    }
    const int lc = threadIdx.x / m, i = threadIdx.x - lc * m;
    const int b = blockIdx.x * curves_per_block + lc;
    const bool valid = lc < curves_per_block && b < B;
    float d[3] = {0.f, 0.f, 0.f}, qn[4] = {0.f, 0.f, 0.f, 0.f}, nrm = 1.f;
    size_t p = 0;
    if (valid) {
        p = (size_t)b * m + i;
        quat_axis0(reinterpret_cast<const float4*>(rot_raw)[p], d, nrm, qn);
    }
    s_d[threadIdx.x][0] = d[0]; s_d[threadIdx.x][1] = d[1]; s_d[threadIdx.x][2] = d[2];
    __syncthreads();
    const float n_vis = s_cnt[0], n_w = s_cnt[1];
    const float gate = op_gate ? *op_gate : 1.f;
    const float any_vis = n_vis > 0.f ? 1.f : 0.f;
    const float k_smo = ra.w_smo * any_vis / ((float)B * (float)(m - 1));
    const float k_op = ra.w_op * gate / fmaxf(n_vis, 1.f);
    const float k_w = ra.w_width / fmaxf(n_w, 1.f);
    float v_smo = 0.f, v_op = 0.f, v_w = 0.f, go_term = 0.f;
    if (valid) {
        // smoothness: this sample is x of pair (i, i+1) and y of pair (i-1, i); the function is symmetric in x, y
        float gd[3] = {0.f, 0.f, 0.f}, gt[3];
        if (i + 1 < m) {
            v_smo = smooth_pair(s_d[threadIdx.x], s_d[threadIdx.x + 1], gt);
            gd[0] += gt[0]; gd[1] += gt[1]; gd[2] += gt[2];
        }
        if (i > 0) {
            (void)smooth_pair(s_d[threadIdx.x], s_d[threadIdx.x - 1], gt);
            gd[0] += gt[0]; gd[1] += gt[1]; gd[2] += gt[2];
        }
        gd[0] *= k_smo; gd[1] *= k_smo; gd[2] *= k_smo;
        // d -> normalised quaternion -> raw quaternion (same chain as k_attrs_bwd)
        const float r = qn[0], qi = qn[1], qj = qn[2], qk = qn[3];
        const float s2 = r * r + qi * qi + qj * qj + qk * qk, two_s = 2.0f / s2;
        const float e0 = qj * qj + qk * qk, e1 = qi * qj + qk * r, e2 = qi * qk - qj * r;
        const float g_two_s = -gd[0] * e0 + gd[1] * e1 + gd[2] * e2;
        const float g_s2 = g_two_s * (-2.0f / (s2 * s2));
        float gq[4];
        gq[0] = two_s * (gd[1] * qk - gd[2] * qj) + g_s2 * 2.f * r;
        gq[1] = two_s * (gd[1] * qj + gd[2] * qk) + g_s2 * 2.f * qi;
        gq[2] = two_s * (-2.f * gd[0] * qj + gd[1] * qi - gd[2] * r) + g_s2 * 2.f * qj;
        gq[3] = two_s * (-2.f * gd[0] * qk + gd[1] * r + gd[2] * qi) + g_s2 * 2.f * qk;
        float4 gr;
        if (nrm > 1e-12f) {
            const float dq = gq[0] * r + gq[1] * qi + gq[2] * qj + gq[3] * qk;
            gr = make_float4((gq[0] - r * dq) / nrm, (gq[1] - qi * dq) / nrm, (gq[2] - qj * dq) / nrm, (gq[3] - qk * dq) / nrm);
        } else {
            gr = make_float4(gq[0] / 1e-12f, gq[1] / 1e-12f, gq[2] / 1e-12f, gq[3] / 1e-12f);
        }
        reinterpret_cast<float4*>(g_rot_raw)[p] = gr;
        // opacity: every visible splat of curve b contributes log(1 + o^2 / 0.5)
        if (radii[p] > 0) {
            const float o = sigmoid_reg(opacity_logit[b]);
            v_op = logf(1.f + o * o / 0.5f);
            go_term = k_op * (2.f * o / 0.5f) / (1.f + o * o / 0.5f) * o * (1.f - o);
        }
        if (i == 0) {   // width: one term per curve
            const float w = expf(width[b]);
            const bool sel = w >= ra.width_thr;
            v_w = sel ? w - ra.width_thr : 0.f;
            g_width[b] = sel ? k_w * w : 0.f;
        }
    }
    s_go[threadIdx.x] = go_term;
    __syncthreads();
    if (valid && i == 0) {
        float sum = 0.f;
        for (int q = 0; q < m; q++) sum += s_go[threadIdx.x + q];
        g_opacity_logit[b] = sum;
    }
    // block sums of the three values -> partial slots
    float v[3] = {v_smo, v_op, v_w};
    __shared__ float s_w[3][4];
#pragma unroll
    for (int t = 0; t < 3; t++) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[t] += __shfl_xor(v[t], off, 64);
        if ((threadIdx.x & 63) == 0) s_w[t][threadIdx.x >> 6] = v[t];
    }
    __syncthreads();
    if (threadIdx.x < 3)
        atomicAdd(&sums[threadIdx.x * REG_SLOTS + blockIdx.x % REG_SLOTS],
                  (double)s_w[threadIdx.x][0] + (double)s_w[threadIdx.x][1] + (double)s_w[threadIdx.x][2] + (double)s_w[threadIdx.x][3]);
}

__global__ void __launch_bounds__(64) k_reg_finish(int B, int m, unsigned int* __restrict__ counts,
                                                   double* __restrict__ sums, RegArgs ra,
                                                   const float* __restrict__ op_gate, float* __restrict__ loss) {
    const int t = threadIdx.x;
    unsigned int c = counts[t];
    counts[t] = 0u;   // self-cleaning workspace
#pragma unroll
    for (int off = REG_SLOTS / 2; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    const float n_vis = (float)__shfl((int)c, 0, 64), n_w = (float)__shfl((int)c, REG_SLOTS, 64);
    double part[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        double s = t < REG_SLOTS ? sums[q * REG_SLOTS + t] : 0.0;
        if (t < REG_SLOTS) sums[q * REG_SLOTS + t] = 0.0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        part[q] = s;
    }
    if (t == 0) {
        const float gate = op_gate ? *op_gate : 1.f;
        const double smo = (double)ra.w_smo * (n_vis > 0.f ? 1.0 : 0.0) * part[0] / ((double)B * (double)(m - 1));
        const double op = (double)ra.w_op * gate * part[1] / fmax((double)n_vis, 1.0);
        const double wd = (double)ra.w_width * part[2] / fmax((double)n_w, 1.0);
        *loss = (float)(smo + op + wd);
    }
}

size_t curve_reg_workspace_bytes() { return 2 * REG_SLOTS * sizeof(unsigned int) + 3 * REG_SLOTS * sizeof(double); }
void launch_curve_regularizers(hipStream_t s, int B, int m, const float* rot_raw, const float* opacity_logit,
                               const float* width, const int* radii, float w_op, const float* op_gate, float w_smo,
                               float w_width, float width_thr, void* workspace, float* loss, float* g_rot_raw,
                               float* g_opacity_logit, float* g_width) {
    unsigned int* counts = reinterpret_cast<unsigned int*>(workspace);
    double* sums = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + 2 * REG_SLOTS * sizeof(unsigned int));
    const RegArgs ra{w_op, w_smo, w_width, width_thr};
    const int P = B * m;
    const int cpb = 256 / m;
    { ProfScope p("reg_count", s); hipLaunchKernelGGL(k_reg_count, dim3(std::min((P + 255) / 256, 512)), dim3(256), 0, s, P, B, radii, width, width_thr, counts); }
    { ProfScope p("reg_main", s); hipLaunchKernelGGL(k_reg_main, dim3((B + cpb - 1) / cpb), dim3(256), 0, s, B, m, cpb, rot_raw, opacity_logit, width, radii, counts, ra, op_gate, sums, g_rot_raw, g_opacity_logit, g_width); }
    { ProfScope p("reg_finish", s); hipLaunchKernelGGL(k_reg_finish, dim3(1), dim3(64), 0, s, B, m, counts, sums, ra, op_gate, loss); }
}

// ------------------------------------------------------------------------------------------------ end-point connection loss
// train.py:133-146: the 2B curve end points (B starts, then B ends), every ordered pair (i, j) of DIFFERENT curves closer
// than dis_thr; loss = weight * mean of those distances.  The reference materialises the full (2B)^2 cdist matrix and
// its masks (O(B^2) memory: 111 GB at B = 83 k); here every thread owns one point, finds its neighbours on a hashed
// uniform grid and keeps count, distance sum and the direction sum  g_i = sum_j (p_i - p_j) / d_ij  in registers.  The mean's
// denominator is only known at the end: a finish kernel scales, dL/dp_i = weight * 2 g_i / count (each unordered pair
// appears twice in the mean; a zero distance has zero gradient, as in torch.cdist).
constexpr int CONN_SLOTS = 64;
// Neighbour search on a hashed uniform grid: cells of edge dis_thr (+0.01 %: two points closer than dis_thr then differ by
// at most one cell per axis whatever the rounding), every point in the linked list of its cell's hash bucket; a query visits
// the 27 cells around its own and tests the points of those buckets whose cell really is the visited one (hash collisions
// only add candidates; a bucket shared by two visited cells is not counted twice).  O(B) time and memory.
struct ConnGrid { float inv_cell; uint32_t mask; };
__device__ __forceinline__ int3 conn_cell(float x, float y, float z, float inv_cell) {
    return make_int3((int)floorf(x * inv_cell), (int)floorf(y * inv_cell), (int)floorf(z * inv_cell));
}
__device__ __forceinline__ uint32_t conn_hash(int3 c, uint32_t mask) {
    return ((uint32_t)c.x * 73856093u ^ (uint32_t)c.y * 19349663u ^ (uint32_t)c.z * 83492791u) & mask;
}
__device__ __forceinline__ float4 conn_point(const float* __restrict__ cp, int B, int k) {
    // k < B: first control point of curve k; else last control point of curve k - B
    const float* q = cp + (size_t)(k < B ? k : k - B) * 12 + (k < B ? 0 : 9);
    return make_float4(q[0], q[1], q[2], 0.f);
}
// one launch initialises the workspace: list heads = ~0 (empty), partial slots and per-point direction sums = 0
__global__ void __launch_bounds__(256) k_conn_init(size_t n_zero, uint32_t* __restrict__ zero, size_t n_heads,
                                                   uint32_t* __restrict__ heads) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_zero; i += stride) zero[i] = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_heads; i += stride) heads[i] = ~0u;
}
// bucket = singly linked list through `next`, built with one atomic exchange per point (no count / scan / fill passes)
__global__ void __launch_bounds__(256) k_conn_build(int B, const float* __restrict__ cp, ConnGrid g,
                                                    uint32_t* __restrict__ heads, uint32_t* __restrict__ next) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * B) return;
    const float4 p = conn_point(cp, B, i);
    next[i] = atomicExch(&heads[conn_hash(conn_cell(p.x, p.y, p.z, g.inv_cell), g.mask)], (uint32_t)i);
}
// one thread per (point, neighbour cell): 27 N short independent list walks instead of N long dependent ones
__global__ void __launch_bounds__(256) k_conn_main(int B, const float* __restrict__ cp, float thr, ConnGrid g,
                                                   const uint32_t* __restrict__ heads, const uint32_t* __restrict__ next,
                                                   float* __restrict__ g_pt, unsigned long long* __restrict__ cnt_slots,
                                                   double* __restrict__ sum_slots) {
    const int N = 2 * B;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    unsigned int cnt = 0;
    float sum = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
    const int i = (int)(t / 27);
    if (i < N) {
        const int k = (int)(t - (long long)i * 27);
        const float4 pi = conn_point(cp, B, i);
        const int ci = i < B ? i : i - B;
        const int3 c0 = conn_cell(pi.x, pi.y, pi.z, g.inv_cell);
        const int3 c = make_int3(c0.x + k % 3 - 1, c0.y + (k / 3) % 3 - 1, c0.z + k / 9 - 1);
        for (uint32_t j = heads[conn_hash(c, g.mask)]; j != ~0u; j = next[j]) {
            const float4 pj = conn_point(cp, B, (int)j);
            const int3 cj = conn_cell(pj.x, pj.y, pj.z, g.inv_cell);
            if (cj.x != c.x || cj.y != c.y || cj.z != c.z) continue;     // another cell of this bucket
            const float ex = pi.x - pj.x, ey = pi.y - pj.y, ez = pi.z - pj.z;
            const float d = sqrtf(ex * ex + ey * ey + ez * ez);
            if (d < thr && ((int)j < B ? (int)j : (int)j - B) != ci) {
                cnt++;
                sum += d;
                if (d > 0.f) {
                    const float rr = 1.0f / d;
                    gx += ex * rr; gy += ey * rr; gz += ez * rr;
                }
            }
        }
        if (cnt) { atomicAdd(&g_pt[3 * i], gx); atomicAdd(&g_pt[3 * i + 1], gy); atomicAdd(&g_pt[3 * i + 2], gz); }
    }
    // block totals -> partial slots
    __shared__ unsigned int s_c[4];
    __shared__ float s_s[4];
    unsigned int c = cnt;
    float v = sum;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { c += __shfl_xor(c, off, 64); v += __shfl_xor(v, off, 64); }
    if ((threadIdx.x & 63) == 0) { s_c[threadIdx.x >> 6] = c; s_s[threadIdx.x >> 6] = v; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int ct = s_c[0] + s_c[1] + s_c[2] + s_c[3];
        if (ct) {
            atomicAdd(&cnt_slots[blockIdx.x % CONN_SLOTS], (unsigned long long)ct);
            atomicAdd(&sum_slots[blockIdx.x % CONN_SLOTS], (double)s_s[0] + (double)s_s[1] + (double)s_s[2] + (double)s_s[3]);
        }
    }
}
__global__ void __launch_bounds__(256) k_conn_finish(int B, float weight, const float* __restrict__ g_pt,
                                                     const unsigned long long* __restrict__ cnt_slots,
                                                     const double* __restrict__ sum_slots, float* __restrict__ loss,
                                                     float* __restrict__ dL_dcp, int accumulate) {
    unsigned long long c = cnt_slots[threadIdx.x & (CONN_SLOTS - 1)];
    double s = sum_slots[threadIdx.x & (CONN_SLOTS - 1)];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { c += __shfl_xor(c, off, 64); s += __shfl_xor(s, off, 64); }
    const double count = (double)c;
    if (blockIdx.x == 0 && threadIdx.x == 0) *loss = c ? (float)((double)weight * s / count) : 0.f;   // `if valid_mask.any()`
    const float scale = c ? (float)(2.0 * (double)weight / count) : 0.f;
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float* row = dL_dcp + (size_t)b * 12;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float gs = scale * g_pt[3 * b + k], ge = scale * g_pt[3 * (B + b) + k];
        if (accumulate) { row[k] += gs; row[9 + k] += ge; }
        else { row[k] = gs; row[3 + k] = 0.f; row[6 + k] = 0.f; row[9 + k] = ge; }
    }
}
// workspace: [slots: 64 u64 + 64 f64][g_pt 3N floats][next N][heads Hs]
static uint32_t conn_hash_size(int B) {
    uint32_t h = 1024;
    while (h < 4u * (uint32_t)B && h < (1u << 24)) h <<= 1;   // >= 2 buckets per point
    return h;
}
size_t endpoint_connection_workspace_bytes(int B) {
    const size_t Hs = conn_hash_size(B), N = 2 * (size_t)B;
    return CONN_SLOTS * 16 + N * 12 + N * 4 + Hs * 4 + 256;
}
void launch_endpoint_connection(hipStream_t s, int B, const float* cp, float thr, float weight, void* workspace, float* loss,
                                float* dL_dcp, int accumulate) {
    const uint32_t Hs = conn_hash_size(B);
    const int N = 2 * B;
    char* w = reinterpret_cast<char*>(workspace);
    unsigned long long* cnt_slots = reinterpret_cast<unsigned long long*>(w);
    double* sum_slots = reinterpret_cast<double*>(w + CONN_SLOTS * 8);
    float* g_pt = reinterpret_cast<float*>(w + CONN_SLOTS * 16);
    uint32_t* next = reinterpret_cast<uint32_t*>(g_pt + 3 * (size_t)N);
    uint32_t* heads = next + N;
    const ConnGrid g{1.0f / (thr * 1.0001f), Hs - 1u};
    const size_t zero_words = (CONN_SLOTS * 16) / 4 + 3 * (size_t)N;    // slots + g_pt
    const dim3 blk(256);
    { ProfScope p("conn_grid", s);
      hipLaunchKernelGGL(k_conn_init, dim3((unsigned)std::min<size_t>((std::max<size_t>(zero_words, Hs) + 255) / 256, 2048)), blk, 0, s,
                         zero_words, reinterpret_cast<uint32_t*>(w), (size_t)Hs, heads);
      hipLaunchKernelGGL(k_conn_build, dim3((N + 255) / 256), blk, 0, s, B, cp, g, heads, next); }
    { ProfScope p("conn_main", s);
      hipLaunchKernelGGL(k_conn_main, dim3((unsigned)(((long long)N * 27 + 255) / 256)), blk, 0, s, B, cp, thr, g, heads, next, g_pt, cnt_slots, sum_slots); }
    { ProfScope p("conn_finish", s); hipLaunchKernelGGL(k_conn_finish, dim3((B + 255) / 256), blk, 0, s, B, weight, g_pt, cnt_slots, sum_slots, loss, dL_dcp, accumulate); }
}

// ------------------------------------------------------------------------------------------------ flat Adam
// torch.optim.Adam (default, non-amsgrad, no weight decay) over ONE flat parameter buffer with per-segment learning
// rates -- the reference steps 6 parameter groups with ~8 foreach kernels each (GaussianCurveModel.training_setup,
// scene/gaussian_curve_model.py:200-213; train.py:235); here it is one launch.
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// The segment table travels BY VALUE in the kernel arguments: learning rates change every iteration (exponential
// schedule of the curve points, update_learning_rate :234-244) and a device-side table would need a host-to-device
// copy -- a stream-blocking hipMemcpy from pageable memory -- per step.
constexpr int ADAM_MAX_SEGS = 16;
struct AdamSeg { long long begin; float lr; float pad; };
struct AdamSegTable { AdamSeg s[ADAM_MAX_SEGS]; };
__global__ void __launch_bounds__(256) k_adam_flat(long long n, float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, AdamSegTable segs,
                                                   int nseg, float b1, float b2, float eps, float bc1, float sqrt_bc2,
                                                   int zero_grad) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float lr = segs.s[0].lr;
#pragma unroll
        for (int s = 1; s < ADAM_MAX_SEGS; s++) lr = (s < nseg && i >= segs.s[s].begin) ? segs.s[s].lr : lr;
        const float gi = g[i];
        if (zero_grad) g[i] = 0.f;  // optimizer.zero_grad() folded in (grads stay allocated: views of the flat buffer)
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);          // torch: exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;          // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;
        p[i] = p[i] - (lr / bc1) * (mi / denom);
    }
}
// Graph-replayable variant: the per-step scalars (segment learning rates, bias corrections) come from DEVICE memory
// (refreshed by a stream-ordered copy before each replay), and the update is skipped -- gradients still cleared --
// when *skip_flag != 0 (the captured forward raised its bucket-overflow flag: the host redoes that step eagerly).
struct AdamDevState { AdamSeg s[ADAM_MAX_SEGS]; float bc1; float sqrt_bc2; float pad[2]; };
__global__ void __launch_bounds__(256) k_adam_flat_dev(long long n, float* __restrict__ p, float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       const AdamDevState* __restrict__ st, int nseg, float b1, float b2,
                                                       float eps, int zero_grad, const unsigned int* __restrict__ skip_flag,
                                                       unsigned int* __restrict__ report_seq,
                                                       unsigned int* __restrict__ report_ring, int report_len) {
    const bool skip = skip_flag && *skip_flag != 0u;
    // Optional report for replayed iterations: the n-th execution (counted in device memory) leaves "was skipped" in entry
    // n % report_len of a ring the host can read (pinned host memory mapped into the device: no device-to-host copy in the
    // stream between one replay and the next).
    if (report_ring && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned int c = *report_seq;
        *report_seq = c + 1u;
        report_ring[c % (unsigned int)report_len] = skip ? 1u : 0u;
    }
    const float bc1 = st->bc1, sqrt_bc2 = st->sqrt_bc2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        if (zero_grad) g[i] = 0.f;
        if (skip) continue;
        float lr = st->s[0].lr;
        for (int s = 1; s < nseg; s++) lr = i >= st->s[s].begin ? st->s[s].lr : lr;
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;
        p[i] = p[i] - (lr / bc1) * (mi / denom);
    }
}
size_t adam_state_bytes() { return sizeof(AdamDevState); }
void launch_adam_flat_dev(hipStream_t s, long long n, float* p, float* g, float* m, float* v, const void* dev_state,
                          int nseg, float b1, float b2, float eps, int zero_grad, const unsigned int* skip_flag,
                          unsigned int* report_seq, unsigned int* report_ring, int report_len) {
    ProfScope pr("adam_flat", s);
    const int blocks = (int)std::min<long long>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_adam_flat_dev, dim3(blocks), dim3(256), 0, s, n, p, g, m, v,
                       reinterpret_cast<const AdamDevState*>(dev_state), nseg, b1, b2, eps, zero_grad, skip_flag, report_seq,
                       report_ring, report_len);
}
int adam_max_segments() { return ADAM_MAX_SEGS; }
void launch_adam_flat(hipStream_t s, long long n, float* p, float* g, float* m, float* v, const void* host_segs, int nseg,
                      float b1, float b2, float eps, float bc1, float sqrt_bc2, int zero_grad) {
    ProfScope pr("adam_flat", s);
    AdamSegTable t{};
    memcpy(t.s, host_segs, sizeof(AdamSeg) * (size_t)nseg);
    const int blocks = (int)std::min<long long>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_adam_flat, dim3(blocks), dim3(256), 0, s, n, p, g, m, v, t, nseg, b1, b2, eps, bc1, sqrt_bc2,
                       zero_grad);
}

}  // namespace cgs
