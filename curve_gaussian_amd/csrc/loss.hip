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
                                                       float eps, int zero_grad, const unsigned int* __restrict__ skip_flag) {
    const bool skip = skip_flag && *skip_flag != 0u;
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
                          int nseg, float b1, float b2, float eps, int zero_grad, const unsigned int* skip_flag) {
    ProfScope pr("adam_flat", s);
    const int blocks = (int)std::min<long long>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_adam_flat_dev, dim3(blocks), dim3(256), 0, s, n, p, g, m, v,
                       reinterpret_cast<const AdamDevState*>(dev_state), nseg, b1, b2, eps, zero_grad, skip_flag);
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
