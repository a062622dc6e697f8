// Fused class-balanced edge loss (reference utils/loss_utils.py:94-115, used at train.py:101) -- ~15 PyTorch kernels
// and two host-visible reductions in the reference, two kernels here:
//   k_edge_count : n_pos = #{ mean_c gt > thr }
//   k_edge_loss  : loss = mean_{c,y,x} (image - gt)^2 * w(y,x),  w = 5 (n_neg+1)/N if edge else (n_pos+1)/N, N = H W
//                  and d loss / d image in the same pass.
#include <algorithm>

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
    unsigned int w = cnt;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) w += __shfl_xor(w, off, 64);
    if ((threadIdx.x & 63) == 0 && w) atomicAdd(n_pos, w);
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

void launch_edge_aware_loss(hipStream_t s, int C, int H, int W, const float* image, const float* gt, float thr,
                            void* scratch16, float* grad) {
    const int HW = H * W;
    unsigned int* n_pos = reinterpret_cast<unsigned int*>(scratch16);
    double* loss_sum = reinterpret_cast<double*>(reinterpret_cast<char*>(scratch16) + 8);
    const int blocks = std::min((HW + 255) / 256, 512);
    { ProfScope p("edge_count", s); hipLaunchKernelGGL(k_edge_count, dim3(blocks), dim3(256), 0, s, C, HW, gt, thr, n_pos); }
    { ProfScope p("edge_loss", s); hipLaunchKernelGGL(k_edge_loss, dim3(blocks), dim3(256), 0, s, C, HW, image, gt, thr, n_pos, loss_sum, grad); }
}

}  // namespace cgs
