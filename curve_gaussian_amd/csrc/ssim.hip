// Fused SSIM forward (+ partial-derivative maps) and backward.
// Replaces the reference's fusedssimCUDA / fusedssim_backwardCUDA (submodules/fused-ssim/ssim.cu:187-286, :288-366):
// separable 11-tap Gaussian window (sigma 1.5, taps ssim.cu:9-19), zero padding ("same").
//
// MI355X mapping: one 256-thread workgroup per 32x32 output tile of one (batch, channel) plane.  The 42x42 halo of
// both images is staged once in LDS; the horizontal pass produces the five filtered rows (x1, x2, x1^2, x2^2, x1 x2)
// for all 42 rows into LDS, the vertical pass finishes them in registers (4 outputs per thread), and the SSIM map
// plus the three derivative maps are written with 128-byte row segments.  The reference re-loads and re-filters the
// tile five times with a barrier-separated scratch flush in between (ssim.cu:213-260); here every input pixel is read
// from HBM once per tile and every LDS element is written once.
#include "kernels.h"

namespace cgs {

__device__ constexpr float SSIM_G[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f,
                                         0.10936068743467331f,  0.21300552785396576f,   0.26601171493530273f,
                                         0.21300552785396576f,  0.10936068743467331f,   0.036000773310661316f,
                                         0.0075987582094967365f, 0.001028380123898387f};
constexpr int STX = 32, STY = 32, SR = 5, SSX = STX + 2 * SR, SSY = STY + 2 * SR;

__device__ __forceinline__ float pix_or_zero(const float* __restrict__ img, int y, int x, int H, int W) {
    return (x >= W || y >= H || x < 0 || y < 0) ? 0.0f : img[(size_t)y * W + x];  // ssim.cu:36-42
}

__global__ void __launch_bounds__(256) k_ssim_fwd(int H, int W, float C1, float C2, const float* __restrict__ img1,
                                                  const float* __restrict__ img2, float* __restrict__ ssim_map,
                                                  float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq,
                                                  float* __restrict__ dm_dsigma12) {
    __shared__ float s1[SSY][SSX + 1];
    __shared__ float s2[SSY][SSX + 1];
    __shared__ float hq[5][SSY][STX + 1];
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float* p1 = img1 + plane;
    const float* p2 = img2 + plane;
    const int x0 = blockIdx.x * STX, y0 = blockIdx.y * STY;
    const int tid = threadIdx.x;
    for (int t = tid; t < SSY * SSX; t += 256) {
        const int ly = t / SSX, lx = t - ly * SSX;
        s1[ly][lx] = pix_or_zero(p1, y0 + ly - SR, x0 + lx - SR, H, W);
        s2[ly][lx] = pix_or_zero(p2, y0 + ly - SR, x0 + lx - SR, H, W);
    }
    __syncthreads();
    const int tx = tid & 31, ty = tid >> 5;
    // horizontal pass: 42 rows x 32 columns
    for (int r = ty; r < SSY; r += 8) {
        float a1 = 0.f, a2 = 0.f, a11 = 0.f, a22 = 0.f, a12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float u = s1[r][tx + k], v = s2[r][tx + k], g = SSIM_G[k];
            a1 += g * u; a2 += g * v; a11 += g * (u * u); a22 += g * (v * v); a12 += g * (u * v);
        }
        hq[0][r][tx] = a1; hq[1][r][tx] = a2; hq[2][r][tx] = a11; hq[3][r][tx] = a22; hq[4][r][tx] = a12;
    }
    __syncthreads();
    // vertical pass + SSIM
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int oy = ty + 8 * j;
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float g = SSIM_G[k];
            mu1 += g * hq[0][oy + k][tx]; mu2 += g * hq[1][oy + k][tx];
            e11 += g * hq[2][oy + k][tx]; e22 += g * hq[3][oy + k][tx]; e12 += g * hq[4][oy + k][tx];
        }
        const int px = x0 + tx, py = y0 + oy;
        if (px < W && py < H) {
            const float sigma1_sq = e11 - mu1 * mu1, sigma2_sq = e22 - mu2 * mu2, sigma12 = e12 - mu1 * mu2;
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
            const float C = 2.0f * mu1_mu2 + C1, D = 2.0f * sigma12 + C2;
            const float A = mu1_sq + mu2_sq + C1, B = sigma1_sq + sigma2_sq + C2;
            const size_t o = plane + (size_t)py * W + px;
            ssim_map[o] = (C * D) / (A * B);
            if (dm_dmu1) {  // ssim.cu:274-283
                dm_dmu1[o] = (mu2 * 2.0f * D) / (A * B) - (mu2 * 2.0f * C) / (A * B) - (mu1 * 2.0f * C * D) / (A * A * B) +
                             (mu1 * 2.0f * C * D) / (A * B * B);
                dm_dsigma1_sq[o] = (-C * D) / (A * B * B);
                dm_dsigma12[o] = (2 * C) / (A * B);
            }
        }
    }
}

// dL/dimg1 = G*(dL_dmap dm_dmu1) + 2 img1 G*(dL_dmap dm_dsigma1_sq) + img2 G*(dL_dmap dm_dsigma12)   (ssim.cu:315-365)
__global__ void __launch_bounds__(256) k_ssim_bwd(int H, int W, const float* __restrict__ img1,
                                                  const float* __restrict__ img2, const float* __restrict__ dL_dmap,
                                                  const float* __restrict__ dm_dmu1,
                                                  const float* __restrict__ dm_dsigma1_sq,
                                                  const float* __restrict__ dm_dsigma12, float* __restrict__ dL_dimg1) {
    __shared__ float s[3][SSY][SSX + 1];
    __shared__ float hq[3][SSY][STX + 1];
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * STX, y0 = blockIdx.y * STY;
    const int tid = threadIdx.x;
    for (int t = tid; t < SSY * SSX; t += 256) {
        const int ly = t / SSX, lx = t - ly * SSX;
        const int y = y0 + ly - SR, x = x0 + lx - SR;
        float a = 0.f, b = 0.f, c = 0.f;
        if (x >= 0 && y >= 0 && x < W && y < H) {
            const size_t o = plane + (size_t)y * W + x;
            const float g = dL_dmap[o];
            a = dm_dmu1[o] * g; b = dm_dsigma1_sq[o] * g; c = dm_dsigma12[o] * g;
        }
        s[0][ly][lx] = a; s[1][ly][lx] = b; s[2][ly][lx] = c;
    }
    __syncthreads();
    const int tx = tid & 31, ty = tid >> 5;
    for (int r = ty; r < SSY; r += 8) {
        float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float g = SSIM_G[k];
            a += g * s[0][r][tx + k]; b += g * s[1][r][tx + k]; c += g * s[2][r][tx + k];
        }
        hq[0][r][tx] = a; hq[1][r][tx] = b; hq[2][r][tx] = c;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int oy = ty + 8 * j;
        float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float g = SSIM_G[k];
            a += g * hq[0][oy + k][tx]; b += g * hq[1][oy + k][tx]; c += g * hq[2][oy + k][tx];
        }
        const int px = x0 + tx, py = y0 + oy;
        if (px < W && py < H) {
            const size_t o = plane + (size_t)py * W + px;
            float dL = a;
            dL += img1[o] * 2.0f * b;
            dL += img2[o] * c;
            dL_dimg1[o] = dL;
        }
    }
}

void launch_ssim_fwd(hipStream_t s, int planes, int H, int W, float C1, float C2, const float* img1, const float* img2,
                     float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12) {
    ProfScope p("ssim_fwd", s);
    hipLaunchKernelGGL(k_ssim_fwd, dim3((W + STX - 1) / STX, (H + STY - 1) / STY, planes), dim3(256), 0, s, H, W, C1, C2,
                       img1, img2, ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12);
}
void launch_ssim_bwd(hipStream_t s, int planes, int H, int W, const float* img1, const float* img2,
                     const float* dL_dmap, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                     float* dL_dimg1) {
    ProfScope p("ssim_bwd", s);
    hipLaunchKernelGGL(k_ssim_bwd, dim3((W + STX - 1) / STX, (H + STY - 1) / STY, planes), dim3(256), 0, s, H, W, img1,
                       img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
}

}  // namespace cgs
