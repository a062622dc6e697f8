// Fused SSIM forward (+ partial-derivative maps) and backward.
// Replaces the reference's fusedssimCUDA / fusedssim_backwardCUDA (submodules/fused-ssim/ssim.cu:187-286, :288-366):
// separable 11-tap Gaussian window (sigma 1.5, taps ssim.cu:9-19), zero padding ("same").
//
// MI355X mapping: one 512-thread workgroup per 32 x 54 output tile of one (batch, channel) plane.  The 42 x 64 halo of
// both images is staged once in LDS; the horizontal pass produces the five filtered rows (x1, x2, x1^2, x2^2, x1 x2)
// for all 64 rows into LDS, the vertical pass finishes them in registers (4 outputs per thread), and the SSIM map
// plus the three derivative maps are written with 128-byte row segments.  The reference re-loads and re-filters the
// tile five times with a barrier-separated scratch flush in between (ssim.cu:213-260); here every input pixel is read
// from HBM once per tile and every LDS element is written once.
#include "kernels.h"

namespace cgs {

__device__ constexpr float SSIM_G[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f,
                                         0.10936068743467331f,  0.21300552785396576f,   0.26601171493530273f,
                                         0.21300552785396576f,  0.10936068743467331f,   0.036000773310661316f,
                                         0.0075987582094967365f, 0.001028380123898387f};
// Tile: 32 columns x 54 rows of output per 512-thread workgroup.  The halo is then 42 x 64: the horizontal pass has exactly
// 64 rows x 8 groups of four columns = 512 items = one per thread (the 32 x 32 tile of rounds 1-2 filtered 42 rows for 32 of
// output, 1.31x, in two rounds with a third of the threads idle in the second: 1.56x issue slots), and every staged row is
// one 42-lane row segment of a wave (row-wise staging: one add of the pitch per element instead of a division by 42).
constexpr int STX = 32, STY = 54, SR = 5, SSX = STX + 2 * SR, SSY = STY + 2 * SR;
constexpr int STH = 512;                   // threads per workgroup: one horizontal item each
constexpr int VR = 4;                      // output rows per thread in the vertical pass (16 row groups x 4 >= 54)
constexpr int SROWS = SSY / (STH / 64);    // halo rows staged per wave
static_assert(SSY == 64 && SSY * (STX / 4) == STH && (STH / 32) * VR >= STY, "tile geometry");

__device__ __forceinline__ float pix_or_zero(const float* __restrict__ img, int y, int x, int H, int W) {
    return (x >= W || y >= H || x < 0 || y < 0) ? 0.0f : img[(size_t)y * W + x];  // ssim.cu:36-42
}

// Training-step fusion (cgs_photometric_loss): FUSED = true additionally
//   forward : clamps img1 to [0,1] on load (render().clamp(0,1), gaussian_renderer/__init__.py:138), skips the
//             ssim_map store and accumulates SUM ssim_map into 64 partial f64 slots (fused_ssim(...).mean());
//   backward: uses a constant dL/dmap, adds the edge_aware_loss gradient (utils/loss_utils.py:94-115) in the epilogue,
//             accumulates its value, and applies the clamp's gradient mask -- one store of d loss / d image.
constexpr int PHOTO_SLOTS = 64;
struct PhotoArgs {
    int clamp;                 // img1 is the UNclamped render: clamp on load, mask the gradient
    float dmap_const;          // d loss / d ssim_map (constant): -lambda_b / N
    float edge_scale;          // lambda_a * 2 / N
    float thr;                 // edge threshold on gt
    const unsigned int* n_pos; // #{gt > thr} (device scalar, cached per gt image; a table when view_index is set)
    const int* view_index;     // optional device scalar v: img2 is a stack of images and n_pos a table, use entry v
    double* ssim_slots;        // [PHOTO_SLOTS]
    double* edge_slots;        // [PHOTO_SLOTS]
};
__device__ __forceinline__ void block_sum_to_slot(float v, double* slots) {
    __shared__ float s_w[STH / 64];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < STH / 64; w++) t += (double)s_w[w];
        atomicAdd(&slots[(blockIdx.x + blockIdx.y * gridDim.x) % PHOTO_SLOTS], t);
    }
}
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// Tile of this workgroup.  Workgroups are dealt to the 8 XCDs round-robin by their linear index, each XCD with its own L2:
// with the natural order horizontally adjacent tiles -- whose 42-column halos overlap by 10 columns -- never share an L2
// (measured on the fused forward at 1600^2: 73 % of its L2 requests miss, 1.56x the image bytes fetched).  Here XCD k takes
// the k-th eighth of the tiles in row-major order, so neighbours meet in one L2.
#ifndef CGS_SSIM_XCD_ORDER
#define CGS_SSIM_XCD_ORDER 1
#endif
__device__ __forceinline__ void ssim_tile(int& bx, int& by) {
    bx = (int)blockIdx.x; by = (int)blockIdx.y;
    if (CGS_SSIM_XCD_ORDER) {
        const uint32_t gx = gridDim.x, T = gridDim.x * gridDim.y, lin = blockIdx.y * gx + blockIdx.x, q = T / 8u;
        if (lin < 8u * q) {
            const uint32_t t = (lin % 8u) * q + lin / 8u;
            bx = (int)(t % gx); by = (int)(t / gx);
        }
    }
}

template <bool FUSED>
__global__ void __launch_bounds__(STH) k_ssim_fwd(int H, int W, float C1, float C2, const float* __restrict__ img1,
                                                  const float* __restrict__ img2, float* __restrict__ ssim_map,
                                                  float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq,
                                                  float* __restrict__ dm_dsigma12, PhotoArgs pa) {
    // the staged halos (s1, s2) are dead once every thread holds its horizontal windows in registers, so the filtered
    // rows (hq) reuse their LDS: 42 KB per workgroup
    __shared__ float smem[5 * SSY * (STX + 1)];
    float (*s1)[SSX + 1] = reinterpret_cast<float (*)[SSX + 1]>(smem);
    float (*s2)[SSX + 1] = reinterpret_cast<float (*)[SSX + 1]>(smem + SSY * (SSX + 1));
    float (*hq)[SSY][STX + 1] = reinterpret_cast<float (*)[SSY][STX + 1]>(smem);
    const size_t plane = (size_t)blockIdx.z * H * W;
    const size_t view = (FUSED && pa.view_index) ? (size_t)pa.view_index[0] : 0;
    const float* p1 = img1 + plane;
    const float* p2 = img2 + plane + view * ((size_t)H * W);
    int tbx, tby;
    ssim_tile(tbx, tby);
    const int x0 = tbx * STX, y0 = tby * STY;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // halo staging, row-wise: wave w stages halo rows 16 w .. 16 w + 15, lane = halo column (42 of 64 lanes).  Two phases --
    // all global loads first (addresses clamped into the image, out-of-image values zeroed afterwards: ssim.cu:36-42), then
    // the LDS stores: a loop that waits for its loads every trip costs a memory round trip per element
    {
        const int lx = min(lane, SSX - 1);
        const int x = x0 + lx - SR;
        const bool xin = lane < SSX && x >= 0 && x < W;
        const int xc = min(max(x, 0), W - 1);
        float g1[SROWS], g2[SROWS];
#pragma unroll
        for (int e = 0; e < SROWS; e++) {
            const int y = y0 + wave * SROWS + e - SR;
            const size_t o = (size_t)min(max(y, 0), H - 1) * W + xc;
            const float v1 = p1[o], v2 = p2[o];
            const bool inb = xin && y >= 0 && y < H;
            g1[e] = inb ? v1 : 0.f;
            g2[e] = inb ? v2 : 0.f;
        }
        if (lane < SSX) {
#pragma unroll
            for (int e = 0; e < SROWS; e++) {
                s1[wave * SROWS + e][lane] = (FUSED && pa.clamp) ? clamp01(g1[e]) : g1[e];
                s2[wave * SROWS + e][lane] = g2[e];
            }
        }
    }
    __syncthreads();
    // Both passes are register-blocked along the filter direction: a thread produces neighbouring outputs from a sliding
    // window (horizontal: 4 outputs from 14 values, 3.5 LDS reads per output and quantity instead of 11); every output still
    // sums its 11 taps in the reference's order.
    // horizontal pass: 64 rows x 8 groups of 4 columns = 512 items, two per thread; windows first, then (after a barrier,
    // because hq overwrites s1/s2) the filtering
    float uw[1][14], vw[1][14];
#pragma unroll
    for (int e = 0; e < 1; e++) {
        const int it = tid;
        const int r = it >> 3, c0 = (it & 7) * 4;
#pragma unroll
        for (int k = 0; k < 14; k++) { uw[e][k] = s1[r][c0 + k]; vw[e][k] = s2[r][c0 + k]; }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 1; e++) {
        const int it = tid;
        const int r = it >> 3, c0 = (it & 7) * 4;
        const float (&u)[14] = uw[e];
        const float (&v)[14] = vw[e];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float a1 = 0.f, a2 = 0.f, a11 = 0.f, a22 = 0.f, a12 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float uu = u[j + k], vv = v[j + k], g = SSIM_G[k];
                a1 += g * uu; a2 += g * vv; a11 += g * (uu * uu); a22 += g * (vv * vv); a12 += g * (uu * vv);
            }
            hq[0][r][c0 + j] = a1; hq[1][r][c0 + j] = a2; hq[2][r][c0 + j] = a11; hq[3][r][c0 + j] = a22; hq[4][r][c0 + j] = a12;
        }
    }
    __syncthreads();
    // vertical pass + SSIM: thread = (column tx, VR consecutive rows), one quantity at a time (14-value window -> 4 outputs)
    const int tx = tid & 31, rg = tid >> 5;
    float res[5][VR];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        float win[VR + 10];
#pragma unroll
        for (int k = 0; k < VR + 10; k++) win[k] = hq[q][min(VR * rg + k, SSY - 1)][tx];   // (rows past the halo feed no valid output)
#pragma unroll
        for (int j = 0; j < VR; j++) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) a += SSIM_G[k] * win[j + k];
            res[q][j] = a;
        }
    }
    float ssim_acc = 0.f;
#pragma unroll
    for (int j = 0; j < VR; j++) {
        const int oy = VR * rg + j;
        const float mu1 = res[0][j], mu2 = res[1][j], e11 = res[2][j], e22 = res[3][j], e12 = res[4][j];
        const int px = x0 + tx, py = y0 + oy;
        if (oy < STY && px < W && py < H) {
            const float sigma1_sq = e11 - mu1 * mu1, sigma2_sq = e22 - mu2 * mu2, sigma12 = e12 - mu1 * mu2;
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
            const float C = 2.0f * mu1_mu2 + C1, D = 2.0f * sigma12 + C2;
            const float A = mu1_sq + mu2_sq + C1, B = sigma1_sq + sigma2_sq + C2;
            const size_t o = plane + (size_t)py * W + px;
            // One reciprocal (v_rcp_f32 + a Newton step: <= 1 ulp) instead of the reference's seven divisions per pixel
            // (ssim.cu:268-283); the four terms of dm/dmu1 are regrouped over the common denominator:
            //   2 mu2 D/(AB) - 2 mu2 C/(AB) - 2 mu1 CD/(A^2 B) + 2 mu1 CD/(A B^2) = 2/(AB) [mu2 (D - C) - mu1 CD (B - A)/(AB)].
            const float AB = A * B;
            float rAB = __builtin_amdgcn_rcpf(AB);
            rAB = fmaf(fmaf(-AB, rAB, 1.0f), rAB, rAB);
            const float CD = C * D;
            const float val = CD * rAB;
            if (FUSED) ssim_acc += val; else ssim_map[o] = val;
            if (dm_dmu1) {  // ssim.cu:274-283
                dm_dmu1[o] = 2.0f * rAB * (mu2 * (D - C) - mu1 * (val * (B - A)));
                dm_dsigma1_sq[o] = -val * (A * rAB);
                dm_dsigma12[o] = 2.0f * C * rAB;
            }
        }
    }
    if (FUSED) block_sum_to_slot(ssim_acc, pa.ssim_slots);
}

// dL/dimg1 = G*(dL_dmap dm_dmu1) + 2 img1 G*(dL_dmap dm_dsigma1_sq) + img2 G*(dL_dmap dm_dsigma12)   (ssim.cu:315-365)
template <bool FUSED>
__global__ void __launch_bounds__(STH) k_ssim_bwd(int H, int W, const float* __restrict__ img1,
                                                  const float* __restrict__ img2, const float* __restrict__ dL_dmap,
                                                  const float* __restrict__ dm_dmu1,
                                                  const float* __restrict__ dm_dsigma1_sq,
                                                  const float* __restrict__ dm_dsigma12, float* __restrict__ dL_dimg1,
                                                  PhotoArgs pa) {
    __shared__ float smem[3 * SSY * (SSX + 1)];   // s[3][64][43]; hq[3][64][33] reuses it (see k_ssim_fwd)
    float (*s)[SSY][SSX + 1] = reinterpret_cast<float (*)[SSY][SSX + 1]>(smem);
    float (*hq)[SSY][STX + 1] = reinterpret_cast<float (*)[SSY][STX + 1]>(smem);
    const size_t plane = (size_t)blockIdx.z * H * W;
    int tbx, tby;
    ssim_tile(tbx, tby);
    const int x0 = tbx * STX, y0 = tby * STY;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (row-wise two-phase halo staging, see k_ssim_fwd)
    {
        const int lx = min(lane, SSX - 1);
        const int x = x0 + lx - SR;
        const bool xin = lane < SSX && x >= 0 && x < W;
        const int xc = min(max(x, 0), W - 1);
        float ga[SROWS], gb[SROWS], gc[SROWS];
#pragma unroll
        for (int e = 0; e < SROWS; e++) {
            const int y = y0 + wave * SROWS + e - SR;
            const size_t o = plane + (size_t)min(max(y, 0), H - 1) * W + xc;
            const bool inb = xin && y >= 0 && y < H;
            const float g = !inb ? 0.f : (FUSED ? pa.dmap_const : dL_dmap[o]);
            ga[e] = dm_dmu1[o] * g; gb[e] = dm_dsigma1_sq[o] * g; gc[e] = dm_dsigma12[o] * g;
        }
        if (lane < SSX) {
#pragma unroll
            for (int e = 0; e < SROWS; e++) {
                s[0][wave * SROWS + e][lane] = ga[e]; s[1][wave * SROWS + e][lane] = gb[e]; s[2][wave * SROWS + e][lane] = gc[e];
            }
        }
    }
    // the epilogue's image values, requested now: their latency hides behind the two filter passes
    const int tx = tid & 31, rg = tid >> 5;
    const size_t gt_view = (FUSED && pa.view_index) ? (size_t)pa.view_index[0] * ((size_t)H * W) : 0;
    float xs[VR], ys[VR];
#pragma unroll
    for (int j = 0; j < VR; j++) {
        const int px = min(x0 + tx, W - 1), py = min(y0 + VR * rg + j, H - 1);
        const size_t o = plane + (size_t)py * W + px;
        xs[j] = img1[o];
        ys[j] = img2[o + gt_view];
    }
    __syncthreads();
    // register-blocked passes (see k_ssim_fwd)
    float ww[1][3][14];
#pragma unroll
    for (int e = 0; e < 1; e++) {
        const int it = tid;
        const int r = it >> 3, c0 = (it & 7) * 4;
#pragma unroll
        for (int k = 0; k < 14; k++) { ww[e][0][k] = s[0][r][c0 + k]; ww[e][1][k] = s[1][r][c0 + k]; ww[e][2][k] = s[2][r][c0 + k]; }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 1; e++) {
        const int it = tid;
        const int r = it >> 3, c0 = (it & 7) * 4;
        const float (&w0)[14] = ww[e][0];
        const float (&w1)[14] = ww[e][1];
        const float (&w2)[14] = ww[e][2];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float g = SSIM_G[k];
                a += g * w0[j + k]; b += g * w1[j + k]; c += g * w2[j + k];
            }
            hq[0][r][c0 + j] = a; hq[1][r][c0 + j] = b; hq[2][r][c0 + j] = c;
        }
    }
    __syncthreads();
    float res[3][VR];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        float win[VR + 10];
#pragma unroll
        for (int k = 0; k < VR + 10; k++) win[k] = hq[q][min(VR * rg + k, SSY - 1)][tx];
#pragma unroll
        for (int j = 0; j < VR; j++) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) a += SSIM_G[k] * win[j + k];
            res[q][j] = a;
        }
    }
    float w_pos = 0.f, w_neg = 0.f, edge_acc = 0.f;
    if (FUSED) {  // loss_utils.py:100-108 (weights from the class balance of the gt edge mask)
        const float n_pos = (float)pa.n_pos[pa.view_index ? pa.view_index[0] : 0], n_neg = (float)H * (float)W - n_pos;
        w_pos = 5.f * (n_neg + 1.f) / (n_pos + n_neg);
        w_neg = 1.0f * (n_pos + 1.f) / (n_pos + n_neg);
    }
#pragma unroll
    for (int j = 0; j < VR; j++) {
        const int oy = VR * rg + j;
        const float a = res[0][j], b = res[1][j], c = res[2][j];
        const int px = x0 + tx, py = y0 + oy;
        if (oy < STY && px < W && py < H) {
            const size_t o = plane + (size_t)py * W + px;
            const float x = xs[j], y = ys[j];
            const float xc = (FUSED && pa.clamp) ? clamp01(x) : x;
            float dL = a;
            dL += xc * 2.0f * b;
            dL += y * c;
            if (FUSED) {
                const float d = xc - y, w = y > pa.thr ? w_pos : w_neg;
                edge_acc += d * d * w;
                dL += pa.edge_scale * d * w;
                if (pa.clamp && (x < 0.f || x > 1.f)) dL = 0.f;  // clamp backward: gradient only where 0 <= x <= 1
            }
            dL_dimg1[o] = dL;
        }
    }
    if (FUSED) block_sum_to_slot(edge_acc, pa.edge_slots);
}

// loss = lambda_a * edge_sum / N + lambda_b * (1 - ssim_sum / N); the slots are cleared for the next call
__global__ void __launch_bounds__(64) k_photo_finish(double* __restrict__ ssim_slots, double* __restrict__ edge_slots,
                                                     float lambda_a, float lambda_b, double inv_n,
                                                     float* __restrict__ loss) {
    double s = ssim_slots[threadIdx.x], e = edge_slots[threadIdx.x];
    ssim_slots[threadIdx.x] = 0.0;
    edge_slots[threadIdx.x] = 0.0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s += __shfl_xor(s, off, 64);
        e += __shfl_xor(e, off, 64);
    }
    if (threadIdx.x == 0) *loss = (float)((double)lambda_a * e * inv_n + (double)lambda_b - (double)lambda_b * s * inv_n);
}

void launch_ssim_fwd(hipStream_t s, int planes, int H, int W, float C1, float C2, const float* img1, const float* img2,
                     float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12) {
    ProfScope p("ssim_fwd", s);
    hipLaunchKernelGGL(k_ssim_fwd<false>, dim3((W + STX - 1) / STX, (H + STY - 1) / STY, planes), dim3(STH), 0, s, H, W,
                       C1, C2, img1, img2, ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, PhotoArgs{});
}
void launch_ssim_bwd(hipStream_t s, int planes, int H, int W, const float* img1, const float* img2,
                     const float* dL_dmap, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                     float* dL_dimg1) {
    ProfScope p("ssim_bwd", s);
    hipLaunchKernelGGL(k_ssim_bwd<false>, dim3((W + STX - 1) / STX, (H + STY - 1) / STY, planes), dim3(STH), 0, s, H, W,
                       img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1, PhotoArgs{});
}

// workspace: [3 * H*W floats: dm_dmu1, dm_dsigma1_sq, dm_dsigma12][2 * PHOTO_SLOTS doubles, zero on first use]
size_t photometric_workspace_bytes(int H, int W) {
    return (((size_t)3 * H * W * sizeof(float) + 127) & ~(size_t)127) + 2 * PHOTO_SLOTS * sizeof(double);
}
void launch_photometric_loss(hipStream_t s, int H, int W, const float* image, const float* gt, const int* view_index,
                             float thr, const unsigned int* n_pos, float lambda_a, float lambda_b, int clamp,
                             void* workspace, float* grad, float* loss) {
    const size_t N = (size_t)H * W;
    float* dm1 = reinterpret_cast<float*>(workspace);
    float* dm2 = dm1 + N;
    float* dm3 = dm2 + N;
    double* slots = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + ((3 * N * sizeof(float) + 127) & ~(size_t)127));
    PhotoArgs pa;
    pa.clamp = clamp;
    pa.dmap_const = -lambda_b / (float)N;
    pa.edge_scale = lambda_a * 2.f / (float)N;
    pa.thr = thr;
    pa.n_pos = n_pos;
    pa.view_index = view_index;
    pa.ssim_slots = slots;
    pa.edge_slots = slots + PHOTO_SLOTS;
    const dim3 grid((W + STX - 1) / STX, (H + STY - 1) / STY, 1);
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;  // fused_ssim/__init__.py: C1 = 0.01**2, C2 = 0.03**2
    { ProfScope p("ssim_fwd", s); hipLaunchKernelGGL(k_ssim_fwd<true>, grid, dim3(STH), 0, s, H, W, C1, C2, image, gt, nullptr, dm1, dm2, dm3, pa); }
    { ProfScope p("ssim_bwd", s); hipLaunchKernelGGL(k_ssim_bwd<true>, grid, dim3(STH), 0, s, H, W, image, gt, nullptr, dm1, dm2, dm3, grad, pa); }
    { ProfScope p("photo_finish", s); hipLaunchKernelGGL(k_photo_finish, dim3(1), dim3(64), 0, s, pa.ssim_slots, pa.edge_slots, lambda_a, lambda_b, 1.0 / (double)N, loss); }
}

}  // namespace cgs
