// f32-input MFMA beside VALU work (gfx950): fragment layouts of the two forms the compositors use, and whether an MFMA
// issued between VALU instructions costs VALU issue time.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// ---- layouts: every lane supplies a = f(lane), b = g(lane) and the full D is dumped
__global__ void k_layout32(float* out) {
    const int l = threadIdx.x;
    v16f c = {};
    // A[i][k] = 100 i + k + 1,  B[k][j] = 1000 (k + 1) + j   (asymmetric)  with i = l & 31, k = l >> 5 (A), k = l >> 5, j = l & 31 (B)
    const float a = 100.f * (l & 31) + (l >> 5) + 1.f;
    const float b = 1000.f * ((l >> 5) + 1) + (l & 31);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) out[l * 16 + r] = c[r];
}
__global__ void k_layout4(float* out) {
    const int l = threadIdx.x;
    v4f c = {};
    // block = l >> 2;  A[block][i = l & 3] = 10 l + 1,  B[block][j = l & 3] = 1000 + l
    const float a = 10.f * l + 1.f;
    const float b = 1000.f + l;
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[l * 4 + r] = c[r];
}

// ---- issue cost: NV v_fma per trip on 8 independent chains, plus NM MFMAs of the given form on rotating accumulators
template <int NV, int NM4, int NM32>
__global__ void __launch_bounds__(256) k_mix(float* out, float a, float b, int iters) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 0.001f + i;
    v4f acc4[4] = {};
    v16f acc32[2] = {};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NV; i++) x[i & 7] = __builtin_fmaf(x[i & 7], a, b);
#pragma unroll
        for (int m = 0; m < NM4; m++) acc4[m & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[m & 7], b, acc4[m & 3], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NM32; m++) acc32[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[m & 7], b, acc32[m & 1], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
#pragma unroll
    for (int m = 0; m < 4; m++) s += acc4[m][0] + acc4[m][3];
    s += acc32[0][0] + acc32[1][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int NM4, int NM32>
static void run_mix(const char* name, float* out, int waves_per_simd) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const int blocks = 256 * waves_per_simd;   // 256-thread blocks: one wave per SIMD each
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_mix<NV, NM4, NM32>), dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double trips_per_simd = (double)blocks * 4 * iters / 1024.0;
    printf("%-44s %d waves/SIMD  %.3f ms -> %.1f cycles per trip per SIMD (2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / trips_per_simd);
}

int main() {
    float* out; hipMalloc(&out, 256 * 4096 * 4 * 4);
    {
        hipLaunchKernelGGL(k_layout32, dim3(1), dim3(64), 0, 0, out);
        std::vector<float> h(64 * 16);
        hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < 16; r++) {
                const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                double e = 0;
                for (int k = 0; k < 2; k++) e += (100.0 * row + k + 1) * (1000.0 * (k + 1) + col);
                if (h[l * 16 + r] != (float)e) { if (bad < 5) printf("  32x32x2 lane %d reg %d: got %.1f want %.1f\n", l, r, h[l * 16 + r], e); bad++; }
            }
        printf("mfma_f32_32x32x2f32 layout (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D col=l&31 row=(r&3)+8(r>>2)+4(l>>5)): %s\n", bad ? "MISMATCH" : "confirmed");
    }
    {
        hipLaunchKernelGGL(k_layout4, dim3(1), dim3(64), 0, 0, out);
        std::vector<float> h(64 * 4);
        hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < 4; r++) {
                const int blk = l >> 2, j = l & 3, i = r;     // hypothesis: lane = (block, column j), reg = row i
                const double e = (10.0 * (4 * blk + i) + 1) * (1000.0 + 4 * blk + j);
                if (h[l * 4 + r] != (float)e) { if (bad < 5) printf("  4x4x1 lane %d reg %d: got %.1f want %.1f\n", l, r, h[l * 4 + r], e); bad++; }
            }
        printf("mfma_f32_4x4x1f32 layout (block=l>>2, A row i=l&3, B col j=l&3, D lane=(block,j) reg=i): %s\n", bad ? "MISMATCH" : "confirmed");
        if (bad) for (int l = 0; l < 8; l++) printf("  lane %d: %.1f %.1f %.1f %.1f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    for (int w : {8, 4, 2}) {
        run_mix<16, 0, 0>("16 v_fma", out, w);
        run_mix<16, 1, 0>("16 v_fma + 1 mfma_4x4x1", out, w);
        run_mix<16, 2, 0>("16 v_fma + 2 mfma_4x4x1", out, w);
        run_mix<16, 4, 0>("16 v_fma + 4 mfma_4x4x1", out, w);
        run_mix<0, 4, 0>("4 mfma_4x4x1", out, w);
        run_mix<64, 0, 1>("64 v_fma + 1 mfma_32x32x2", out, w);
        run_mix<64, 0, 0>("64 v_fma", out, w);
        run_mix<0, 0, 2>("2 mfma_32x32x2", out, w);
    }
    return 0;
}
