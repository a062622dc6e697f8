// Does a bf16 MFMA issued between VALU instructions cost VALU time on gfx950?  (The f32-input forms do: mfma_probe.hip.)
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o mfma_bf16_probe mfma_bf16_probe.hip && ./mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int NV, int NM32, int NM16>
__global__ void __launch_bounds__(256) k_mix(float* out, float a, float b, int iters) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 0.001f + i;
    v16f acc32[2] = {};
    v4f acc16[4] = {};
    v8bf fa, fb;
#pragma unroll
    for (int i = 0; i < 8; i++) { fa[i] = (__bf16)(a + i); fb[i] = (__bf16)(b - i); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NV; i++) x[i & 7] = __builtin_fmaf(x[i & 7], a, b);
#pragma unroll
        for (int m = 0; m < NM32; m++) acc32[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc32[m & 1], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NM16; m++) acc16[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc16[m & 3], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
    s += acc32[0][0] + acc32[1][15] + acc16[0][0] + acc16[1][1] + acc16[2][2] + acc16[3][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int NM32, int NM16>
static void run_mix(const char* name, float* out, int waves_per_simd) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    const int blocks = 256 * waves_per_simd;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_mix<NV, NM32, NM16>), dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double trips_per_simd = (double)blocks * 4 * iters / 1024.0;
    printf("%-44s %d waves/SIMD  %.3f ms -> %.1f cycles per trip per SIMD (2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / trips_per_simd);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 4096 * 4 * 4);
    for (int w : {8, 4}) {
        run_mix<32, 0, 0>("32 v_fma", out, w);
        run_mix<32, 1, 0>("32 v_fma + 1 mfma_32x32x16_bf16", out, w);
        run_mix<32, 2, 0>("32 v_fma + 2 mfma_32x32x16_bf16", out, w);
        run_mix<0, 2, 0>("2 mfma_32x32x16_bf16", out, w);
        run_mix<32, 0, 2>("32 v_fma + 2 mfma_16x16x32_bf16", out, w);
        run_mix<32, 0, 4>("32 v_fma + 4 mfma_16x16x32_bf16", out, w);
        run_mix<0, 0, 4>("4 mfma_16x16x32_bf16", out, w);
    }
    return 0;
}
