#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2_ __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, float b, int iters) {
    __shared__ float4 sm[256]; __shared__ float smf[1024];
    if (MODE == 16) sm[threadIdx.x] = make_float4(a, b, a, b);
    if (MODE == 17) for (int q = threadIdx.x; q < 1024; q += 256) smf[q] = a;
    __syncthreads();
    float x[8]; float2_ y[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.001f + i; y[i] = (float2_){x[i], x[i] + 1.f}; }
    float2_ av = {a, a}, bv = {b, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
                if (MODE == 1) y[i] = __builtin_elementwise_fma(y[i], av, bv);
                if (MODE == 2) x[i] = __builtin_amdgcn_exp2f(x[i]);
                if (MODE == 3) x[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[i]), 0xB1, 0xF, 0xF, true));
                if (MODE == 4) x[i] = (x[i] > a) ? b : x[i];
                if (MODE == 5) { auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x[i]), __builtin_bit_cast(unsigned, x[(i + 1) & 7]), false, false); x[i] = __builtin_bit_cast(float, r[0]); x[(i + 1) & 7] = __builtin_bit_cast(float, r[1]); }
                if (MODE == 6) x[i] = __builtin_amdgcn_rcpf(x[i]);
                if (MODE == 7) x[i] = x[i] * a;
                if (MODE == 8) x[i] = __builtin_fminf(x[i], a);
                if (MODE == 9) x[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((threadIdx.x * 4 + 4) & 255, __builtin_bit_cast(int, x[i])));
                if (MODE == 10) { auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x[i]), __builtin_bit_cast(unsigned, x[(i + 1) & 7]), false, false); x[i] = __builtin_bit_cast(float, r[0]); x[(i + 1) & 7] = __builtin_bit_cast(float, r[1]); }
                if (MODE == 11) { x[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x[i]), 3)) + x[i]; }
                if (MODE == 13) x[i] = __builtin_amdgcn_fmed3f(__builtin_fmaf(x[i], a, b), 0.f, 1.f);
                if (MODE == 14) { unsigned u = __builtin_bit_cast(unsigned, x[i]), v = (unsigned)it, w = __builtin_bit_cast(unsigned, b); x[i] = __builtin_bit_cast(float, max(min(u, v), min(max(u, v), w))); }
                if (MODE == 15) x[i] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, x[i]), __builtin_bit_cast(int, a) + it));
                if (MODE == 16) { float4 v = sm[(it * 8 + i) & 255]; x[i] += v.x + v.w; }
                if (MODE == 17) { float v = smf[((it * 8 + i) * 64 + threadIdx.x) & 1023]; x[i] += v; }
                if (MODE == 12) { x[i] = (__builtin_bit_cast(int, x[i]) > it) ? x[i] : b; }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000; const int blocks = 256 * 8;  // 8 blocks/CU = 32 waves/CU
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_add_f32_dpp", "v_cmp+v_cndmask", "permlane16_swap", "v_rcp_f32", "v_mul_f32", "v_min_f32", "ds_bpermute", "permlane32_swap", "readlane+add", "cmp_int+cndmask", "fma clamp", "med3_u32", "max_i32", "ds_read_b128 uniform(+2add)", "ds_read_b32 (+add)"};
    for (int m = 0; m < 18; m++) for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
        if (m == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
        if (m == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
#define L(M) if (m == M) hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
        L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17)
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double winstr = (double)blocks * 4 * iters * 32;  // wave-instructions
        double per_simd = winstr / 1024.0;
        if (rep) printf("%-16s %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", names[m], ms, ms * 1e-3 * 2.4e9 / per_simd);
    }
    return 0;
}
