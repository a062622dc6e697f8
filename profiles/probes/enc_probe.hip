// cost (issue cycles per wave64 instruction per SIMD) of specific gfx950 VALU encodings, pinned with inline asm
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, float b, int iters) {
    float x[8], y[8], z[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.001f + i; y[i] = x[i] * 0.5f + a; z[i] = x[i] + b; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#define A(i) if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y[i]), "v"(z[i]));
            REP8(A)
#undef A
#define A(i) if (MODE == 1) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(z[i]));
            REP8(A)
#undef A
#define A(i) if (MODE == 2) asm volatile("v_fma_f32 %0, %1, %2, %0 clamp" : "+v"(x[i]) : "v"(y[i]), "v"(z[i]));
            REP8(A)
#undef A
#define A(i) if (MODE == 3) asm volatile("v_mul_f32_e64 %0, %0, %1 clamp" : "+v"(x[i]) : "v"(y[i]));
            REP8(A)
#undef A
#define A(i) if (MODE == 4) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
            REP8(A)
#undef A
#define A(i) if (MODE == 5) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(x[i]), "v"(y[i]) : "vcc");
            REP8(A)
#undef A
#define A(i) if (MODE == 6) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y[i]) : );
            REP8(A)
#undef A
#define A(i) if (MODE == 7) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(z[i]));
            REP8(A)
#undef A
#define A(i) if (MODE == 8) asm volatile("v_fma_f32 %0, %1, %2, 1.0" : "+v"(x[i]) : "v"(y[i]), "s"(a));
            REP8(A)
#undef A
#define A(i) if (MODE == 9) asm volatile("v_max_i32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
            REP8(A)
#undef A
#define A(i) if (MODE == 10) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" :: "v"(x[i]), "v"(y[i]) : "s20", "s21");
            REP8(A)
#undef A
#define A(i) if (MODE == 11) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
            REP8(A)
#undef A
#define A(i) if (MODE == 12) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(x[i]) : "s"(it));
            REP8(A)
#undef A
#define A(i) if (MODE == 13) asm volatile("v_add_f32_e64 %0, %0, %1 clamp" : "+v"(x[i]) : "v"(y[i]));
            REP8(A)
#undef A
#define A(i) if (MODE == 14) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(x[i]) : "v"(y[i]));
            REP8(A)
#undef A
#define A(i) if (MODE == 15) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(x[i]) : "s"(it));
            REP8(A)
#undef A
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000; const int blocks = 256 * 8;
    const char* names[] = {"v_fma_f32 (vop3 3 vgpr)", "v_fmac_f32_e32", "v_fma_f32 clamp", "v_mul_f32_e64 clamp", "v_mul_f32_e32", "v_cmp_lt_f32_e32 (vcc)", "v_cndmask_b32_e32", "v_med3_f32", "v_fma_f32 v,s,1.0", "v_max_i32_e32", "v_cmp_lt_f32_e64 (sgpr)", "v_sub_f32_e32", "v_mov_b32 v,s", "v_add_f32_e64 clamp", "v_fma_f32 (2 distinct vgpr)", "v_writelane_b32"};
    for (int m = 0; m < 16; m++) for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
#define L(M) if (m == M) hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
        L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15)
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double per_simd = (double)blocks * 4 * iters * 32 / 1024.0;
        if (rep) printf("%-28s %.3f ms  -> %.2f cycles/wave-instr/SIMD at 2.4 GHz (%.2f at 2.1)\n", names[m], ms, ms * 1e-3 * 2.4e9 / per_simd, ms * 1e-3 * 2.1e9 / per_simd);
    }
    return 0;
}
