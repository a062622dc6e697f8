// does scalar / LDS / branch work share issue slots with VALU work on a gfx950 SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, float b, int iters) {
    __shared__ float4 sm[256];
    sm[threadIdx.x] = make_float4(a, b, a, b);
    __syncthreads();
    float x[8], y[8], z[8];
    int zero = 0; asm volatile("" : "+v"(zero));
    int s0 = iters, s1 = 3, s2 = 5, s3 = 7;
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.001f + i; y[i] = x[i] * 0.5f + a; z[i] = x[i] + b; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#define A(i) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
#define S(i) asm volatile("s_add_i32 %0, %0, %1" : "+s"(s1) : "s"(s2) : "scc");
#define B(i) asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc0 1f\n1:" :: "s"(s0) : "scc");
#define W(i) asm volatile("s_waitcnt lgkmcnt(0)");
#define D(i) { f4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(zero)); asm volatile("" :: "v"(v)); }
#define N(i) asm volatile("s_nop 0");
            if (MODE == 0) { REP8(A) }
            if (MODE == 1) { A(0) S(0) A(1) S(0) A(2) S(0) A(3) S(0) A(4) S(0) A(5) S(0) A(6) S(0) A(7) S(0) }   // 8 valu + 8 salu
            if (MODE == 2) { A(0) B(0) A(1) B(0) A(2) B(0) A(3) B(0) A(4) B(0) A(5) B(0) A(6) B(0) A(7) B(0) }   // 8 valu + 8 (cmp+branch not taken)
            if (MODE == 3) { A(0) W(0) A(1) W(0) A(2) W(0) A(3) W(0) A(4) W(0) A(5) W(0) A(6) W(0) A(7) W(0) }   // 8 valu + 8 waitcnt
            if (MODE == 4) { A(0) D(0) A(1) A(2) A(3) D(0) A(4) A(5) A(6) A(7) }   // 8 valu + 2 ds_read_b128
            if (MODE == 5) { REP8(S) }
            if (MODE == 6) { A(0) N(0) A(1) N(0) A(2) N(0) A(3) N(0) A(4) N(0) A(5) N(0) A(6) N(0) A(7) N(0) }
            if (MODE == 7) { A(0) S(0) S(0) A(1) S(0) S(0) A(2) S(0) S(0) A(3) S(0) S(0) A(4) S(0) S(0) A(5) S(0) S(0) A(6) S(0) S(0) A(7) S(0) S(0) }
        }
    }
    float s = s1 + s3;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000; const int blocks = 256 * 8;
    const char* names[] = {"8 v_mul", "8 v_mul + 8 s_add", "8 v_mul + 8 (s_cmp,s_cbranch)", "8 v_mul + 8 s_waitcnt", "8 v_mul + 2 ds_read_b128", "8 s_add", "8 v_mul + 8 s_nop", "8 v_mul + 16 s_add"};
    for (int m = 0; m < 8; m++) for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
#define L(M) if (m == M) hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
        L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7)
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double groups = (double)blocks * 4 * iters * 4 / 1024.0;   // groups of "8 v_mul + extras" per SIMD
        if (rep) printf("%-32s %.3f ms  -> %.1f cycles per group per SIMD at 2.1 GHz\n", names[m], ms, ms * 1e-3 * 2.1e9 / groups);
    }
    return 0;
}
