// Layout probe of v_mfma_f32_4x4x1_16b_f32 on gfx950: 16 blocks of four lanes; lane (b, r) supplies A[b][r] and B[b][r];
// the four result registers of lane (b, j) hold D[b][i][j], i = register index -- i.e. register i of lane 4 b + j is
// sum_k A[b][i] * B[b][j].  Prints the mismatches against that reading (0 = confirmed).
//   hipcc --offload-arch=gfx950 -O2 profiles/probes/mfma4x4_probe.hip -o /tmp/mfma4x4_probe && /tmp/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[64 + l], b[64 + l], acc, 0, 0, 0);
    for (int i = 0; i < 4; i++) d[l * 4 + i] = acc[i];
}
int main() {
    float ha[128], hb[128], hd[256];
    for (int i = 0; i < 128; i++) { ha[i] = 1.f + 0.37f * (float)((i * 7) % 23); hb[i] = 0.5f + 0.11f * (float)((i * 5) % 19); }
    float *a, *b, *d;
    hipMalloc(&a, sizeof ha); hipMalloc(&b, sizeof hb); hipMalloc(&d, sizeof hd);
    hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, sizeof hd, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int i = 0; i < 4; i++) {
            const int blk = l / 4, j = l % 4;
            const float want = ha[blk * 4 + i] * hb[blk * 4 + j] + ha[64 + blk * 4 + i] * hb[64 + blk * 4 + j];
            if (fabsf(hd[l * 4 + i] - want) > 1e-4f * fabsf(want)) bad++;
        }
    printf("mfma_f32_4x4x1 layout: %d mismatches of 256 (register i of lane 4b+j = sum A[b][i] B[b][j])\n", bad);
    return bad != 0;
}
