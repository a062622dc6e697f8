// What v_exp_f32 / v_rcp_f32 do at the edges of the normal range on gfx950, and how much accuracy the biased exponent of
// csrc/p2_mfma.h (EXP_BIAS_SLOT in a spare K slot of the second MFMA) costs.  The forward compositor relies on
//   (1) v_exp_f32(x) == 0 for every x < -126 (denormal results are flushed whatever the FP mode) and a normal float for x >= -126,
//   (2) the biased exponent being as accurate as the unbiased one up to the final rounding at magnitude ~118.
//   hipcc --offload-arch=gfx950 -O3 -I curve_gaussian_amd/csrc -o exp_flush_probe profiles/probes/exp_flush_probe.hip
#include "p2_mfma.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
using namespace cgs;

__global__ void k_edges(const float* x, float* e, float* r, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        e[i] = __builtin_amdgcn_exp2f(x[i]);
        r[i] = __builtin_amdgcn_rcpf(x[i]);
    }
}
__global__ void k_mfma(const float* sp, float X0, float Y0, float* out, int biased) {   // sp: 16 x {cx, cy, A2, B2, C2, cadd}
    const int lane = threadIdx.x;
    const P2Frag pix = p2_pixel_operand(lane, biased != 0);
    const int s = p2_row_splat(lane), h = p2_row_half(lane);
    const float* q = sp + 6 * s;
    const uint32_t slot = (biased && lane < 32) ? EXP_BIAS_SLOT : 0u;
    const P2Frag a = p2_splat_operand(lane, q[0], q[1], q[2], q[3], q[4], q[5] + (biased ? EXP_BIAS_FRAC : 0.f), X0 + 3.5f,
                                      Y0 + 4.f * h + 1.5f, slot);
    const f32x16 d = p2_mfma(a, pix);
    for (int r = 0; r < 16; r++) out[lane * 16 + r] = d[r];
}

static float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
    int bad = 0;
    {   // ---- (1) edges
        const int N = 64;
        float hx[N];
        int n = 0;
        const uint32_t b126 = bits(-126.0f);
        for (int d = -6; d <= 6; d++) hx[n++] = from_bits(b126 + d);   // d > 0: more negative than -126
        hx[n++] = -127.f; hx[n++] = -130.f; hx[n++] = -149.f; hx[n++] = -1000.f; hx[n++] = -1118.f;
        const uint32_t b128 = bits(128.0f);
        for (int d = -3; d <= 1; d++) hx[n++] = from_bits(b128 + d);
        hx[n++] = INFINITY; hx[n++] = 3.0e38f; hx[n++] = 0x1p126f; hx[n++] = 0x1p127f;
        float *dx, *de, *dr, he[N], hr[N];
        hipMalloc(&dx, N * 4); hipMalloc(&de, N * 4); hipMalloc(&dr, N * 4);
        hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_edges, dim3(1), dim3(64), 0, 0, dx, de, dr, n);
        hipMemcpy(he, de, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hr, dr, n * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; i++) {
            printf("x = %-16.9g (0x%08x)  v_exp_f32 = %-14.6g (0x%08x)   v_rcp_f32 = %-14.6g (0x%08x)\n", hx[i], bits(hx[i]), he[i],
                   bits(he[i]), hr[i], bits(hr[i]));
            if (hx[i] < -126.0f && he[i] != 0.0f) { printf("  ^ NOT flushed\n"); bad++; }
            if (hx[i] >= -126.0f && hx[i] < -125.0f && !(he[i] >= 0x1p-126f)) { printf("  ^ not a normal float\n"); bad++; }
        }
    }
    {   // ---- (2) accuracy of the biased exponent against float64 (same scenes as p2_mfma_probe.hip)
        srand(1);
        float h_sp[16 * 6];
        double worst[2] = {0, 0}, worst_thr[2] = {0, 0};
        float *d_sp, *d_out;
        hipMalloc(&d_sp, sizeof(h_sp)); hipMalloc(&d_out, 64 * 16 * 4);
        for (int trial = 0; trial < 400; trial++) {
            const float X0 = 16.f * (rand() % 100), Y0 = 16.f * (rand() % 100) + 8.f;
            for (int s = 0; s < 16; s++) {
                const double sig1 = 0.55 + (rand() % 1000) / 1000.0 * (trial % 4 == 0 ? 0.5 : 12.0), sig2 = 0.55 + (rand() % 1000) / 1000.0 * 8.0;
                const double th = (rand() % 1000) / 1000.0 * 3.14159;
                const double a = cos(th) * cos(th) / (sig1 * sig1) + sin(th) * sin(th) / (sig2 * sig2);
                const double c = sin(th) * sin(th) / (sig1 * sig1) + cos(th) * cos(th) / (sig2 * sig2);
                const double b = sin(th) * cos(th) * (1 / (sig1 * sig1) - 1 / (sig2 * sig2));
                const double L2E = 1.4426950408889634;
                h_sp[6 * s + 0] = X0 + 3.5f + ((rand() % 2000) / 1000.f - 1.f) * (float)(3.2 * sig1 + 5);
                h_sp[6 * s + 1] = Y0 + 3.5f + ((rand() % 2000) / 1000.f - 1.f) * (float)(3.2 * sig2 + 5);
                h_sp[6 * s + 2] = (float)(-0.5 * L2E * a); h_sp[6 * s + 3] = (float)(-L2E * b); h_sp[6 * s + 4] = (float)(-0.5 * L2E * c);
                h_sp[6 * s + 5] = log2f(0.05f + (rand() % 1000) / 1000.f * 0.9f);
            }
            hipMemcpy(d_sp, h_sp, sizeof(h_sp), hipMemcpyHostToDevice);
            for (int biased = 0; biased < 2; biased++) {
                hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, d_sp, X0, Y0, d_out, biased);
                float h_out[64 * 16];
                hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
                for (int l = 0; l < 64; l++) for (int s = 0; s < 16; s++) {
                    const float* q = h_sp + 6 * s;
                    const double px = X0 + (l & 7), py = Y0 + (l >> 3);
                    const double dx = (double)q[0] - px, dy = (double)q[1] - py;
                    const double ref = (double)q[2] * dx * dx + (double)q[3] * dx * dy + (double)q[4] * dy * dy + (double)q[5];
                    const double got = (double)h_out[l * 16 + s] - (biased ? (log2(255.0) - 126.0) : 0.0);
                    if (ref > -14.0) worst[biased] = fmax(worst[biased], fabs(got - ref));
                    if (ref > -8.5 && ref < -7.5) worst_thr[biased] = fmax(worst_thr[biased], fabs(got - ref));
                }
            }
        }
        printf("max |p_mfma - p_f64| over exponents > -14:  unbiased %.3e   biased %.3e   (near the 1/255 threshold: %.3e / %.3e)\n",
               worst[0], worst[1], worst_thr[0], worst_thr[1]);
        if (worst[1] > 2.5e-5) bad++;
    }
    printf(bad ? "FAILED (%d)\n" : "OK\n", bad);
    return bad ? 1 : 0;
}
