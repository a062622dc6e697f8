// Checks curve_gaussian_amd/csrc/p2_mfma.h on the GPU: the exponent of 16 splats at the 64 pixels of a quadrant from two
// bf16 MFMAs against the direct f32 evaluation the compositors used before and against f64.
//   hipcc --offload-arch=gfx950 -O3 -I curve_gaussian_amd/csrc -o p2_mfma_probe profiles/probes/p2_mfma_probe.hip
#include "p2_mfma.h"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
using namespace cgs;
__global__ void k(const float* sp, float X0, float Y0, float* out) {   // sp: 16 x {cx, cy, A2, B2, C2, cadd}
    const int lane = threadIdx.x;
    const P2Frag pix = p2_pixel_operand(lane);
    const int s = p2_row_splat(lane), h = p2_row_half(lane);
    const float* q = sp + 6 * s;
    const P2Frag a = p2_splat_operand(lane, q[0], q[1], q[2], q[3], q[4], q[5], X0 + 3.5f, Y0 + 4.f * h + 1.5f);
    const f32x16 d = p2_mfma(a, pix);
    for (int r = 0; r < 16; r++) out[lane * 16 + r] = d[r];
}
int main() {
    srand(1);
    float h_sp[16 * 6];
    double worst = 0, worst_direct = 0;
    float *d_sp, *d_out;
    hipMalloc(&d_sp, sizeof(h_sp)); hipMalloc(&d_out, 64 * 16 * 4);
    for (int trial = 0; trial < 200; trial++) {
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
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_sp, X0, Y0, d_out);
        float h_out[64 * 16];
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++) for (int s = 0; s < 16; s++) {
            const float* q = h_sp + 6 * s;
            const double px = X0 + (l & 7), py = Y0 + (l >> 3);
            const double dx = (double)q[0] - px, dy = (double)q[1] - py;
            const double ref = (double)q[2] * dx * dx + (double)q[3] * dx * dy + (double)q[4] * dy * dy + (double)q[5];
            const float fdx = q[0] - (float)px, fdy = q[1] - (float)py;
            const float direct = fdx * (q[2] * fdx + q[3] * fdy) + q[4] * fdy * fdy + q[5];
            if (ref > -14.0) {   // only exponents that can matter (alpha >= 1/255 needs p >= -8)
                worst = fmax(worst, fabs(h_out[l * 16 + s] - ref));
                worst_direct = fmax(worst_direct, fabs(direct - ref));
            }
        }
    }
    printf("max |p_mfma - p_f64| = %.3e   (direct f32 evaluation: %.3e) over exponents > -14\n", worst, worst_direct);
    return worst < 2e-5 ? 0 : 1;
}
