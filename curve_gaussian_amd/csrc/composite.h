// Helpers shared by the tile compositors (render.hip, render_unit_bwd.hip): tile / quadrant geometry of a 256-thread
// workgroup, the exact splat-vs-quadrant reach test behind the per-quadrant lists, and the staged (log2-scaled) conic.
#pragma once
#include "common.h"

namespace cgs {

constexpr float ALPHA_MIN = 1.0f / 255.0f;
// n_contrib word of the saved image state: bits 0..30 = the reference's n_contrib (1-based list position of the last blended
// splat, forward.cu:353,394,403; the unit-colour forward stores the position in front of the TERMINATING entry instead --
// equally valid, see below), bit 31 = the pixel TERMINATED (T < 1e-4 stopped it, forward.cu:371-376).  The backward's
// position cut (backward.cu:576-578) only matters for a terminated pixel: any other pixel evaluated every entry of its list
// and blended exactly those that pass the alpha test, which the backward repeats on the same exponent bits -- so its cut
// is "the whole list", and a quadrant whose pixels all ran to the end needs no per-pixel position test at all.
constexpr uint32_t NCONTRIB_TERMINATED = 0x80000000u;
__device__ __forceinline__ uint32_t backward_cut(uint32_t word, uint32_t total) {
    return (word & NCONTRIB_TERMINATED) ? (word & ~NCONTRIB_TERMINATED) : total;
}
constexpr float L2_NEVER = -1000.f;   // log2 "opacity" of the padding entry: alpha = exp2(-1000) = 0

struct TileGeom {
    uint32_t tile, tx, ty;
    int wave, lane;
    int px, py;  // this lane's pixel
    bool inside;
    uint32_t pix_id;
};
__device__ __forceinline__ TileGeom tile_geom(int W, int H, int grid_x) {
    TileGeom g;
    g.tile = blockIdx.x;
    g.tx = g.tile % grid_x;
    g.ty = g.tile / grid_x;
    g.wave = threadIdx.x >> 6;
    g.lane = threadIdx.x & 63;
    const int lx = ((g.wave & 1) << 3) | (g.lane & 7);
    const int ly = ((g.wave >> 1) << 3) | (g.lane >> 3);
    g.px = g.tx * TILE + lx;
    g.py = g.ty * TILE + ly;
    g.inside = g.px < W && g.py < H;
    g.pix_id = (uint32_t)(W * g.py + g.px);
    return g;
}

// min over the box dx in [l,r], dy in [b,t] of q(dx,dy) = A dx^2 + 2 B dx dy + C dy^2  (A,C > 0, AC > B^2)
__device__ __forceinline__ float quad_min_box(float A, float B, float C, float rA, float rC, float l, float r, float b,
                                              float t) {
    if (l <= 0.f && r >= 0.f && b <= 0.f && t >= 0.f) return 0.f;
    float m;
    {
        const float d = fminf(fmaxf(-B * l * rC, b), t);
        m = A * l * l + (2.f * B * l + C * d) * d;
    }
    {
        const float d = fminf(fmaxf(-B * r * rC, b), t);
        m = fminf(m, A * r * r + (2.f * B * r + C * d) * d);
    }
    {
        const float d = fminf(fmaxf(-B * b * rA, l), r);
        m = fminf(m, C * b * b + (2.f * B * b + A * d) * d);
    }
    {
        const float d = fminf(fmaxf(-B * t * rA, l), r);
        m = fminf(m, C * t * t + (2.f * B * t + A * d) * d);
    }
    return m;
}

// 4-bit mask: bit q set iff the splat may reach alpha >= 1/255 somewhere in quadrant q of the tile at (X0,Y0).
// power = -0.5 q  and  alpha = op * exp(power) >= 1/255  <=>  q <= 2 ln(255 op) =: tau2 (stored in rec.d.z).
__device__ __forceinline__ uint32_t quadrant_mask(const float4 a, const float4 b, float tau2, float X0, float Y0) {
    if (!(tau2 >= 0.f)) return 0u;  // opacity < 1/255 (or NaN): never blended
    const float A = a.z, B = a.w, C = b.x;
    const float rA = __builtin_amdgcn_rcpf(A), rC = __builtin_amdgcn_rcpf(C);
    // Conservative acceptance: slack = fixed margin + a bound on the float cancellation error of the quadratic form
    // (both here and in the compositor's per-pixel evaluation), which scales with the magnitude of its terms.
    const float lim = tau2 * 1.001f + 1e-3f;
    const float l0 = X0 - a.x, b0 = Y0 - a.y;
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float l = l0 + (float)((q & 1) * 8), bb = b0 + (float)((q >> 1) * 8);
        const float v = quad_min_box(A, B, C, rA, rC, l, l + 7.f, bb, bb + 7.f);
        const float X = fmaxf(fabsf(l), fabsf(l + 7.f)), Y = fmaxf(fabsf(bb), fabsf(bb + 7.f));
        const float mag = A * X * X + 2.f * fabsf(B) * X * Y + C * Y * Y;
        m |= (v <= lim + 8e-6f * mag) ? (1u << q) : 0u;
    }
    return m;
}

__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// Staged per-splat constants (LDS): the conic is pre-scaled so the compositor evaluates
//   log2(G) = dx (A2 dx + B2 dy) + C2 dy dy   with  A2 = -0.5 A log2e, B2 = -B log2e, C2 = -0.5 C log2e
// and G = exp2(.) is a single v_exp_f32.  (power > 0  <=>  log2(G) > 0.)
constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ void stage_splat(const float4 a, const float4 b, float4& sa, float4& sb) {
    sa = make_float4(a.x, a.y, (-0.5f * LOG2E) * a.z, -LOG2E * a.w);
    sb = make_float4((-0.5f * LOG2E) * b.x, b.y, b.z, b.w);
}

}  // namespace cgs
