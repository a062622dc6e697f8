// The compositors' per-(pixel, splat) exponent as a dense contraction on the matrix cores.
//
// log2(alpha) at pixel (x, y) of an 8x8 quadrant is a quadratic polynomial in the pixel coordinates,
//     p(u, v) = c0 + cu u + cv v + A2 u^2 + B2 u v + C2 v^2,
// i.e. a rank-6 product [pixels x 6 monomials] . [6 coefficients x splats].  One v_mfma_f32_32x32x16_bf16 pair evaluates it for
// 16 splats x 64 pixels: the monomials (u = x - 3.5, v = y - 1.5 within a 8x4 half quadrant: half-integers, their products
// <= 12.25) are exact in bf16, every f32 coefficient is split into three bf16 terms (hi + mid + lo = the f32 value to
// 2^-24), products are exact and the accumulation is f32 -- f32-roundoff-class accuracy at the bf16 matrix rate, which runs
// beside the vector ALU instead of on it (profiles/probes/mfma_bf16_probe.hip; the f32-input MFMA forms do NOT:
// profiles/probes/mfma_probe.hip).
//
// Layout (v_mfma_f32_32x32x16_bf16: A[m = l & 31][k = 8 (l >> 5) + i], B[k = 8 (l >> 5) + i][n = l & 31], D: lane l holds
// column n = l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5), r = 0..15):
//   column n      = pixel n of a half quadrant (x = n & 7, y' = n >> 3); lanes 32..63 are the pixels of the lower half
//   row m         = 8 j + 4 h + r'  <->  splat s = 4 j + r' of the group of 16, expanded about the centre of half h
//   => D register s of lane l = p of splat s at pixel l (x = l & 7, y = l >> 3): lane = pixel, register = splat.
//   k slots (two instructions, K = 32, 18 used):
//     lanes  0..31 supply k 0..7 / 16..23:  c0 hi mid lo | cv hi mid lo | B2 hi mid   /  B2 lo, 0 ...   x  1 1 1 | v v v | uv uv / uv
//     lanes 32..63 supply k 8..15 / 24..31: cu hi mid lo | A2 hi mid lo | C2 hi mid   /  C2 lo, 0 ...   x  u u u | uu uu uu | vv vv / vv
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cgs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct P2Frag { u32x4 k0, k1; };   // one operand of the two MFMAs (8 bf16 each)

// three-way bf16 split of an f32 (truncation): x = hi + mid + lo up to 2^-24 |x|; returned as f32 bit patterns whose low
// 16 bits are zero
__device__ __forceinline__ void split3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(hi);
    mid = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mid);
    lo = __float_as_uint(r2) & 0xffff0000u;
}
// two bf16 (given as f32 bit patterns) -> one dword, element 0 in the low half
__device__ __forceinline__ uint32_t pack2(uint32_t e0, uint32_t e1) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }

__device__ __forceinline__ P2Frag p2_pack(float t0, float t1, float t2) {   // [t0 x3 | t1 x3 | t2 x2] / [t2, 0 ...]
    uint32_t a0, a1, a2, b0, b1, b2, c0, c1, c2;
    split3(t0, a0, a1, a2);
    split3(t1, b0, b1, b2);
    split3(t2, c0, c1, c2);
    P2Frag f;
    f.k0 = u32x4{pack2(a0, a1), pack2(a2, b0), pack2(b1, b2), pack2(c0, c1)};
    f.k1 = u32x4{c2 >> 16, 0u, 0u, 0u};
    return f;
}
// The lane's pixel-side operand (loop invariant): lane l supplies column n = l & 31 (x = n & 7, y' = n >> 3).
__device__ __forceinline__ P2Frag p2_pixel_operand(int lane) {
    const float u = (float)(lane & 7) - 3.5f, v = (float)((lane & 31) >> 3) - 1.5f;
    const bool hi_half = lane >= 32;
    const float m0 = hi_half ? u : 1.0f, m1 = hi_half ? u * u : v, m2 = hi_half ? v * v : u * v;
    const uint32_t b0 = __float_as_uint(m0), b1 = __float_as_uint(m1), b2 = __float_as_uint(m2);   // exact in bf16
    P2Frag f;
    f.k0 = u32x4{pack2(b0, b0), pack2(b0, b1), pack2(b1, b1), pack2(b2, b2)};
    f.k1 = u32x4{b2 >> 16, 0u, 0u, 0u};
    return f;
}
// Which splat of the group and which half this lane's A row stands for.
__device__ __forceinline__ int p2_row_splat(int lane) { const int m = lane & 31; return 4 * (m >> 3) + (m & 3); }
__device__ __forceinline__ int p2_row_half(int lane) { return (lane >> 2) & 1; }
// The lane's splat-side operand.  (cx, cy): splat centre; (A2, B2, C2): conic pre-scaled so that the exponent is in log2
// units, p = A2 dx^2 + B2 dx dy + C2 dy^2 with d = centre - pixel; c_add: added to the constant term (log2 opacity);
// (hx, hy): centre of the lane's half quadrant (quadrant origin + (3.5, 4 h + 1.5)).
__device__ __forceinline__ P2Frag p2_splat_operand(int lane, float cx, float cy, float A2, float B2, float C2, float c_add,
                                                   float hx, float hy) {
    const float dxc = cx - hx, dyc = cy - hy;
    const float c0 = dxc * (A2 * dxc + B2 * dyc) + C2 * dyc * dyc + c_add;
    const float cu = -(2.f * A2 * dxc + B2 * dyc);
    const float cv = -(B2 * dxc + 2.f * C2 * dyc);
    const bool hi_half = lane >= 32;
    return p2_pack(hi_half ? cu : c0, hi_half ? A2 : cv, hi_half ? C2 : B2);
}
__device__ __forceinline__ f32x16 p2_mfma(const P2Frag& a, const P2Frag& b) {
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a.k0), __builtin_bit_cast(bf16x8, b.k0), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a.k1), __builtin_bit_cast(bf16x8, b.k1), acc, 0, 0, 0);
    return acc;
}

}  // namespace cgs
