// Backward compositor of the unit-colour training instance (what cgs_view_backward runs; reference K8,
// backward.cu:451-675, for colours == 1 and only dL/dcolour flowing in).
//
// With unit colours the walk carries no state from pair to pair (render.hip, UNIT):
//     g(pixel, splat) = alpha_u dL/dalpha = alpha_u K(pixel) / (1 - alpha),    K = (1 - bg) T_final dL/dpixel,
// for every listed splat in front of the pixel's cut with alpha >= 1/255, and the per-splat output is the six moments
// sum_pixels g {1, dx, dy, dx^2, dx dy, dy^2}.  Nothing orders the pairs, so the layout is turned around with respect to the
// forward: LANE = (splat, quadrant) pair, REGISTER = pixel.
//   * pooled list: the (splat, quadrant) pairs of the whole tile -- quadrants the splat's alpha >= 1/255 ellipse reaches
//     and that lie in front of the deepest cut of that quadrant -- form ONE list per batch of 256 staged splats, in (splat,
//     quadrant) order; the four waves take chunks of 32 pairs in turn.  No wave is tied to a quadrant: the chunks are full
//     (one padded chunk per batch instead of one padded group per quadrant and batch) and the waves finish together.
//   * exponents: log2(alpha_u) of 32 pairs x 32 pixels (an 8x4 half of the quadrant) per v_mfma_f32_32x32x16_bf16 pair,
//     D[pixel row][pair column]: the pixel monomials about the quadrant centre are the A operand (loop invariant), the
//     pair's six coefficients, split three-way into bf16 like p2_mfma.h, the B operand -- ONE coefficient set per pair
//     (the pixel-major kernels build one per pair and half quadrant).
//   * walk: exp2, clamp, alpha test, rcp, two multiplies -- per register; the pixel's K (and cut) come from LDS, four
//     pixels per read.
//   * moments: every lane owns four rows of eight pixels of ITS pair, so the six sums are plain register arithmetic
//     (suffix sums: adds only) -- no slot buffer, no LDS transposition, no barriers inside the walk.
//   * leaving the workgroup: the two lane halves of a pair (different pixel rows) are added with v_permlane32_swap, the
//     up-to-four consecutive lanes of one splat with two DPP steps, and the last lane of each run hands the instance's
//     sums to (entry, field) lanes through a small per-wave LDS array: ONE 6-float atomic request per (tile, splat)
//     instance and chunk, as in k_render_bwd3 (the L2 executes ~20 requests per ns whatever their width).
#include "kernels.h"
#include "composite.h"
#include "p2_mfma.h"

namespace cgs {

// What-if builds (profiles/probes/kernel_times.py, profiles/r06_experiments.md): 1 no pixel rows, 2 no output phase (shift to the
// centre, swaps, run sums, atomics), 4 no chunks at all (prologue, staging and the pair list stay).  0 in every product build.
#ifndef CGS_UB_WHATIF
#define CGS_UB_WHATIF 0
#endif
#ifndef CGS_UBWD_WAVES
#define CGS_UBWD_WAVES 6
#endif
constexpr int UB = 256;                 // splats staged per batch (one per thread)
constexpr int CH = 32;                  // pairs per chunk = columns of one MFMA
constexpr int KROW = 36;                // floats per (quadrant, lane half) row of the per-pixel tables: 32 + 4 of pad, so that
                                        // the eight rows a ds_read_b128 can touch at once sit in disjoint banks
// list entry (u16): staged index J [0..8] (UB = the padding entry), quadrant [9..10], index of the pair inside its
// splat's run [11..12], run length - 1 [13..14], [15]: some pixel of the quadrant may have its cut in front of this entry
constexpr uint32_t ENT_PAD = (uint32_t)UB;

__device__ __forceinline__ float dpp_wave_shr1(float v) {   // lane l <- lane l - 1 (lane 0 <- 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float swap32_sum(float v) {      // v(lane) + v(lane ^ 32)
    const unsigned x = __float_as_uint(v);
    auto s = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}
// lanes 0..31: lo(lane) + lo(lane + 32);  lanes 32..63: hi(lane - 32) + hi(lane)
__device__ __forceinline__ float swap32_pair_sum(float lo, float hi) {
    auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Fragments of the transposed exponent product (layout of v_mfma_f32_32x32x16_bf16 as in p2_mfma.h).  K slots of the
// first instruction, eight per lane half:   lanes 0..31:  c0 hi mid | cv hi mid lo | B2 hi mid lo   x  1 1 | v v v | uv uv uv
//                                           lanes 32..63: cu hi mid | A2 hi mid lo | C2 hi mid lo   x  u u | uu uu uu | vv vv vv
// and of the second:                        lanes 0..31:  c0 lo, 0 ...  x  1, 0 ...;     lanes 32..63: cu lo, 0 ...  x  u, 0 ...
// (the second instruction's pixel operand does not depend on v: shared by the two halves of the quadrant).
__device__ __forceinline__ P2Frag ub_pack(float t0, float t1, float t2) {
    uint32_t a0, a1, a2, b0, b1, b2, c0, c1, c2;
    split3(t0, a0, a1, a2);
    split3(t1, b0, b1, b2);
    split3(t2, c0, c1, c2);
    P2Frag f;
    f.k0 = u32x4{pack2(a0, a1), pack2(b0, b1), pack2(b2, c0), pack2(c1, c2)};
    f.k1 = u32x4{a2 >> 16, 0u, 0u, 0u};
    return f;
}
// pixel row m of an 8x4 block: x = (m & 3) + 4 ((m >> 3) & 1), y' = ((m >> 2) & 1) + 2 (m >> 4) -- chosen so that
// accumulator register r of lane half h is the pixel x = r & 7, y' = h + 2 (r >> 3): two full rows of eight per lane
__device__ __forceinline__ P2Frag ub_pixel_operand(int lane, int block) {
    const int m = lane & 31;
    const float u = (float)((m & 3) + 4 * ((m >> 3) & 1)) - 3.5f;
    const float v = (float)(4 * block + ((m >> 2) & 1) + 2 * (m >> 4)) - 3.5f;
    const bool hi_half = lane >= 32;
    const float m0 = hi_half ? u : 1.0f, m1 = hi_half ? u * u : v, m2 = hi_half ? v * v : u * v;
    const uint32_t b0 = __float_as_uint(m0), b1 = __float_as_uint(m1), b2 = __float_as_uint(m2);   // exact in bf16
    P2Frag f;
    f.k0 = u32x4{pack2(b0, b0), pack2(b1, b1), pack2(b1, b2), pack2(b2, b2)};
    f.k1 = u32x4{b0 >> 16, 0u, 0u, 0u};
    return f;
}

// Constants of the walk, kept in VGPRs on purpose (opaque to the compiler): a VOP2 with a literal or scalar operand issues
// slower than one on registers (profiles/probes/enc_probe.hip).
struct WalkConsts { float one, floor01, amin, ramin; };
__device__ __forceinline__ WalkConsts walk_consts() {
    WalkConsts k;
    k.one = 1.0f;
    k.floor01 = 1.0f - 0.99f;    // 1 - min(0.99, e) = max(1 - e, 1 - 0.99): the same float as the reference's 1 - alpha at the clamp
    k.amin = ALPHA_MIN;
    k.ramin = 255.0f;
    asm volatile("" : "+v"(k.one), "+v"(k.floor01), "+v"(k.amin), "+v"(k.ramin));
    return k;
}

// s += v in the lanes where the pair is blended at this pixel: alpha >= 1/255 (reference backward.cu:595: alpha < 1/255 ->
// continue; evaluated as !(e < 1/255) like the C oracle on e = the unclamped alpha -- the 0.99 clamp cannot change the
// outcome -- or, LEAN, as !(1/e > 255) on the reciprocal) and, SLOW, the entry lies in front of the pixel's cut.  v_cmpx narrows
// EXEC, the add runs under it, EXEC is restored: one instruction less than compare + select + add (120 -> 111 us).  (The walk
// runs in wave-uniform control flow: EXEC is all ones on entry -- a PRECONDITION of this routine, which restores EXEC to -1
// rather than to a saved copy; every call site sits in loops whose bounds are wave-uniform (chunk counts, row indices).)
template <bool SLOW, bool LEAN>
__device__ __forceinline__ void masked_add(float& s, float v, float x, float thr, uint32_t pos, uint32_t cut) {
#define CGS_UB_CMPX_E "v_cmpx_nlt_f32_e32 vcc, %[x], %[thr]\n\t"
#define CGS_UB_CMPX_R "v_cmpx_ngt_f32_e32 vcc, %[x], %[thr]\n\t"
#define CGS_UB_TAIL "v_add_f32_e32 %[s], %[s], %[v]\n\ts_mov_b64 exec, -1"
    if (SLOW) {
        if (LEAN)
            asm volatile(CGS_UB_CMPX_R "v_cmpx_lt_u32_e32 vcc, %[pos], %[cut]\n\t" CGS_UB_TAIL
                         : [s] "+v"(s) : [x] "v"(x), [thr] "v"(thr), [pos] "v"(pos), [cut] "v"(cut), [v] "v"(v) : "vcc");
        else
            asm volatile(CGS_UB_CMPX_E "v_cmpx_lt_u32_e32 vcc, %[pos], %[cut]\n\t" CGS_UB_TAIL
                         : [s] "+v"(s) : [x] "v"(x), [thr] "v"(thr), [pos] "v"(pos), [cut] "v"(cut), [v] "v"(v) : "vcc");
    } else {
        if (LEAN) asm volatile(CGS_UB_CMPX_R CGS_UB_TAIL : [s] "+v"(s) : [x] "v"(x), [thr] "v"(thr), [v] "v"(v) : "vcc");
        else asm volatile(CGS_UB_CMPX_E CGS_UB_TAIL : [s] "+v"(s) : [x] "v"(x), [thr] "v"(thr), [v] "v"(v) : "vcc");
    }
#undef CGS_UB_CMPX_E
#undef CGS_UB_CMPX_R
#undef CGS_UB_TAIL
}

// One row of eight pixels of the lane's pair (registers r0 .. r0 + 7 of the block's exponents): g per pixel, consumed on
// the spot by the row's three sums  R0 = sum g, R1 = sum x g, R2 = sum x^2 g  (x = 0..7).  The sums are built as running
// suffix sums from x = 7 down -- s += g; u += s; w += u -- which needs adds only and no register per pixel:
//     s = sum_{x>=1} g_x,  u = sum_k s_k = sum x g_x,  w = sum_k u_k = sum x (x + 1) / 2 g_x   =>   R2 = 2 w - u.
// nP holds -log2(alpha_u) (the coefficient sets are built negated).  LEAN -- no splat of the chunk has opacity >= 0.99, so the
// reference's alpha = min(0.99, alpha_u) never clamps: with E = 1 / alpha_u = exp2(nP),
//     g = alpha_u K / (1 - alpha_u) = K / (E - 1) = 1 / (E Kinv - Kinv):   exp2, one fma, rcp  (Kinv = 1 / K from the second
// per-pixel table; 1e30 where K = 0: the quotient is then ~1e-30 K-units, i.e. nothing).  The two transcendentals cost next
// to nothing here (what-if builds without them run as fast); the plain vector instructions are what the kernel pays for.
template <bool SLOW, bool LEAN>
__device__ __forceinline__ void ub_walk_row(const f32x16& nP, int r0, const float* __restrict__ krow, const uint32_t* __restrict__ lrow,
                                            uint32_t pos, const WalkConsts& k, float& R0, float& R1, float& R2) {
    float s = 0.f, u = 0.f, w = 0.f;
#pragma unroll
    for (int rr = 1; rr >= 0; rr--) {
        const float4 K4 = *reinterpret_cast<const float4*>(krow + r0 + 4 * rr);
        uint4 L4 = make_uint4(0u, 0u, 0u, 0u);
        if (SLOW) L4 = *reinterpret_cast<const uint4*>(lrow + r0 + 4 * rr);
        const float Kv[4] = {K4.x, K4.y, K4.z, K4.w};
        const uint32_t Lv[4] = {L4.x, L4.y, L4.z, L4.w};
#pragma unroll
        for (int t = 3; t >= 0; t--) {
            if (LEAN) {   // (krow holds 1 / K here)
                const float E = __builtin_amdgcn_exp2f(nP[r0 + 4 * rr + t]);     // 1 / alpha_u
                const float v = __builtin_amdgcn_rcpf(fmaf(E, Kv[t], -Kv[t]));   // g = K / (1 / alpha_u - 1) = 1 / ((E - 1) / K)
                masked_add<SLOW, true>(s, v, E, k.ramin, pos, Lv[t]);
            } else {
                const float e = __builtin_amdgcn_exp2f(-nP[r0 + 4 * rr + t]);    // alpha_u = opacity G
                const float om = fmaxf(k.one - e, k.floor01);                      // 1 - alpha, alpha = min(0.99, e)  (forward.cu:368)
                const float v = (e * Kv[t]) * __builtin_amdgcn_rcpf(om);           // g = alpha_u K / (1 - alpha)
                masked_add<SLOW, false>(s, v, e, k.amin, pos, Lv[t]);
            }
            if (4 * rr + t > 0) { u += s; w += u; }
        }
    }
    R0 = s;
    R1 = u;
    R2 = (w + w) - u;
}

// STRIDE: floats per accumulator record -- ACC_STRIDE_VIEW on the view path (32-byte records), ACC_STRIDE behind the operator
// API (the 64-byte records k_preprocess_bwd reads).  GATED (operator API): the forward's device-side verdict on the colours
// decides at run time between this kernel and the general training instance launched beside it.
template <int STRIDE, bool GATED>
__global__ void __launch_bounds__(256, CGS_UBWD_WAVES) k_render_bwd_unit(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int grid_x,
    const float* __restrict__ bg_color, const SplatRec* __restrict__ rec, const float* __restrict__ final_Ts,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels, float* __restrict__ grad_acc,
    const uint32_t* __restrict__ nonunit_gate, const float* __restrict__ clamp_raw) {
    if (GATED && *nonunit_gate != 0u) return;   // some visible splat has a colour or all_map[3] other than 1
    __shared__ float4 s_geo[UB + 1];     // {cx, cy, A2, B2}; entry UB: padding (never blended)
    __shared__ float4 s_at[UB + 1];      // {C2, log2 opacity, splat id bits, -}
    __shared__ uint16_t s_list[UB * 4 + CH];
    __shared__ __attribute__((aligned(16))) float s_K[8 * KROW];        // [quadrant][lane half][block][register]
    __shared__ __attribute__((aligned(16))) float s_Kinv[8 * KROW];     // 1 / K (clamp-free walk)
    __shared__ __attribute__((aligned(16))) uint32_t s_last[8 * KROW];  // the same layout: the pixel's cut (list position)
    __shared__ uint32_t s_wcount[4];
    __shared__ uint32_t s_qlast[8];      // [q]: deepest cut of the quadrant, [4 + q]: shallowest
    __shared__ __attribute__((aligned(16))) float s_out[4][CH][8];      // per wave: sums of the chunk's instances

    const TileGeom g = tile_geom(W, H, grid_x);
    const int lane = g.lane;
    const uint2 range = ranges[g.tile];
    const int total = (int)(range.y - range.x);
    if (total == 0) return;
    const float X0 = (float)(g.tx * TILE), Y0 = (float)(g.ty * TILE);

    // ---- per-pixel constants (thread = pixel, as in the forward: wave = quadrant, lane -> x = l & 7, y = l >> 3)
    {
        float K = 0.f;
        uint32_t last = 0u;
        if (g.inside) {
            const float T_final = final_Ts[g.pix_id];
            float dL = dL_dpixels[g.pix_id];
            if (clamp_raw) {   // the caller's image went through clamp(0, 1): its gradient passes where 0 <= raw <= 1
                const float x = clamp_raw[g.pix_id];
                dL = (x >= 0.f && x <= 1.f) ? dL : 0.f;
            }
            K = T_final * dL - T_final * (bg_color[0] * dL);     // (1 - bg) T_final dL/dpixel  (backward.cu:649-652)
            last = backward_cut(n_contrib[g.pix_id], (uint32_t)total);
        }
        const int x = lane & 7, y = lane >> 3;
        const int b = y >> 2, yy = y & 3, hh = yy & 1, r = x + 8 * (yy >> 1);
        const int idx = (g.wave * 2 + hh) * KROW + b * 16 + r;
        s_K[idx] = K;
        s_Kinv[idx] = fabsf(K) > 1e-30f ? __builtin_amdgcn_rcpf(K) : 1e30f;
        s_last[idx] = last;
        uint32_t mx = last, mn = g.inside ? last : 0xffffffffu;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
            mn = min(mn, (uint32_t)__shfl_xor((int)mn, off, 64));
        }
        if (lane == 0) {
            s_qlast[g.wave] = mx;
            s_qlast[4 + g.wave] = mn;
        }
        if (threadIdx.x == 0) {
            s_geo[UB] = make_float4(0.f, 0.f, 0.f, 0.f);
            s_at[UB] = make_float4(0.f, L2_NEVER, 0.f, 0.f);
        }
    }
    __syncthreads();
    uint32_t qmax[4], qmin[4];   // (wave-uniform: kept in scalar registers)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        qmax[q] = __builtin_amdgcn_readfirstlane(s_qlast[q]);
        qmin[q] = __builtin_amdgcn_readfirstlane(s_qlast[4 + q]);
    }
    const int nb = min(total, (int)max(max(qmax[0], qmax[1]), max(qmax[2], qmax[3])));   // nothing behind the deepest cut
    const int rounds = (nb + UB - 1) / UB;

    // loop-invariant matrix-core operands (pixel side)
    const P2Frag pix0 = ub_pixel_operand(lane, 0), pix1 = ub_pixel_operand(lane, 1);
    const int n = lane & 31, hh = lane >> 5;
    const WalkConsts wk = walk_consts();
    float* const outw = &s_out[g.wave][0][0];

    for (int i = 0; i < rounds; i++) {
        if (i > 0) __syncthreads();   // every wave is done with the previous batch's staged data and list
        // ---- staging: thread t gathers list entry i * UB + t (front to back: the order does not matter here)
        const int pos = i * UB + (int)threadIdx.x;
        uint32_t qm = 0, cutm = 0;
        if (pos < nb) {
            const uint32_t tagged = point_list[range.x + pos];   // quadrant mask in the top bits (the forward's, composite.h)
            const uint32_t id = tagged & LIST_ID_MASK;
            const SplatRec* r = rec + id;
            const float4 a = r->a, b = r->b;
            float4 sa, sb;
            stage_splat(a, b, sa, sb);
            s_geo[threadIdx.x] = sa;
            s_at[threadIdx.x] = make_float4(sb.x, __builtin_amdgcn_logf(sb.y), __uint_as_float(id), 0.f);   // v_log_f32 = log2
            qm = tagged >> LIST_TAG_SHIFT;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if ((uint32_t)pos >= qmax[q]) qm &= ~(1u << q);   // behind everything this quadrant blended
                if ((uint32_t)pos >= qmin[q]) cutm |= 1u << q;    // behind the shallowest cut: per-pixel position test needed
            }
        }
        // pooled pair list in (splat, quadrant) order: position = pairs of the lower waves + pairs of the lower lanes
        uint32_t before = 0, wave_pairs = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint64_t bal = ballot64((qm >> q) & 1u);
            before += __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            wave_pairs += (uint32_t)__builtin_popcountll(bal);
        }
        if (lane == 0) s_wcount[g.wave] = wave_pairs;
        __syncthreads();
        uint32_t n_pairs = 0;
        {
            uint32_t base = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const uint32_t c = s_wcount[w];
                base += w < g.wave ? c : 0u;
                n_pairs += c;
            }
            if (qm) {
                const uint32_t cnt = (uint32_t)__builtin_popcount(qm);
                uint32_t p = base + before, k = 0;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if ((qm >> q) & 1u) {
                        s_list[p + k] = (uint16_t)(threadIdx.x | ((uint32_t)q << 9) | (k << 11) | ((cnt - 1u) << 13) |
                                                   (((cutm >> q) & 1u) << 15));
                        k++;
                    }
            }
            if (threadIdx.x < CH) s_list[n_pairs + threadIdx.x] = (uint16_t)ENT_PAD;
        }
        __syncthreads();
        const int nch = (int)((n_pairs + CH - 1) / CH);
        for (int c = g.wave; c < ((CGS_UB_WHATIF & 4) ? 0 : nch); c += 4) {
            // ---- the chunk's 32 pairs: lane (n, hh) builds K half hh of pair n's coefficient set
            const uint32_t ent = s_list[c * CH + n];
            const uint32_t J = ent & 511u, q = (ent >> 9) & 3u, run_k = (ent >> 11) & 3u, run_last = (ent >> 13) & 3u;
            const bool valid = (uint32_t)(c * CH + n) < n_pairs;
            const float4 ge = s_geo[J];
            const float4 at = s_at[J];
            const float qox = X0 + (float)((q & 1u) << 3), qoy = Y0 + (float)((q >> 1) << 3);
            P2Frag bf;
            {
                const float dxc = ge.x - (qox + 3.5f), dyc = ge.y - (qoy + 3.5f);
                const float A2 = ge.z, B2 = ge.w, C2 = at.x;
                const float c0 = dxc * (A2 * dxc + B2 * dyc) + C2 * dyc * dyc + at.y;
                const float cu = -(2.f * A2 * dxc + B2 * dyc);
                const float cv = -(B2 * dxc + 2.f * C2 * dyc);
                bf = ub_pack(hh ? -cu : -c0, hh ? -A2 : -cv, hh ? -C2 : -B2);   // negated: the product is -log2(alpha_u)
            }
            const uint32_t lpos = (uint32_t)(i * UB) + J;                       // the pair's list position
            const bool slow = ballot64((ent >> 15) != 0u) != 0ull;              // some pixel of some pair may be cut
            // clamp-free walk only when NO alpha of the chunk can reach the 0.99 clamp: log2(0.99) = -0.0144996, and the exponent
            // from the matrix cores is within ~2e-5 of the exact one -- the margin (0.0055 in log2 units: opacity < 0.9862)
            // covers that rounding hundreds of times over
            const bool lean = ballot64(at.y >= -0.02f) == 0ull;
            const float* const krow = (lean ? s_Kinv : s_K) + (q * 2u + (uint32_t)hh) * KROW;
            const uint32_t* const lrow = s_last + (q * 2u + (uint32_t)hh) * KROW;
            float N00 = 0.f, X1 = 0.f, X2 = 0.f, Y1 = 0.f, Y2 = 0.f, XY = 0.f;   // moments about (qox, qoy + hh), y in steps of 2
#pragma unroll
            for (int b = 0; b < 2; b++) {
                f32x16 P = {};
                const P2Frag& px = b ? pix1 : pix0;
                P = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, px.k0), __builtin_bit_cast(bf16x8, bf.k0), P, 0, 0, 0);
                P = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pix0.k1), __builtin_bit_cast(bf16x8, bf.k1), P, 0, 0, 0);
                float r0, r1, r2;
                if (CGS_UB_WHATIF & 1) { r0 = P[0]; r1 = P[7]; r2 = P[15]; }
                const int variant = (CGS_UB_WHATIF & 1) ? 4 : (slow ? 2 : 0) + (lean ? 1 : 0);   // wave-uniform
#define CGS_UB_ROW(R)                                                                                        \
    if (variant == 3) ub_walk_row<true, true>(P, R, krow + 16 * b, lrow + 16 * b, lpos, wk, r0, r1, r2);        \
    else if (variant == 2) ub_walk_row<true, false>(P, R, krow + 16 * b, lrow + 16 * b, lpos, wk, r0, r1, r2);  \
    else if (variant == 1) ub_walk_row<false, true>(P, R, krow + 16 * b, lrow + 16 * b, lpos, wk, r0, r1, r2);  \
    else if (variant == 0) ub_walk_row<false, false>(P, R, krow + 16 * b, lrow + 16 * b, lpos, wk, r0, r1, r2)
                CGS_UB_ROW(0);    // row j = 2 b
                N00 += r0; X1 += r1; X2 += r2;
                if (b) { Y1 = fmaf(2.f, r0, Y1); Y2 = fmaf(4.f, r0, Y2); XY = fmaf(2.f, r1, XY); }
                CGS_UB_ROW(8);    // row j = 2 b + 1
#undef CGS_UB_ROW
                N00 += r0; X1 += r1; X2 += r2;
                const float j = (float)(2 * b + 1);
                Y1 = fmaf(j, r0, Y1); Y2 = fmaf(j * j, r0, Y2); XY = fmaf(j, r1, XY);
            }
            if (CGS_UB_WHATIF & 2) {
                if (N00 + X1 + X2 + Y1 + Y2 + XY == 12345.f) grad_acc[lane] = 1.f;
                continue;
            }
            // ---- to the splat centre: pixel = (qox + x, qoy + hh + 2 j), d = centre - pixel (backward.cu:655-672 are
            // linear in these sums; applied once per splat in splat_math.h::splat_backward)
            const float Dx = ge.x - qox, Dy = ge.y - (qoy + (float)hh);
            float Sg = N00;
            float Sx = fmaf(Dx, N00, -X1);
            float Sy = fmaf(Dy, N00, -2.f * Y1);
            float Sxx = fmaf(Dx, Sx - X1, X2);
            float Syy = fmaf(Dy, Sy - 2.f * Y1, 4.f * Y2);
            float Sxy = fmaf(Dx, Sy, fmaf(-Dy, X1, 2.f * XY));
            // the other lane half holds the other four pixel rows of the same pair; from here on lanes 0..31 carry fields 0..2
            // of their pair and lanes 32..63 fields 3..5.  ONE v_permlane32_swap per field pair does both: it exchanges the
            // upper half of its first operand with the lower half of its second, so afterwards the two registers hold, in
            // the lower lanes, both halves' Sg and, in the upper lanes, both halves' Sxx (three swaps and adds instead of six
            // plus three selects).
            float v0 = swap32_pair_sum(Sg, Sxx), v1 = swap32_pair_sum(Sx, Sxy), v2 = swap32_pair_sum(Sy, Syy);
            // ---- the up-to-four pairs of one splat sit in consecutive lanes: add along the run
            const uint32_t k_eff = min(run_k, (uint32_t)n);       // a run that started in the previous chunk restarts here
            {
                const float t0 = dpp_wave_shr1(v0), t1 = dpp_wave_shr1(v1), t2 = dpp_wave_shr1(v2);
                if (k_eff >= 1u) { v0 += t0; v1 += t1; v2 += t2; }
                const float u0 = dpp_wave_shr1(dpp_wave_shr1(v0)), u1 = dpp_wave_shr1(dpp_wave_shr1(v1)), u2 = dpp_wave_shr1(dpp_wave_shr1(v2));
                if (k_eff >= 2u) { v0 += u0; v1 += u1; v2 += u2; }
            }
            const bool leader = valid && (run_k == run_last || n == CH - 1);   // last lane of the run inside this chunk
            const uint32_t lmask = (uint32_t)ballot64(leader);                 // (lanes 0..31; the upper half is identical)
            const uint32_t slot = (uint32_t)__builtin_popcount(lmask & ((1u << n) - 1u));
            const int n_lead = __builtin_popcount(lmask);
            if (leader) {
                float* o = outw + slot * 8u + (hh ? 3 : 0);
                o[0] = v0; o[1] = v1; o[2] = v2;
                if (!hh) outw[slot * 8u + 6u] = at.z;
            }
            wave_lds_fence();
            // lane (entry, field): eight consecutive floats of the splat's 64-byte accumulator record = one L2 request
            for (int e0 = 0; e0 < n_lead; e0 += 8) {
                const int e = e0 + (lane >> 3), f = lane & 7;
                if (e < n_lead && f < 6) {
                    const float v = outw[e * 8 + f];
                    const uint32_t id = __float_as_uint(outw[e * 8 + 6]);
                    if (v != 0.f) atomicAdd(grad_acc + (size_t)id * STRIDE + f, v);
                }
            }
            wave_lds_fence();   // the array is rewritten by the next chunk
        }
    }
}

void launch_render_bwd_unit(hipStream_t s, int tiles, const uint2* ranges, const uint32_t* point_list, int W, int H,
                            int grid_x, const float* bg_color, const SplatRec* rec, const float* final_Ts,
                            const uint32_t* n_contrib, const float* dL_dpixels, float* grad_acc, int acc_stride,
                            const uint32_t* nonunit_gate, const float* clamp_raw) {
    ProfScope p(nonunit_gate ? "render_bwd_unit_gated" : "render_bwd", s);
    if (acc_stride == ACC_STRIDE_VIEW && !nonunit_gate)
        hipLaunchKernelGGL((k_render_bwd_unit<ACC_STRIDE_VIEW, false>), dim3(tiles), dim3(256), 0, s, ranges, point_list, W, H,
                           grid_x, bg_color, rec, final_Ts, n_contrib, dL_dpixels, grad_acc, nullptr, clamp_raw);
    else if (acc_stride == ACC_STRIDE && nonunit_gate)
        hipLaunchKernelGGL((k_render_bwd_unit<ACC_STRIDE, true>), dim3(tiles), dim3(256), 0, s, ranges, point_list, W, H, grid_x,
                           bg_color, rec, final_Ts, n_contrib, dL_dpixels, grad_acc, nonunit_gate, clamp_raw);
    else
        set_error("launch_render_bwd_unit: unsupported (stride %d, gate %p) combination", acc_stride, (const void*)nonunit_gate);
}

}  // namespace cgs
