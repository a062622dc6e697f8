// Shared device/host helpers for libcurvegs (gfx950 / CDNA4 only: wave64, DPP, permlane swaps).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/curvegs.h"

namespace cgs {

constexpr int TILE = 16;            // tile edge in pixels (reference config.h:17-18)
constexpr int TILE_PIX = TILE * TILE;
constexpr int NUM_ALL_MAP = 4;      // reference config.h:16

// Per-splat render record written by preprocess and gathered by the render kernels: one 64-byte line.
//   a = {mean2D.x, mean2D.y, conic.x, conic.y}
//   b = {conic.z, opacity(*AA scale), colour, 1/depth}
//   c = all_map[0..3]
//   d = {depth, radius, 2 ln(255 opacity), -}
struct __attribute__((aligned(64))) SplatRec {
    float4 a, b, c, d;
};

// Packed per-splat gradient accumulators filled by the backward compositor (raw sums; the per-splat linear maps to
// dL/dmean2D, dL/dconic are applied once per splat in k_preprocess_bwd).  One 64-byte line per splat:
//   [0] Sg = sum g, g = opacity G dL/dalpha (dL/dopacity = Sg / opacity)   [1] Sx = sum g dx   [2] Sy = sum g dy
//   [3] Sxx  [4] Sxy  [5] Syy   [6] colour   [7] inv-depth   [8..11] all_map   [12..15] unused
constexpr int ACC_STRIDE = 16;
// The view entry points (cgs_view_forward / cgs_view_backward) only ever fill the six geometric sums: their records are 32 bytes
// (stride 8, fields 6 and 7 stay zero) inside the same buffer -- four splats per 128-byte line instead of two, so more of the
// backward compositor's per-instance atomic requests coalesce (neighbours on a curve share tiles): 107 -> 102 us at cfg3.
constexpr int ACC_STRIDE_VIEW = 8;
constexpr int ACC_COL = 6, ACC_INVD = 7, ACC_MAP = 8;

// The forward compositors tag every tile-list entry they stage with the splat's quadrant mask (which of the tile's four 8x8
// quadrants its alpha >= 1/255 ellipse reaches): bits 28..31; the splat index keeps bits 0..27.  The view entry points always
// do (cgs_view_forward rejects P >= 2^28), the operator API whenever P < 2^28 (k_render_fwd3<.., TAG>).
constexpr uint32_t LIST_TAG_SHIFT = 28u;
constexpr uint32_t LIST_ID_MASK = (1u << LIST_TAG_SHIFT) - 1u;

struct GeomState {            // carved from the geometry buffer
    SplatRec* rec;            // [P]
    float* grad_acc;          // [P][ACC_STRIDE]  zeroed and filled by the backward
    float* rgb;               // [P]   SH-evaluated colour (SH path only)
    uint8_t* clamped;         // [P]
};
constexpr int WORK_WORDS = 64 * 32;   // statistics of the checked forward [4..8), non-unit verdict [8], visible counts per chunk [16..80)
constexpr int TOTAL_PARTS = 256;
constexpr int TOTAL_WORDS = 4 + 2 * TOTAL_PARTS;
struct ImageState {           // carved from the image buffer
    float* final_T;           // [H*W]
    uint32_t* n_contrib;      // [H*W]
    uint2* ranges;            // [tiles]  [start,end) in the sorted list
    uint32_t* tile_count;     // [tiles]
    uint32_t* tile_cursor;    // [tiles]
    uint32_t* work;           // [WORK_WORDS]  cleared with the histogram (api.hip: statistics, verdict and count words)
    uint32_t* total;          // [TOTAL_WORDS]  [0] = R, [1] = longest list (exact path); [4 + 2k], [5 + 2k] = partial
                              //                sum / max of the tile lists with tile % TOTAL_PARTS == k (bucket path)
};
struct BinState {             // carved from the binning buffer
    uint64_t* keys;           // [R]  (depth_bits << 32) | splat_idx, bucketed by tile, unsorted inside a bucket
    uint32_t* point_list;     // [R]  splat idx, sorted by (tile, depth_bits, idx)
};

template <typename T>
static inline void carve(char*& chunk, T*& ptr, size_t count) {
    uintptr_t off = (reinterpret_cast<uintptr_t>(chunk) + 127) & ~uintptr_t(127);
    ptr = reinterpret_cast<T*>(off);
    chunk = reinterpret_cast<char*>(ptr + count);
}
static inline GeomState geom_from_chunk(char*& chunk, size_t P) {
    GeomState g;
    carve(chunk, g.rec, P);
    carve(chunk, g.grad_acc, P * ACC_STRIDE);
    carve(chunk, g.rgb, P);
    carve(chunk, g.clamped, P);
    return g;
}
static inline ImageState image_from_chunk(char*& chunk, size_t npix, size_t tiles) {
    ImageState s;
    carve(chunk, s.final_T, npix);
    carve(chunk, s.n_contrib, npix);
    carve(chunk, s.ranges, tiles);
    carve(chunk, s.tile_count, tiles);
    carve(chunk, s.tile_cursor, tiles);
    carve(chunk, s.work, WORK_WORDS);          // (between the cursors and the status words: inside the span the forward clears)
    carve(chunk, s.total, TOTAL_WORDS + 32);   // + 32 words the library never clears: [TOTAL_WORDS] = sticky overflow count
    return s;
}
static inline BinState bin_from_chunk(char*& chunk, size_t R) {
    BinState b;
    carve(chunk, b.point_list, R);  // first: its address must not depend on the capacity the buffer was sized for
    carve(chunk, b.keys, R);        // (the backward only knows R, the forward may have allocated R + slack)
    return b;
}

// ---------------------------------------------------------------- device math shared by fwd and bwd
// Contraction is OFF in the projection / covariance chain: a*b+c fused or not changes the last bits, the 2D covariance
// inverse amplifies them (~40 ulp in the conic), and the same splat would get different alpha-threshold decisions in the
// general and in the fused kernels (different inlining contexts contract differently).  Separate roundings are also
// what the C oracle (-ffp-contract=off) and the reference's host-side expectations are written against.
#ifdef __HIPCC__
__device__ __forceinline__ float3 xform4x3(const float3 p, const float* __restrict__ m) {
#pragma clang fp contract(off)
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform4x4(const float3 p, const float* __restrict__ m) {
#pragma clang fp contract(off)
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}
// reference auxiliary.h:40-43 (double because of the 1.0 / 0.5 literals)
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

// reference auxiliary.h:45-55
__device__ __forceinline__ void get_rect(float px, float py, int max_radius, int gx, int gy, uint2& rmin, uint2& rmax) {
    rmin.x = (uint32_t)min(gx, max(0, (int)((px - (float)max_radius) / (float)TILE)));
    rmin.y = (uint32_t)min(gy, max(0, (int)((py - (float)max_radius) / (float)TILE)));
    rmax.x = (uint32_t)min(gx, max(0, (int)((px + (float)max_radius + (float)(TILE - 1)) / (float)TILE)));
    rmax.y = (uint32_t)min(gy, max(0, (int)((py + (float)max_radius + (float)(TILE - 1)) / (float)TILE)));
}

// Rotation of the UN-normalised quaternion q = (r,x,y,z), rows R[0..2] (reference forward.cu:127-138: its glm
// matrix is the transpose of this one).
__device__ __forceinline__ void quat_rows(const float4 q, float R[3][3]) {
#pragma clang fp contract(off)
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}
// Sigma = Rq diag(s^2) Rq^T, packed upper triangle (reference forward.cu:118-152)
__device__ __forceinline__ void cov3d_from_scale_rot(const float3 scale, float mod, const float4 q, float cov[6]) {
#pragma clang fp contract(off)
    float R[3][3];
    quat_rows(q, R);
    const float s[3] = {mod * scale.x, mod * scale.y, mod * scale.z};
    float Mm[3][3];  // Mm[k][a] = s_k * R[a][k]
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int a = 0; a < 3; a++) Mm[k][a] = s[k] * R[a][k];
    auto sg = [&](int a, int b) { return Mm[0][a] * Mm[0][b] + Mm[1][a] * Mm[1][b] + Mm[2][a] * Mm[2][b]; };
    cov[0] = sg(0, 0); cov[1] = sg(0, 1); cov[2] = sg(0, 2); cov[3] = sg(1, 1); cov[4] = sg(1, 2); cov[5] = sg(2, 2);
}
// EWA projection terms shared by preprocess fwd (forward.cu:78-113) and bwd (backward.cu:172-205):
// t = clamped view-space mean, Mt = Jm * Rwc (2x3), cov = Mt Sigma Mt^T as (a,b,c).
__device__ __forceinline__ void cov2d_terms(const float3 mean, float fx, float fy, float tan_fovx, float tan_fovy,
                                            const float cov3D[6], const float* __restrict__ vm, float3& t,
                                            float Mt[2][3], float3& cov, float& txtz, float& tytz) {
#pragma clang fp contract(off)
    t = xform4x3(mean, vm);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    txtz = t.x / t.z;
    tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    const float J00 = fx / t.z, J02 = -(fx * t.x) / (t.z * t.z);
    const float J11 = fy / t.z, J12 = -(fy * t.y) / (t.z * t.z);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        Mt[0][j] = vm[j * 4 + 0] * J00 + vm[j * 4 + 2] * J02;
        Mt[1][j] = vm[j * 4 + 1] * J11 + vm[j * 4 + 2] * J12;
    }
    const float V[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}};
    float U[2][3];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) U[i][j] = Mt[i][0] * V[0][j] + Mt[i][1] * V[1][j] + Mt[i][2] * V[2][j];
    cov.x = U[0][0] * Mt[0][0] + U[0][1] * Mt[0][1] + U[0][2] * Mt[0][2];
    cov.y = U[0][0] * Mt[1][0] + U[0][1] * Mt[1][1] + U[0][2] * Mt[1][2];
    cov.z = U[1][0] * Mt[1][0] + U[1][1] * Mt[1][1] + U[1][2] * Mt[1][2];
}

// ---------------------------------------------------------------- wave64 reductions (DPP, no LDS)
// Sum over the 16 lanes of each DPP row; every lane of the row ends up holding the row sum.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}
// Full wave64 sum, valid in every lane (4 DPP steps + 2 cross-row steps through readlane).
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}
// Cooperative rect walk: the 64 lanes of a wave visit the tiles of ONE splat's rect at a time (lane = tile), so the
// atomics of a rect row hit consecutive addresses and coalesce into one L2 request per row (measured ~7x the rate
// of one-splat-per-lane loops, whose 64 lanes scatter over 64 unrelated cache lines).  `has` = this lane's splat
// is visible; rect = its tile rect.  fn(splat_lane, tile_x, tile_y) is called with a wave-uniform splat_lane.
template <typename F>
__device__ __forceinline__ void for_each_rect_tile_coop(bool has, uint2 rmin, uint2 rmax, F fn) {
    uint64_t todo = __ballot(has);
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t x0 = __builtin_amdgcn_readlane(rmin.x, src), y0 = __builtin_amdgcn_readlane(rmin.y, src);
        const uint32_t w = __builtin_amdgcn_readlane(rmax.x, src) - x0, h = __builtin_amdgcn_readlane(rmax.y, src) - y0;
        const uint32_t n = w * h;
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t t = base + (uint32_t)lane;
            if (t < n) {
                const uint32_t ty = t / w, tx = t - ty * w;
                fn(src, x0 + tx, y0 + ty);
            }
        }
    }
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// ---------------------------------------------------------------- exact tile-level culling
// One edge of the box minimisation below, written with explicit fmaf / separate roundings only (contraction off) so
// the count pass (k_preprocess_fwd) and the scatter pass (k_scatter) take bit-identical decisions.
__device__ __forceinline__ float edge_min_det(float P, float B, float Q, float rQ, float e, float lo, float hi) {
#pragma clang fp contract(off)
    // min over d in [lo,hi] of P e^2 + 2 B e d + Q d^2
    const float Be = B * e;
    const float d = fminf(fmaxf(-Be * rQ, lo), hi);
    return __builtin_fmaf(P * e, e, __builtin_fmaf(2.f * Be, d, (Q * d) * d));
}
// true iff the splat (centre c, conic A,B,C, tau2 = 2 ln(255 opacity)) may reach alpha >= 1/255 at some pixel of the
// TILE x TILE tile whose first pixel is (X0, Y0).  Conservative: same slack as the compositor's quadrant masks.
__device__ __forceinline__ bool tile_reach_det(float cx, float cy, float A, float B, float C, float tau2, float X0,
                                               float Y0) {
#pragma clang fp contract(off)
    if (!(tau2 >= 0.f)) return false;  // opacity < 1/255 (or NaN): never blended anywhere
    const float l = X0 - cx, r = l + (float)(TILE - 1), b = Y0 - cy, t = b + (float)(TILE - 1);
    if (l <= 0.f && r >= 0.f && b <= 0.f && t >= 0.f) return true;
    const float rA = __builtin_amdgcn_rcpf(A), rC = __builtin_amdgcn_rcpf(C);
    float m = edge_min_det(A, B, C, rC, l, b, t);
    m = fminf(m, edge_min_det(A, B, C, rC, r, b, t));
    m = fminf(m, edge_min_det(C, B, A, rA, b, l, r));
    m = fminf(m, edge_min_det(C, B, A, rA, t, l, r));
    const float X = fmaxf(fabsf(l), fabsf(r)), Y = fmaxf(fabsf(b), fabsf(t));
    const float mag = __builtin_fmaf(A * X, X, __builtin_fmaf(2.f * fabsf(B) * X, Y, (C * Y) * Y));
    const float lim = __builtin_fmaf(tau2, 1.001f, 1e-3f);
    return m <= __builtin_fmaf(8e-6f, mag, lim);
}
// ---------------------------------------------------------------- per-tile rank sort helpers (binning.hip, render.hip)
constexpr uint32_t RANK_MAX = 1024;  // longest list the rank sort handles (bitonic network beyond)
// rank[q] += #{ k in s[0..n) : k < mine[q] }, RANK_U broadcast keys per iteration, the next RANK_U already in flight.
// s is padded with the maximum value up to a multiple of RANK_U.
#ifndef CGS_RANK_U
#define CGS_RANK_U 4
#endif
constexpr uint32_t RANK_U = CGS_RANK_U;
template <int NQ, typename K>
__device__ __forceinline__ void rank_loop(const K* s, uint32_t n, const K (&mine)[4], uint32_t (&rank)[4]) {
    const uint32_t nu = (n + RANK_U - 1) / RANK_U * RANK_U;
    K k[RANK_U], p[RANK_U];
#pragma unroll
    for (uint32_t e = 0; e < RANK_U; e++) k[e] = s[e];
    for (uint32_t u = RANK_U; u < nu; u += RANK_U) {
#pragma unroll
        for (uint32_t e = 0; e < RANK_U; e++) p[e] = s[u + e];  // uniform address: LDS broadcast
#pragma unroll
        for (int q = 0; q < NQ; q++)
#pragma unroll
            for (uint32_t e = 0; e < RANK_U; e++) rank[q] += (uint32_t)(k[e] < mine[q]);
#pragma unroll
        for (uint32_t e = 0; e < RANK_U; e++) k[e] = p[e];
    }
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (uint32_t e = 0; e < RANK_U; e++) rank[q] += (uint32_t)(k[e] < mine[q]);
}
// The same count for ONE key per thread when the keys are positive finite floats (depths): [k < mine] is
// sat((mine - k) * 2^64) -- the scaling is exact, any non-zero difference of two depths (> 0.2) times 2^64 exceeds 1 -- i.e.
// one fma with clamp plus one add per compare (~5 SIMD cycles) instead of v_cmp + add-with-carry (~8: compares and carry
// adds issue at a quarter of the fma rate; the sorting forward spent 11 of its 122 us in them at cfg3).
// s is padded with +inf up to a multiple of 4; four independent partial counts keep the adds off one dependency chain.
__device__ __forceinline__ uint32_t rank_loop_f32(const float* s, uint32_t n, float mine) {
    float big = 0x1p64f, nbig = -0x1p64f;   // depths in (0.2, 1.8e19): no overflow, and ulp(0.2) * 2^64 = 2.7e11 >> 1
    asm volatile("" : "+v"(big), "+v"(nbig));   // in VGPRs: a three-VGPR fma issues faster than one with a literal
    const float mb = mine * big;
    const uint32_t nu = (n + 3u) & ~3u;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
    float4 k = *reinterpret_cast<const float4*>(s);
    for (uint32_t u = 4; u < nu; u += 4) {
        const float4 p = *reinterpret_cast<const float4*>(s + u);   // uniform address: LDS broadcast, in flight over the fmas
        r0 += __builtin_amdgcn_fmed3f(fmaf(k.x, nbig, mb), 0.f, 1.f);
        r1 += __builtin_amdgcn_fmed3f(fmaf(k.y, nbig, mb), 0.f, 1.f);
        r2 += __builtin_amdgcn_fmed3f(fmaf(k.z, nbig, mb), 0.f, 1.f);
        r3 += __builtin_amdgcn_fmed3f(fmaf(k.w, nbig, mb), 0.f, 1.f);
        k = p;
    }
    r0 += __builtin_amdgcn_fmed3f(fmaf(k.x, nbig, mb), 0.f, 1.f);
    r1 += __builtin_amdgcn_fmed3f(fmaf(k.y, nbig, mb), 0.f, 1.f);
    r2 += __builtin_amdgcn_fmed3f(fmaf(k.z, nbig, mb), 0.f, 1.f);
    r3 += __builtin_amdgcn_fmed3f(fmaf(k.w, nbig, mb), 0.f, 1.f);
    return (uint32_t)((r0 + r1) + (r2 + r3));
}
template <typename K>
__device__ __forceinline__ void rank_dispatch(int nq, const K* s, uint32_t n, const K (&mine)[4], uint32_t (&rank)[4]) {
    switch (nq) {  // the compare loop is specialised: no per-key branches inside it
        case 1: rank_loop<1>(s, n, mine, rank); break;
        case 2: rank_loop<2>(s, n, mine, rank); break;
        case 3: rank_loop<3>(s, n, mine, rank); break;
        default: rank_loop<4>(s, n, mine, rank); break;
    }
}

// ---- sort of one tile's (<= RANK_MAX) keys by one 256-thread workgroup, shared by k_tile_rank_sort and the fused
// sort of k_render_fwd<.., SORT>.  Keys are (depth_bits << 32 | splat_idx), unique; on return thread t knows, for each of
// its slots q (element i = t + 256 q < n of the order it ended up holding), the splat index idx[q] and its final position
// rank[q] in the reference's stable (depth, then index) order.
//
// Ranking is done on the 32-bit depth alone (a v_cmp_lt_u32 + add-with-carry per compare; the 64-bit compare of the full
// key is several times slower).  Distinct depths give distinct ranks; if two splats of the tile share a depth, two keys
// collide on a rank -- detected through a claim array in LDS -- and the tile is redone with the full keys.  Lists longer than RANK_PARTITION_MIN are first partitioned into RANK_NB depth buckets (a monotonic map
// of the depth bits, so bucket order is depth order) laid out bucket by bucket in LDS; thread i then takes the i-th key
// of that layout and a wave only ranks against the buckets its own 64 keys fall into -- everything before is smaller,
// everything after larger: O(n^2 / RANK_NB) instead of O(n^2) compares (cfg5, mean list 686: 0.45 -> 0.14 ms).
#ifndef CGS_RANK_PARTITION_MIN
#define CGS_RANK_PARTITION_MIN 192
#endif
constexpr uint32_t RANK_PARTITION_MIN = CGS_RANK_PARTITION_MIN;
#ifndef CGS_RANK_NB
#define CGS_RANK_NB 64
#endif
constexpr uint32_t RANK_NB = CGS_RANK_NB;   // depth buckets (<= 64: their prefix sum is one wave)
struct RankScratch {               // LDS
    uint32_t* sd;                  // [RANK_MAX + RANK_U] depths, then the claim array
    uint32_t* si;                  // [RANK_MAX] splat indices in the order the threads hold them (partition, tie fallback)
    uint32_t* hist;                // [RANK_NB]
    uint32_t* start;               // [RANK_NB + 1]
    uint32_t* mm;                  // [8] per-wave min / max depth
};
// Every thread of the workgroup calls this (waves that hold no key just pass the barriers), n > 0 block-uniform.
// KPT = keys per thread: lists of up to 256 * KPT entries (the scratch arrays are sized accordingly by the caller).
template <int KPT>
__device__ __forceinline__ void tile_rank_sort(const uint64_t* __restrict__ gk, uint32_t n, const RankScratch S,
                                               uint32_t (&rank)[KPT], uint32_t (&idx)[KPT]) {
    const uint32_t tid = threadIdx.x;
    const int lane = (int)(tid & 63u), wave = (int)(tid >> 6);
    const bool has_keys = ((uint32_t)__builtin_amdgcn_readfirstlane((int)tid) & ~63u) < n;   // wave-uniform, in an SGPR
    const int nq = (int)((n + 255u) / 256u);         // keys per thread actually in use (block-uniform)
    (void)nq;
    uint32_t mine_d[KPT];
    uint32_t rbase[KPT], rlen[KPT];   // wave-uniform: the slice of the LDS order each of this wave's key rounds ranks against
#pragma unroll
    for (int q = 0; q < KPT; q++) {
        rbase[q] = 0u;
        rlen[q] = n;
        const uint32_t i = tid + 256u * q;
        const uint64_t k = i < n ? gk[i] : ~0ull;
        mine_d[q] = (uint32_t)(k >> 32);
        idx[q] = (uint32_t)k;
        rank[q] = 0u;
    }
    if (KPT == 4 && n <= RANK_PARTITION_MIN) {   // block-uniform (longer per-thread key sets always partition)
#pragma unroll
        for (int q = 0; q < KPT; q++) {
            const uint32_t i = tid + 256u * q;
            if (i < n) S.sd[i] = mine_d[q];
        }
        if (tid < RANK_U) S.sd[n + tid] = ~0u;       // +inf padding: never "less than" a real key
        __syncthreads();
        if constexpr (KPT == 4) {
            if (has_keys) rank_dispatch(nq, S.sd, n, mine_d, rank);
        }
    } else {
        uint32_t lo = ~0u, hi = 0u;
#pragma unroll
        for (int q = 0; q < KPT; q++)
            if (tid + 256u * q < n) { lo = min(lo, mine_d[q]); hi = max(hi, mine_d[q]); }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, 64));
        }
        if (lane == 0) { S.mm[wave] = lo; S.mm[4 + wave] = hi; }
        if (tid < RANK_NB) S.hist[tid] = 0u;
        __syncthreads();
        uint32_t dmin = S.mm[0], dmax = S.mm[4];
        const int nwaves = (int)min(4u, (n + 63u) / 64u);   // waves that hold keys (the others may have left the kernel)
        for (int w = 1; w < nwaves; w++) { dmin = min(dmin, S.mm[w]); dmax = max(dmax, S.mm[4 + w]); }
        const float scale = (float)RANK_NB / ((float)(dmax - dmin) + 1.0f);
        auto bucket_of = [&](uint32_t d) {   // conversion, product and truncation are all monotonic in d
            return min(RANK_NB - 1u, (uint32_t)((float)(d - dmin) * scale));
        };
        uint32_t bq[KPT], pq[KPT];
#pragma unroll
        for (int q = 0; q < KPT; q++)
            if (tid + 256u * q < n) { bq[q] = bucket_of(mine_d[q]); pq[q] = atomicAdd(&S.hist[bq[q]], 1u); }
        __syncthreads();
        if (tid < 64) {              // exclusive prefix of the bucket counts
            uint32_t v = tid < RANK_NB ? S.hist[tid] : 0u;
#pragma unroll
            for (int off = 1; off < (int)RANK_NB; off <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)v, off, 64);
                if (lane >= off) v += t;
            }
            if (tid < RANK_NB) S.start[tid + 1] = v;
            if (tid == 0) S.start[0] = 0u;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < KPT; q++)
            if (tid + 256u * q < n) {
                const uint32_t slot = S.start[bq[q]] + pq[q];
                S.sd[slot] = mine_d[q];
                S.si[slot] = idx[q];
            }
        if (tid < RANK_U) S.sd[n + tid] = 0x7f800000u;   // +inf as a float, and above every depth as an integer
        __syncthreads();
#pragma unroll
        for (int q = 0; q < KPT; q++) {
            const uint32_t i = tid + 256u * q;
            mine_d[q] = i < n ? S.sd[i] : ~0u;
            idx[q] = i < n ? S.si[i] : ~0u;
        }
#pragma unroll
        for (int q = 0; q < KPT; q++) {
            const uint32_t i0 = ((uint32_t)__builtin_amdgcn_readfirstlane((int)tid) & ~63u) + 256u * q;   // wave-uniform
            if (i0 < n) {
                const uint32_t i1 = min(n - 1u, i0 + 63u);
                const uint32_t b0 = bucket_of(S.sd[i0]), b1 = bucket_of(S.sd[i1]);
                const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(S.start[b0] & ~(RANK_U - 1u)));
                const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.start[b1 + 1]) - base;
                // keys before `base` are all smaller, keys after all larger (the loop may read up to three of the latter)
                rank[q] = base + rank_loop_f32(reinterpret_cast<const float*>(S.sd + base), len, __uint_as_float(mine_d[q]));
                rbase[q] = base;
                rlen[q] = len;
            }
        }
    }
    __syncthreads();                 // everyone is done reading the depths: sd becomes the claim array
#pragma unroll
    for (int q = 0; q < KPT; q++) {
        const uint32_t i = tid + 256u * q;
        if (i < n) S.sd[rank[q]] = i;
    }
    __syncthreads();
    bool lost = false;
#pragma unroll
    for (int q = 0; q < KPT; q++) {
        const uint32_t i = tid + 256u * q;
        if (i < n) lost |= S.sd[rank[q]] != i;
    }
    if (__syncthreads_or(lost)) {    // equal depths in this tile: rank on the full keys (block-uniform branch)
        // Not rare enough to be slow: with ~160 float depths per tile about one tile in 600 has a pair of equal ones
        // (a dozen tiles per cfg3 view), and a workgroup that re-read its keys from global memory one by one sat in the
        // kernel's tail.  The claim array overwrote the depths: restage (depth, index) from registers and rank in LDS.
        uint64_t mine[KPT];
#pragma unroll
        for (int q = 0; q < KPT; q++) {
            const uint32_t i = tid + 256u * q;
            mine[q] = ((uint64_t)mine_d[q] << 32) | idx[q];
            rank[q] = 0u;
            if (i < n) { S.sd[i] = mine_d[q]; S.si[i] = idx[q]; }
        }
        __syncthreads();
        if (has_keys) {              // same slices as the first pass (ties share a bucket), full (depth, index) compare
#pragma unroll
            for (int q = 0; q < KPT; q++) {
                if (((uint32_t)__builtin_amdgcn_readfirstlane((int)tid) & ~63u) + 256u * q >= n) continue;
                const uint32_t u1 = min(n, rbase[q] + rlen[q]);
                uint32_t r = rbase[q];
                for (uint32_t u = rbase[q]; u < u1; u++) {
                    const uint64_t k = ((uint64_t)S.sd[u] << 32) | S.si[u];   // uniform address: LDS broadcast
                    r += (uint32_t)(k < mine[q]);
                }
                rank[q] = r;
            }
        }
        __syncthreads();             // the caller may overwrite sd / si now
    }
}

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
#endif  // __HIPCC__

// ---------------------------------------------------------------- host-side launch bookkeeping
void set_error(const char* fmt, ...);
struct ProfScope {  // brackets one kernel launch with events when profiling is enabled
    ProfScope(const char* name, hipStream_t s);
    ~ProfScope();
    const char* name;
    hipStream_t stream;
    hipEvent_t e0, e1;
    bool on;
};
bool check_launch(const char* what, bool debug, hipStream_t s);

}  // namespace cgs
