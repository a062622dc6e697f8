// Tile compositing kernels: forward alpha-composite (reference K6, forward.cu:279-417) and its backward
// (reference K8, backward.cu:451-675).
//
// Mapping (gfx950, wave64): one workgroup = 256 threads = 4 waves per 16x16 tile; wave w owns the 8x8 pixel
// quadrant (w&1, w>>1) (lane l -> x = l&7, y = l>>3).  Per batch of staged splats:
//   1. staging: thread t gathers splat t's 64-byte record into LDS and tests the splat's alpha>=1/255 ellipse
//      against each of the four quadrant rectangles (exact convex-quadratic minimum over a box, conservative
//      margin).  The four per-quadrant 64-bit ballots of each staging wave are stored in LDS.
//   2. lists: every wave compacts the set bits of its own quadrant's ballots (mbcnt) into a private list of LDS
//      offsets, so a splat that cannot touch a quadrant costs that wave nothing; the walk over it is a counted loop.
//   3. exponents: log2(alpha) of 16 listed splats x 64 pixels comes out of two bf16 MFMAs (p2_mfma.h): lane = pixel,
//      accumulator register = splat.  The vector ALU is left with exp2, the alpha tests and the blend.
// Results are identical to visiting every splat: a culled (splat, quadrant) pair has alpha < 1/255 at all 64 pixels.
//
// Backward reduction (k_render_bwd3): the per-pixel g = alpha_u dL/dalpha of eight pairs are parked in a per-wave LDS
// slot buffer, folded into the six moments sum g {1, dx, dy, dx^2, dx dy, dy^2} splat-parallel (lane = (slot, pixel
// row)), and the per-(quadrant, splat) sums are parked at the pair's list position; one combining pass per batch adds
// the up-to-four quadrant sums of an instance and issues ONE 6-float atomic request per (tile, splat) instance (the
// reference issues 12 atomics per pixel pair, backward.cu:613-672).  The unit-colour training instance of the view entry
// points has its own kernel with lane = (splat, quadrant) pair: render_unit_bwd.hip.
#include "kernels.h"
#include "p2_mfma.h"
#include "composite.h"

namespace cgs {

constexpr int BATCH = 256;
// [x > t] as a 0/1 float with one instruction (v_fma_f32 ... clamp): scaling by 2^100 is exact and any non-zero difference
// of two floats of these magnitudes times 2^100 exceeds 1, so the saturated product is exactly the step function.
// [x >= thr] is [x > pred(thr)], pred = the next float below.
// The constants live in VGPRs on purpose (opaque to the compiler): an fma with an SGPR or inline-constant operand
// issues in ~4 cycles, with three VGPRs in ~2.9 (profiles/probes/enc_probe.hip).
__device__ __forceinline__ float sat01(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 1.f); }  // folds into "clamp"
struct StepConsts {
    float big, cA;       // [alpha >= 1/255] = sat(alpha * 2^100 - pred(1/255) * 2^100)
};
__device__ __forceinline__ StepConsts step_consts() {
    StepConsts k;
    k.big = 0x1p100f;
    k.cA = -__uint_as_float(0x3b808080u) * 0x1p100f;   // pred(1/255f = 0x3b808081)
    asm volatile("" : "+v"(k.big), "+v"(k.cA));
    return k;
}
constexpr int SLOTS = 8;          // accepted splats buffered per wave between two splat-parallel moment passes

// ------------------------------------------------------------------------------------------------ forward
// What the SORT variant needs to turn a tile's unsorted bucket into its depth-ordered list (single-pass bucket binning).
struct BucketSort {
    const uint32_t* tile_count;  // [tiles] instances per tile (may exceed cap: overflow)
    const uint64_t* keys;        // [tiles * cap] (depth_bits << 32 | splat_idx), unsorted inside a bucket
    uint32_t* point_list;        // [tiles * cap] out: sorted splat indices (the backward reads them)
    uint2* ranges;               // [tiles] out
    uint32_t* total;             // status words: partial sums / maxima, overflow flag
    uint32_t cap;                // bucket capacity (<= RANK_MAX for this kernel)
};

// render()'s epilogue (gaussian_renderer/__init__.py:138-145) written by the forward itself when the caller asks for it: the
// clamped image beside the raw one (torch.clamp's gradient mask needs the raw value) and the direction map taken from view to
// world space, out_i = sum_k all_map[k] wv[4 i + k] -- the pixel's sums are in registers at that point, so the separate pass
// over the 1600^2 image (82 MB of traffic, one launch) disappears.
struct ViewEpilogue {
    float* color_clamped;   // [H*W] or NULL
    float* dir_out;         // [3,H*W] or NULL
    const float* wv;        // world_view_transform, row-major 4x4 (device memory); read when dir_out != NULL
};

// ------------------------------------------------------------------------------------------------ forward, v3
// Same per-pixel arithmetic after the exponent, but the exponent itself -- log2(alpha) of every (pixel, splat) pair, a
// quadratic polynomial in the pixel coordinates -- comes from the matrix cores (p2_mfma.h): two bf16 MFMAs evaluate it for
// 16 splats x 64 pixels, at f32-class accuracy, beside the vector ALU instead of on it.  Per pair that removes the seven
// VALU instructions of the quadratic form, the opacity multiply and 24 of the 48 bytes every lane used to pull out of LDS
// (the kernel is bound by exactly those two: VALU issue and LDS return traffic).  Each wave compacts the staged entries
// its quadrant accepted into a private list and walks it in groups of 16.
constexpr int GROUP = 16;
// What-if builds (profiles/probes/kernel_times.py, profiles/r06_experiments.md): the forward with ONE cost removed -- wrong
// images on purpose, only the times mean something.  Bits: 1 no pair walk (operands + MFMAs stay), 2 no groups at all (sort,
// staging and lists stay), 4 walk without the per-pair LDS read of the splat's channels, 8 walk without v_exp, 16 walk without the
// termination test, 32 fused rank + stage path without the ranking loop, 64 no per-pixel output stores, 128 staging without the
// quadrant reach test (every mask 15), 256 staging without the record gather, 512 no list building.  0 in every product build.
#ifndef CGS_WHATIF
#define CGS_WHATIF 0
#endif
// 6 waves per SIMD (80 VGPRs, a dozen spills): 136 us at the natural 104 VGPRs / 4 waves, 124 at 5, 122 at 6, 128 at 7
#ifndef CGS_FWD3_WAVES
#define CGS_FWD3_WAVES 6
#endif
// the general (arbitrary colours) instances carry two more accumulators and the last-contributor tracking
#ifndef CGS_FWD3_WAVES_GENERAL
#define CGS_FWD3_WAVES_GENERAL CGS_FWD3_WAVES
#endif
// fused rank + stage path of the sorting forward for tiles whose list fits one batch (0: always the general sort)
#ifndef CGS_FWD_FAST
#define CGS_FWD_FAST 1
#endif
// backward, training configuration: 6 waves per SIMD like the forward (80 VGPRs, 5 spills; 26.2 KB of LDS with the first 104
// list positions of each quadrant parked).  Alone it runs as fast with 5 waves, 95 VGPRs and all 128 positions parked
// (176 us); with three views in flight the matching footprints let forward and backward workgroups of neighbouring views
// share CUs evenly: 567 -> 597 Msplats/s.  (A branch-free walk -- eight pairs' exp2 / rcp / colour reads in flight before
// the sequential recurrence -- was 8 us faster at 5 waves, but needs 32 more live registers: 589 at 6 waves with spills.)
// What-if builds of the general backward (profiles/r06_experiments.md section 8; 0 in product builds): 1 = no flush (the
// splat-parallel moment passes), 2 = no recurrences in the walk (exp2 + tests + the store stay), 4 = no combining pass
#ifndef CGS_BWD3_WHATIF
#define CGS_BWD3_WHATIF 0
#endif
#ifndef CGS_BWD3_WAVES
#define CGS_BWD3_WAVES 6
#endif
#ifndef CGS_BWD3_CAP
#define CGS_BWD3_CAP 104
#endif
constexpr int BWD_BATCH = 128;   // splats staged per round by the backward (its LDS also holds the per-quadrant sums)
// the park-or-atomics decision is taken per group of SLOTS list positions, the combining pass reads every position < CAP
static_assert(CGS_BWD3_CAP % 8 == 0 && CGS_BWD3_CAP <= BWD_BATCH, "CGS_BWD3_CAP: a multiple of SLOTS (8), at most BWD_BATCH");
constexpr uint32_t BWD_PAD_OFF = (BWD_BATCH + 1) * 16;
constexpr uint32_t PAD_OFF = (BATCH + 1) * 16;   // byte offset of the padding entry in the staged arrays

// UNIT: the caller guarantees colour == 1 and all_map[3] == 1 for every splat (the view entry point builds both itself:
// unit features, gaussian_renderer/__init__.py:97,104).  Then sum w c = sum w = 1 - T (w_i = T_i - T_{i+1} telescopes), and
// the two accumulators are not carried through the walk -- two of the six fmas per pair.
// TAG: every list entry this kernel stages is rewritten with the splat's quadrant mask in its top four bits (composite.h,
// LIST_TAG_SHIFT) for a pair-major backward of the same forward; UNIT implies it.  The operator API sets it whenever P < 2^28
// so that the device-side "unit colours" decision (api.hip) can hand the backward to k_render_bwd_unit.
template <bool GEO, bool SORT, bool UNIT = false, bool TAG = UNIT>
__global__ void __launch_bounds__(256, UNIT ? CGS_FWD3_WAVES : CGS_FWD3_WAVES_GENERAL) k_render_fwd3(const uint2* __restrict__ ranges,
                                                     const uint32_t* __restrict__ point_list, int W, int H, int grid_x,
                                                     const SplatRec* __restrict__ rec, float* __restrict__ final_T,
                                                     uint32_t* __restrict__ n_contrib, const float* __restrict__ bg_color,
                                                     float* __restrict__ out_color, float* __restrict__ out_invdepth,
                                                     float* __restrict__ out_all_map, BucketSort bs, ViewEpilogue epi) {
    // UNIT without GEO is the image-only instance: the caller wants neither inverse depth nor all_map (a training iteration
    // reads `render` only, train.py:98-107) -- the walk then carries the transmittance and nothing else.
    constexpr bool IMAGE_ONLY = UNIT && !GEO;
    // staged entry j of the batch lives at index j + 1 (offset 0 = "nothing blended yet"); index BATCH + 1 is the padding
    // entry (never blended, and behind every real one: the list offsets stay sorted)
    __shared__ float4 s_geo[BATCH + 2];   // {cx, cy, A2, B2}
    __shared__ float4 s_at[BATCH + 2];    // {colour, 1/depth, C2, log2 opacity}
    __shared__ float4 s_c[GEO ? BATCH + 2 : 1];
    __shared__ uint64_t s_qmask[4][4];
    __shared__ __attribute__((aligned(16))) uint32_t s_list[4][BATCH + GROUP];   // per wave: 16 * (staged index + 1); dwords: a
                                                                                 // broadcast read hands the walk ready LDS addresses
    __shared__ __attribute__((aligned(16))) uint32_t s_ord[SORT ? RANK_MAX + RANK_U : 1];
    __shared__ uint32_t s_si[SORT ? RANK_MAX : 1];
    __shared__ uint32_t s_hist[SORT ? RANK_NB : 1], s_start[SORT ? RANK_NB + 1 : 1], s_mm[8];
    if (threadIdx.x == 0) {
        s_geo[BATCH + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_at[BATCH + 1] = make_float4(0.f, 0.f, 0.f, L2_NEVER);
        if (GEO) s_c[GEO ? BATCH + 1 : 0] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const TileGeom g = tile_geom(W, H, grid_x);
    const int lane = g.lane;
    const float X0 = (float)(g.tx * TILE), Y0 = (float)(g.ty * TILE);
    uint2 range;
    bool prestaged = false;   // block-uniform: the tile's (single) batch was staged by the fused rank + stage path below
    if (SORT) {
        const uint32_t tid = threadIdx.x;
        const uint32_t base = g.tile * bs.cap;
        // the thread's key is requested before the tile's count is known (slot tid exists whenever tid < cap): the two
        // loads are in flight together instead of one behind the other
        uint64_t key0 = ~0ull;
        if (CGS_FWD_FAST && tid < bs.cap) key0 = bs.keys[base + tid];
        const uint32_t cnt = bs.tile_count[g.tile];
        const uint32_t n = min(cnt, bs.cap);
        range = make_uint2(base, base + n);
        if (tid == 0) {
            bs.ranges[g.tile] = range;
            if (cnt) {
                uint32_t* part = bs.total + 4 + 2 * (g.tile % TOTAL_PARTS);
                atomicAdd(&part[0], n);
                atomicMax(&part[1], cnt);
                if (cnt > bs.cap) {
                    bs.total[2] = 1u;
                    atomicAdd(&bs.total[TOTAL_WORDS], 1u);   // sticky: survives the next forward's clear
                }
            }
        }
        if (CGS_FWD_FAST && n > 0 && n <= (uint32_t)BATCH) {   // block-uniform
            // ---- single-batch tile (9 tiles in 10 at cfg3): rank and stage in one pass.  Thread t owns key t: its splat's
            // record is requested BEFORE the ranking loop (the index is in the key), the loop runs while the gather is in
            // flight, and the thread stages the record straight into slot rank(t) of the batch arrays -- no ordered index
            // array, no second dependent gather, two workgroup barriers instead of eight.  Equal depths (two keys claiming one
            // slot: about one tile in 600) are detected by the returning LDS exchange that publishes the quadrant mask, and
            // the tile is redone by the general path below.
            const bool has = tid < n;
            const uint32_t depth = has ? (uint32_t)(key0 >> 32) : ~0u;
            const uint32_t id = has ? (uint32_t)key0 : 0u;
            const SplatRec* r = rec + id;
            float4 ra = make_float4(0.f, 0.f, 1.f, 0.f), rb = make_float4(1.f, 0.f, 0.f, 0.f), rc = ra;
            float tau2 = -1.f;
            if (has && !(CGS_WHATIF & 256)) {
                ra = r->a;
                rb = r->b;
                if (GEO) rc = r->c;
                tau2 = r->d.z;
            }
            if (has) s_ord[tid] = depth;
            if (tid < 4u) s_ord[n + tid] = 0x7f800000u;   // +inf padding of the broadcast loop
            s_si[tid] = 0u;                           // per list position: 0x100 | quadrant mask once claimed
            __syncthreads();
            uint32_t lost = 0u;
            if (((uint32_t)__builtin_amdgcn_readfirstlane((int)tid) & ~63u) < n) {   // wave-uniform
                // (depths are positive finite floats -- view-space z > 0.2 -- whose order is the order of their bit patterns)
                const uint32_t rk[1] = {(CGS_WHATIF & 32) ? tid : rank_loop_f32(reinterpret_cast<const float*>(s_ord), n, __uint_as_float(depth))};
                if (has) {
                    float4 sa, sb;
                    stage_splat(ra, rb, sa, sb);
                    const uint32_t slot = rk[0] + 1u;
                    s_geo[slot] = sa;
                    s_at[slot] = make_float4(sb.z, sb.w, sb.x, __builtin_amdgcn_logf(sb.y));
                    if (GEO) s_c[GEO ? slot : 0] = UNIT ? make_float4(rc.x, rc.y, rc.z, sb.w) : rc;
                    const uint32_t qm = (CGS_WHATIF & 128) ? 15u : quadrant_mask(ra, rb, tau2, X0, Y0);
                    lost = atomicExch(&s_si[rk[0]], 0x100u | qm);
                    bs.point_list[base + rk[0]] = TAG ? (id | (qm << LIST_TAG_SHIFT)) : id;
                }
            }
            prestaged = !__syncthreads_or((int)lost);
        }
        if (n > 0 && !prestaged) {   // block-uniform
            uint32_t rank[4], idx[4];
            tile_rank_sort<4>(bs.keys + base, n, RankScratch{s_ord, s_si, s_hist, s_start, s_mm}, rank, idx);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t i = tid + 256u * q;
                if (i < n) {
                    s_ord[rank[q]] = idx[q];
                    // UNIT: the entries are written again, tagged with their quadrant masks, when their batch is staged; a
                    // batch that is never staged (every pixel terminated before it) keeps these untagged entries: valid
                    // indices with an empty mask for whoever reads the whole range
                    if (!TAG || rank[q] >= (uint32_t)BATCH) bs.point_list[base + rank[q]] = idx[q];
                }
            }
            __syncthreads();
        }
    } else {
        range = ranges[g.tile];
    }
    const int total = (int)(range.y - range.x);
    const int rounds = (total + BATCH - 1) / BATCH;
    const StepConsts k = step_consts();
    float T_dead = 0.0f;
    float Tw = 1.0f;
    float cA = g.inside ? k.cA : -0x1p126f;
    uint32_t last_contributor = 0;
    float C = 0.f, Dacc = 0.f;
    float A0 = 0.f, A1 = 0.f, A2 = 0.f, A3 = 0.f;
    bool wave_done = ballot64(cA > -0x1p120f) == 0ull;
    // matrix-core operands: this lane's pixel monomials (loop invariant), and what its splat-side row stands for
    const P2Frag pix = p2_pixel_operand(lane);
    const int row_splat = p2_row_splat(lane);
    const float hx = X0 + (float)((g.wave & 1) << 3) + 3.5f;
    const float hy = Y0 + (float)((g.wave >> 1) << 3) + 4.f * (float)p2_row_half(lane) + 1.5f;
    uint32_t* const list = s_list[g.wave];
    const char* const geo_bytes = reinterpret_cast<const char*>(s_geo);
    const char* const at_bytes = reinterpret_cast<const char*>(s_at);
    const char* const c_bytes = reinterpret_cast<const char*>(s_c);

    for (int i = 0; i < rounds; i++) {
        if (!prestaged) {   // block-uniform (prestaged: one round, staged above)
        if (!__syncthreads_or(!wave_done)) break;
        const int progress = i * BATCH + threadIdx.x;
        uint32_t qm = 0;
        if (progress < total) {
            const uint32_t id = SORT ? s_ord[progress] : point_list[range.x + progress];
            const SplatRec* r = rec + id;
            const float4 a = r->a, b = r->b;
            float4 sa, sb;
            stage_splat(a, b, sa, sb);
            s_geo[threadIdx.x + 1] = sa;
            s_at[threadIdx.x + 1] = make_float4(sb.z, sb.w, sb.x, __builtin_amdgcn_logf(sb.y));   // v_log_f32 = log2
            // UNIT: all_map[3] == 1 is not read back, its slot carries 1/depth -- one 16-byte read per pair instead of two reads
            if (GEO) s_c[threadIdx.x + 1] = UNIT ? make_float4(r->c.x, r->c.y, r->c.z, sb.w) : r->c;
            qm = quadrant_mask(a, b, r->d.z, X0, Y0);   // (0 unless opacity >= 1/255: the log is finite for every listed entry)
            // UNIT (view entry points): the list entry carries the quadrant mask in its top four bits for the backward of the
            // same view (LIST_ID_MASK / LIST_TAG_SHIFT, composite.h) -- its staging then needs no reach test.  Entries of
            // batches this workgroup never stages (every pixel terminated before) lie behind every pixel's cut.
            if (TAG) const_cast<uint32_t*>(SORT ? bs.point_list : point_list)[range.x + progress] = id | (qm << LIST_TAG_SHIFT);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint64_t bal = ballot64((qm >> q) & 1u);
            if (lane == 0) s_qmask[q][g.wave] = bal;
        }
        __syncthreads();
        }
        if (wave_done || (CGS_WHATIF & 512)) continue;
        // ---- this wave's list: the staged entries its quadrant accepted, in list order, padded to a multiple of 16
        int n = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint64_t m = prestaged ? ballot64((s_si[SORT ? c * 64 + lane : 0] >> g.wave) & 1u) : uniform64(s_qmask[g.wave][c]);
            if ((m >> lane) & 1ull) {
                const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                list[pos] = (uint32_t)((c * 64 + lane + 1) * 16);
            }
            n += __builtin_popcountll(m);
        }
        if (lane < GROUP) list[n + lane] = PAD_OFF;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t last_off = 0;   // byte offset (16 * (staged index + 1)) of the last splat this pixel blended in this batch
        for (int g0 = 0; g0 < ((CGS_WHATIF & 2) ? 0 : n); g0 += GROUP) {
            // exponents of the group's 16 splats at this wave's 64 pixels.  (Issuing the NEXT group's MFMAs before this
            // group's blends was tried: 16 more live registers, 122 -> 135 us.)
            f32x16 P;
            {
                const uint32_t joff = list[g0 + row_splat];
                const float4 ge = *reinterpret_cast<const float4*>(geo_bytes + joff);
                const float2 cl = *reinterpret_cast<const float2*>(at_bytes + joff + 8);
                P = p2_mfma(p2_splat_operand(lane, ge.x, ge.y, ge.z, ge.w, cl.x, cl.y, hx, hy), pix);
            }
            const int cnt = (CGS_WHATIF & 1) ? 0 : min(GROUP, n - g0);
            if (CGS_WHATIF & 1) Dacc += P[0] + P[15];
            uint4 w4 = make_uint4(0u, 0u, 0u, 0u);   // four list offsets at a time, same in every lane
#pragma unroll
            for (int s = 0; s < GROUP; s += 2) {
                if (s < cnt) {                                                  // wave-uniform
                if ((s & 3) == 0) w4 = *reinterpret_cast<const uint4*>(list + g0 + s);
                const uint32_t j0 = (s & 2) ? w4.z : w4.x, j1 = (s & 2) ? w4.w : w4.y;
                float2 t0 = make_float2(0.f, 0.f), t1 = t0;
                if (!UNIT) {
                    t0 = *reinterpret_cast<const float2*>(at_bytes + j0);
                    t1 = *reinterpret_cast<const float2*>(at_bytes + j1);
                }
                float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
                if (GEO && !(CGS_WHATIF & 4)) {
                    c0 = *reinterpret_cast<const float4*>(c_bytes + j0);
                    c1 = *reinterpret_cast<const float4*>(c_bytes + j1);
                } else if (GEO) {
                    c0 = make_float4(hx, hy, k.big, k.cA);
                    c1 = make_float4(hy, hx, k.cA, k.big);
                }
                // alpha = min(0.99, opacity * G) = min(0.99, exp2(P)); reference: alpha < 1/255 -> skip
                const float al0 = fminf(0.99f, (CGS_WHATIF & 8) ? P[s] : __builtin_amdgcn_exp2f(P[s]));
                const float al1 = fminf(0.99f, (CGS_WHATIF & 8) ? P[s + 1] : __builtin_amdgcn_exp2f(P[s + 1]));
                const float a0 = al0 * sat01(fmaf(al0, k.big, cA));
                const float a1 = al1 * sat01(fmaf(al1, k.big, cA));
                // two splats blended together, one termination test (see k_render_fwd)
                float wa = a0 * Tw;
                float T1 = fmaf(-Tw, a0, Tw);
                float wb = a1 * T1;
                float T2 = fmaf(-T1, a1, T1);
                if (!(CGS_WHATIF & 16) && __builtin_expect(ballot64(T2 < 0.0001f) != 0ull, 0)) {
                    const bool d0 = T1 < 0.0001f;
                    T_dead = d0 ? Tw : T_dead;
                    wa = d0 ? 0.f : wa;
                    T1 = d0 ? 1.0f : T1;
                    const float a1e = d0 ? 0.f : a1;
                    wb = a1e * T1;
                    T2 = fmaf(-T1, a1e, T1);
                    const bool d1 = T2 < 0.0001f;
                    T_dead = d1 ? T1 : T_dead;
                    wb = d1 ? 0.f : wb;
                    T2 = d1 ? 1.0f : T2;
                    cA = (d0 || d1) ? -0x1p126f : cA;
                    if (UNIT) last_off = d0 ? j0 : d1 ? j1 : last_off;   // the splat that terminated the pixel (at most once)
                }
                Tw = T2;
                if (!UNIT) C = fmaf(t0.x, wa, C);
                if (!IMAGE_ONLY) Dacc = fmaf(UNIT ? c0.w : t0.y, wa, Dacc);
                if (GEO) { A0 = fmaf(c0.x, wa, A0); A1 = fmaf(c0.y, wa, A1); A2 = fmaf(c0.z, wa, A2); if (!UNIT) A3 = fmaf(c0.w, wa, A3); }
                if (!UNIT) C = fmaf(t1.x, wb, C);
                if (!IMAGE_ONLY) Dacc = fmaf(UNIT ? c1.w : t1.y, wb, Dacc);
                if (GEO) { A0 = fmaf(c1.x, wb, A0); A1 = fmaf(c1.y, wb, A1); A2 = fmaf(c1.z, wb, A2); if (!UNIT) A3 = fmaf(c1.w, wb, A3); }
                // the offsets grow along the list and w > 0 exactly when a splat was blended (its bit pattern then
                // exceeds any offset): the median of the three keeps the offset of the last blended splat
                // UNIT: the backward's position cut only matters for a pixel that TERMINATED (any other pixel blended every
                // entry that passes the alpha test, and the backward repeats that test on the same exponent bits): the
                // state then holds the position of the terminating splat instead, recorded in the rare branch above --
                // entries between the last blended one and it failed the alpha test anyway.
                if (!UNIT) {
                    last_off = max(min(last_off, j0), min(max(last_off, j0), __float_as_uint(wa)));   // v_med3_u32
                    last_off = max(min(last_off, j1), min(max(last_off, j1), __float_as_uint(wb)));
                }
            }
            }
            if (ballot64(cA > -0x1p120f) == 0ull) {   // every pixel of the quadrant has terminated
                wave_done = true;
                break;
            }
        }
        if (last_off) last_contributor = (uint32_t)(i * BATCH) + (last_off >> 4) - (UNIT ? 1u : 0u);   // 1-based list position (UNIT: of the entry before the terminating one)
    }
    const bool terminated = !(cA > -0x1p120f);
    if (UNIT && !terminated) last_contributor = (uint32_t)total;   // never terminated: no cut
    if ((CGS_WHATIF & 64) ? (g.inside && Tw + Dacc + A0 + A1 + A2 == 12345.f) : g.inside) {
        const size_t HW = (size_t)H * W;
        const float T = terminated ? T_dead : Tw;
        final_T[g.pix_id] = T;
        n_contrib[g.pix_id] = last_contributor | (terminated ? NCONTRIB_TERMINATED : 0u);   // (composite.h)
        if (UNIT) C = A3 = 1.f - T;
        const float Cout = C + T * bg_color[0];
        out_color[g.pix_id] = Cout;
        if (SORT && epi.color_clamped) epi.color_clamped[g.pix_id] = fminf(fmaxf(Cout, 0.f), 1.f);
        if (SORT && GEO && epi.dir_out) {
            const float* wv = epi.wv;
            epi.dir_out[g.pix_id] = A0 * wv[0] + A1 * wv[1] + A2 * wv[2];
            epi.dir_out[HW + g.pix_id] = A0 * wv[4] + A1 * wv[5] + A2 * wv[6];
            epi.dir_out[2 * HW + g.pix_id] = A0 * wv[8] + A1 * wv[9] + A2 * wv[10];
        }
        if (!IMAGE_ONLY) out_invdepth[g.pix_id] = Dacc;
        if (IMAGE_ONLY) {
            // (no other outputs)
        } else if (GEO) {
            out_all_map[g.pix_id] = A0;
            out_all_map[HW + g.pix_id] = A1;
            out_all_map[2 * HW + g.pix_id] = A2;
            out_all_map[3 * HW + g.pix_id] = A3;
        } else {
            out_all_map[g.pix_id] = 0.f;
            out_all_map[HW + g.pix_id] = 0.f;
            out_all_map[2 * HW + g.pix_id] = 0.f;
            out_all_map[3 * HW + g.pix_id] = 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// column-wise sum over the four 16-lane rows, result replicated in every row
__device__ __forceinline__ float rows_sum(float v) {
    const unsigned x = __float_as_uint(v);
    auto s16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);  // [x0 x0 x2 x2], [x1 x1 x3 x3]
    const float t = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
    const unsigned y = __float_as_uint(t);
    auto s32 = __builtin_amdgcn_permlane32_swap(y, y, false, false);  // [lo lo], [hi hi]
    return __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
}

// ------------------------------------------------------------------------------------------------ backward
// Per (pixel, splat) pair the backward needs, with g := alpha_u dL/dalpha (alpha_u = opacity * G, the unclamped alpha), the
// sums  Sg = sum g, Sx = sum g dx, Sy = sum g dy, Sxx = sum g dx dx, Sxy = sum g dx dy, Syy = sum g dy dy  over the pixels,
// from which (reference backward.cu:655-672, linear in the sums; applied once per splat in splat_math.h::splat_backward):
//   dL/dmean2D = -(A Sx + B Sy) W/2, -(C Sy + B Sx) H/2;   dL/dconic = -1/2 (Sxx, Sxy, Syy);   dL/dopacity = Sg / opacity.
//
// Structure (the kernel is bound by vector issue and by the bytes every lane pulls out of LDS, profiles/r02_*):
//   * log2(alpha_u) of 16 splats x 64 pixels comes from two bf16 MFMAs (p2_mfma.h), as in k_render_fwd3: per pair that
//     leaves exp2, the alpha tests and the transmittance / colour-behind recurrences on the vector ALU, and ONE dword (the
//     splat's colour) to read from LDS instead of seven;
//   * each wave compacts the staged entries its quadrant accepted (and that lie before the last splat any of its pixels
//     blended) into a private list and walks it as a counted loop -- no scalar bit walk, no per-pair slot bookkeeping;
//   * phase 1 (pixel-parallel) parks one scalar g per (pixel, splat) in a per-wave LDS slot buffer, eight splats at a time;
//     phase 2 (splat-parallel): lane (slot, pixel row) folds its 8 pixels into the six moments about the row's first pixel
//     and shifts them to the splat centre, the 8 row results of a slot are summed through LDS by the (slot, field) lanes;
//   * leaving the workgroup: in the training instances those lanes PARK the sums at the pair's list position (s_res below)
//     and one combining pass per batch issues a single atomic request per tile instance; the instances with extra sums
//     issue the global f32 atomics directly -- 8 consecutive floats per splat = one L2 request (the reference issues 12
//     atomics per PIXEL pair, backward.cu:613-672);
//   * the "power > 0" skip (backward.cu:583-585) is not evaluated: for a positive-definite conic it can only fire on
//     rounding noise at pixels where G = 1 to 1e-6 (deviations).
// LDS layouts of the per-wave slot buffer, bank-conflict free for the flush (searched by brute force over strides and row
// pads against the ds_read_b128 lane groups of MI355X_MICROARCH.md; the plain 8-floats-per-row layout cost 25 % of the
// kernel's LDS cycles in conflicts):
//   g values:  slot u, pixel (x, row q)  at  u * SSTRIDE + slot_row(q) + x      (rows 8 floats, 4 floats of pad after rows 1, 5)
//   row sums:  slot s, field f, row q    at  s * LSTRIDE + sum_field(f) + q     (4 floats of pad after field 3)
constexpr int SSTRIDE = 80, LSTRIDE = 68;
__device__ __forceinline__ int slot_row(int q) { return 8 * q + 4 * (q >= 2) + 4 * (q >= 6); }
__device__ __forceinline__ int sum_field(int f) { return 8 * f + 4 * (f >= 4); }

// (The unit-colour training instance -- colour == 1, only dL/dcolour flowing in -- has a closed form without recurrences and
// its own pair-major kernel, render_unit_bwd.hip; the operator API reaches it through a device-side verdict on the colours.)
//
// Instances with extra sums (round 5; COLG / INVD / GEO: dL/dcolour, dL/d(1/depth), dL/dall_map of the splats):
//   * ONE recurrence whatever the number of channels.  The reference carries a "colour behind" accumulator per channel
//     (backward.cu:605-640) and adds (c_ch - behind_ch) dL/dchannel_ch over the channels.  That sum is the colour-behind
//     recurrence of the single PROJECTED colour  c^ = sum_ch c_ch dL/dchannel_ch(pixel)  (the recurrence is linear in the
//     colour, the upstream gradient is a per-pixel constant): one 6-term dot product and one accumulator per pair instead of
//     six subtract / fma / fma triples.
//   * every channel's per-splat gradient is  sum_pixels w dL/dchannel_ch  with the SAME blend weight w = alpha T, so the walk
//     parks w next to g (one more LDS store per pair) and the splat-parallel flush folds the row of eight w against the eight
//     upstream gradients of its pixel row, which lane (slot, row) holds in registers for the whole kernel: the extra sums are
//     extra moment rows.  (Before: one multiply, a four-step DPP row reduction and a select per channel and PAIR -- the
//     all-gradient instance took 675 us at cfg3 against 200 for the training instance.)  With the colour as the only channel
//     the parked value is w dL/dpixel itself and the flush only adds.
//   * the sums are parked per (quadrant, list position) and combined like the training instance's: one 8- or 12-float atomic
//     request per (tile, splat) instance instead of one per (quadrant, splat) pair.
// TAGGED (id_mask strips bits): the list entries carry the forward's quadrant masks (composite.h) -- no reach test here.
template <bool GEO, bool INVD, bool COLG>
#ifndef CGS_BWD3_WAVES_GEO
#define CGS_BWD3_WAVES_GEO 3
#endif
#ifndef CGS_BWD3_WAVES_EXTRA
#define CGS_BWD3_WAVES_EXTRA 4
#endif
#ifndef CGS_BWD3_CAP_EXTRA
#define CGS_BWD3_CAP_EXTRA 96
#endif
#ifndef CGS_BWD3_CAP_GEO
#define CGS_BWD3_CAP_GEO 64
#endif
__global__ void __launch_bounds__(256, GEO ? CGS_BWD3_WAVES_GEO : (INVD || COLG) ? CGS_BWD3_WAVES_EXTRA : CGS_BWD3_WAVES) k_render_bwd3(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int grid_x,
    const float* __restrict__ bg_color, const SplatRec* __restrict__ rec, const float* __restrict__ final_Ts,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dout_invdepth, const float* __restrict__ dL_dout_all_map,
    float* __restrict__ grad_acc, int acc_stride, uint32_t id_mask, const uint32_t* __restrict__ nonunit_gate) {
    // nonunit_gate (operator API, training instance): the forward's device-side verdict on the colours; zero = every visible
    // splat has unit colour and all_map[3] == 1, and k_render_bwd_unit -- launched beside this kernel -- does the work
    if (nonunit_gate && *nonunit_gate == 0u) return;
    constexpr bool EXTRA = COLG || INVD || GEO;   // sums beyond the six geometric ones
    constexpr bool MULTI = INVD || GEO;           // more than one channel flows in: projected colour, w parked
    static_assert(!MULTI || COLG, "the instances with depth / all_map gradients also produce the colour gradient");
    constexpr int NF = GEO ? 12 : EXTRA ? 8 : 6;  // fields of the packed per-splat accumulator record that can be non-zero
    constexpr int NX = (COLG ? 1 : 0) + (INVD ? 1 : 0) + (GEO ? 4 : 0);   // channels
    constexpr int BB = BWD_BATCH, NC = BB / 64;
    // list positions per wave whose sums are combined in LDS (see s_res)
    constexpr int CAP = GEO ? CGS_BWD3_CAP_GEO : EXTRA ? CGS_BWD3_CAP_EXTRA : CGS_BWD3_CAP;
    static_assert(CAP % 8 == 0 && CAP <= BB, "CAP: a multiple of SLOTS (8), at most BWD_BATCH");
    // staged entry j of the batch lives at index j + 1; index BB + 1 is the padding entry (alpha = 0)
    __shared__ float4 s_geo[BB + 2];   // {cx, cy, A2, B2}
    __shared__ float4 s_at[BB + 2];    // {colour, 1/depth (INVD) or the splat id's bits, C2, log2 opacity}
    __shared__ float4 s_c[GEO ? BB + 2 : 1];
    __shared__ uint32_t s_id[INVD ? BB + 2 : 1];
    __shared__ uint64_t s_qmask[4][NC];   // [quadrant][64-entry chunk]: staged entries whose ellipse reaches the quadrant
    __shared__ uint64_t s_tmask[4][NC];   // the same, minus the entries behind everything the quadrant blended = its list
    __shared__ __attribute__((aligned(16))) uint32_t s_list[4][BB + GROUP];   // per wave: 16 * (staged index + 1)
    __shared__ __attribute__((aligned(16))) float s_g[4][SLOTS * SSTRIDE];    // per wave: g of 8 slots x 64 pixels; then the row sums
    // per wave: the blend weight w = alpha T of the same 8 x 64 pairs (COLG alone: w dL/dpixel); then the row sums of fields 8..11
    __shared__ __attribute__((aligned(16))) float s_w[EXTRA ? 4 : 1][EXTRA ? SLOTS * SSTRIDE : 1];
    // Per-(quadrant, list position) gradient sums of the batch.  The L2 executes ~20 scattered atomic requests per ns
    // chip-wide, whatever their scope or width up to a line; one request per (quadrant, splat) pair -- 3.8 M per cfg3 view --
    // is a 190 us floor (no-atomics experiment: 231 -> 144 us).  The four quadrant waves therefore park their sums here with
    // plain stores (consecutive list positions = consecutive addresses), and after the batch lane (entry, field) adds the
    // up-to-four quadrant sums of its entry -- list positions recomputed from the quadrant masks -- and issues ONE request per
    // tile instance (1.6 M per view).  Adding in LDS with ds_add_f32 instead costs more than it saves (63 us of this kernel).
    // List positions >= CAP keep the direct one-request-per-pair path.
    __shared__ __attribute__((aligned(16))) float s_res[4][CAP * NF];
    // MULTI: the upstream gradients of the wave's 64 pixels, per channel, row-major (pixel (x, row q) at 8 q + x): the flush
    // reads its row of eight per channel from here (registers instead: 48 more live VGPRs = one wave per SIMD less, and these
    // instances are latency-bound -- 3 waves: 375 us, 2 waves: 447 us for the colour + all_map instance at cfg3)
    __shared__ __attribute__((aligned(16))) float s_d[MULTI ? 4 : 1][MULTI ? NX : 1][64];
    const TileGeom g = tile_geom(W, H, grid_x);
    const int lane = g.lane;
    const float X0 = (float)(g.tx * TILE), Y0 = (float)(g.ty * TILE);
    const uint2 range = ranges[g.tile];
    const int total = (int)(range.y - range.x);
    if (total == 0) return;
    const int rounds = (total + BB - 1) / BB;
    const size_t HW = (size_t)H * W;
    const bool tagged = id_mask != 0xffffffffu;
    if (threadIdx.x == 0) {
        s_geo[BB + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_at[BB + 1] = make_float4(0.f, 0.f, 0.f, L2_NEVER);
        if (GEO) s_c[GEO ? BB + 1 : 0] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (INVD) s_id[BB + 1] = 0u;
    }

    const float T_final = g.inside ? final_Ts[g.pix_id] : 0.f;
    float T = T_final;
    const uint32_t last_contributor = g.inside ? backward_cut(n_contrib[g.pix_id], (uint32_t)total) : 0u;
    // largest list position any pixel of this quadrant can have blended: everything behind it is skipped wave-wide
    uint32_t wave_last = last_contributor;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, off, 64));
    wave_last = __builtin_amdgcn_readfirstlane(wave_last);

    float accum_rec = 0.f;   // colour behind (MULTI: of the projected colour)
    float dL_dpixel = 0.f, dL_invd = 0.f, dm0 = 0.f, dm1 = 0.f, dm2 = 0.f, dm3 = 0.f;
    if (g.inside) {
        dL_dpixel = dL_dpixels[g.pix_id];
        if (INVD) dL_invd = dL_dout_invdepth[g.pix_id];
        if (GEO) {
            dm0 = dL_dout_all_map[g.pix_id];
            dm1 = dL_dout_all_map[HW + g.pix_id];
            dm2 = dL_dout_all_map[2 * HW + g.pix_id];
            dm3 = dL_dout_all_map[3 * HW + g.pix_id];
        }
    }
    const float nTf_bg = -T_final * (bg_color[0] * dL_dpixel);             // backward.cu:649-652
    float Tp = T_final * dL_dpixel;                                        // T dL/dpixel (single-channel configurations)
    uint32_t* const list = s_list[g.wave];
    const StepConsts k = step_consts();
    float* const sg = s_g[g.wave];
    float* const sw = s_w[EXTRA ? g.wave : 0];
    float* const res = s_res[g.wave];
    const char* const geo_bytes = reinterpret_cast<const char*>(s_geo);
    const char* const at_bytes = reinterpret_cast<const char*>(s_at);
    const char* const c_bytes = reinterpret_cast<const char*>(s_c);
    // matrix-core operands (see k_render_fwd3)
    const P2Frag pix = p2_pixel_operand(lane);
    const int row_splat = p2_row_splat(lane);
    const float qx0 = (float)(g.tx * TILE + ((g.wave & 1) << 3));
    const float hx = qx0 + 3.5f;
    const float hy = Y0 + (float)((g.wave >> 1) << 3) + 4.f * (float)p2_row_half(lane) + 1.5f;
    // flush roles: lane (sl, q) folds pixel row q of slot sl; lane (fs, ff) sums field ff of slot fs over the rows
    const int sl = lane & (SLOTS - 1), q = lane >> 3;
    const int fs = lane >> 3, ff = lane & 7;
    const float qyr = (float)(g.ty * TILE + ((g.wave >> 1) << 3) + q);
    const int pix_off = slot_row(lane >> 3) + (lane & 7);   // where this lane's pixel sits inside a slot
    if (MULTI) {
        const float mine[6] = {dL_dpixel, INVD ? dL_invd : dm0, INVD ? dm0 : dm1, INVD ? dm1 : dm2, INVD ? dm2 : dm3, dm3};
#pragma unroll
        for (int c = 0; c < NX; c++) s_d[MULTI ? g.wave : 0][MULTI ? c : 0][lane] = mine[c];   // (lane = 8 y + x)
        // (read by this wave only, after the first batch's __syncthreads)
    }
    const float* const drow = &s_d[MULTI ? g.wave : 0][0][8 * q];

    for (int i = 0; i < rounds; i++) {
        if (i > 0) __syncthreads();  // every wave is done with the previous batch's staged data
        const int progress = i * BB + threadIdx.x;
        uint32_t qm = 0;
        if (threadIdx.x < BB && progress < total) {
            const uint32_t ent = point_list[range.y - progress - 1];  // back to front (backward.cu:554)
            const uint32_t id = ent & id_mask;                        // id_mask strips the forward's tags
            const SplatRec* r = rec + id;
            const float4 a = r->a, b = r->b;
            float4 sa, sb;
            stage_splat(a, b, sa, sb);
            if (INVD) s_id[threadIdx.x + 1] = id;
            s_geo[threadIdx.x + 1] = sa;
            s_at[threadIdx.x + 1] = make_float4(sb.z, INVD ? sb.w : __uint_as_float(id), sb.x, __builtin_amdgcn_logf(sb.y));   // v_log_f32 = log2
            if (GEO) s_c[threadIdx.x + 1] = r->c;
            // (an entry of a batch the forward never staged is untagged = empty mask: it lies behind every pixel's cut)
            qm = tagged ? (ent >> LIST_TAG_SHIFT) : quadrant_mask(a, b, r->d.z, X0, Y0);
        }
        if (g.wave < NC) {   // (wave-uniform)
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const uint64_t bal = ballot64((qm >> qq) & 1u);
                if (lane == 0) s_qmask[qq][g.wave] = bal;
            }
        }
        __syncthreads();
        // ---- this wave's list.  Staged index J (0..BB-1) of batch i sits at 0-based list position
        // total-1-(i*BB+J); it can matter to this wave only if that is < wave_last  <=>  J >= first_J
        const int first_J = total - (int)wave_last - i * BB;
        int n = 0;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            uint64_t m = uniform64(s_qmask[g.wave][c]);
            const int lo = first_J - c * 64;
            if (lo >= 64) m = 0;
            else if (lo > 0) m &= ~((1ull << lo) - 1ull);
            if (lane == 0) s_tmask[g.wave][c] = m;
            if ((m >> lane) & 1ull) {
                const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                list[pos] = (uint32_t)((c * 64 + lane + 1) * 16);
            }
            n += __builtin_popcountll(m);
        }
        if (lane < GROUP) list[n + lane] = BWD_PAD_OFF;
        // lane-private: staged entry J matters to this pixel iff its list position < last_contributor  <=>  J >= first_lane
        // <=>  16 (J + 1) - 16 max(first_lane, 0) >= 16: on these integer-valued floats the saturated difference IS the 0 / 1 step
        const int first_lane = total - (int)last_contributor - i * BB;
        const float jmin_f = (float)(16 * max(first_lane, 0));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int g0 = 0; g0 < n; g0 += GROUP) {
            f32x16 P;   // log2 of the unclamped alpha of the group's 16 splats at this lane's pixel
            {
                const uint32_t joff = list[g0 + row_splat];
                const float4 ge = *reinterpret_cast<const float4*>(geo_bytes + joff);
                const float2 cl = *reinterpret_cast<const float2*>(at_bytes + joff + 8);
                P = p2_mfma(p2_splat_operand(lane, ge.x, ge.y, ge.z, ge.w, cl.x, cl.y, hx, hy), pix);
            }
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int k0 = g0 + half * SLOTS;
                if (k0 < n) {   // wave-uniform
                    const uint4 wa4 = *reinterpret_cast<const uint4*>(list + k0);      // eight offsets, same address in every lane
                    const uint4 wb4 = *reinterpret_cast<const uint4*>(list + k0 + 4);
                    const uint32_t wv[SLOTS] = {wa4.x, wa4.y, wa4.z, wa4.w, wb4.x, wb4.y, wb4.z, wb4.w};
                    // the eight splats' colours: requested together, ahead of the walk (inside it every read would expose an LDS
                    // round trip behind the previous pair's arithmetic)
                    float2 ci[SLOTS];
                    float4 cm[GEO ? SLOTS : 1];
#pragma unroll
                    for (int u = 0; u < SLOTS; u++) {
                        if (MULTI) ci[u] = *reinterpret_cast<const float2*>(at_bytes + wv[u]);   // {colour, 1/depth}
                        else ci[u] = make_float2(*reinterpret_cast<const float*>(at_bytes + wv[u]), 0.f);
                        if (GEO) cm[GEO ? u : 0] = *reinterpret_cast<const float4*>(c_bytes + wv[u]);
                    }
                    // Branch-free walk: the reference's two tests (backward.cu:576-578 position, :595 alpha < 1/255) become 0 / 1
                    // factors on the float pipe -- [e >= 1/255] = sat(e 2^100 - pred(1/255) 2^100) like the forward,
                    // [position in front of the pixel's cut] = sat(16 (J + 1) - 16 jmin) -- and a pair that fails either runs the
                    // recurrences with alpha = 0, which leaves every carried quantity bit for bit as it was (1 / (1 - 0) = 1).
#pragma unroll
                    for (int u = 0; u < SLOTS; u++) {
                        const float e = __builtin_amdgcn_exp2f(P[half * SLOTS + u]);
                        const float m = sat01(fmaf(e, k.big, k.cA)) * sat01((float)wv[u] - jmin_f);
                        const float alpha_u = e * m;
                        const float alpha = fminf(0.99f, alpha_u);
                        if (CGS_BWD3_WHATIF & 2) {
                            sg[u * SSTRIDE + pix_off] = alpha;
                            if (EXTRA) sw[u * SSTRIDE + pix_off] = alpha_u;
                            continue;
                        }
                        // The reference keeps (last_alpha, last_colour) and folds them into the "colour behind" accumulator at
                        // the start of the next step (backward.cu:605,620,631); folding right after use is the same
                        // recurrence -- acc' = acc + alpha (c - acc) -- with one fma per channel.
                        const float rcp_1ma = __builtin_amdgcn_rcpf(1.f - alpha);
                        float dL_dalpha, v_w = 0.f;
                        if (!MULTI) {   // one channel: carry Tp = T dL/dpixel instead of T
                            Tp = Tp * rcp_1ma;
                            const float d_c = ci[u].x - accum_rec;
                            accum_rec = fmaf(alpha, d_c, accum_rec);
                            if (COLG) v_w = alpha * Tp;   // = w dL/dpixel: this pixel's share of dL/dcolour
                            dL_dalpha = fmaf(nTf_bg, rcp_1ma, d_c * Tp);
                        } else {        // projected colour: c^ = sum_ch c_ch dL/dchannel_ch
                            T = T * rcp_1ma;
                            float chat = ci[u].x * dL_dpixel;
                            if (INVD) chat = fmaf(ci[u].y, dL_invd, chat);
                            if (GEO) {
                                const float4 c4 = cm[GEO ? u : 0];
                                chat = fmaf(c4.x, dm0, chat); chat = fmaf(c4.y, dm1, chat);
                                chat = fmaf(c4.z, dm2, chat); chat = fmaf(c4.w, dm3, chat);
                            }
                            const float d_c = chat - accum_rec;
                            accum_rec = fmaf(alpha, d_c, accum_rec);
                            v_w = alpha * T;              // blend weight (dchannel_dcolor, backward.cu:607)
                            dL_dalpha = fmaf(nTf_bg, rcp_1ma, d_c * T);
                        }
                        sg[u * SSTRIDE + pix_off] = alpha_u * dL_dalpha;
                        if (EXTRA) sw[u * SSTRIDE + pix_off] = v_w;
                    }
                    // ---- flush the eight slots
                    if (!(CGS_BWD3_WHATIF & 1)) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    float Sg, Sx, Sy, Sxx, Sxy, Syy;
                    float E[EXTRA ? NX : 1];
                    {
                        const uint32_t joff = list[k0 + sl];
                        const float2 cxy = *reinterpret_cast<const float2*>(geo_bytes + joff);
                        const float dx0 = cxy.x - qx0, dyr = cxy.y - qyr;
                        const float4 g0v = *reinterpret_cast<const float4*>(sg + sl * SSTRIDE + slot_row(q));
                        const float4 g1v = *reinterpret_cast<const float4*>(sg + sl * SSTRIDE + slot_row(q) + 4);
                        // moments of the row's 8 values about its first pixel (weights 0..7 and 0,1,4,..49 are instruction
                        // constants), then shifted to the splat centre: sum g (d-c) = d M0 - M1, sum g (d-c)^2 = d (d M0 - 2 M1) + M2
                        float M0 = g0v.x + g0v.y, M1 = g0v.y, M2 = g0v.y;
                        M0 += g0v.z; M1 = fmaf(g0v.z, 2.f, M1); M2 = fmaf(g0v.z, 4.f, M2);
                        M0 += g0v.w; M1 = fmaf(g0v.w, 3.f, M1); M2 = fmaf(g0v.w, 9.f, M2);
                        M0 += g1v.x; M1 = fmaf(g1v.x, 4.f, M1); M2 = fmaf(g1v.x, 16.f, M2);
                        M0 += g1v.y; M1 = fmaf(g1v.y, 5.f, M1); M2 = fmaf(g1v.y, 25.f, M2);
                        M0 += g1v.z; M1 = fmaf(g1v.z, 6.f, M1); M2 = fmaf(g1v.z, 36.f, M2);
                        M0 += g1v.w; M1 = fmaf(g1v.w, 7.f, M1); M2 = fmaf(g1v.w, 49.f, M2);
                        const float Rx = fmaf(dx0, M0, -M1);
                        const float Rxx = fmaf(dx0, Rx - M1, M2);
                        Sg = M0; Sx = Rx; Sxx = Rxx;
                        Sy = dyr * M0; Sxy = dyr * Rx; Syy = (dyr * dyr) * M0;
                        if (EXTRA) {   // the row's share of the channel gradients: sum_x w dL/dchannel
                            const float4 w0v = *reinterpret_cast<const float4*>(sw + sl * SSTRIDE + slot_row(q));
                            const float4 w1v = *reinterpret_cast<const float4*>(sw + sl * SSTRIDE + slot_row(q) + 4);
                            if (!MULTI) {
                                E[0] = ((w0v.x + w0v.y) + (w0v.z + w0v.w)) + ((w1v.x + w1v.y) + (w1v.z + w1v.w));
                            } else {
#pragma unroll
                                for (int c = 0; c < NX; c++) {
                                    const float4 d0 = *reinterpret_cast<const float4*>(drow + 64 * c);
                                    const float4 d1 = *reinterpret_cast<const float4*>(drow + 64 * c + 4);
                                    float e = w0v.x * d0.x;
                                    e = fmaf(w0v.y, d0.y, e); e = fmaf(w0v.z, d0.z, e); e = fmaf(w0v.w, d0.w, e);
                                    e = fmaf(w1v.x, d1.x, e); e = fmaf(w1v.y, d1.y, e); e = fmaf(w1v.z, d1.z, e);
                                    E[c] = fmaf(w1v.w, d1.w, e);
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();   // every lane has read its slot rows: the buffers become [slot][field][row]
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    {
                        float* rp = sg + sl * LSTRIDE + q;
                        rp[sum_field(0)] = Sg; rp[sum_field(1)] = Sx; rp[sum_field(2)] = Sy;
                        rp[sum_field(3)] = Sxx; rp[sum_field(4)] = Sxy; rp[sum_field(5)] = Syy;
                        if (EXTRA) {   // record fields: 6 colour, 7 inverse depth, 8..11 all_map
                            rp[sum_field(ACC_COL)] = E[0];
                            rp[sum_field(ACC_INVD)] = INVD ? E[1] : 0.f;
                            if (GEO) {
                                float* xp = sw + sl * LSTRIDE + q;
                                constexpr int M0I = INVD ? 2 : 1;
                                xp[sum_field(0)] = E[GEO ? M0I : 0]; xp[sum_field(1)] = E[GEO ? M0I + 1 : 0];
                                xp[sum_field(2)] = E[GEO ? M0I + 2 : 0]; xp[sum_field(3)] = E[GEO ? M0I + 3 : 0];
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    {
                        float v = 0.f, v2 = 0.f;
                        if (ff < (EXTRA ? 8 : 6)) {
                            const float4 r0 = *reinterpret_cast<const float4*>(sg + fs * LSTRIDE + sum_field(ff));
                            const float4 r1 = *reinterpret_cast<const float4*>(sg + fs * LSTRIDE + sum_field(ff) + 4);
                            v = ((r0.x + r0.y) + (r0.z + r0.w)) + ((r1.x + r1.y) + (r1.z + r1.w));
                        }
                        if (GEO && ff < 4) {
                            const float4 r0 = *reinterpret_cast<const float4*>(sw + fs * LSTRIDE + sum_field(ff));
                            const float4 r1 = *reinterpret_cast<const float4*>(sw + fs * LSTRIDE + sum_field(ff) + 4);
                            v2 = ((r0.x + r0.y) + (r0.z + r0.w)) + ((r1.x + r1.y) + (r1.z + r1.w));
                        }
                        if (CAP >= BB || k0 + SLOTS <= CAP) {   // (wave-uniform) park: 8 slots x NF fields, consecutive
                            if (ff < (NF < 8 ? NF : 8)) res[(k0 + fs) * NF + ff] = v;
                            if (GEO && ff < 4) res[(k0 + fs) * NF + 8 + ff] = v2;
                        } else {
                            // the 6-8 atomics of one splat hit 8 consecutive floats of its 64-byte accumulator record and
                            // coalesce into ONE L2 request (measured 7x the rate of one field per instruction)
                            const uint32_t joff = list[k0 + fs];
                            const uint32_t id = INVD ? *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(s_id) + (joff >> 2))
                                                     : __float_as_uint(*reinterpret_cast<const float*>(at_bytes + joff + 4));
                            if (v != 0.f) atomicAdd(grad_acc + (size_t)id * acc_stride + ff, v);
                            if (GEO && v2 != 0.f) atomicAdd(grad_acc + (size_t)id * acc_stride + 8 + ff, v2);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();   // the slot buffers may be overwritten from here on
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                }
            }
        }
        // ---- the batch's per-splat sums leave the workgroup: lane (entry, field) -> consecutive floats of the splat's
        // 64-byte accumulator record = one L2 request per entry
        if (!(CGS_BWD3_WHATIF & 4)) {
            __syncthreads();
            const int nb = min(BB, total - i * BB);
            uint64_t tm[4][NC];
            int base[4][NC];
#pragma unroll
            for (int w = 0; w < 4; w++) {
                int acc = 0;
#pragma unroll
                for (int c = 0; c < NC; c++) { tm[w][c] = s_tmask[w][c]; base[w][c] = acc; acc += __builtin_popcountll(tm[w][c]); }
            }
            constexpr int FL = NF > 8 ? 16 : 8;     // lanes per entry
            constexpr int EPP = 256 / FL;           // entries per pass of the workgroup
            const int f = (int)(threadIdx.x & (FL - 1));
#pragma unroll
            for (int c = 0; c < NC; c++) {
#pragma unroll
                for (int h = 0; h < 64 / EPP; h++) {
                    const int bit = h * EPP + (int)(threadIdx.x / FL), e = c * 64 + bit;
                    if (e < nb && f < NF) {
                        float v = 0.f;
                        bool any = false;
#pragma unroll
                        for (int w = 0; w < 4; w++) {
                            const uint64_t m = tm[w][c];
                            const int pos = base[w][c] + __builtin_popcountll(m & ((1ull << bit) - 1ull));
                            if (((m >> bit) & 1ull) && pos < CAP) { v += s_res[w][pos * NF + f]; any = true; }
                        }
                        if (any && v != 0.f) {
                            const uint32_t id = INVD ? s_id[e + 1] : __float_as_uint(s_at[e + 1].y);
                            atomicAdd(grad_acc + (size_t)id * acc_stride + f, v);
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
#define CGS_FWD3(G, S, U, T, R, PL, BS)                                                                               \
    hipLaunchKernelGGL((k_render_fwd3<G, S, U, T>), dim3(tiles), dim3(256), 0, s, R, PL, W, H, grid_x, rec, final_T,  \
                       n_contrib, bg_color, out_color, out_invdepth, out_all_map, BS, epi)
void launch_render_fwd(hipStream_t s, bool geo, int tiles, const uint2* ranges, const uint32_t* point_list, int W,
                       int H, int grid_x, const SplatRec* rec, float* final_T, uint32_t* n_contrib,
                       const float* bg_color, float* out_color, float* out_invdepth, float* out_all_map, bool unit, bool tag) {
    ProfScope p("render_fwd", s);
    const ViewEpilogue epi{nullptr, nullptr, nullptr};
    if (geo && unit) CGS_FWD3(true, false, true, true, ranges, point_list, BucketSort{});
    else if (unit) CGS_FWD3(false, false, true, true, ranges, point_list, BucketSort{});
    else if (geo && tag) CGS_FWD3(true, false, false, true, ranges, point_list, BucketSort{});
    else if (tag) CGS_FWD3(false, false, false, true, ranges, point_list, BucketSort{});
    else if (geo) CGS_FWD3(true, false, false, false, ranges, point_list, BucketSort{});
    else CGS_FWD3(false, false, false, false, ranges, point_list, BucketSort{});
}
// (Round 5: ordering only the lists beyond RANK_MAX in a pre-pass and letting this kernel sort the rest was measured at cfg5
// -- render_fwd 183 + tile_sort 74 -> 248 us serial, but 1 484 -> 1 456 Msplats/s with three views in flight -- and dropped.)
bool render_fwd_can_sort(uint32_t cap) { return cap <= RANK_MAX; }
void launch_render_fwd_sorting(hipStream_t s, bool geo, int tiles, const uint32_t* tile_count, const uint64_t* keys,
                               uint32_t cap, uint2* ranges, uint32_t* total, uint32_t* point_list, int W, int H,
                               int grid_x, const SplatRec* rec, float* final_T, uint32_t* n_contrib,
                               const float* bg_color, float* out_color, float* out_invdepth, float* out_all_map, bool unit,
                               bool tag, float* color_clamped, float* dir_out, const float* wv) {
    ProfScope p("render_fwd", s);
    const BucketSort bs{tile_count, keys, point_list, ranges, total, cap};
    const ViewEpilogue epi{color_clamped, geo ? dir_out : nullptr, wv};
    const uint2* no_ranges = nullptr;
    const uint32_t* no_list = nullptr;
    if (geo && unit) CGS_FWD3(true, true, true, true, no_ranges, no_list, bs);
    else if (unit) CGS_FWD3(false, true, true, true, no_ranges, no_list, bs);
    else if (geo && tag) CGS_FWD3(true, true, false, true, no_ranges, no_list, bs);
    else if (tag) CGS_FWD3(false, true, false, true, no_ranges, no_list, bs);
    else if (geo) CGS_FWD3(true, true, false, false, no_ranges, no_list, bs);
    else CGS_FWD3(false, true, false, false, no_ranges, no_list, bs);
}
#undef CGS_FWD3
void launch_render_bwd(hipStream_t s, bool geo, bool invd, bool colg, int tiles, const uint2* ranges,
                       const uint32_t* point_list, int W, int H, int grid_x, const float* bg_color,
                       const SplatRec* rec, const float* final_Ts, const uint32_t* n_contrib, const float* dL_dpixels,
                       const float* dL_dout_invdepth, const float* dL_dout_all_map, float* grad_acc, int acc_stride,
                       uint32_t id_mask, const uint32_t* nonunit_gate) {
    ProfScope p("render_bwd", s);
#define CGS_BWD(G, I, C)                                                                                         \
    hipLaunchKernelGGL((k_render_bwd3<G, I, C>), dim3(tiles), dim3(256), 0, s, ranges, point_list, W, H, grid_x, \
                       bg_color, rec, final_Ts, n_contrib, dL_dpixels, dL_dout_invdepth, dL_dout_all_map, grad_acc, acc_stride, id_mask, nonunit_gate)
    if (geo && invd) CGS_BWD(true, true, true);        // full-gradient configuration
    else if (geo) CGS_BWD(true, false, true);
    else if (invd) CGS_BWD(false, true, true);
    else if (colg) CGS_BWD(false, false, true);
    else CGS_BWD(false, false, false);                  // training configuration: only dL/dcolor upstream, unit colours
#undef CGS_BWD
}

}  // namespace cgs
