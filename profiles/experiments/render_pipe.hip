// Forward compositor of the sync-free bucket path as a PERSISTENT, role-split kernel (reference K4 + K5 + K6:
// rasterizer_impl.cu:98-138,309-324 and forward.cu:279-417 for one tile list at a time).
//
// Why: in k_render_fwd3<.., SORT> (render.hip) one workgroup per tile first loads its bucket, gathers the splat records,
// sorts, stages and only then composites.  Measured on cfg3 (profiles/r04_experiments.md): of that kernel's 122 us the walk
// is 61 -- the other half is the front end: dependent memory round trips (count -> keys -> records, ~1.3 us each, 8.7 us of
// kernel time per trip at six workgroups per CU), the launch of 10 000 workgroups (6.5 us), barriers and LDS hand-offs, during
// which a workgroup holds 80 VGPRs x 256 threads and 26 KB of LDS and issues next to nothing.
//
// Here a workgroup is five waves that stay resident and pull tiles from 64 interleaved queues:
//   * wave 4, the PREFETCHER, claims the next tile, reads its count and keys, writes depths and indices into LDS and brings the
//     splat records of the list into LDS with direct global -> LDS loads (global_load_lds_dwordx4: no registers, all of a
//     fill's requests in flight at once).  Lists longer than one fill (257..1024 entries) it first sorts itself with a
//     wave-local bitonic network on the full 64-bit keys and then fetches fill by fill in depth order;
//   * waves 0..3, the WALKERS (one 8x8 quadrant each, lane = pixel), never touch global memory except for their outputs:
//     thread i ranks entry i against the tile's depths (rank = list position; equal depths are detected and re-ranked on the
//     full keys), stages its record at that position and evaluates the quadrant reach test; then every wave builds its list
//     and composites exactly as k_render_fwd3 does (exponents of 16 splats x 64 pixels from two bf16 MFMAs, p2_mfma.h);
//   * two workgroup barriers per fill: A "records of fill k are in LDS, the walk of k-1 is over", B "fill k is staged, the
//     raw buffers are free": the prefetcher fetches fill k+1 between B(k) and A(k+1), under the walk of fill k.
// Results are bit-identical to k_render_fwd3<.., SORT>: same staging arithmetic, same order, same walk.
#include "kernels.h"
#include "p2_mfma.h"
#include "composite.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace cgs {

namespace {
constexpr int PB = 256;                           // entries per fill
constexpr int PGROUP = 16;                        // entries per MFMA group
constexpr uint32_t PPAD_OFF = (PB + 1) * 16;      // byte offset of the padding entry in the staged arrays
constexpr uint32_t PIPE_END = 0xFFFFFFFFu;
constexpr int PIPE_WAVES = 5;
constexpr uint32_t PIPE_QUEUES = 64, PIPE_QSTRIDE = 32;   // tile queues and the distance of their counters in u32 words
#ifndef CGS_PIPE_WAVES_PER_SIMD
#define CGS_PIPE_WAVES_PER_SIMD 5
#endif
constexpr uint32_t SI_OFF = 512;                  // u32 index of the splat indices inside the key array (ranked lists)

__device__ __forceinline__ float psat01(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 1.f); }
__device__ __forceinline__ void wave_fence() {   // same-wave LDS hand-off: the LDS queue is in order per wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Workgroup barrier that orders LDS only.  __syncthreads() also drains the wave's GLOBAL stores (s_waitcnt vmcnt(0): its
// release fence covers every address space), which would put the latency of the walkers' eight output stores per pixel on
// the critical path of every tile; nothing this kernel writes to global memory is read back inside it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// 16 (4) bytes per lane from global memory straight into LDS: lane l's data lands at dst + 16 l (4 l); dst is wave-uniform
__device__ __forceinline__ void dma16(const void* src, void* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const void* src, void* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
}
// Ascending bitonic network over k[0..n) run by ONE wave ("flip + disperse": the smaller key goes to the lower index, so
// indices >= n count as +inf and are skipped; n2 = n rounded up to a power of two).
__device__ __forceinline__ void wave_bitonic(uint64_t* k, uint32_t n, uint32_t n2, uint32_t lane) {
    for (uint32_t lsize = 1; (1u << lsize) <= n2; lsize++) {
        const uint32_t size = 1u << lsize;
        for (uint32_t t = lane; t < n2 / 2; t += 64) {
            const uint32_t base = (t >> (lsize - 1)) << lsize, j = t & ((size >> 1) - 1);
            const uint32_t lo = base + j, hi = base + size - 1 - j;
            if (hi < n) {
                const uint64_t a = k[lo], b = k[hi];
                if (a > b) { k[lo] = b; k[hi] = a; }
            }
        }
        wave_fence();
        for (int ld = (int)lsize - 2; ld >= 0; ld--) {
            const uint32_t d = 1u << ld;
            for (uint32_t t = lane; t < n2 / 2; t += 64) {
                const uint32_t lo = ((t >> ld) << (ld + 1)) + (t & (d - 1)), hi = lo + d;
                if (hi < n) {
                    const uint64_t a = k[lo], b = k[hi];
                    if (a > b) { k[lo] = b; k[hi] = a; }
                }
            }
            wave_fence();
        }
    }
}
}  // namespace

// work[c * 32]: next position of tile queue c (zeroed with the tile histogram by the preprocess / view kernel of the same forward)
template <bool GEO, bool UNIT>
__global__ void __launch_bounds__(64 * PIPE_WAVES, CGS_PIPE_WAVES_PER_SIMD) k_render_fwd_pipe(
    int tiles, int W, int H, int grid_x, const SplatRec* __restrict__ rec, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, const float* __restrict__ bg_color, float* __restrict__ out_color,
    float* __restrict__ out_invdepth, float* __restrict__ out_all_map, const uint32_t* __restrict__ tile_count,
    const uint64_t* __restrict__ keys, uint32_t* __restrict__ point_list, uint2* __restrict__ ranges,
    uint32_t* __restrict__ total, uint32_t cap, uint32_t* __restrict__ work) {
    constexpr bool IMAGE_ONLY = UNIT && !GEO;
    // staged entry j of a fill lives at index j + 1 (offset 0 = "nothing blended yet"); index PB + 1 is the padding entry
    __shared__ float4 s_geo[PB + 2];   // {cx, cy, A2, B2}
    __shared__ float4 s_at[PB + 2];    // {colour, 1/depth, C2, log2 opacity}
    __shared__ float4 s_c[GEO ? PB + 2 : 1];
    // the fill's splat records as the prefetcher's direct loads leave them: entry i of the bucket (or of the sorted list)
    __shared__ float4 r_a[PB], r_b[PB], r_c[GEO ? PB : 1];
    __shared__ float r_t[PB];
    __shared__ uint32_t s_tag[2][PB];  // per list position of the fill: 0x100 | quadrant mask (0: no entry); fills alternate
    __shared__ uint32_t s_meta[4];     // tile (PIPE_END: no more work), entries in the fill, first list position, flags
    __shared__ uint32_t s_tie;
    __shared__ __attribute__((aligned(16))) uint32_t s_list[4][PB + PGROUP];
    // the tile's keys: lists of one fill as two u32 arrays (depths at [0, n + 4), splat indices at [SI_OFF, SI_OFF + n)) for
    // the walkers' rank loop; longer lists as u64 keys, sorted in place by the prefetcher
    __shared__ __attribute__((aligned(16))) uint64_t s_key[RANK_MAX + 2];
    uint32_t* const sd = reinterpret_cast<uint32_t*>(s_key);
    uint32_t* const si = sd + SI_OFF;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) {
        s_geo[PB + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_at[PB + 1] = make_float4(0.f, 0.f, 0.f, L2_NEVER);
        if (GEO) s_c[GEO ? PB + 1 : 0] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_tie = 0u;
    }
    if (threadIdx.x < PB) { s_tag[0][threadIdx.x] = 0u; s_tag[1][threadIdx.x] = 0u; }

    // ================================================================================================ prefetcher state
    uint32_t p_tile = 0, p_n = 0, p_fill = 0, p_nfills = 0;
    bool p_sorted = false;
    uint32_t p_id[4] = {0u, 0u, 0u, 0u};   // ranked lists: the splat indices of this lane's four bucket entries
    // Tiles are dealt from PIPE_QUEUES interleaved queues (queue c owns tiles c, c + Q, c + 2Q, ...), each behind its own
    // counter on its own 128-byte line: atomics on ONE address retire at ~20-40 ns each chip-wide (10 000 claims = 0.4 ms,
    // measured: a single counter made this kernel take 499 us), on 64 lines they run side by side.  A workgroup drains its
    // home queue first and then steals from the others in order.
    uint32_t p_queue = blockIdx.x % PIPE_QUEUES;   // the queue this workgroup is drawing from
    uint32_t p_claim = 0;   // lane 0: the position claimed in p_queue for the NEXT new tile (an atomic in flight until first read)
    if (wave == 4 && lane == 0) p_claim = atomicAdd(&work[p_queue * PIPE_QSTRIDE], 1u);

    auto fetch = [&]() {
        if (p_fill >= p_nfills) {   // wave-uniform: a new tile
            uint32_t pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)p_claim);
            uint32_t t = p_queue + pos * PIPE_QUEUES;
            while (t >= (uint32_t)tiles) {   // this queue is empty: steal from the next queue that still has tiles
                // lane c looks at queue c's counter (one round trip for all 64), then one claim on the first live queue
                // behind the current one; a queue that runs dry in between just sends us round again
                const uint32_t seen = __hip_atomic_load(&work[(uint32_t)lane * PIPE_QSTRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint64_t live = ballot64((uint32_t)lane + seen * PIPE_QUEUES < (uint32_t)tiles);
                if (live == 0ull) break;
                const uint32_t sh = (p_queue + 1u) % PIPE_QUEUES;
                const uint64_t rot = sh ? ((live >> sh) | (live << (64u - sh))) : live;   // bit k = queue (sh + k) % 64
                p_queue = (sh + (uint32_t)__builtin_ctzll(rot)) % PIPE_QUEUES;
                if (lane == 0) p_claim = atomicAdd(&work[p_queue * PIPE_QSTRIDE], 1u);
                pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)p_claim);
                t = p_queue + pos * PIPE_QUEUES;
            }
            if (t >= (uint32_t)tiles) {
                if (lane == 0) s_meta[0] = PIPE_END;
                return;
            }
            if (lane == 0) p_claim = atomicAdd(&work[p_queue * PIPE_QSTRIDE], 1u);   // the claim after this one: in flight meanwhile
            const uint32_t base = t * cap;
            // keys requested before the count is known (slot i exists whenever i < cap)
            uint64_t k4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t i = (uint32_t)lane + 64u * q;
                k4[q] = i < cap ? keys[base + i] : ~0ull;
            }
            const uint32_t cnt = tile_count[t];
            const uint32_t n = min(cnt, cap);
            if (lane == 0) {
                ranges[t] = make_uint2(base, base + n);
                if (cnt) {
                    uint32_t* part = total + 4 + 2 * (t % TOTAL_PARTS);
                    atomicAdd(&part[0], n);
                    atomicMax(&part[1], cnt);
                    if (cnt > cap) {
                        total[2] = 1u;
                        atomicAdd(&total[TOTAL_WORDS], 1u);   // sticky: survives the next forward's clear
                    }
                }
            }
            p_sorted = n > (uint32_t)PB;
            if (!p_sorted) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t i = (uint32_t)lane + 64u * q;
                    p_id[q] = (uint32_t)k4[q];
                    if (i < n) { sd[i] = (uint32_t)(k4[q] >> 32); si[i] = p_id[q]; }
                }
                if (lane < 4) sd[n + lane] = 0x7f800000u;   // +inf padding of the walkers' broadcast loop
            } else {
                for (uint32_t i = (uint32_t)lane; i < n; i += 64u) s_key[i] = keys[base + i];
                wave_fence();
                uint32_t n2 = 2u;
                while (n2 < n) n2 <<= 1;
                wave_bitonic(s_key, n, n2, (uint32_t)lane);
            }
            p_tile = t; p_n = n; p_fill = 0u;
            p_nfills = max(1u, (n + (uint32_t)PB - 1u) / (uint32_t)PB);
        }
        // ---- the records of fill p_fill, straight into LDS
        const uint32_t first = p_fill * (uint32_t)PB;
        const uint32_t m = min((uint32_t)PB, p_n - first);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t j = (uint32_t)lane + 64u * q;
            if (j < m) {
                const uint32_t id = p_sorted ? (uint32_t)s_key[first + j] : p_id[q];
                const SplatRec* r = rec + id;
                dma16(&r->a, &r_a[64 * q]);
                dma16(&r->b, &r_b[64 * q]);
                if (GEO) dma16(&r->c, &r_c[GEO ? 64 * q : 0]);
                dma4(&r->d.z, &r_t[64 * q]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            s_meta[0] = p_tile;
            s_meta[1] = m;
            s_meta[2] = first;
            s_meta[3] = ((p_fill + 1u == p_nfills) ? 1u : 0u) | (p_sorted ? 2u : 0u);
        }
        p_fill++;
    };

    // ================================================================================================ walker state
    const float cA_live = -__uint_as_float(0x3b808080u) * 0x1p100f;   // [alpha >= 1/255] = sat(alpha 2^100 - pred(1/255) 2^100)
    float kbig = 0x1p100f, kcA = cA_live;
    asm volatile("" : "+v"(kbig), "+v"(kcA));   // constants in VGPRs: a three-VGPR fma issues faster (profiles/probes/enc_probe.hip)
    float T_dead = 0.f, Tw = 1.f, cA = kcA;
    uint32_t last_contributor = 0;
    float C = 0.f, Dacc = 0.f, A0 = 0.f, A1 = 0.f, A2 = 0.f, A3 = 0.f;
    bool wave_done = false, inside = false;
    uint32_t pix_id = 0;
    float hx = 0.f, hy = 0.f;
    const P2Frag pix = p2_pixel_operand(lane);
    const int row_splat = p2_row_splat(lane);

    auto walk = [&](int tb, uint32_t tile, uint32_t m, uint32_t first, bool last) {
        if (first == 0u) {   // a new tile: this lane's pixel and its running state
            const uint32_t tx = tile % (uint32_t)grid_x, ty = tile / (uint32_t)grid_x;
            const int px = (int)(tx * TILE) + (((wave & 1) << 3) | (lane & 7));
            const int py = (int)(ty * TILE) + (((wave >> 1) << 3) | (lane >> 3));
            inside = px < W && py < H;
            pix_id = (uint32_t)(W * py + px);
            hx = (float)(tx * TILE) + (float)((wave & 1) << 3) + 3.5f;
            hy = (float)(ty * TILE) + (float)((wave >> 1) << 3) + 4.f * (float)p2_row_half(lane) + 1.5f;
            T_dead = 0.f; Tw = 1.f;
            cA = inside ? kcA : -0x1p126f;
            last_contributor = 0u;
            C = Dacc = A0 = A1 = A2 = A3 = 0.f;
            wave_done = ballot64(cA > -0x1p120f) == 0ull;
        }
        if (!wave_done && m > 0u) {
            uint32_t* const list = s_list[wave];
            const char* const geo_bytes = reinterpret_cast<const char*>(s_geo);
            const char* const at_bytes = reinterpret_cast<const char*>(s_at);
            const char* const c_bytes = reinterpret_cast<const char*>(s_c);
            // ---- this wave's list: the staged entries its quadrant accepted, in list order, padded to a multiple of 16
            int n = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint64_t mk = ballot64((s_tag[tb][c * 64 + lane] >> wave) & 1u);
                if ((mk >> lane) & 1ull) {
                    const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                    list[pos] = (uint32_t)((c * 64 + lane + 1) * 16);
                }
                n += __builtin_popcountll(mk);
            }
            if (lane < PGROUP) list[n + lane] = PPAD_OFF;
            wave_fence();
            uint32_t last_off = 0;   // byte offset (16 * (staged index + 1)) of the last splat this pixel blended in this fill
            for (int g0 = 0; g0 < n; g0 += PGROUP) {
                f32x16 P;
                {
                    const uint32_t joff = list[g0 + row_splat];
                    const float4 ge = *reinterpret_cast<const float4*>(geo_bytes + joff);
                    const float2 cl = *reinterpret_cast<const float2*>(at_bytes + joff + 8);
                    P = p2_mfma(p2_splat_operand(lane, ge.x, ge.y, ge.z, ge.w, cl.x, cl.y, hx, hy), pix);
                }
                const int cnt = min(PGROUP, n - g0);
                uint4 w4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int s = 0; s < PGROUP; s += 2) {
                    if (s < cnt) {                                                  // wave-uniform
                    if ((s & 3) == 0) w4 = *reinterpret_cast<const uint4*>(list + g0 + s);
                    const uint32_t j0 = (s & 2) ? w4.z : w4.x, j1 = (s & 2) ? w4.w : w4.y;
                    float2 t0 = make_float2(0.f, 0.f), t1 = t0;
                    if (!UNIT) {
                        t0 = *reinterpret_cast<const float2*>(at_bytes + j0);
                        t1 = *reinterpret_cast<const float2*>(at_bytes + j1);
                    }
                    float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
                    if (GEO) {
                        c0 = *reinterpret_cast<const float4*>(c_bytes + j0);
                        c1 = *reinterpret_cast<const float4*>(c_bytes + j1);
                    }
                    // alpha = min(0.99, opacity * G) = min(0.99, exp2(P)); reference: alpha < 1/255 -> skip (forward.cu:366-368)
                    const float al0 = fminf(0.99f, __builtin_amdgcn_exp2f(P[s]));
                    const float al1 = fminf(0.99f, __builtin_amdgcn_exp2f(P[s + 1]));
                    const float a0 = al0 * psat01(fmaf(al0, kbig, cA));
                    const float a1 = al1 * psat01(fmaf(al1, kbig, cA));
                    // two splats blended together, one termination test (forward.cu:371-376 applied in the rare branch)
                    float wa = a0 * Tw;
                    float T1 = fmaf(-Tw, a0, Tw);
                    float wb = a1 * T1;
                    float T2 = fmaf(-T1, a1, T1);
                    if (__builtin_expect(ballot64(T2 < 0.0001f) != 0ull, 0)) {
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
                    if (!UNIT) {   // offset of the last blended splat (w > 0 exactly when blended; offsets grow along the list)
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
            // 1-based list position (UNIT: of the entry before the terminating one)
            if (last_off) last_contributor = first + (last_off >> 4) - (UNIT ? 1u : 0u);
        }
        if (last) {
            if (UNIT && cA > -0x1p120f) last_contributor = first + m;   // never terminated: no cut
            if (inside) {
                const size_t HW = (size_t)H * W;
                const float T = cA > -0x1p120f ? Tw : T_dead;
                final_T[pix_id] = T;
                n_contrib[pix_id] = last_contributor;
                if (UNIT) C = A3 = 1.f - T;
                out_color[pix_id] = C + T * bg_color[0];
                if (!IMAGE_ONLY) out_invdepth[pix_id] = Dacc;
                if (IMAGE_ONLY) {
                    // (no other outputs)
                } else if (GEO) {
                    out_all_map[pix_id] = A0;
                    out_all_map[HW + pix_id] = A1;
                    out_all_map[2 * HW + pix_id] = A2;
                    out_all_map[3 * HW + pix_id] = A3;
                } else {
                    out_all_map[pix_id] = 0.f;
                    out_all_map[HW + pix_id] = 0.f;
                    out_all_map[2 * HW + pix_id] = 0.f;
                    out_all_map[3 * HW + pix_id] = 0.f;
                }
            }
        }
    };

    // ---- walkers: thread i stages entry i of the fill at its list position (see the header); `full`: rank on (depth, index)
    auto convert = [&](int tb, uint32_t tile, uint32_t m, uint32_t first, bool presorted, bool full, uint32_t base) {
        const uint32_t i = threadIdx.x;
        const bool valid = i < m;
        uint32_t slot = i, id = 0u;
        if (presorted) {
            if (valid) id = (uint32_t)s_key[first + i];
        } else {
            const uint32_t d = valid ? sd[i] : ~0u;
            if (valid) id = si[i];
            if (((uint32_t)__builtin_amdgcn_readfirstlane((int)i) & ~63u) < m) {   // wave-uniform
                if (!full) {
                    slot = rank_loop_f32(reinterpret_cast<const float*>(sd), m, __uint_as_float(d));
                } else {
                    uint32_t r = 0u;
                    for (uint32_t u = 0; u < m; u++) {   // uniform addresses: LDS broadcast
                        const uint32_t du = sd[u], iu = si[u];
                        r += (uint32_t)(du < d || (du == d && iu < id));
                    }
                    slot = r;
                }
            }
        }
        if (valid) {
            const float X0 = (float)((tile % (uint32_t)grid_x) * TILE), Y0 = (float)((tile / (uint32_t)grid_x) * TILE);
            const float4 ra = r_a[i], rb = r_b[i];
            float4 sa, sb;
            stage_splat(ra, rb, sa, sb);
            s_geo[slot + 1] = sa;
            s_at[slot + 1] = make_float4(sb.z, sb.w, sb.x, __builtin_amdgcn_logf(sb.y));   // v_log_f32 = log2
            // UNIT: all_map[3] == 1 is not read back, its slot carries 1/depth -- one 16-byte read per pair in the walk
            if (GEO) {
                const float4 rc = r_c[GEO ? i : 0];
                s_c[GEO ? slot + 1 : 0] = UNIT ? make_float4(rc.x, rc.y, rc.z, sb.w) : rc;
            }
            const uint32_t qm = quadrant_mask(ra, rb, r_t[i], X0, Y0);
            // two keys claiming one position = equal depths (about one tile in 600 at 160 entries): everybody re-ranks
            if (atomicExch(&s_tag[tb][slot], 0x100u | qm) != 0u) s_tie = 1u;
            // UNIT (view entry points): the list entry carries the quadrant mask in its top four bits for the backward of
            // the same view (LIST_TAG_SHIFT, composite.h)
            point_list[base + first + slot] = UNIT ? (id | (qm << LIST_TAG_SHIFT)) : id;
        }
    };

    // ================================================================================================ the pipeline
    if (wave == 4) fetch();
    lds_barrier();                                          // A(0)
    for (int it = 0;; it++) {
        const int tb = it & 1;
        const uint32_t tile = s_meta[0], m = s_meta[1], first = s_meta[2], flags = s_meta[3];
        if (tile == PIPE_END) break;   // block-uniform
        const uint32_t base = tile * cap;
        if (wave < 4) convert(tb, tile, m, first, (flags & 2u) != 0u, false, base);
        lds_barrier();                                      // B: the fill is staged, the raw buffers are free
        if (s_tie != 0u) {   // block-uniform: equal depths -- clear the claims, rank again on the full keys
            if (threadIdx.x < PB) s_tag[tb][threadIdx.x] = 0u;
            lds_barrier();
            if (threadIdx.x == 0) s_tie = 0u;
            if (wave < 4) convert(tb, tile, m, first, false, true, base);
            lds_barrier();
        }
        if (wave == 4) {
            fetch();                                          // fill k + 1, under the walk of fill k
        } else {
            walk(tb, tile, m, first, (flags & 1u) != 0u);
            s_tag[tb ^ 1][threadIdx.x] = 0u;                  // (last read before A(k): free for the fill after this one)
        }
        lds_barrier();                                      // A: fill k + 1 is in LDS, the walk of fill k is over
    }
}

bool render_fwd_pipe_ok(uint32_t cap) { return cap <= RANK_MAX; }
void launch_render_fwd_pipe(hipStream_t s, bool geo, int tiles, const uint32_t* tile_count, const uint64_t* keys, uint32_t cap,
                            uint2* ranges, uint32_t* total, uint32_t* point_list, int W, int H, int grid_x,
                            const SplatRec* rec, float* final_T, uint32_t* n_contrib, const float* bg_color, float* out_color,
                            float* out_invdepth, float* out_all_map, bool unit, uint32_t* work) {
    ProfScope p("render_fwd", s);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    static int per_cu = 0;
    if (per_cu == 0) {   // resident workgroups per CU of the headline instance (LDS-bound: 4)
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_render_fwd_pipe<true, true>, 64 * PIPE_WAVES, 0) != hipSuccess || nb <= 0) nb = 4;
        per_cu = nb;
        if (getenv("CGS_PIPE_DEBUG")) fprintf(stderr, "render_fwd_pipe: %d CUs, %d workgroups per CU\n", cus, per_cu);
    }
    const int grid = std::min(tiles, cus * per_cu);
#define CGS_PIPE(G, U)                                                                                                   \
    hipLaunchKernelGGL((k_render_fwd_pipe<G, U>), dim3(grid), dim3(64 * PIPE_WAVES), 0, s, tiles, W, H, grid_x, rec, final_T, \
                       n_contrib, bg_color, out_color, out_invdepth, out_all_map, tile_count, keys, point_list, ranges, total, \
                       cap, work)
    if (geo && unit) CGS_PIPE(true, true);
    else if (unit) CGS_PIPE(false, true);
    else if (geo) CGS_PIPE(true, false);
    else CGS_PIPE(false, false);
#undef CGS_PIPE
}

}  // namespace cgs
