// Tile binning: replaces the reference's scan + duplicateWithKeys + 64-bit global radix sort + identifyTileRanges
// (K2-K5: rasterizer_impl.cu:70-138, :283-324) with a counting sort by tile followed by a per-tile LDS sort.
//
//   k_preprocess_fwd   counts instances per tile (atomics on a [tiles] histogram)
//   k_scan_tiles       exclusive scan of the histogram -> ranges[tile] = [start,end), total R
//   k_scatter          each splat claims a slot in every tile bucket of its rect and writes the 64-bit key
//                      (depth_bits << 32) | splat_idx
//   k_tile_sort        one workgroup per tile sorts its bucket in LDS and emits point_list
//
// Ordering contract: the reference's stable radix sort on (tile << 32 | depth_bits) leaves ties in emission
// order = ascending splat index.  Sorting each tile's bucket by the unique key
// (depth_bits, splat_idx) reproduces exactly that order, independent of the (non-deterministic) slot order
// produced by the atomics in k_scatter.
#include "kernels.h"

namespace cgs {

// Single workgroup, 1024 threads; tiles can be any size (loops with a running carry).
__global__ void __launch_bounds__(1024) k_scan_tiles(int tiles, const uint32_t* __restrict__ tile_count,
                                                     uint2* __restrict__ ranges, uint32_t* __restrict__ total) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t wave_max[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    uint32_t vmax = 0;
    for (int base = 0; base < tiles; base += 1024) {
        const int i = base + tid;
        const uint32_t v = (i < tiles) ? tile_count[i] : 0u;
        vmax = max(vmax, v);
        // inclusive scan inside the wave
        uint32_t s = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t n = __shfl_up(s, off, 64);
            if (lane >= off) s += n;
        }
        if (lane == 63) wave_tot[wid] = s;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wid; w++) woff += wave_tot[w];
        const uint32_t carry = carry_s;
        const uint32_t incl = carry + woff + s;
        if (i < tiles) ranges[i] = make_uint2(incl - v, incl);
        __syncthreads();
        if (tid == 1023) carry_s = incl;
        __syncthreads();
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vmax = max(vmax, (uint32_t)__shfl_xor((int)vmax, off, 64));
    if (lane == 0) wave_max[wid] = vmax;
    __syncthreads();
    if (tid == 0) {
        uint32_t mx = 0;
        for (int w = 0; w < 16; w++) mx = max(mx, wave_max[w]);
        total[0] = carry_s;  // R
        total[1] = mx;       // longest tile list (selects the sort path on the host)
    }
}

// Every iteration of the cooperative walk is a dependent chain of memory round trips (ranges load -> returning atomic
// -> key store), so a wave's run time is (splats per wave) x (two L2 latencies): the kernel is latency-bound, not
// throughput-bound.  Only the first SCATTER_SPW lanes of a wave own a splat, which shortens the serial chain and
// multiplies the number of waves in flight by 64 / SCATTER_SPW.
#ifndef CGS_SCATTER_SPW
#define CGS_SCATTER_SPW 16
#endif
constexpr int SCATTER_SPW = CGS_SCATTER_SPW;
// BUCKET = false: exact layout -- slot inside the tile's range from the scan.  BUCKET = true: single-pass layout --
// every tile owns a fixed-capacity bucket [tile * cap, (tile + 1) * cap); tile_cursor doubles as the instance count
// (no count pass, no scan); instances beyond `cap` are counted but not stored (the sort flags the overflow).
template <bool BUCKET>
__global__ void __launch_bounds__(256) k_scatter(int P, const int* __restrict__ radii,
                                                 const SplatRec* __restrict__ rec, int grid_x, int grid_y,
                                                 const uint2* __restrict__ ranges, uint32_t* __restrict__ tile_cursor,
                                                 uint64_t* __restrict__ keys, uint32_t cap, int cull,
                                                 uint32_t* __restrict__ nonunit) {
    const int lane = threadIdx.x & 63;
    const int idx = (blockIdx.x * 4 + (threadIdx.x >> 6)) * SCATTER_SPW + lane;
    const int radius = (lane < SCATTER_SPW && idx < P) ? radii[idx] : 0;
    uint2 rmin = make_uint2(0, 0), rmax = make_uint2(0, 0);
    uint32_t key_lo = 0, key_hi = 0;
    float4 ra = make_float4(0.f, 0.f, 1.f, 0.f);
    float conic_z = 1.f, tau2 = -1.f;
    if (radius > 0) {
        ra = rec[idx].a;
        get_rect(ra.x, ra.y, radius, grid_x, grid_y, rmin, rmax);
        const float4 d = rec[idx].d;
        key_hi = __float_as_uint(d.x);  // depth bits (positive floats order like unsigned ints)
        key_lo = (uint32_t)idx;
        if (nonunit && d.w != 0.f) *nonunit = 1u;   // a visible splat whose colour / all_map[3] is not 1 (splat_record)
        if (cull) {
            conic_z = rec[idx].b.x;
            tau2 = d.z;
        }
    }
    for_each_rect_tile_coop(radius > 0, rmin, rmax, [&](int src, uint32_t tx, uint32_t ty) {
        if (cull && !tile_reach_det(readlane_f(ra.x, src), readlane_f(ra.y, src), readlane_f(ra.z, src),
                                    readlane_f(ra.w, src), readlane_f(conic_z, src), readlane_f(tau2, src),
                                    (float)(tx * TILE), (float)(ty * TILE)))
            return;
        const uint32_t tile = ty * (uint32_t)grid_x + tx;
        const uint64_t key = ((uint64_t)__builtin_amdgcn_readlane(key_hi, src) << 32) | __builtin_amdgcn_readlane(key_lo, src);
        if (BUCKET) {
            const uint32_t slot = atomicAdd(&tile_cursor[tile], 1u);
            if (slot < cap) keys[(size_t)tile * cap + slot] = key;
        } else {
            const uint2 rg = ranges[tile];
            if (rg.y > cap) return;  // speculative buffer too small for this tile: the host re-runs with the exact size
            const uint32_t slot = atomicAdd(&tile_cursor[tile], 1u);
            if (rg.x + slot < rg.y) keys[rg.x + slot] = key;  // (always true; keeps a count/scatter disagreement memory-safe)
        }
    });
}

// Ascending bitonic network in its "flip + disperse" form: every compare-exchange puts the smaller key at the
// lower index, so indices >= n can be treated as +inf and simply skipped -- any n works without padding.
// WAVE_LOCAL (LDS only, 256 threads, t = tid + 256 k): the 64 pairs a wave handles in one trip of a pass cover one
// aligned run of 128 elements whenever the pass spans <= 128 elements (flip) or has stride <= 64 (disperse), and it is
// the same run in every such pass -- so between two such passes a wave only has to order its own LDS accesses, no
// workgroup barrier: 6 instead of 55 barriers for 1024 keys.
template <bool WAVE_LOCAL, typename Ptr>
__device__ __forceinline__ void bitonic_any_n(Ptr k, uint32_t n, uint32_t n2, uint32_t tid, uint32_t nthreads) {
    auto sync = [&](bool done_local, bool next_local) {
        if (WAVE_LOCAL && done_local && next_local) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    };
    // (strides are powers of two: shifts and masks, no integer division in the index arithmetic)
    for (uint32_t lsize = 1; (1u << lsize) <= n2; lsize++) {
        const uint32_t size = 1u << lsize;
        // flip: i <-> (block_end - 1 - i)
        for (uint32_t t = tid; t < n2 / 2; t += nthreads) {
            const uint32_t base = (t >> (lsize - 1)) << lsize, j = t & ((size >> 1) - 1);
            const uint32_t lo = base + j, hi = base + size - 1 - j;
            if (hi < n) {
                const uint64_t a = k[lo], b = k[hi];
                if (a > b) { k[lo] = b; k[hi] = a; }
            }
        }
        // next pass: the first disperse (stride size/4) if there is one, else the flip of the next size
        sync(size <= 128, size >= 4 ? (size >> 2) <= 64 : 2 * size <= 128);
        for (int ld = (int)lsize - 2; ld >= 0; ld--) {
            const uint32_t d = 1u << ld;
            for (uint32_t t = tid; t < n2 / 2; t += nthreads) {
                const uint32_t lo = ((t >> ld) << (ld + 1)) + (t & (d - 1)), hi = lo + d;
                if (hi < n) {
                    const uint64_t a = k[lo], b = k[hi];
                    if (a > b) { k[lo] = b; k[hi] = a; }
                }
            }
            sync(d <= 64, d > 1 ? (d >> 1) <= 64 : 2 * size <= 128);
        }
    }
    if (WAVE_LOCAL) __syncthreads();   // (the last pass may have ended on a wave-level fence)
}

// Small buckets (n <= 1024, i.e. every tile of the BASELINE configs): counting rank sort.  Keys are unique, so
// rank(key) = #{keys < key} is its final position.  Every thread ranks up to 4 keys against wave-uniform (broadcast)
// LDS reads of all n keys: no barriers inside the O(n^2/256) loop, 8 KiB of LDS, 8 workgroups per CU (the bitonic
// network spends its time in 36+ barrier-separated LDS round trips at 5 workgroups per CU).
// Measured on cfg3 (mean list 225): 81 us vs 90 us for the LDS bitonic network; unrolling the broadcast loop or
// splitting the compare into 32-bit depth/idx parts was slower (103 / 117 us).
// Bucket layout: derive this tile's range from its instance count, publish it for the compositor and fold the
// count into the partial sums / maxima of `total` (num_rendered, longest list; longest > cap <=> overflow).
__device__ __forceinline__ uint2 bucket_range(const uint32_t* __restrict__ tile_count, uint2* __restrict__ ranges_out,
                                              uint32_t* __restrict__ total, uint32_t cap, bool publish) {
    const uint32_t cnt = tile_count[blockIdx.x];
    const uint32_t n = min(cnt, cap);
    const uint32_t base = blockIdx.x * cap;
    if (publish && threadIdx.x == 0) {
        ranges_out[blockIdx.x] = make_uint2(base, base + n);
        if (cnt) {  // same-address atomics from every tile serialise (~20 ns each): spread over TOTAL_PARTS partial slots
            uint32_t* part = total + 4 + 2 * (blockIdx.x % TOTAL_PARTS);
            atomicAdd(&part[0], n);
            atomicMax(&part[1], cnt);  // > cap  <=>  this bucket overflowed
            if (cnt > cap) {               // device-visible overflow flag (read by sync-free consumers)
                total[2] = 1u;
                atomicAdd(&total[TOTAL_WORDS], 1u);   // sticky: survives the next forward's clear (caller resets it)
            }
        }
    }
    return make_uint2(base, base + n);
}

// ---------------------------------------------------------------------------------------- quad-walk bucket scatter
// One splat's rect holds ~12 tiles (cfg3: radius 19 px median, <= 49 tiles), so the one-splat-per-step cooperative walk
// runs its ~150 instructions per step (tile index division, exact reach test, atomic, store) with 1/5 of the lanes
// useful -- the bucket scatter was bound by VALU issue, not by its atomics.  Here a wave walks FOUR splats per step:
// lane group g = lane / 16 takes the g-th pending splat, lane % 16 is the tile inside its rect (16 tiles per step and
// group), and the per-splat constants reach the group through ds_bpermute.
struct SplatWalk {                  // per-lane: this lane's own splat; fetch(): the splat of lane `src`
    uint32_t x0, y0, w, nt;
    bool nonunit;                   // (own splat only) colour or all_map[3] differs from 1
    float cx, cy, A, B, C, tau2;
    uint32_t khi, klo;
    __device__ __forceinline__ SplatWalk fetch(int src) const {
        SplatWalk o;
        o.x0 = (uint32_t)__shfl((int)x0, src, 64); o.y0 = (uint32_t)__shfl((int)y0, src, 64);
        o.w = (uint32_t)__shfl((int)w, src, 64); o.nt = (uint32_t)__shfl((int)nt, src, 64);
        o.cx = __shfl(cx, src, 64); o.cy = __shfl(cy, src, 64); o.A = __shfl(A, src, 64); o.B = __shfl(B, src, 64);
        o.C = __shfl(C, src, 64); o.tau2 = __shfl(tau2, src, 64);
        o.khi = (uint32_t)__shfl((int)khi, src, 64); o.klo = (uint32_t)__shfl((int)klo, src, 64);
        o.nonunit = false;
        return o;
    }
};
__device__ __forceinline__ SplatWalk load_splat_walk(bool owner, int idx, int P, const int* __restrict__ radii,
                                                     const SplatRec* __restrict__ rec, int grid_x, int grid_y) {
    SplatWalk s{};
    s.w = 1; s.A = 1.f; s.C = 1.f; s.tau2 = -1.f;
    const int radius = (owner && idx < P) ? radii[idx] : 0;
    if (radius > 0) {
        const float4 a = rec[idx].a;
        uint2 rmin, rmax;
        get_rect(a.x, a.y, radius, grid_x, grid_y, rmin, rmax);
        const float4 d = rec[idx].d;
        s.x0 = rmin.x; s.y0 = rmin.y; s.w = max(rmax.x - rmin.x, 1u); s.nt = (rmax.x - rmin.x) * (rmax.y - rmin.y);
        s.cx = a.x; s.cy = a.y; s.A = a.z; s.B = a.w; s.C = rec[idx].b.x; s.tau2 = d.z;
        s.khi = __float_as_uint(d.x);  // depth bits (positive floats order like unsigned ints)
        s.klo = (uint32_t)idx;
        s.nonunit = d.w != 0.f;
    }
    return s;
}
// fn(pass, tile, key) is called by ALL lanes in every step (wave-uniform control flow); pass marks a real instance
template <typename F>
__device__ __forceinline__ void quad_walk(const SplatWalk& me, int grid_x, int cull, F fn) {
    const int lane = lane_id();
    const int g = lane >> 4, sub = lane & 15;
    uint64_t todo = __ballot(me.nt > 0);
    while (todo) {
        int src = 64;  // the g-th lowest pending splat lane (64: this group idles in this step)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int s = todo ? __builtin_ctzll(todo) : 64;
            todo &= todo - 1;  // (0 & ~0 stays 0)
            src = g == k ? s : src;
        }
        const bool gvalid = src < 64;
        const SplatWalk sp = me.fetch(gvalid ? src : lane);
        const uint32_t nt = gvalid ? sp.nt : 0u;
        for (uint32_t base = 0; __ballot(base < nt) != 0ull; base += 16) {
            const uint32_t t = base + (uint32_t)sub;
            bool pass = false;
            uint32_t tile = 0, gx_ = 0, gy_ = 0;
            if (t < nt) {
                const uint32_t ty = t / sp.w, tx = t - ty * sp.w;
                const uint32_t gxx = sp.x0 + tx, gyy = sp.y0 + ty;
                pass = !cull || tile_reach_det(sp.cx, sp.cy, sp.A, sp.B, sp.C, sp.tau2, (float)(gxx * TILE), (float)(gyy * TILE));
                tile = gyy * (uint32_t)grid_x + gxx;
                gx_ = gxx; gy_ = gyy;
            }
            fn(pass, tile, ((uint64_t)sp.khi << 32) | sp.klo, gx_, gy_);
        }
    }
}
// ---------------------------------------------------------------------------------------- grouped bucket scatter
// With the walk out of the way the scatter is bound by its returning atomics (the L2 retires ~20 atomic requests per ns
// chip-wide: ~50 us for 1.6 M instances).  Consecutive splats lie on the same curve and mostly fall into the same few
// tiles, so a wave first collects the instances of its GW_SPW splats in LDS, groups them by tile, claims a RUN of slots per
// tile with a single atomic (add = run length) and stores the run's keys behind it.
// Measured on cfg3 (1.62 M instances): one-splat walk 54 us, quad walk 53 us (atomic-bound); grouping by a per-wave rank
// sort on tile << 8 | position (round 1) 33 us at ~600 vector instructions per wave; grouping through the window table
// below 26 us at ~60 (cfg5: 159 -> 109 us).  4 / 16 / 32 splats per wave: 30 / 40 / 52 us.
__device__ __forceinline__ void wave_lds_fence() {  // same-wave LDS hand-off: the LDS queue is in order per wave, so
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // only the compiler has to be kept from reordering
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Splats whose rect holds more than BIG_TILES tiles (a near-camera splat of a room-scale scene can cover the whole screen:
// 3 225 tiles at cfg4) would keep one wave walking for tens of microseconds -- and the splats of one curve are neighbours,
// so a wave tends to hold eight of them: the kernel's tail (54 us at cfg4 for 165 k instances).  They are counted in
// big_count and, when the caller passes a queue, deferred to k_scatter_big (one workgroup per splat).
constexpr uint32_t BIG_TILES = 96;
// The splats of a wave are neighbours on a curve, so their tiles fall into a small window of the tile grid.  A WIN x WIN table of counters in LDS, anchored at the smallest tile coordinates of the wave's rects,
// groups the instances without sorting them: a returning LDS add hands every instance its rank inside its (wave, tile)
// cell, one lane per non-empty cell claims the cell's run of bucket slots with ONE global atomic, and every instance stores
// its key at run base + rank.  Instances outside the window (and beyond the list capacity) take one global atomic each.
// Splats per wave: 12 = the samples of ONE curve (the model's default n_gaussians: their rects share one window by
// construction).  Round 4, cfg3 / cfg5 / cfg4 / cfg2: 4: 29.7 us; 6: 29.1; 8: 26.2 / 109.7 / 21.8 / 12.4; 12: 24.4 / 97.2 / 23.8 /
// 13.0; 16: 39.8 / 167 / 23.3 / 16.9 (fewer, longer-running waves).
// Chosen per launch (template parameter, launch_scatter_bucket): 12 for dense views, 8 when the previous forward of the shape
// binned few instances (latency-bound either way; a sparse view wants more, shorter waves).
constexpr uint32_t GW_WIN = 16, GW_CELLS = GW_WIN * GW_WIN;
#ifndef CGS_GW_LCAP
#define CGS_GW_LCAP 192
#endif
constexpr uint32_t GW_LCAP = CGS_GW_LCAP;
template <int GW_SPW>
__global__ void __launch_bounds__(256) k_scatter_window(int P, const int* __restrict__ radii,
                                                        const SplatRec* __restrict__ rec, int grid_x, int grid_y,
                                                        uint32_t* __restrict__ tile_count, uint64_t* __restrict__ keys,
                                                        uint32_t cap, int cull, uint32_t* __restrict__ big_count,
                                                        uint32_t* __restrict__ big_queue, uint32_t big_cap,
                                                        uint32_t* __restrict__ nonunit) {
    __shared__ uint32_t s_cell[4][GW_CELLS];   // per wave: instances per window cell; then the cell's first bucket slot
    __shared__ uint16_t s_ent[4][GW_LCAP];     // per wave and instance: cell | rank inside the cell << 8
    __shared__ uint64_t s_key[4][GW_LCAP];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* wcell = s_cell[wave];
    uint16_t* went = s_ent[wave];
    uint64_t* wkey = s_key[wave];
    const int idx = (blockIdx.x * 4 + (int)wave) * GW_SPW + (int)lane;
    SplatWalk me = load_splat_walk((int)lane < GW_SPW, idx, P, radii, rec, grid_x, grid_y);
    // the operator API's device-side "unit colours" decision: any visible splat whose colour or all_map[3] is not exactly 1
    // raises the word (benign race: every writer stores the same value)
    if (nonunit && me.nonunit) *nonunit = 1u;
    if (me.nt > BIG_TILES) {
        const uint32_t q = atomicAdd(big_count, 1u);
        if (big_queue && q < big_cap) {
            big_queue[q] = (uint32_t)idx;
            me.nt = 0u;                                    // handled by k_scatter_big
        }
    }
    if (__ballot(me.nt > 0) == 0ull) return;
    // window origin: smallest tile coordinates over the wave's rects
    uint32_t wx0 = me.nt > 0 ? me.x0 : 0xFFFFFFFFu, wy0 = me.nt > 0 ? me.y0 : 0xFFFFFFFFu;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        wx0 = min(wx0, (uint32_t)__shfl_xor((int)wx0, off, 64));
        wy0 = min(wy0, (uint32_t)__shfl_xor((int)wy0, off, 64));
    }
    wx0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wx0);
    wy0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wy0);
#pragma unroll
    for (uint32_t q = 0; q < GW_CELLS / 64; q++) wcell[lane + 64 * q] = 0u;
    wave_lds_fence();
    // ---- A: collect the instances, rank them inside their window cell (cnt stays wave-uniform)
    uint32_t cnt = 0;
    quad_walk(me, grid_x, cull, [&](bool pass, uint32_t tile, uint64_t key, uint32_t gxx, uint32_t gyy) {
        const uint32_t dx = gxx - wx0, dy = gyy - wy0;
        const bool inwin = pass && dx < GW_WIN && dy < GW_WIN;
        const uint64_t bal = __ballot(inwin);
        const uint32_t pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (inwin && pos < GW_LCAP) {
            const uint32_t cell = dy * GW_WIN + dx;
            const uint32_t r = atomicAdd(&wcell[cell], 1u);          // ds_add_rtn_u32: rank inside the cell (< GW_SPW)
            went[pos] = (uint16_t)(cell | (r << 8));
            wkey[pos] = key;
        } else if (pass) {  // outside the window / list full: one atomic per instance
            const uint32_t slot = atomicAdd(&tile_count[tile], 1u);
            if (slot < cap) keys[(size_t)tile * cap + slot] = key;
        }
        cnt += (uint32_t)__builtin_popcountll(bal);
    });
    const uint32_t n = min(cnt, GW_LCAP);
    if (n == 0) return;
    wave_lds_fence();
    // ---- B: one atomic per non-empty cell claims its run of bucket slots
#pragma unroll
    for (uint32_t q = 0; q < GW_CELLS / 64; q++) {
        const uint32_t cell = lane + 64 * q;
        const uint32_t c = wcell[cell];
        if (c) {
            const uint32_t tile = (wy0 + cell / GW_WIN) * (uint32_t)grid_x + wx0 + cell % GW_WIN;
            wcell[cell] = atomicAdd(&tile_count[tile], c);
        }
    }
    wave_lds_fence();
    // ---- C: every instance stores its key at run base + rank
    for (uint32_t j = lane; j < n; j += 64) {
        const uint32_t e = went[j], cell = e & 255u;
        const uint32_t slot = wcell[cell] + (e >> 8);
        const uint32_t tile = (wy0 + cell / GW_WIN) * (uint32_t)grid_x + wx0 + cell % GW_WIN;
        if (slot < cap) keys[(size_t)tile * cap + slot] = wkey[j];
    }
}

// One workgroup per deferred splat: 256 tiles of its rect per step, one atomic per instance (its instances all fall into
// different tiles: nothing to group).
__global__ void __launch_bounds__(256) k_scatter_big(const uint32_t* __restrict__ big_count,
                                                     const uint32_t* __restrict__ big_queue, uint32_t big_cap,
                                                     const int* __restrict__ radii, const SplatRec* __restrict__ rec,
                                                     int grid_x, int grid_y, uint32_t* __restrict__ tile_count,
                                                     uint64_t* __restrict__ keys, uint32_t cap, int cull) {
    const uint32_t n = min(*big_count, big_cap);
    for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
        const int idx = (int)big_queue[e];
        const float4 a = rec[idx].a;
        const float4 d = rec[idx].d;
        const float C = rec[idx].b.x;
        uint2 rmin, rmax;
        get_rect(a.x, a.y, radii[idx], grid_x, grid_y, rmin, rmax);
        const uint32_t w = max(rmax.x - rmin.x, 1u), nt = (rmax.x - rmin.x) * (rmax.y - rmin.y);
        const uint64_t key = ((uint64_t)__float_as_uint(d.x) << 32) | (uint32_t)idx;
        for (uint32_t t = threadIdx.x; t < nt; t += 256) {
            const uint32_t ty = t / w, tx = t - ty * w;
            const uint32_t gxx = rmin.x + tx, gyy = rmin.y + ty;
            if (cull && !tile_reach_det(a.x, a.y, a.z, a.w, C, d.z, (float)(gxx * TILE), (float)(gyy * TILE))) continue;
            const uint32_t tile = gyy * (uint32_t)grid_x + gxx;
            const uint32_t slot = atomicAdd(&tile_count[tile], 1u);
            if (slot < cap) keys[(size_t)tile * cap + slot] = key;
        }
    }
}

// Lists longer than this go to the bitonic network instead of the rank sort: ranking is O(n^2) compares (the kernel is
// VALU-bound on them for long lists: 0.45 ms at cfg5's mean list of 686), the network O(n log^2 n) with a barrier per pass
#ifndef CGS_RANK_SPLIT
#define CGS_RANK_SPLIT 1024
#endif
constexpr uint32_t RANK_SPLIT = CGS_RANK_SPLIT;
// The shared rank sort (common.h, tile_rank_sort) for lists of (min_n, 256 * KPT] entries: KPT = 4 covers every list of
// the 200k-splat configs, a second launch with KPT = 8 the 1025..2048-entry lists of denser scenes (cfg5), the bitonic
// network below the rest.  Bucket layout: the first launch (PUBLISH) derives each tile's range from its instance count
// and publishes it, with the partial totals, for the compositor.
constexpr uint32_t RANK_MAX2 = 2048;
template <bool BUCKET, int KPT, bool PUBLISH>
__global__ void __launch_bounds__(256) k_tile_rank_sort(const uint2* __restrict__ ranges,
                                                        const uint64_t* __restrict__ keys,
                                                        uint32_t* __restrict__ point_list, uint32_t cap,
                                                        const uint32_t* __restrict__ tile_count,
                                                        uint2* __restrict__ ranges_out, uint32_t* __restrict__ total,
                                                        uint32_t min_n) {
    __shared__ __attribute__((aligned(16))) uint32_t sd[256 * KPT + RANK_U];
    __shared__ uint32_t si[256 * KPT];
    __shared__ uint32_t s_hist[RANK_NB], s_start[RANK_NB + 1], s_mm[8];
    const uint2 rg = BUCKET ? bucket_range(tile_count, ranges_out, total, cap, PUBLISH) : ranges[blockIdx.x];
    const uint32_t n = rg.y - rg.x;
    if (n <= min_n || n > 256u * KPT || n > (KPT == 4 ? RANK_SPLIT : ~0u) || (!BUCKET && rg.y > cap)) return;
    const uint32_t tid = threadIdx.x;
    // a wave that owns no key leaves at once (the hardware drops finished waves from the workgroup barriers): its
    // slot goes to the next tile's workgroup -- the kernel is bound by dependent-load latency, i.e. by tiles in flight
    if ((tid & ~63u) >= n) return;
    uint32_t rank[KPT], idx[KPT];
    tile_rank_sort<KPT>(keys + rg.x, n, RankScratch{sd, si, s_hist, s_start, s_mm}, rank, idx);
    uint32_t* out = point_list + rg.x;
#pragma unroll
    for (int q = 0; q < KPT; q++) {
        const uint32_t i = tid + 256u * q;
        if (i < n) out[rank[q]] = idx[q];
    }
}

constexpr uint32_t SORT_LDS_KEYS = 4096;  // 32 KiB of LDS per workgroup

template <bool BUCKET>
__global__ void __launch_bounds__(256) k_tile_sort(const uint2* __restrict__ ranges, uint64_t* __restrict__ keys,
                                                   uint32_t* __restrict__ point_list, uint32_t min_n, uint32_t cap,
                                                   const uint32_t* __restrict__ tile_count) {
    __shared__ uint64_t sk[SORT_LDS_KEYS];
    const uint2 rg = BUCKET ? bucket_range(tile_count, nullptr, nullptr, cap, false) : ranges[blockIdx.x];
    const uint32_t n = rg.y - rg.x;
    if (n <= min_n) return;  // handled by k_tile_rank_sort
    const uint32_t tid = threadIdx.x;
    uint64_t* gk = keys + rg.x;
    uint32_t* out = point_list + rg.x;
    if (n == 1) {
        if (tid == 0) out[0] = (uint32_t)gk[0];
        return;
    }
    uint32_t n2 = 2;
    while (n2 < n) n2 <<= 1;
    if (n <= SORT_LDS_KEYS) {
        for (uint32_t i = tid; i < n; i += 256) sk[i] = gk[i];
        __syncthreads();
        bitonic_any_n<true>(sk, n, n2, tid, 256u);
        for (uint32_t i = tid; i < n; i += 256) out[i] = (uint32_t)sk[i];
    } else {
        // rare oversized bucket: same network directly on global memory (one workgroup => __syncthreads orders it)
        bitonic_any_n<false>(gk, n, n2, tid, 256u);
        for (uint32_t i = tid; i < n; i += 256) out[i] = (uint32_t)gk[i];
    }
}


// ------------------------------------------------------------------------------------------------ launchers
void launch_scan_tiles(hipStream_t s, int tiles, const uint32_t* tile_count, uint2* ranges, uint32_t* total) {
    ProfScope p("scan_tiles", s);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, tiles, tile_count, ranges, total);
}
void launch_scatter(hipStream_t s, int P, const int* radii, const SplatRec* rec, int grid_x, int grid_y,
                    const uint2* ranges, uint32_t* tile_cursor, uint64_t* keys, uint32_t cap, int cull, uint32_t* nonunit) {
    ProfScope p("scatter", s);
    hipLaunchKernelGGL(k_scatter<false>, dim3((P + 4 * SCATTER_SPW - 1) / (4 * SCATTER_SPW)), dim3(256), 0, s, P,
                       radii, rec, grid_x, grid_y, ranges, tile_cursor, keys, cap, cull, nonunit);
}
void launch_tile_sort_small(hipStream_t s, int tiles, const uint2* ranges, uint64_t* keys, uint32_t* point_list,
                            uint32_t cap) {
    ProfScope p("tile_sort", s);
    hipLaunchKernelGGL((k_tile_rank_sort<false, 4, false>), dim3(tiles), dim3(256), 0, s, ranges, keys, point_list, cap,
                       nullptr, nullptr, nullptr, 0u);
}
void launch_tile_sort_big(hipStream_t s, int tiles, const uint2* ranges, uint64_t* keys, uint32_t* point_list,
                          uint32_t max_count) {
    if (max_count <= RANK_SPLIT) return;  // some tile list is longer than that
    uint32_t done = RANK_SPLIT;
    if (RANK_SPLIT == RANK_MAX) {         // 1025..2048 entries: rank sort with 8 keys per thread
        ProfScope p("tile_sort_mid", s);
        hipLaunchKernelGGL((k_tile_rank_sort<false, 8, false>), dim3(tiles), dim3(256), 0, s, ranges, keys, point_list,
                           ~0u, nullptr, nullptr, nullptr, RANK_MAX);
        done = RANK_MAX2;
    }
    ProfScope p("tile_sort_big", s);
    if (max_count > done)                 // beyond: bitonic network (LDS or global)
        hipLaunchKernelGGL(k_tile_sort<false>, dim3(tiles), dim3(256), 0, s, ranges, keys, point_list, done, 0u, nullptr);
}
// Single-pass bucket binning: scatter straight into fixed-capacity tile buckets, then sort each bucket and publish
// ranges / num_rendered / longest list / overflow flag from the sort kernel.
void launch_scatter_bucket(hipStream_t s, int P, const int* radii, const SplatRec* rec, int grid_x, int grid_y,
                           uint32_t* tile_count, uint64_t* keys, uint32_t cap, int cull, uint32_t* big_count,
                           uint32_t* big_queue, uint32_t big_cap, uint32_t* nonunit, int splats_per_wave) {
    ProfScope p("scatter", s);
    if (splats_per_wave == 8)
        hipLaunchKernelGGL(k_scatter_window<8>, dim3((P + 4 * 8 - 1) / (4 * 8)), dim3(256), 0, s, P, radii, rec, grid_x,
                           grid_y, tile_count, keys, cap, cull, big_count, big_queue, big_cap, nonunit);
    else
        hipLaunchKernelGGL(k_scatter_window<12>, dim3((P + 4 * 12 - 1) / (4 * 12)), dim3(256), 0, s, P, radii, rec, grid_x,
                           grid_y, tile_count, keys, cap, cull, big_count, big_queue, big_cap, nonunit);
    if (big_queue)
        hipLaunchKernelGGL(k_scatter_big, dim3(512), dim3(256), 0, s, big_count, big_queue, big_cap, radii, rec, grid_x,
                           grid_y, tile_count, keys, cap, cull);
}
void launch_tile_sort_bucket(hipStream_t s, int tiles, const uint32_t* tile_count, uint2* ranges, uint32_t* total,
                             uint64_t* keys, uint32_t* point_list, uint32_t cap) {
    {
        ProfScope p("tile_sort", s);
        hipLaunchKernelGGL((k_tile_rank_sort<true, 4, true>), dim3(tiles), dim3(256), 0, s, nullptr, keys, point_list, cap,
                           tile_count, ranges, total, 0u);
    }
    if (cap > RANK_SPLIT) {
        uint32_t done = RANK_SPLIT;
        if (RANK_SPLIT == RANK_MAX) {
            ProfScope p("tile_sort_mid", s);
            hipLaunchKernelGGL((k_tile_rank_sort<true, 8, false>), dim3(tiles), dim3(256), 0, s, nullptr, keys, point_list,
                               cap, tile_count, nullptr, nullptr, RANK_MAX);
            done = RANK_MAX2;
        }
        ProfScope p("tile_sort_big", s);
        if (cap > done)
            hipLaunchKernelGGL(k_tile_sort<true>, dim3(tiles), dim3(256), 0, s, nullptr, keys, point_list, done, cap,
                               tile_count);
    }
}
uint32_t bucket_cap_limit() { return SORT_LDS_KEYS; }

}  // namespace cgs
