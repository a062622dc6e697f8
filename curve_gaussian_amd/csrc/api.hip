// C ABI of libcurvegs.so (see include/curvegs.h): host-side sequencing of the HIP kernels on the caller's stream.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"

namespace cgs {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------- per-kernel timing (bench.py roofline leg)
static bool g_prof_on = false;
struct ProfRec { hipEvent_t e0, e1; const char* name; };
static std::vector<ProfRec> g_pending;
static std::map<std::string, std::pair<double, int64_t>> g_totals;
static std::vector<std::string> g_names_storage;
static std::mutex g_prof_mu;

ProfScope::ProfScope(const char* n, hipStream_t s) : name(n), stream(s), on(g_prof_on) {
    if (on) {
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, stream);
    }
}
ProfScope::~ProfScope() {
    if (on) {
        (void)hipEventRecord(e1, stream);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_pending.push_back({e0, e1, name});
    }
}

bool check_launch(const char* what, bool debug, hipStream_t s) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && debug) e = hipStreamSynchronize(s);  // reference CHECK_CUDA semantics, auxiliary.h:178-185
    if (e != hipSuccess) {
        set_error("%s failed: %s", what, hipGetErrorString(e));
        return false;
    }
    return true;
}

// Binning capacity hints, one set per workload shape (P, width, height): a process that alternates train and test cameras,
// two resolutions or two models keeps a separate history for each instead of thrashing one (each mismatch used to cost a
// bucket overflow and an exact-path redo).  Small LRU table behind a mutex; the three numbers of an entry:
//   R     num_rendered of the previous forward of this shape (speculative binning capacity of the exact path)
//   max   longest tile list seen recently (decaying maximum): sizes the fixed-capacity buckets of the single-pass binning
//   big   splats with oversized tile rects seen by the previous blocking forward: non-zero switches their deferral to
//         k_scatter_big on (one more launch, only worth it when there are any -- room-scale scenes with near-camera splats)
struct BinHints { int64_t R = 0, max = 0, big = 0; };
// bucket scatter: splats per wave from the instance count the shape binned last time (profiles/r04_experiments.md #19: 12 wins
// at cfg3 / cfg5 -- 1.6 M / 8 M instances --, 8 at cfg2 / cfg4 -- 0.4 M / 0.17 M)
static inline int scatter_spw(const BinHints& h) { return (h.R > 0 && h.R < 800000) ? 8 : 12; }
struct HintEntry { int dev = -1, P = -1, W = 0, H = 0; uint64_t stamp = 0; BinHints h; };   // dev: the HIP device the shape was seen on
static std::mutex g_hint_mu;
static HintEntry g_hint_tab[32];
static uint64_t g_hint_clock = 0;
// Exact match, or nullptr.  (A pure lookup never inserts: cgs_view_forward / cgs_rasterize_forward_static only ask.)
static int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    return d;
}
static HintEntry* hint_find_locked(int dev, int P, int W, int H) {
    for (auto& e : g_hint_tab)
        if (e.dev == dev && e.P == P && e.W == W && e.H == H) { e.stamp = ++g_hint_clock; return &e; }
    return nullptr;
}
// Most recently used entry of the same resolution but another splat count: a topology edit (densify / prune / split) changes
// P by a few curves, and the new cloud bins almost like the old one -- its history seeds the new shape (R scaled by the
// splat ratio) instead of sending the next forward of every resolution through the exact path again.
static const HintEntry* hint_neighbour_locked(int dev, int W, int H) {
    const HintEntry* best = nullptr;
    for (auto& e : g_hint_tab)
        if (e.dev == dev && e.P > 0 && e.W == W && e.H == H && (!best || e.stamp > best->stamp)) best = &e;
    return best;
}
static BinHints hint_seed_locked(int dev, int P, int W, int H) {
    BinHints h;
    if (const HintEntry* nb = hint_neighbour_locked(dev, W, H)) {
        h = nb->h;
        h.R = (int64_t)((double)nb->h.R * (double)P / (double)nb->P);
    }
    return h;
}
static HintEntry* hint_entry_locked(int dev, int P, int W, int H) {   // find or insert (LRU eviction)
    if (HintEntry* e = hint_find_locked(dev, P, W, H)) return e;
    const BinHints seed = hint_seed_locked(dev, P, W, H);
    HintEntry* lru = &g_hint_tab[0];
    for (auto& e : g_hint_tab)
        if (e.stamp < lru->stamp) lru = &e;
    *lru = HintEntry{};
    lru->dev = dev; lru->P = P; lru->W = W; lru->H = H; lru->stamp = ++g_hint_clock; lru->h = seed;
    return lru;
}
// (the shape's history belongs to the CURRENT device: two ranks' or two models' views on different GPUs of one process do not
// share bucket capacities; `dev` < 0 = ask the runtime)
static BinHints hints_load(int P, int W, int H, int dev = -1) {
    if (dev < 0) dev = current_device();
    std::lock_guard<std::mutex> lk(g_hint_mu);
    if (HintEntry* e = hint_find_locked(dev, P, W, H)) return e->h;
    return hint_seed_locked(dev, P, W, H);
}
// R < 0 / big < 0: leave that field; longest: folded into the decaying maximum
static void hints_update(int P, int W, int H, int64_t R, uint32_t longest, int64_t big, int dev = -1) {
    if (dev < 0) dev = current_device();
    std::lock_guard<std::mutex> lk(g_hint_mu);
    BinHints& h = hint_entry_locked(dev, P, W, H)->h;
    if (R >= 0) h.R = R;
    if (big >= 0) h.big = big;
    // Slowly decaying maximum.  A bucket overflow costs a whole second forward through the exact path, spare capacity only
    // memory (12 bytes per slot and tile), and a training loop cycles through dozens of views whose longest lists differ by
    // tens of per cent: at the round-4 rate of 1/16 per call seven sparser views in a row shrank the capacity by a third
    // and the next dense view overflowed (the general route's eager time was bimodal, 0.40 / 0.53 ms at cfg3).  1/1024 per
    // call (at least one entry: below 1 024 the shift alone would never decay) keeps 94 % after 64 calls -- inside the 25 %
    // margin -- and still follows a cloud that thins out for good.
    h.max = std::max<int64_t>((int64_t)longest, h.max - std::max<int64_t>(1, h.max >> 10));
}
static thread_local int64_t g_last_visible = -1;   // radii > 0 count of the last cgs_view_forward_checked
static thread_local int64_t g_last_stats[3] = {0, 0, 0};  // num_rendered, longest tile list, binning path (0 exact, 1 bucket)

// Device-side zero fill.  hipMemsetAsync is NOT used anywhere in the library: captured into a hipGraph (ROCm 7.0 runtime
// under PyTorch 2.10) its memset node cleared the buffer on the first replay only -- later replays ran on stale
// histograms / partial sums.  A plain kernel node replays correctly, and hipMemsetAsync is a fill kernel anyway.
__global__ void __launch_bounds__(256) k_zero_words(uint32_t* __restrict__ p, size_t words) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) p[i] = 0u;
}
__global__ void __launch_bounds__(256) k_zero_vec(uint4* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
static hipError_t zero_async(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if (!(bytes & 15u) && !(reinterpret_cast<uintptr_t>(p) & 15u)) {
        const size_t n = bytes / 16;
        const int blocks = (int)std::min<size_t>((n + 255) / 256, 8192);
        hipLaunchKernelGGL(k_zero_vec, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint4*>(p), n);
        return hipGetLastError();
    }
    if ((bytes & 3u) || (reinterpret_cast<uintptr_t>(p) & 3u)) return hipMemsetAsync(p, 0, bytes, s);  // never the case here
    const size_t words = bytes / 4;
    const int blocks = (int)std::min<size_t>((words + 255) / 256, 4096);
    hipLaunchKernelGGL(k_zero_words, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p), words);
    return hipGetLastError();
}

// Operator API (cgs_rasterize_forward / _backward): the forward tags its tile lists and lets the scatter raise a device word
// when some visible splat's colour or all_map[3] is not exactly 1; a backward that needs neither colour nor depth / all_map
// gradients then launches the pair-major unit-colour kernel AND the general training instance, and that word decides on the
// device which of the two runs (no host sync).  CGS_OPT_GENERAL_BACKWARD in that call's `debug` bits keeps the general instance only (A/B, tests).
static inline bool list_tags_fit(int P) { return (long long)P < (1ll << LIST_TAG_SHIFT); }
constexpr int NONUNIT_WORD = 8;            // index into ImageState::work (cleared with the tile histogram)
// tile sort inside the forward compositor; CGS_FUSED_TILE_SORT=0 (read once) selects the separate sort launch for A/B runs
static inline bool fuse_sort() {
    static const bool on = [] { const char* e = getenv("CGS_FUSED_TILE_SORT"); return !(e && e[0] == '0'); }();
    return on;
}
// Shared curve sampling for several views of ONE parameter state (cgs_view_forward_shared, CGS_VIEW_SHARED in
// cgs_view_backward's flags): the grid-wide norm pass of the forward and the last pass of the sampling backward run once per
// view BATCH (cgs_view_shared_begin / _end) instead of once per view -- the parameters do not change inside a batch and that
// backward pass is linear in the per-splat gradients.  A per-call choice: nothing process-wide changes what another caller's
// cgs_view_forward / cgs_view_backward does.
constexpr int VIEW_MODE_MASK = 3, VIEW_MODE_SHARED = 4;
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace cgs

using namespace cgs;

// Longest tile list of a finished scatter, for the checked (blocking) view forward: one small launch between the scatter
// and the compositor, so the host's wait ends when the BINNING is done and the compositor is still running.
constexpr int STAT_BLOCKS = 64;            // blocks of k_count_stats = chunks of k_visible_compact
constexpr int VIS_COUNT_WORD = 16;         // work[16 .. 16 + STAT_BLOCKS): splats with radii > 0 per chunk of ceil(P / STAT_BLOCKS)
__global__ void __launch_bounds__(256) k_count_stats(const uint32_t* __restrict__ tile_count, int tiles, const int* __restrict__ radii,
                                                     int P, const uint32_t* __restrict__ big, uint32_t* __restrict__ out4,
                                                     uint32_t* __restrict__ vis_counts) {
    // (same-address atomics serialise at ~15 ns each: one per wave -- 768 of them -- made this reduction take 12.7 us of the
    // host's critical path, profiles/r05_kernel_stats.csv; one per BLOCK after an LDS step: 64 blocks x 3)
    __shared__ uint32_t s_part[3][4];
    uint32_t mx = 0, sum = 0, vis = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < tiles; i += gridDim.x * 256) {
        const uint32_t c = tile_count[i];
        mx = max(mx, c);
        sum += c;
    }
    // visible splats of this block's CONTIGUOUS chunk: the per-chunk counts are what k_visible_compact needs to write
    // (radii > 0).nonzero() in order without a scan of its own
    const int chunk = (P + (int)gridDim.x - 1) / (int)gridDim.x;
    const int lo = blockIdx.x * chunk, hi = min(P, lo + chunk);
    for (int i = lo + threadIdx.x; i < hi; i += 256) vis += radii[i] > 0 ? 1u : 0u;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
        sum += (uint32_t)__shfl_xor((int)sum, off, 64);
        vis += (uint32_t)__shfl_xor((int)vis, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_part[0][threadIdx.x >> 6] = sum;
        s_part[1][threadIdx.x >> 6] = mx;
        s_part[2][threadIdx.x >> 6] = vis;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t bvis = s_part[2][0] + s_part[2][1] + s_part[2][2] + s_part[2][3];
        vis_counts[blockIdx.x] = bvis;
        atomicAdd(&out4[0], s_part[0][0] + s_part[0][1] + s_part[0][2] + s_part[0][3]);                  // num_rendered
        atomicMax(&out4[1], max(max(s_part[1][0], s_part[1][1]), max(s_part[1][2], s_part[1][3])));      // longest tile list
        atomicAdd(&out4[2], bvis);   // splats with radii > 0 (sizes render()'s visibility_filter without a host sync)
        if (blockIdx.x == 0) out4[3] = *big;   // splats with oversized tile rects (final after the scatter)
    }
}
// (radii > 0).nonzero() (gaussian_renderer/__init__.py:150) in one launch: block b writes the indices of its chunk's
// visible splats, in order, behind those of the chunks before it (their counts come from k_count_stats).
__global__ void __launch_bounds__(256) k_visible_compact(const int* __restrict__ radii, int P, const uint32_t* __restrict__ vis_counts,
                                                         long long* __restrict__ out) {
    __shared__ uint32_t s_wave[4];
    uint32_t base = 0;
    for (int b = 0; b < (int)blockIdx.x; b++) base += vis_counts[b];   // (<= 63 uniform loads)
    const int chunk = (P + (int)gridDim.x - 1) / (int)gridDim.x;
    const int lo = blockIdx.x * chunk, hi = min(P, lo + chunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i0 = lo; i0 < hi; i0 += 256) {
        const int i = i0 + (int)threadIdx.x;
        const bool v = i < hi && radii[i] > 0;
        const uint64_t bal = __ballot(v);
        if (lane == 0) s_wave[wave] = (uint32_t)__builtin_popcountll(bal);
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint32_t c = s_wave[w];
            before += w < wave ? c : 0u;
            all += c;
        }
        if (v) out[base + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = (long long)i;
        base += all;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------- render() epilogue
// gaussian_renderer/__init__.py:138-145 for the fused view route in ONE launch: the clamp of the image and the view -> world
// transform of the direction map (the reference runs a clamp kernel and a [H*W,3] x [3,3] matmul on a 1600^2 image), and the
// clamp's gradient mask for the way back.
__global__ void __launch_bounds__(256) k_render_epilogue(size_t npix, const float* __restrict__ color_raw, const float* __restrict__ all_map,
                                                         const float* __restrict__ wv, int clamp, float* __restrict__ color_out,
                                                         float* __restrict__ dir_out) {
    float w00 = 0.f, w01 = 0.f, w02 = 0.f, w10 = 0.f, w11 = 0.f, w12 = 0.f, w20 = 0.f, w21 = 0.f, w22 = 0.f;
    if (dir_out) {   // (viewmatrix may be NULL for a clamp-only call)
        w00 = wv[0]; w01 = wv[1]; w02 = wv[2]; w10 = wv[4]; w11 = wv[5]; w12 = wv[6]; w20 = wv[8]; w21 = wv[9]; w22 = wv[10];
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        if (color_out) {
            const float c = color_raw[i];
            color_out[i] = clamp ? fminf(fmaxf(c, 0.f), 1.f) : c;
        }
        if (dir_out) {   // out_i = sum_k d_k wv[i][k]   (rendered_dir.permute(1, 2, 0) @ world_view_transform[:3, :3].T)
            const float d0 = all_map[i], d1 = all_map[npix + i], d2 = all_map[2 * npix + i];
            dir_out[i] = d0 * w00 + d1 * w01 + d2 * w02;
            dir_out[npix + i] = d0 * w10 + d1 * w11 + d2 * w12;
            dir_out[2 * npix + i] = d0 * w20 + d1 * w21 + d2 * w22;
        }
    }
}
__global__ void __launch_bounds__(256) k_clamp_backward(size_t n, const float* __restrict__ raw, const float* __restrict__ g_in,
                                                        float* __restrict__ g_out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float x = raw[i];
        g_out[i] = (x >= 0.f && x <= 1.f) ? g_in[i] : 0.f;   // torch.clamp's backward: the gradient passes where min <= x <= max
    }
}
extern "C" {

const char* cgs_last_error(void) { return g_err; }
int cgs_version(void) { return 100; }
const char* cgs_target_arch(void) { return "gfx950"; }

size_t cgs_geometry_bytes(int P) {
    char* c = nullptr;
    geom_from_chunk(c, (size_t)(P > 0 ? P : 1));
    return (size_t)c + 128;
}
size_t cgs_image_bytes(int width, int height) {
    char* c = nullptr;
    const size_t tiles = (size_t)((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
    image_from_chunk(c, (size_t)width * height, tiles);
    return (size_t)c + 128;
}
size_t cgs_binning_bytes(int64_t R) {
    char* c = nullptr;
    bin_from_chunk(c, (size_t)(R > 0 ? R : 1));
    return (size_t)c + 128;
}

void cgs_reset_binning_hints(void) {
    std::lock_guard<std::mutex> lk(g_hint_mu);
    for (auto& e : g_hint_tab) e = HintEntry{};
}
void cgs_prof_enable(int on) { g_prof_on = on != 0; }
void cgs_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_pending) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_pending.clear();
    g_totals.clear();
}
int cgs_prof_collect(const char** names, double* total_ms, int64_t* launches, int cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_pending) {
        (void)hipEventSynchronize(r.e1);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            auto& t = g_totals[r.name];
            t.first += ms;
            t.second += 1;
        }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_pending.clear();
    g_names_storage.clear();
    for (auto& kv : g_totals) g_names_storage.push_back(kv.first);
    int n = 0;
    for (auto& kv : g_totals) {
        if (n < cap) {
            names[n] = g_names_storage[n].c_str();
            total_ms[n] = kv.second.first;
            launches[n] = kv.second.second;
        }
        n++;
    }
    return n;
}

int64_t cgs_rasterize_forward(cgs_alloc_fn geometry_alloc, void* geometry_user, cgs_alloc_fn binning_alloc,
                              void* binning_user, cgs_alloc_fn image_alloc, void* image_user, int P, int D, int M,
                              const float* background, int width, int height, const float* means3D, const float* shs,
                              const float* colors_precomp, const float* opacities, const float* scales,
                              float scale_modifier, const float* rotations, const float* cov3D_precomp,
                              const float* all_map, const float* viewmatrix, const float* projmatrix,
                              const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                              float* out_invdepth, float* out_all_map, int antialiasing, int render_geo, int* radii,
                              int debug, void* stream_) {
    (void)prefiltered;
    hipStream_t s = (hipStream_t)stream_;
    g_last_visible = -1;   // (set again by the bucket path, whose status readback carries the count)
    if (P < 0 || width <= 0 || height <= 0 || !out_color || !out_invdepth || !out_all_map || !background ||
        !viewmatrix || !projmatrix) {
        set_error("cgs_rasterize_forward: invalid argument (P=%d W=%d H=%d or NULL output/camera pointer)", P, width, height);
        return CGS_ERR_INVALID_ARGUMENT;
    }
    const size_t npix = (size_t)width * height;
    if (P == 0) {  // rasterize_points.cu:91: outputs stay zero-filled, nothing is rendered (not even background)
        if (zero_async(out_color, npix * 4, s) != hipSuccess || zero_async(out_invdepth, npix * 4, s) != hipSuccess ||
            zero_async(out_all_map, npix * 16, s) != hipSuccess) {
            set_error("zero_async failed");
            return CGS_ERR_HIP;
        }
        return 0;
    }
    if (!means3D || !opacities || !radii || (!shs == !colors_precomp) ||
        (cov3D_precomp ? (scales || rotations) : (!scales || !rotations)) || (render_geo && !all_map) ||
        (shs && (!cam_pos || M <= 0))) {
        set_error("cgs_rasterize_forward: inconsistent inputs (need exactly one of shs/colors_precomp and one of "
                  "(scales,rotations)/cov3D_precomp; all_map is required with render_geo)");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    if (!aligned16(rotations) || !aligned16(all_map)) {
        set_error("cgs_rasterize_forward: rotations/all_map must be 16-byte aligned");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE;
    const int tiles = gx * gy;
    const float focal_y = height / (2.0f * tan_fovy);  // rasterizer_impl.cu:227-228
    const float focal_x = width / (2.0f * tan_fovx);

    char* gchunk = (char*)geometry_alloc(geometry_user, cgs_geometry_bytes(P));
    char* ichunk = (char*)image_alloc(image_user, cgs_image_bytes(width, height));
    if (!gchunk || !ichunk) {
        set_error("cgs_rasterize_forward: geometry/image allocation callback returned NULL");
        return CGS_ERR_ALLOC;
    }
    GeomState geom = geom_from_chunk(gchunk, (size_t)P);
    ImageState img = image_from_chunk(ichunk, npix, (size_t)tiles);

    // tile_count and tile_cursor are adjacent 128B-aligned carve-outs: clear both (+total) with one memset
    const size_t clear_bytes = (size_t)((char*)(img.total + TOTAL_WORDS) - (char*)img.tile_count);
    const int cull = (debug & CGS_OPT_NO_TILE_CULLING) ? 0 : 1;   // per call (include/curvegs.h)
    debug &= CGS_OPT_DEBUG;
    // pinned copy of img.total + the event behind it: per calling thread AND per device (an event only records on streams of
    // the device it was created on)
    constexpr int MAX_DEV = 16;
    static thread_local uint32_t* h_tot_dev[MAX_DEV] = {};
    static thread_local hipEvent_t ev_dev[MAX_DEV] = {};
    const int dev_ix = current_device();
    if (dev_ix < 0 || dev_ix >= MAX_DEV) {
        set_error("cgs_rasterize_forward: device index %d out of range", dev_ix);
        return CGS_ERR_INVALID_ARGUMENT;
    }
    if (!h_tot_dev[dev_ix]) {
        if (hipHostMalloc((void**)&h_tot_dev[dev_ix], TOTAL_WORDS * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&ev_dev[dev_ix], hipEventDisableTiming) != hipSuccess) {
            set_error("pinned readback buffer / event creation failed");
            if (h_tot_dev[dev_ix]) (void)hipHostFree(h_tot_dev[dev_ix]);
            h_tot_dev[dev_ix] = nullptr;
            return CGS_ERR_HIP;
        }
    }
    uint32_t* const h_tot = h_tot_dev[dev_ix];
    const hipEvent_t ev = ev_dev[dev_ix];
    auto read_totals = [&]() -> bool {  // async copy of img.total + event; the caller waits on the event later
        hipError_t e = hipMemcpyAsync(h_tot, img.total, TOTAL_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipEventRecord(ev, s);
        if (e != hipSuccess) set_error("reading num_rendered failed: %s", hipGetErrorString(e));
        return e == hipSuccess;
    };
    auto wait_totals = [&]() -> bool {
        const hipError_t e = hipEventSynchronize(ev);
        if (e != hipSuccess) set_error("reading num_rendered failed: %s", hipGetErrorString(e));
        return e == hipSuccess;
    };
    // tile_count == NULL (bucket binning): the kernel does not count, so it can clear the histogram/cursors/status words
    // itself; with counting, they are cleared by a separate launch first.  Either way it zeroes the gradient accumulators.
    auto preprocess = [&](uint32_t* tile_count) -> bool {
        launch_preprocess_fwd(s, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, geom.clamped,
                              cov3D_precomp, colors_precomp, render_geo ? all_map : nullptr, viewmatrix, projmatrix,
                              cam_pos, width, height, tan_fovx, tan_fovy, focal_x, focal_y, radii, geom.rec, geom.rgb,
                              gx, gy, tile_count, antialiasing, cull, geom.grad_acc, img.tile_count,
                              tile_count ? 0 : clear_bytes / sizeof(uint32_t));
        return check_launch("preprocess_fwd", debug, s);
    };
    const bool tag = list_tags_fit(P);
    uint32_t* const nonunit = img.work + NONUNIT_WORD;
    auto render = [&](const uint32_t* point_list) -> bool {
        launch_render_fwd(s, render_geo != 0, tiles, img.ranges, point_list, width, height, gx, geom.rec, img.final_T,
                          img.n_contrib, background, out_color, out_invdepth, out_all_map, false, tag);
        return check_launch("render_fwd", debug, s);
    };

    // ---- path B: single-pass bucket binning (default once a previous forward has told us how long tile lists get).
    // Every tile owns a fixed-capacity bucket, so neither the per-tile count pass, nor the scan, nor num_rendered is
    // needed before the compositor can be queued: the whole forward is enqueued back to back and the host only waits
    // for the 16-byte readback (num_rendered is part of the reference's API) while the compositor is already running.
    // A tile that outgrows its bucket raises the overflow flag and the call falls through to the exact path below.
    bool preprocessed = false;
    const BinHints hints = hints_load(P, width, height);
    const int64_t max_hint = hints.max;
    if (cull && !debug && max_hint > 0) {
        const uint64_t cap = (((uint64_t)max_hint * 5 / 4 + 64) + 63) & ~63ull;
        if (cap <= bucket_cap_limit() && cap * (uint64_t)tiles < (1ull << 31)) {
            if (!preprocess(nullptr)) return CGS_ERR_HIP;
            preprocessed = true;
            char* bchunk = (char*)binning_alloc(binning_user, cgs_binning_bytes((int64_t)(cap * tiles)));
            if (!bchunk) {
                set_error("cgs_rasterize_forward: binning allocation callback returned NULL");
                return CGS_ERR_ALLOC;
            }
            BinState bin = bin_from_chunk(bchunk, (size_t)(cap * tiles));
            const bool defer_big = hints.big > 0;
            launch_scatter_bucket(s, P, radii, geom.rec, gx, gy, img.tile_count, bin.keys, (uint32_t)cap, cull, img.total + 3,
                                  defer_big ? img.tile_cursor : nullptr, (uint32_t)tiles, nonunit, scatter_spw(hints));   // (cursors: unused here)
            // num_rendered (part of the reference's return value) and the longest tile list are known once the SCATTER is done:
            // one small launch reduces the tile histogram, 16 bytes travel to the host, and the forward compositor -- which
            // sorts every tile's bucket itself, like the sync-free forward's -- is queued behind them before the host waits.
            // (Round 4 ran a separate sort launch here so that the readback could follow it: 30 us of kernel per view.)
            uint32_t* const stat = img.work + 4;   // four words of the (cleared) work block
            hipLaunchKernelGGL(k_count_stats, dim3(STAT_BLOCKS), dim3(256), 0, s, img.tile_count, tiles, radii, P, img.total + 3, stat,
                               img.work + VIS_COUNT_WORD);
            {
                hipError_t e = hipMemcpyAsync(h_tot, stat, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
                if (e == hipSuccess) e = hipEventRecord(ev, s);
                if (e != hipSuccess) {
                    set_error("reading num_rendered failed: %s", hipGetErrorString(e));
                    return CGS_ERR_HIP;
                }
            }
            if (fuse_sort() && render_fwd_can_sort((uint32_t)cap)) {
                launch_render_fwd_sorting(s, render_geo != 0, tiles, img.tile_count, bin.keys, (uint32_t)cap, img.ranges, img.total,
                                          bin.point_list, width, height, gx, geom.rec, img.final_T, img.n_contrib, background,
                                          out_color, out_invdepth, out_all_map, false, tag);
                if (!check_launch("render_fwd", debug, s)) return CGS_ERR_HIP;
            } else {
                launch_tile_sort_bucket(s, tiles, img.tile_count, img.ranges, img.total, bin.keys, bin.point_list, (uint32_t)cap);
                if (!render(bin.point_list)) return CGS_ERR_HIP;
            }
            if (!wait_totals()) return CGS_ERR_HIP;
            const uint32_t longest = h_tot[1];
            const int64_t Rb = (int64_t)h_tot[0];   // (= the sum of the list lengths whenever no bucket overflowed)
            g_last_visible = (int64_t)h_tot[2];
            hints_update(P, width, height, (uint64_t)longest <= cap ? Rb : -1, longest, (int64_t)h_tot[3]);
            if ((uint64_t)longest <= cap) {
                g_last_stats[0] = Rb; g_last_stats[1] = (int64_t)longest; g_last_stats[2] = 1;
                return Rb;
            }
            // overflow: the image just rendered is incomplete -- redo the binning with exact sizes
        }
    }

    // ---- path A: exact layout (count -> scan -> scatter -> sort); bit-identical to the reference's binning when
    // tile culling is off.  Used for the first forward, in debug mode, with culling off and after a bucket overflow.
    if (zero_async(img.tile_count, clear_bytes, s) != hipSuccess) {
        set_error("zero_async(tile histogram) failed");
        return CGS_ERR_HIP;
    }
    if (!preprocess(img.tile_count)) return CGS_ERR_HIP;
    (void)preprocessed;
    launch_scan_tiles(s, tiles, img.tile_count, img.ranges, img.total);
    if (!check_launch("scan_tiles", debug, s)) return CGS_ERR_HIP;

    // num_rendered has to reach the host (it sizes the binning buffer and is part of the reference's API).  Instead of
    // idling the GPU during that round trip (the reference blocks on a 4-byte cudaMemcpy, rasterizer_impl.cu:287), the
    // binning kernels are launched SPECULATIVELY into a buffer sized from the previous call's R (+25 %) while an event
    // marks the readback; the host then waits on the event only.  If the guess was too small (scene changed a lot)
    // the kernels skipped every tile that would not fit and are re-run on an exact-size buffer.
    if (!read_totals()) return CGS_ERR_HIP;
    const int64_t hint = hints.R;
    int64_t cap = 0;
    char* bchunk = nullptr;
    BinState bin{};
    if (hint > 0 && !debug) {
        cap = hint + hint / 4 + 4096;
        bchunk = (char*)binning_alloc(binning_user, cgs_binning_bytes(cap));
        if (!bchunk) {
            set_error("cgs_rasterize_forward: binning allocation callback returned NULL");
            return CGS_ERR_ALLOC;
        }
        bin = bin_from_chunk(bchunk, (size_t)cap);
        launch_scatter(s, P, radii, geom.rec, gx, gy, img.ranges, img.tile_cursor, bin.keys, (uint32_t)cap, cull, nonunit);
        launch_tile_sort_small(s, tiles, img.ranges, bin.keys, bin.point_list, (uint32_t)cap);
    }
    if (!wait_totals()) return CGS_ERR_HIP;
    const int64_t R = (int64_t)h_tot[0];
    const uint32_t max_count = h_tot[1];
    hints_update(P, width, height, R, max_count, -1);
    if (!bchunk || R > cap) {  // first call, debug mode, or the speculative buffer was too small: exact-size (re)run
        if (cap > 0 && zero_async(img.tile_cursor, (size_t)tiles * sizeof(uint32_t), s) != hipSuccess) {
            set_error("zero_async(tile cursors) failed");
            return CGS_ERR_HIP;
        }
        cap = R;
        bchunk = (char*)binning_alloc(binning_user, cgs_binning_bytes(R));
        if (!bchunk) {
            set_error("cgs_rasterize_forward: binning allocation callback returned NULL");
            return CGS_ERR_ALLOC;
        }
        bin = bin_from_chunk(bchunk, (size_t)(R > 0 ? R : 1));
        if (R > 0) {
            launch_scatter(s, P, radii, geom.rec, gx, gy, img.ranges, img.tile_cursor, bin.keys, (uint32_t)cap, cull, nonunit);
            if (!check_launch("scatter", debug, s)) return CGS_ERR_HIP;
            launch_tile_sort_small(s, tiles, img.ranges, bin.keys, bin.point_list, (uint32_t)cap);
            if (!check_launch("tile_sort", debug, s)) return CGS_ERR_HIP;
        }
    }
    if (R > 0) {
        launch_tile_sort_big(s, tiles, img.ranges, bin.keys, bin.point_list, max_count);  // no-op unless a list > 1024
        if (!check_launch("tile_sort", debug, s)) return CGS_ERR_HIP;
    }
    if (!render(bin.point_list)) return CGS_ERR_HIP;
    g_last_stats[0] = R; g_last_stats[1] = (int64_t)max_count; g_last_stats[2] = 0;
    return R;
}

// Sync-free forward for stream-ordered / hipGraph-captured pipelines: caller-owned buffers, caller-chosen bucket
// capacity, single-pass bucket binning only, nothing read back.  Status lives in the image buffer (see header).
int cgs_rasterize_forward_static(void* geometry_buffer, void* binning_buffer, size_t binning_bytes, void* image_buffer,
                                 uint32_t bucket_capacity, int P, int D, int M, const float* background, int width,
                                 int height, const float* means3D, const float* shs, const float* colors_precomp,
                                 const float* opacities, const float* scales, float scale_modifier,
                                 const float* rotations, const float* cov3D_precomp, const float* all_map,
                                 const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                                 float tan_fovy, float* out_color, float* out_invdepth, float* out_all_map,
                                 int antialiasing, int render_geo, int* radii, void* stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (P <= 0 || width <= 0 || height <= 0 || !out_color || !out_invdepth || !out_all_map || !background ||
        !viewmatrix || !projmatrix || !geometry_buffer || !binning_buffer || !image_buffer || bucket_capacity == 0) {
        set_error("cgs_rasterize_forward_static: invalid argument (P=%d W=%d H=%d, NULL pointer or zero capacity)", P, width, height);
        return CGS_ERR_INVALID_ARGUMENT;
    }
    if (!means3D || !opacities || !radii || (!shs == !colors_precomp) ||
        (cov3D_precomp ? (scales || rotations) : (!scales || !rotations)) || (render_geo && !all_map) ||
        (shs && (!cam_pos || M <= 0)) || !aligned16(rotations) || !aligned16(all_map)) {
        set_error("cgs_rasterize_forward_static: inconsistent or misaligned inputs");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE;
    const int tiles = gx * gy;
    const uint64_t cap = bucket_capacity;
    if (cap > bucket_cap_limit() || cap * (uint64_t)tiles >= (1ull << 31) ||
        binning_bytes < cgs_binning_bytes((int64_t)(cap * tiles))) {
        set_error("cgs_rasterize_forward_static: bucket capacity %u needs %zu binning bytes (got %zu; limit %u per tile)",
                  bucket_capacity, cgs_binning_bytes((int64_t)(cap * tiles)), binning_bytes, bucket_cap_limit());
        return CGS_ERR_INVALID_ARGUMENT;
    }
    const size_t npix = (size_t)width * height;
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    char* gchunk = (char*)geometry_buffer;
    char* bchunk = (char*)binning_buffer;
    char* ichunk = (char*)image_buffer;
    GeomState geom = geom_from_chunk(gchunk, (size_t)P);
    BinState bin = bin_from_chunk(bchunk, (size_t)(cap * tiles));
    ImageState img = image_from_chunk(ichunk, npix, (size_t)tiles);
    const size_t clear_bytes = (size_t)((char*)(img.total + TOTAL_WORDS) - (char*)img.tile_count);
    launch_preprocess_fwd(s, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, geom.clamped,
                          cov3D_precomp, colors_precomp, render_geo ? all_map : nullptr, viewmatrix, projmatrix, cam_pos,
                          width, height, tan_fovx, tan_fovy, focal_x, focal_y, radii, geom.rec, geom.rgb, gx, gy, nullptr,
                          antialiasing, 1, geom.grad_acc, img.tile_count, clear_bytes / sizeof(uint32_t));
    const BinHints hints = hints_load(P, width, height);   // (from the caller's probing forwards of this shape)
    const bool defer_big = hints.big > 0;
    const bool tag = list_tags_fit(P);
    launch_scatter_bucket(s, P, radii, geom.rec, gx, gy, img.tile_count, bin.keys, (uint32_t)cap, 1, img.total + 3,
                          defer_big ? img.tile_cursor : nullptr, (uint32_t)tiles, img.work + NONUNIT_WORD, scatter_spw(hints));
    if (render_fwd_can_sort((uint32_t)cap) && fuse_sort()) {
        launch_render_fwd_sorting(s, render_geo != 0, tiles, img.tile_count, bin.keys, (uint32_t)cap, img.ranges, img.total,
                                  bin.point_list, width, height, gx, geom.rec, img.final_T, img.n_contrib, background,
                                  out_color, out_invdepth, out_all_map, false, tag);
    } else {
        launch_tile_sort_bucket(s, tiles, img.tile_count, img.ranges, img.total, bin.keys, bin.point_list, (uint32_t)cap);
        launch_render_fwd(s, render_geo != 0, tiles, img.ranges, bin.point_list, width, height, gx, geom.rec, img.final_T,
                          img.n_contrib, background, out_color, out_invdepth, out_all_map, false, tag);
    }
    if (!check_launch("rasterize_forward_static", false, s)) return CGS_ERR_HIP;
    return CGS_OK;
}

size_t cgs_image_status_offset(int width, int height) {
    char* c = nullptr;
    const size_t tiles = (size_t)((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
    ImageState img = image_from_chunk(c, (size_t)width * height, tiles);
    return (size_t)((char*)img.total - (char*)nullptr);
}
int cgs_status_words(void) { return TOTAL_WORDS; }
uint32_t cgs_bucket_capacity_limit(void) { return bucket_cap_limit(); }

void cgs_last_forward_stats(int64_t* num_rendered, int64_t* longest_tile_list, int* binning_path) {
    if (num_rendered) *num_rendered = g_last_stats[0];
    if (longest_tile_list) *longest_tile_list = g_last_stats[1];
    if (binning_path) *binning_path = (int)g_last_stats[2];
}

int cgs_rasterize_backward(int P, int D, int M, int64_t R, const float* background, int width, int height,
                           const float* means3D, const float* shs, const float* colors_precomp, const float* all_map,
                           const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                           const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                           const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                           void* geometry_buffer, const void* binning_buffer, const void* image_buffer,
                           const float* dL_dout_color, const float* dL_dout_invdepth, const float* dL_dout_all_map,
                           float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                           float* dL_dinvdepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                           float* dL_drot, float* dL_dall_map, int antialiasing, int render_geo, int debug,
                           void* stream_) {
    (void)colors_precomp;
    (void)all_map;
    hipStream_t s = (hipStream_t)stream_;
    const bool general_only = (debug & CGS_OPT_GENERAL_BACKWARD) != 0;   // per call (include/curvegs.h)
    debug &= CGS_OPT_DEBUG;
    if (P == 0) return CGS_OK;
    if (P < 0 || width <= 0 || height <= 0 || !geometry_buffer || !binning_buffer || !image_buffer || !radii ||
        !dL_dout_color || !dL_dmean2D || !dL_dopacity || (shs && !dL_dcolor) || !dL_dmean3D || !dL_dcov3D ||
        !dL_dall_map || (!dL_dout_invdepth != !dL_dinvdepth) || (scales && (!dL_dscale || !dL_drot)) ||
        (shs && !dL_dsh)) {
        set_error("cgs_rasterize_backward: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    if (!dL_dcolor && ((render_geo && dL_dout_all_map) || dL_dout_invdepth)) {
        set_error("cgs_rasterize_backward: dL_dcolor may only be NULL when no depth / all_map gradients flow in");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    if (!aligned16(rotations) || !aligned16(dL_dconic) || !aligned16(dL_drot) || !aligned16(dL_dall_map)) {
        set_error("cgs_rasterize_backward: rotations/dL_dconic/dL_drot must be 16-byte aligned");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE;
    const int tiles = gx * gy;
    const size_t npix = (size_t)width * height;
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    char* gchunk = (char*)geometry_buffer;
    char* bchunk = (char*)binning_buffer;
    char* ichunk = (char*)image_buffer;
    GeomState geom = geom_from_chunk(gchunk, (size_t)P);
    BinState bin = bin_from_chunk(bchunk, (size_t)(R > 0 ? R : 1));
    ImageState img = image_from_chunk(ichunk, npix, (size_t)tiles);

    // geom.grad_acc is zero here: the forward's preprocess kernel cleared it and k_preprocess_bwd clears it after use
    if (R > 0) {
        const bool geo = render_geo && dL_dout_all_map;
        const bool invd = dL_dout_invdepth != nullptr, colg = dL_dcolor != nullptr;
        const bool tagged = list_tags_fit(P);   // (the forward's own condition: both sides derive it from P)
        const uint32_t id_mask = tagged ? LIST_ID_MASK : 0xffffffffu;
        // Training instance (only dL/dcolour upstream, the colours themselves need no gradient): the reference's own call
        // (gaussian_renderer/__init__.py:96-129) passes all-ones colours and all_map[:, 3] == 1, for which dL/dalpha has the
        // closed form of render_unit_bwd.hip.  The ABI receives tensors and cannot know that on the host; the scatter of the
        // forward raised img.work[NONUNIT_WORD] if any visible splat deviates, and the two kernels test that word on entry.
        const uint32_t* gate = nullptr;
        if (tagged && !geo && !invd && !colg && !general_only) {
            gate = img.work + NONUNIT_WORD;
            launch_render_bwd_unit(s, tiles, img.ranges, bin.point_list, width, height, gx, background, geom.rec, img.final_T,
                                   img.n_contrib, dL_dout_color, geom.grad_acc, ACC_STRIDE, gate);
        }
        launch_render_bwd(s, geo, invd, colg, tiles, img.ranges, bin.point_list, width, height, gx, background, geom.rec,
                          img.final_T, img.n_contrib, dL_dout_color, dL_dout_invdepth, dL_dout_all_map, geom.grad_acc,
                          ACC_STRIDE, id_mask, gate);
        if (!check_launch("render_bwd", debug, s)) return CGS_ERR_HIP;
    }
    launch_preprocess_bwd(s, P, D, M, means3D, radii, shs, geom.clamped, opacities, scales, rotations, scale_modifier,
                          cov3D_precomp, viewmatrix, projmatrix, cam_pos, focal_x, focal_y, tan_fovx, tan_fovy, width,
                          height, geom.rec, geom.grad_acc, dL_dmean2D, dL_dconic, dL_dinvdepth, dL_dopacity, dL_dmean3D,
                          dL_dcolor, dL_dall_map, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, antialiasing);
    if (!check_launch("preprocess_bwd", debug, s)) return CGS_ERR_HIP;
    return CGS_OK;
}

int cgs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream_) {
    (void)projmatrix;
    if (P == 0) return CGS_OK;
    if (P < 0 || !means3D || !viewmatrix || !present) {
        set_error("cgs_mark_visible: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    launch_mark_visible((hipStream_t)stream_, P, means3D, viewmatrix, present);
    if (!check_launch("mark_visible", false, (hipStream_t)stream_)) return CGS_ERR_HIP;
    return CGS_OK;
}


int cgs_sample_curves_forward(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier,
                              const float* coef, float eps, double* norms, float* xyz, float* rotation,
                              float* scaling, void* stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (B == 0) return CGS_OK;
    if (B < 0 || m <= 0 || m > 32 || !curve_points || !width || !coef || !norms || !xyz || !rotation || !scaling ||
        !aligned16(curve_points) || !aligned16(rotation) || !aligned16(coef)) {
        set_error("cgs_sample_curves_forward: invalid argument (NULL or misaligned pointer, B=%d m=%d)", B, m);
        return CGS_ERR_INVALID_ARGUMENT;
    }
    // (no zero fill of norms: k_sample_f12 writes the forward sums and clears the backward's, csrc/curve_math.h)
    launch_sample_forward(s, B, m, curve_points, width, is_bezier, coef, eps, norms, xyz, rotation, scaling);
    if (!check_launch("sample_curves_forward", false, s)) return CGS_ERR_HIP;
    return CGS_OK;
}

// ---------------------------------------------------------------------------------------------- fused per-view path
// One view of the training configuration, curve parameters in, image out (and back): the per-splat chains are fused
// (view.hip), the rasterizer is the sync-free single-pass bucket pipeline of cgs_rasterize_forward_static.
// Status readback of the checked view forward.  Every cgs_view_forward_begin takes a slot of a small pool (pinned 16-byte
// buffer + event, created on first use) and returns its handle; cgs_view_forward_wait(handle) blocks on that slot's event and
// releases it.  Forwards begun by different threads, on different devices or streams, or for different models are
// independent; a caller that drops a handle (an exception between begin and wait) leaks nothing but the slot until
// cgs_view_forward_abandon(handle).
// A slot belongs to the device its event was created on (hipEventRecord rejects an event / stream pair of different devices):
// a forward only takes slots of the current device.
struct ViewStat { uint32_t* h = nullptr; hipEvent_t ev = nullptr; int dev = -1, P = 0, W = 0, H = 0; uint64_t cap = 0; bool busy = false; };
constexpr int VIEW_SLOTS = 64;
static ViewStat g_view_slots[VIEW_SLOTS];
static std::mutex g_view_mu;
static int view_slot_acquire() {   // -> slot index, or a negative status
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(g_view_mu);
    int fresh = -1;
    for (int i = 0; i < VIEW_SLOTS; i++) {
        ViewStat& v = g_view_slots[i];
        if (v.busy) continue;
        if (!v.h) { if (fresh < 0) fresh = i; continue; }   // never used: taken only when no idle slot of this device exists
        if (v.dev != dev) continue;
        v.busy = true;
        return i;
    }
    if (fresh >= 0) {
        ViewStat& v = g_view_slots[fresh];
        if (hipHostMalloc((void**)&v.h, 4 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&v.ev, hipEventDisableTiming) != hipSuccess) {
            set_error("pinned readback buffer / event creation failed");
            if (v.h) (void)hipHostFree(v.h);
            v.h = nullptr;
            return CGS_ERR_HIP;
        }
        v.dev = dev;
        v.busy = true;
        return fresh;
    }
    set_error("cgs_view_forward_begin: no free status slot for device %d (%d slots; every begin needs its cgs_view_forward_wait)", dev,
              VIEW_SLOTS);
    return CGS_ERR_INVALID_ARGUMENT;
}
static bool view_slot_busy(int i) {
    std::lock_guard<std::mutex> lk(g_view_mu);
    return g_view_slots[i].busy;
}
static void view_slot_release(int i) {
    std::lock_guard<std::mutex> lk(g_view_mu);
    g_view_slots[i].busy = false;
}
static int64_t view_forward_wait(int handle, int64_t* n_visible) {
    if (handle < 0 || handle >= VIEW_SLOTS || !view_slot_busy(handle)) {
        set_error("cgs_view_forward_wait: handle %d is not an outstanding checked forward", handle);
        return CGS_ERR_INVALID_ARGUMENT;
    }
    ViewStat& v = g_view_slots[handle];
    const hipError_t e = hipEventSynchronize(v.ev);
    if (e != hipSuccess) {
        view_slot_release(handle);
        set_error("cgs_view_forward_checked: status readback failed: %s", hipGetErrorString(e));
        return CGS_ERR_HIP;
    }
    const uint32_t longest = v.h[1];
    hints_update(v.P, v.W, v.H, (uint64_t)longest <= v.cap ? (int64_t)v.h[0] : -1, longest, (int64_t)v.h[3], v.dev);
    g_last_stats[0] = (int64_t)v.h[0]; g_last_stats[1] = (int64_t)longest; g_last_stats[2] = 1;
    g_last_visible = (int64_t)v.h[2];
    if (n_visible) *n_visible = (int64_t)v.h[2];
    view_slot_release(handle);
    return (int64_t)longest;
}

static int64_t view_forward_impl(int mode, int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier, const float* coef,
                     float eps, double* norms, const float* opacity_logit, const float* mask_logit, float mask_thr,
                     const float* colors_precomp, void* geometry_buffer, void* binning_buffer, size_t binning_bytes,
                     void* image_buffer, uint32_t bucket_capacity, const float* background, int width_px, int height_px,
                     const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                     float* out_color, float* out_invdepth, float* out_all_map, int* radii, float* xyz, float* rotation,
                     float* scaling, void* stream_, float* out_color_clamped = nullptr, float* out_rend_dir = nullptr) {
    hipStream_t s = (hipStream_t)stream_;
    const int P = B * m;
    if (out_rend_dir && !out_all_map) {
        set_error("cgs_view_forward: the direction map needs the all_map output");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    if (B <= 0 || m <= 0 || m > 32 || (long long)B * m >= (1ll << 28) || width_px <= 0 || height_px <= 0 || !curve_points ||
        !width || !coef || !norms ||
        !opacity_logit || !geometry_buffer || !binning_buffer || !image_buffer || bucket_capacity == 0 || !background ||
        !viewmatrix || !projmatrix || !cam_pos || !out_color || (!out_invdepth != !out_all_map) ||
        (!out_all_map && colors_precomp) || !radii || (xyz && (!rotation || !scaling)) || !aligned16(curve_points) || !aligned16(coef) || !aligned16(rotation)) {
        set_error("cgs_view_forward: invalid argument (B=%d m=%d W=%d H=%d, NULL / misaligned pointer or zero capacity)", B, m,
                  width_px, height_px);
        return CGS_ERR_INVALID_ARGUMENT;
    }
    const int gx = (width_px + TILE - 1) / TILE, gy = (height_px + TILE - 1) / TILE;
    const int tiles = gx * gy;
    const uint64_t cap = bucket_capacity;
    if (cap > bucket_cap_limit() || cap * (uint64_t)tiles >= (1ull << 31) ||
        binning_bytes < cgs_binning_bytes((int64_t)(cap * tiles))) {
        set_error("cgs_view_forward: bucket capacity %u needs %zu binning bytes (got %zu; limit %u per tile)", bucket_capacity,
                  cgs_binning_bytes((int64_t)(cap * tiles)), binning_bytes, bucket_cap_limit());
        return CGS_ERR_INVALID_ARGUMENT;
    }
    const size_t npix = (size_t)width_px * height_px;
    const float focal_y = height_px / (2.0f * tan_fovy);
    const float focal_x = width_px / (2.0f * tan_fovx);
    char* gchunk = (char*)geometry_buffer;
    char* bchunk = (char*)binning_buffer;
    char* ichunk = (char*)image_buffer;
    GeomState geom = geom_from_chunk(gchunk, (size_t)P);
    BinState bin = bin_from_chunk(bchunk, (size_t)(cap * tiles));
    ImageState img = image_from_chunk(ichunk, npix, (size_t)tiles);
    const size_t clear_bytes = (size_t)((char*)(img.total + TOTAL_WORDS) - (char*)img.tile_count);
    const bool shared = (mode & VIEW_MODE_SHARED) != 0;
    mode &= VIEW_MODE_MASK;
    // the norm pass writes the three forward sums and clears the backward's two: no zero-fill launch
    if (!shared) launch_sample_norms(s, B, m, curve_points, is_bezier, coef, norms);
    launch_view_forward(s, B, m, curve_points, width, is_bezier, coef, eps, norms, opacity_logit, mask_logit, mask_thr,
                        colors_precomp, cam_pos, viewmatrix, projmatrix, tan_fovx, tan_fovy, focal_x, focal_y, width_px,
                        height_px, gx, gy, xyz, rotation, scaling, radii, geom.rec, geom.grad_acc, img.tile_count,
                        clear_bytes / sizeof(uint32_t));
    // k_view_fwd writes unit colours (no colors_precomp) and all_map[3] = 1 itself: the compositor derives both sums from T
    const bool unit = colors_precomp == nullptr;
    const bool aux = out_all_map != nullptr;   // image-only forward (unit colours required) when the caller passes neither map
    const BinHints hints = hints_load(P, width_px, height_px);
    const bool defer_big = hints.big > 0;
    launch_scatter_bucket(s, P, radii, geom.rec, gx, gy, img.tile_count, bin.keys, (uint32_t)cap, 1, img.total + 3,
                          defer_big ? img.tile_cursor : nullptr, (uint32_t)tiles, nullptr, scatter_spw(hints));
    // checked: the longest tile list (and the instance count, oversized-rect count) travel to the host right behind the
    // scatter; the compositor is queued before the host waits, so the wait overlaps it
    const bool checked = mode != 0;
    int slot = -1;
    if (checked) {
        slot = view_slot_acquire();
        if (slot < 0) return slot;
        ViewStat& vs = g_view_slots[slot];
        // four words of the (cleared) work block
        uint32_t* const stat = img.work + 4;
        hipLaunchKernelGGL(k_count_stats, dim3(STAT_BLOCKS), dim3(256), 0, s, img.tile_count, tiles, radii, P, img.total + 3, stat,
                               img.work + VIS_COUNT_WORD);
        hipError_t e = hipMemcpyAsync(vs.h, stat, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipEventRecord(vs.ev, s);
        if (e != hipSuccess) {
            view_slot_release(slot);
            set_error("cgs_view_forward_checked: status readback failed: %s", hipGetErrorString(e));
            return CGS_ERR_HIP;
        }
        vs.P = P; vs.W = width_px; vs.H = height_px; vs.cap = cap;
    }
    if (render_fwd_can_sort((uint32_t)cap) && fuse_sort()) {
        launch_render_fwd_sorting(s, aux, tiles, img.tile_count, bin.keys, (uint32_t)cap, img.ranges, img.total,
                                  bin.point_list, width_px, height_px, gx, geom.rec, img.final_T, img.n_contrib, background,
                                  out_color, out_invdepth, out_all_map, unit, unit, out_color_clamped, out_rend_dir, viewmatrix);
    } else {
        launch_tile_sort_bucket(s, tiles, img.tile_count, img.ranges, img.total, bin.keys, bin.point_list, (uint32_t)cap);
        launch_render_fwd(s, aux, tiles, img.ranges, bin.point_list, width_px, height_px, gx, geom.rec, img.final_T,
                          img.n_contrib, background, out_color, out_invdepth, out_all_map, unit);
        if (out_color_clamped || out_rend_dir)   // (long-list buckets: the non-sorting forward has no epilogue of its own)
            hipLaunchKernelGGL(k_render_epilogue, dim3((unsigned)std::min<size_t>((npix + 255) / 256, 4096)), dim3(256), 0, s, npix,
                               out_color, out_all_map, viewmatrix, 1, out_color_clamped, out_rend_dir);
    }
    if (!check_launch("view_forward", false, s)) {
        if (checked) view_slot_release(slot);
        return CGS_ERR_HIP;
    }
    if (checked) return mode == 1 ? view_forward_wait(slot, nullptr) : (int64_t)slot;
    return CGS_OK;
}

int cgs_view_forward(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier, const float* coef,
                     float eps, double* norms, const float* opacity_logit, const float* mask_logit, float mask_thr,
                     const float* colors_precomp, void* geometry_buffer, void* binning_buffer, size_t binning_bytes,
                     void* image_buffer, uint32_t bucket_capacity, const float* background, int width_px, int height_px,
                     const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                     float* out_color, float* out_invdepth, float* out_all_map, int* radii, float* xyz, float* rotation,
                     float* scaling, void* stream_) {
    return (int)view_forward_impl(0, B, m, curve_points, width, is_bezier, coef, eps, norms, opacity_logit, mask_logit,
                                  mask_thr, colors_precomp, geometry_buffer, binning_buffer, binning_bytes, image_buffer,
                                  bucket_capacity, background, width_px, height_px, viewmatrix, projmatrix, cam_pos, tan_fovx,
                                  tan_fovy, out_color, out_invdepth, out_all_map, radii, xyz, rotation, scaling, stream_);
}
int64_t cgs_view_forward_checked(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier,
                                 const float* coef, float eps, double* norms, const float* opacity_logit,
                                 const float* mask_logit, float mask_thr, const float* colors_precomp, void* geometry_buffer,
                                 void* binning_buffer, size_t binning_bytes, void* image_buffer, uint32_t bucket_capacity,
                                 const float* background, int width_px, int height_px, const float* viewmatrix,
                                 const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                                 float* out_color, float* out_invdepth, float* out_all_map, int* radii, float* xyz,
                                 float* rotation, float* scaling, void* stream_) {
    return view_forward_impl(1, B, m, curve_points, width, is_bezier, coef, eps, norms, opacity_logit, mask_logit, mask_thr,
                             colors_precomp, geometry_buffer, binning_buffer, binning_bytes, image_buffer, bucket_capacity,
                             background, width_px, height_px, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, out_color,
                             out_invdepth, out_all_map, radii, xyz, rotation, scaling, stream_);
}
int cgs_view_forward_begin(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier,
                                 const float* coef, float eps, double* norms, const float* opacity_logit,
                                 const float* mask_logit, float mask_thr, const float* colors_precomp, void* geometry_buffer,
                                 void* binning_buffer, size_t binning_bytes, void* image_buffer, uint32_t bucket_capacity,
                                 const float* background, int width_px, int height_px, const float* viewmatrix,
                                 const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                                 float* out_color, float* out_invdepth, float* out_all_map, int* radii, float* xyz,
                                 float* rotation, float* scaling, void* stream_) {
    return (int)view_forward_impl(2, B, m, curve_points, width, is_bezier, coef, eps, norms, opacity_logit, mask_logit, mask_thr,
                             colors_precomp, geometry_buffer, binning_buffer, binning_bytes, image_buffer, bucket_capacity,
                             background, width_px, height_px, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, out_color,
                             out_invdepth, out_all_map, radii, xyz, rotation, scaling, stream_);
}
// ... with render()'s epilogue written by the compositor itself (out_color_clamped [H*W], out_rend_dir [3,H*W]; either may be NULL)
int cgs_view_forward_render(int checked, int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier,
                            const float* coef, float eps, double* norms, const float* opacity_logit, const float* mask_logit,
                            float mask_thr, void* geometry_buffer, void* binning_buffer, size_t binning_bytes, void* image_buffer,
                            uint32_t bucket_capacity, const float* background, int width_px, int height_px, const float* viewmatrix,
                            const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, float* out_color,
                            float* out_invdepth, float* out_all_map, int* radii, float* out_color_clamped, float* out_rend_dir,
                            void* stream_) {
    return (int)view_forward_impl(checked ? 2 : 0, B, m, curve_points, width, is_bezier, coef, eps, norms, opacity_logit, mask_logit,
                                  mask_thr, nullptr, geometry_buffer, binning_buffer, binning_bytes, image_buffer, bucket_capacity,
                                  background, width_px, height_px, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, out_color,
                                  out_invdepth, out_all_map, radii, nullptr, nullptr, nullptr, stream_, out_color_clamped,
                                  out_rend_dir);
}
int64_t cgs_view_forward_wait(int handle, int64_t* n_visible) { return view_forward_wait(handle, n_visible); }
void cgs_view_forward_abandon(int handle) {
    if (handle < 0 || handle >= VIEW_SLOTS || !view_slot_busy(handle)) return;
    // the slot's readback may still be in flight: let it land before the pinned words can be handed to another forward (a
    // later forward on ANOTHER stream would otherwise race with it)
    (void)hipEventSynchronize(g_view_slots[handle].ev);
    view_slot_release(handle);
}
int cgs_view_forward_shared(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier, const float* coef,
                     float eps, double* norms, const float* opacity_logit, const float* mask_logit, float mask_thr,
                     const float* colors_precomp, void* geometry_buffer, void* binning_buffer, size_t binning_bytes,
                     void* image_buffer, uint32_t bucket_capacity, const float* background, int width_px, int height_px,
                     const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                     float* out_color, float* out_invdepth, float* out_all_map, int* radii, float* xyz, float* rotation,
                     float* scaling, void* stream_) {
    return (int)view_forward_impl(VIEW_MODE_SHARED, B, m, curve_points, width, is_bezier, coef, eps, norms, opacity_logit, mask_logit,
                                  mask_thr, colors_precomp, geometry_buffer, binning_buffer, binning_bytes, image_buffer,
                                  bucket_capacity, background, width_px, height_px, viewmatrix, projmatrix, cam_pos, tan_fovx,
                                  tan_fovy, out_color, out_invdepth, out_all_map, radii, xyz, rotation, scaling, stream_);
}
int64_t cgs_last_forward_visible(void) { return g_last_visible; }
int cgs_visible_indices(int P, const int* radii, const void* image_buffer, int width, int height, int64_t* out_indices, void* stream_) {
    if (P <= 0 || !radii || !image_buffer || width <= 0 || height <= 0 || !out_indices) {
        set_error("cgs_visible_indices: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    const size_t tiles = (size_t)((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
    char* ichunk = (char*)const_cast<void*>(image_buffer);
    ImageState img = image_from_chunk(ichunk, (size_t)width * height, tiles);
    hipLaunchKernelGGL(k_visible_compact, dim3(STAT_BLOCKS), dim3(256), 0, (hipStream_t)stream_, radii, P, img.work + VIS_COUNT_WORD,
                       (long long*)out_indices);
    if (!check_launch("visible_indices", false, (hipStream_t)stream_)) return CGS_ERR_HIP;
    return CGS_OK;
}
uint32_t cgs_bucket_capacity_hint(int P, int width, int height) {
    const int64_t mx = hints_load(P, width, height).max;
    if (mx <= 0) return 0u;
    const uint64_t cap = (((uint64_t)mx * 5 / 4 + 64) + 63) & ~63ull;
    return (uint32_t)std::min<uint64_t>(cap, bucket_cap_limit());
}

int cgs_view_norms_backward_range(int* first, int* count) {
    if (first) *first = sample_norm_fwd_words();
    if (count) *count = sample_norm_words() - sample_norm_fwd_words();
    return 384;
}
// 13 floats per curve are used (16 asked for: the size stays a multiple of 64 bytes); rounds 2-5: 15 per splat
size_t cgs_view_backward_scratch_floats(int B, int m) { (void)m; return (size_t)(B > 0 ? B : 0) * 16; }

}  // extern "C"
static int view_backward_impl(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier, const float* coef,
                      float eps, double* norms, const float* opacity_logit, const float* mask_logit, float mask_thr,
                      const float* colors_precomp, void* geometry_buffer, const void* binning_buffer, const void* image_buffer, const float* background,
                      int width_px, int height_px, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                      float tan_fovx, float tan_fovy, const int* radii, const float* dL_dout_color,
                      const float* dL_drotation_extra, float* dL_dmeans2D, float* dL_dcurve_points, float* dL_dwidth,
                      float* dL_dopacity_logit, float* dL_dmask_logit, float* scratch, int flags, void* stream_,
                      const float* clamp_raw) {
    hipStream_t s = (hipStream_t)stream_;
    if (clamp_raw && colors_precomp) {
        set_error("cgs_view_backward_render: the folded clamp mask is part of the unit-colour path (no colors_precomp)");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    const int P = B * m;
    if (B <= 0 || m <= 0 || m > 32 || width_px <= 0 || height_px <= 0 || !curve_points || !width || !coef || !norms ||
        !opacity_logit || !geometry_buffer || !binning_buffer || !image_buffer || !background || !viewmatrix || !projmatrix ||
        !cam_pos || !radii || !dL_dout_color || !dL_dmeans2D || !dL_dcurve_points || !dL_dwidth || !dL_dopacity_logit ||
        !scratch || (mask_logit && !dL_dmask_logit) || !aligned16(curve_points) || !aligned16(coef) ||
        !aligned16(dL_drotation_extra) || !aligned16(dL_dcurve_points)) {
        set_error("cgs_view_backward: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    const int gx = (width_px + TILE - 1) / TILE, gy = (height_px + TILE - 1) / TILE;
    const int tiles = gx * gy;
    const size_t npix = (size_t)width_px * height_px;
    const float focal_y = height_px / (2.0f * tan_fovy);
    const float focal_x = width_px / (2.0f * tan_fovx);
    char* gchunk = (char*)geometry_buffer;
    char* bchunk = (char*)binning_buffer;
    char* ichunk = (char*)image_buffer;
    GeomState geom = geom_from_chunk(gchunk, (size_t)P);
    BinState bin = bin_from_chunk(bchunk, 1);
    ImageState img = image_from_chunk(ichunk, npix, (size_t)tiles);
    // scratch: [B,13] per-curve partials of dL/d{curve_points, width} (k_view_bwd -> k_sample_bwd_close; curve_math.h,
    // sample_backward_tail).  Rounds 2-5 sent 15 floats per SPLAT through here.
    // training configuration: only dL/dcolour flows in, the colours themselves need no gradient; the forward wrote unit
    // colours unless it was given colors_precomp (same argument here): closed-form dL/dalpha, no recurrences (render.hip, UNIT)
    if (colors_precomp == nullptr)
        launch_render_bwd_unit(s, tiles, img.ranges, bin.point_list, width_px, height_px, gx, background, geom.rec, img.final_T,
                               img.n_contrib, dL_dout_color, geom.grad_acc, ACC_STRIDE_VIEW, nullptr, clamp_raw);
    else   // arbitrary colours: the general training instance (the forward did not tag the lists)
        launch_render_bwd(s, false, false, false, tiles, img.ranges, bin.point_list, width_px, height_px, gx, background, geom.rec,
                          img.final_T, img.n_contrib, dL_dout_color, nullptr, nullptr, geom.grad_acc, ACC_STRIDE_VIEW);
    launch_view_backward(s, B, m, curve_points, width, is_bezier, coef, eps, norms, opacity_logit, mask_logit, mask_thr,
                         cam_pos, viewmatrix, projmatrix, tan_fovx, tan_fovy, focal_x, focal_y, width_px, height_px, radii,
                         geom.rec, geom.grad_acc, dL_drotation_extra, dL_dmeans2D, dL_dopacity_logit, dL_dmask_logit, scratch,
                         ((flags & CGS_VIEW_ACCUMULATE) ? 1 : 0) | ((flags & CGS_VIEW_SHARED) ? 2 : 0));
    if (!(flags & CGS_VIEW_SHARED))
        launch_sample_backward_close(s, B, m, curve_points, width, is_bezier, coef, eps, norms, scratch, dL_dcurve_points,
                                     dL_dwidth, (flags & CGS_VIEW_ACCUMULATE) ? 1 : 0);
    if (!check_launch("view_backward", false, s)) return CGS_ERR_HIP;
    return CGS_OK;
}
extern "C" {
int cgs_view_backward(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier, const float* coef,
                      float eps, double* norms, const float* opacity_logit, const float* mask_logit, float mask_thr,
                      const float* colors_precomp, void* geometry_buffer, const void* binning_buffer, const void* image_buffer, const float* background,
                      int width_px, int height_px, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                      float tan_fovx, float tan_fovy, const int* radii, const float* dL_dout_color,
                      const float* dL_drotation_extra, float* dL_dmeans2D, float* dL_dcurve_points, float* dL_dwidth,
                      float* dL_dopacity_logit, float* dL_dmask_logit, float* scratch, int flags, void* stream_) {
    return view_backward_impl(B, m, curve_points, width, is_bezier, coef, eps, norms, opacity_logit, mask_logit, mask_thr, colors_precomp,
                              geometry_buffer, binning_buffer, image_buffer, background, width_px, height_px, viewmatrix, projmatrix,
                              cam_pos, tan_fovx, tan_fovy, radii, dL_dout_color, dL_drotation_extra, dL_dmeans2D, dL_dcurve_points,
                              dL_dwidth, dL_dopacity_logit, dL_dmask_logit, scratch, flags, stream_, nullptr);
}
// ... for an image that went through render()'s clamp: dL_dout_color is the gradient of the CLAMPED image and color_raw the
// forward's unclamped one; torch.clamp's gradient mask is applied where the compositor loads the pixel's upstream gradient
int cgs_view_backward_render(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier, const float* coef,
                      float eps, double* norms, const float* opacity_logit, const float* mask_logit, float mask_thr,
                      void* geometry_buffer, const void* binning_buffer, const void* image_buffer, const float* background,
                      int width_px, int height_px, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                      float tan_fovx, float tan_fovy, const int* radii, const float* dL_dout_color, const float* color_raw,
                      float* dL_dmeans2D, float* dL_dcurve_points, float* dL_dwidth,
                      float* dL_dopacity_logit, float* dL_dmask_logit, float* scratch, int flags, void* stream_) {
    return view_backward_impl(B, m, curve_points, width, is_bezier, coef, eps, norms, opacity_logit, mask_logit, mask_thr, nullptr,
                              geometry_buffer, binning_buffer, image_buffer, background, width_px, height_px, viewmatrix, projmatrix,
                              cam_pos, tan_fovx, tan_fovy, radii, dL_dout_color, nullptr, dL_dmeans2D, dL_dcurve_points,
                              dL_dwidth, dL_dopacity_logit, dL_dmask_logit, scratch, flags, stream_, color_raw);
}

int cgs_view_shared_begin(int B, int m, const float* curve_points, const uint8_t* is_bezier, const float* coef, double* norms,
                          float* scratch, void* stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (B <= 0 || m <= 0 || m > 32 || !curve_points || !coef || !norms || !scratch || !aligned16(curve_points) || !aligned16(coef)) {
        set_error("cgs_view_shared_begin: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    if (zero_async(scratch, cgs_view_backward_scratch_floats(B, m) * sizeof(float), s) != hipSuccess) {
        set_error("zero_async failed");
        return CGS_ERR_HIP;
    }
    launch_sample_norms(s, B, m, curve_points, is_bezier, coef, norms);
    if (!check_launch("view_shared_begin", false, s)) return CGS_ERR_HIP;
    return CGS_OK;
}
int cgs_view_shared_end(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier, const float* coef,
                        float eps, double* norms, float* scratch, float* dL_dcurve_points, float* dL_dwidth, int accumulate,
                        void* stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (B <= 0 || m <= 0 || m > 32 || !curve_points || !width || !coef || !norms || !scratch || !dL_dcurve_points || !dL_dwidth ||
        !aligned16(curve_points) || !aligned16(coef) || !aligned16(dL_dcurve_points)) {
        set_error("cgs_view_shared_end: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    launch_sample_backward_close(s, B, m, curve_points, width, is_bezier, coef, eps, norms, scratch, dL_dcurve_points, dL_dwidth,
                                 accumulate);
    if (!check_launch("view_shared_end", false, s)) return CGS_ERR_HIP;
    return CGS_OK;
}

int cgs_sample_curves_backward(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier,
                               const float* coef, float eps, double* norms, const float* dL_dxyz,
                               const float* dL_drotation, const float* dL_dscaling, float* dL_dcurve_points,
                               float* dL_dwidth, float* scratch, void* stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (B == 0) return CGS_OK;
    if (B < 0 || m <= 0 || m > 32 || !curve_points || !width || !coef || !norms || !dL_dcurve_points || !dL_dwidth ||
        (dL_drotation && !scratch) ||
        !aligned16(curve_points) || !aligned16(dL_drotation) || !aligned16(dL_dcurve_points) || !aligned16(coef)) {
        set_error("cgs_sample_curves_backward: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    // (a second backward over the same forward -- retain_graph -- must not see the first one's two sums)
    if (zero_async(norms + sample_norm_fwd_words(), (size_t)(sample_norm_words() - sample_norm_fwd_words()) * sizeof(double), s) != hipSuccess) {
        set_error("zero_async(norms) failed");
        return CGS_ERR_HIP;
    }
    launch_sample_backward(s, B, m, curve_points, width, is_bezier, coef, eps, norms, dL_dxyz, dL_drotation, dL_dscaling,
                           dL_dcurve_points, dL_dwidth, scratch);
    if (!check_launch("sample_curves_backward", false, s)) return CGS_ERR_HIP;
    return CGS_OK;
}

int cgs_splat_attrs_forward(int B, int m, const float* rotation_raw, const float* xyz, const float* opacity_logit,
                            const float* mask_logit, float mask_thr, const float* scaling, const float* campos,
                            const float* viewmatrix, float* rotation_n, float* opacity, float* scaling_out,
                            float* all_map, void* stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (B == 0) return CGS_OK;
    if (B < 0 || m <= 0 || !rotation_raw || !xyz || !opacity_logit || !campos || !viewmatrix || !rotation_n || !opacity ||
        !all_map || (scaling_out && !scaling) || !aligned16(rotation_raw) || !aligned16(rotation_n) || !aligned16(all_map)) {
        set_error("cgs_splat_attrs_forward: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    launch_attrs_forward(s, B, m, rotation_raw, xyz, opacity_logit, mask_logit, mask_thr, scaling, campos, viewmatrix,
                         rotation_n, opacity, scaling_out, all_map);
    if (!check_launch("splat_attrs_forward", false, s)) return CGS_ERR_HIP;
    return CGS_OK;
}

int cgs_splat_attrs_backward(int B, int m, const float* rotation_raw, const float* xyz, const float* opacity_logit,
                             const float* mask_logit, float mask_thr, const float* scaling, const float* campos,
                             const float* viewmatrix, const float* dL_drotation_n, const float* dL_dopacity,
                             const float* dL_dscaling_out, const float* dL_dall_map, float* dL_drotation_raw,
                             float* dL_dopacity_logit, float* dL_dmask_logit, float* dL_dscaling, void* stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (B == 0) return CGS_OK;
    if (B < 0 || m <= 0 || !rotation_raw || !xyz || !opacity_logit || !campos || !viewmatrix || !dL_drotation_raw ||
        !dL_dopacity_logit || !aligned16(rotation_raw) || !aligned16(dL_drotation_n) || !aligned16(dL_dall_map) ||
        !aligned16(dL_drotation_raw)) {
        set_error("cgs_splat_attrs_backward: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    launch_attrs_backward(s, B, m, rotation_raw, xyz, opacity_logit, mask_logit, mask_thr, scaling, campos, viewmatrix,
                          dL_drotation_n, dL_dopacity, dL_dscaling_out, dL_dall_map, dL_drotation_raw, dL_dopacity_logit,
                          dL_dmask_logit, dL_dscaling);
    if (!check_launch("splat_attrs_backward", false, s)) return CGS_ERR_HIP;
    return CGS_OK;
}


int cgs_ssim_forward(int batch, int channels, int height, int width, float C1, float C2, const float* img1,
                     const float* img2, float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                     void* stream_) {
    if (batch * channels == 0 || height == 0 || width == 0) return CGS_OK;
    if (batch < 0 || channels < 0 || height < 0 || width < 0 || !img1 || !img2 || !ssim_map ||
        (dm_dmu1 && (!dm_dsigma1_sq || !dm_dsigma12)) || (long long)batch * channels > 65535) {
        set_error("cgs_ssim_forward: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    launch_ssim_fwd((hipStream_t)stream_, batch * channels, height, width, C1, C2, img1, img2, ssim_map, dm_dmu1,
                    dm_dsigma1_sq, dm_dsigma12);
    if (!check_launch("ssim_forward", false, (hipStream_t)stream_)) return CGS_ERR_HIP;
    return CGS_OK;
}

int cgs_ssim_backward(int batch, int channels, int height, int width, float C1, float C2, const float* img1,
                      const float* img2, const float* dL_dmap, const float* dm_dmu1, const float* dm_dsigma1_sq,
                      const float* dm_dsigma12, float* dL_dimg1, void* stream_) {
    (void)C1;
    (void)C2;
    if (batch * channels == 0 || height == 0 || width == 0) return CGS_OK;
    if (batch < 0 || channels < 0 || height < 0 || width < 0 || !img1 || !img2 || !dL_dmap || !dm_dmu1 ||
        !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1 || (long long)batch * channels > 65535) {
        set_error("cgs_ssim_backward: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    launch_ssim_bwd((hipStream_t)stream_, batch * channels, height, width, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq,
                    dm_dsigma12, dL_dimg1);
    if (!check_launch("ssim_backward", false, (hipStream_t)stream_)) return CGS_ERR_HIP;
    return CGS_OK;
}

int cgs_edge_aware_loss(int channels, int height, int width, const float* image, const float* gt, float threshold,
                        void* scratch16, float* dL_dimage, void* stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (channels <= 0 || height <= 0 || width <= 0 || !image || !gt || !scratch16) {
        set_error("cgs_edge_aware_loss: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    if (zero_async(scratch16, 16, s) != hipSuccess) {
        set_error("zero_async failed");
        return CGS_ERR_HIP;
    }
    launch_edge_aware_loss(s, channels, height, width, image, gt, threshold, scratch16, dL_dimage);
    if (!check_launch("edge_aware_loss", false, s)) return CGS_ERR_HIP;
    return CGS_OK;
}

size_t cgs_photometric_workspace_bytes(int height, int width) {
    return photometric_workspace_bytes(height > 0 ? height : 1, width > 0 ? width : 1);
}
int cgs_render_epilogue(int height, int width, const float* color_raw, const float* all_map, const float* viewmatrix, int clamp,
                        float* color_out, float* dir_out, void* stream_) {
    if (height <= 0 || width <= 0 || (color_out && !color_raw) || (dir_out && (!all_map || !viewmatrix))) {
        set_error("cgs_render_epilogue: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    if (!color_out && !dir_out) return CGS_OK;
    const size_t npix = (size_t)height * width;
    hipLaunchKernelGGL(k_render_epilogue, dim3((unsigned)std::min<size_t>((npix + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream_,
                       npix, color_raw, all_map, viewmatrix, clamp, color_out, dir_out);
    if (!check_launch("render_epilogue", false, (hipStream_t)stream_)) return CGS_ERR_HIP;
    return CGS_OK;
}
int cgs_clamp_backward(int64_t n, const float* raw, const float* g_in, float* g_out, void* stream_) {
    if (n < 0 || (n > 0 && (!raw || !g_in || !g_out))) {
        set_error("cgs_clamp_backward: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(k_clamp_backward, dim3((unsigned)std::min<size_t>(((size_t)n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream_,
                       (size_t)n, raw, g_in, g_out);
    if (!check_launch("clamp_backward", false, (hipStream_t)stream_)) return CGS_ERR_HIP;
    return CGS_OK;
}

int cgs_edge_count(int channels, int height, int width, const float* gt, float threshold, uint32_t* n_pos, void* stream_) {
    if (channels <= 0 || height <= 0 || width <= 0 || !gt || !n_pos) {
        set_error("cgs_edge_count: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = (hipStream_t)stream_;
    if (zero_async(n_pos, sizeof(uint32_t), s) != hipSuccess) {
        set_error("cgs_edge_count: zero_async failed");
        return CGS_ERR_HIP;
    }
    launch_edge_count(s, channels, height * width, gt, threshold, n_pos);
    if (!check_launch("edge_count", false, s)) return CGS_ERR_HIP;
    return CGS_OK;
}
int cgs_photometric_loss(int height, int width, const float* image, const float* gt, float threshold,
                         const uint32_t* n_pos, float lambda_edge, float lambda_ssim, int clamp_input, void* workspace,
                         float* dL_dimage, float* loss, void* stream_) {
    if (height <= 0 || width <= 0 || !image || !gt || !n_pos || !workspace || !dL_dimage || !loss) {
        set_error("cgs_photometric_loss: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = (hipStream_t)stream_;
    launch_photometric_loss(s, height, width, image, gt, nullptr, threshold, n_pos, lambda_edge, lambda_ssim, clamp_input,
                            workspace, dL_dimage, loss);
    if (!check_launch("photometric_loss", false, s)) return CGS_ERR_HIP;
    return CGS_OK;
}
int cgs_photometric_loss_indexed(int height, int width, const float* image, const float* gt_stack, const int* view_index,
                                 float threshold, const uint32_t* n_pos_table, float lambda_edge, float lambda_ssim,
                                 int clamp_input, void* workspace, float* dL_dimage, float* loss, void* stream_) {
    if (height <= 0 || width <= 0 || !image || !gt_stack || !view_index || !n_pos_table || !workspace || !dL_dimage ||
        !loss) {
        set_error("cgs_photometric_loss_indexed: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = (hipStream_t)stream_;
    launch_photometric_loss(s, height, width, image, gt_stack, view_index, threshold, n_pos_table, lambda_edge, lambda_ssim,
                            clamp_input, workspace, dL_dimage, loss);
    if (!check_launch("photometric_loss_indexed", false, s)) return CGS_ERR_HIP;
    return CGS_OK;
}

size_t cgs_curve_regularizers_workspace_bytes(void) { return curve_reg_workspace_bytes(); }
int cgs_curve_regularizers(int B, int m, const float* rotation_raw, const float* opacity_logit, const float* width_log,
                           const int* radii, float w_opacity, const float* opacity_gate, float w_smooth, float w_width,
                           float width_threshold, void* workspace, float* loss, float* dL_drotation_raw,
                           float* dL_dopacity_logit, float* dL_dwidth_log, void* stream_) {
    if (B <= 0 || m < 2 || m > 32 || 256 / m < 1 || !rotation_raw || !opacity_logit || !width_log || !radii || !workspace ||
        !loss || !dL_drotation_raw || !dL_dopacity_logit || !dL_dwidth_log || !aligned16(rotation_raw) ||
        !aligned16(dL_drotation_raw)) {
        set_error("cgs_curve_regularizers: invalid argument (NULL / misaligned pointer, B=%d m=%d)", B, m);
        return CGS_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = (hipStream_t)stream_;
    launch_curve_regularizers(s, B, m, rotation_raw, opacity_logit, width_log, radii, w_opacity, opacity_gate, w_smooth,
                              w_width, width_threshold, workspace, loss, dL_drotation_raw, dL_dopacity_logit,
                              dL_dwidth_log);
    if (!check_launch("curve_regularizers", false, s)) return CGS_ERR_HIP;
    return CGS_OK;
}

int cgs_adam_step_flat(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                       const void* segments, int n_segments, float beta1, float beta2, float eps, int step,
                       int zero_grads, void* stream_) {
    if (n == 0) return CGS_OK;
    if (n < 0 || !params || !grads || !exp_avg || !exp_avg_sq || !segments || n_segments <= 0 ||
        n_segments > adam_max_segments() || step <= 0) {
        set_error("cgs_adam_step_flat: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    launch_adam_flat((hipStream_t)stream_, (long long)n, params, grads, exp_avg, exp_avg_sq, segments, n_segments, beta1,
                     beta2, eps, (float)bc1, (float)sqrt(bc2), zero_grads);
    if (!check_launch("adam_step_flat", false, (hipStream_t)stream_)) return CGS_ERR_HIP;
    return CGS_OK;
}

int cgs_adam_step_flat_dev(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                           const void* device_state, int n_segments, float beta1, float beta2, float eps, int zero_grads,
                           const uint32_t* skip_flag, void* stream_) {
    if (n == 0) return CGS_OK;
    if (n < 0 || !params || !grads || !exp_avg || !exp_avg_sq || !device_state || n_segments <= 0 ||
        n_segments > adam_max_segments()) {
        set_error("cgs_adam_step_flat_dev: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    launch_adam_flat_dev((hipStream_t)stream_, (long long)n, params, grads, exp_avg, exp_avg_sq, device_state, n_segments,
                         beta1, beta2, eps, zero_grads, skip_flag);
    if (!check_launch("adam_step_flat_dev", false, (hipStream_t)stream_)) return CGS_ERR_HIP;
    return CGS_OK;
}
int cgs_adam_step_flat_dev_report(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                                  const void* device_state, int n_segments, float beta1, float beta2, float eps, int zero_grads,
                                  const uint32_t* skip_flag, uint32_t* report_seq, uint32_t* report_ring, int report_len,
                                  void* stream_) {
    if (n == 0) return CGS_OK;
    if (n < 0 || !params || !grads || !exp_avg || !exp_avg_sq || !device_state || n_segments <= 0 ||
        n_segments > adam_max_segments() || !report_seq || !report_ring || report_len <= 0) {
        set_error("cgs_adam_step_flat_dev_report: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    launch_adam_flat_dev((hipStream_t)stream_, (long long)n, params, grads, exp_avg, exp_avg_sq, device_state, n_segments,
                         beta1, beta2, eps, zero_grads, skip_flag, report_seq, report_ring, report_len);
    if (!check_launch("adam_step_flat_dev_report", false, (hipStream_t)stream_)) return CGS_ERR_HIP;
    return CGS_OK;
}
size_t cgs_endpoint_connection_workspace_bytes(int B) { return endpoint_connection_workspace_bytes(B > 0 ? B : 1); }
int cgs_endpoint_connection_loss(int B, const float* curve_points, float distance_threshold, float weight, void* workspace,
                                 float* loss, float* dL_dcurve_points, int accumulate, void* stream_) {
    if (B <= 0 || !curve_points || !workspace || !loss || !dL_dcurve_points || !(distance_threshold > 0.f)) {
        set_error("cgs_endpoint_connection_loss: invalid argument (NULL pointer, B=%d or threshold <= 0)", B);
        return CGS_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = (hipStream_t)stream_;
    launch_endpoint_connection(s, B, curve_points, distance_threshold, weight, workspace, loss, dL_dcurve_points, accumulate);
    if (!check_launch("endpoint_connection_loss", false, s)) return CGS_ERR_HIP;
    return CGS_OK;
}

size_t cgs_adam_state_bytes(void) { return adam_state_bytes(); }

size_t cgs_knn_workspace_bytes(int P) { return knn_workspace_bytes(P); }

int cgs_knn_mean_dist2(int P, const float* points, float* mean_dist2, void* workspace, void* stream_) {
    if (P == 0) return CGS_OK;
    if (P < 0 || !points || !mean_dist2 || !workspace) {
        set_error("cgs_knn_mean_dist2: invalid argument");
        return CGS_ERR_INVALID_ARGUMENT;
    }
    launch_knn((hipStream_t)stream_, P, points, mean_dist2, workspace);
    if (!check_launch("knn_mean_dist2", false, (hipStream_t)stream_)) return CGS_ERR_HIP;
    return CGS_OK;
}

}  // extern "C"
