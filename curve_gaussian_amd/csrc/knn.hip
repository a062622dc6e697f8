// simple-knn: mean squared distance to the 3 nearest neighbours of every point (exact).
// Replaces SimpleKNN::knn (submodules/simple-knn/simple_knn.cu:186-222): bbox reduce (N1, with the reference's
// {0,0,0}-initialised min/max, :192-200), 30-bit Morton codes (N2 :55-71), sort by Morton (N3 :211-214), per-1024
// box AABBs (N4 :79-118) and the pruned exact 3-NN search (N5 :148-184).
//
// The result does not depend on the traversal order (it is the exact 3-NN), so the sort is a plain single-workgroup
// bitonic network on (code << 32 | index) keys -- this runs once per training run on a few thousand points.
#include <cfloat>

#include "kernels.h"

namespace cgs {

constexpr int KNN_BOX = 1024;
struct MinMax { float3 minn, maxx; };

__device__ __forceinline__ uint32_t prep_morton(uint32_t x) {  // simple_knn.cu:45-52
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

// N1: min/max over all points, both reductions seeded with (0,0,0) (reference quirk 14)
__global__ void __launch_bounds__(1024) k_knn_bbox(int P, const float* __restrict__ pts, float* __restrict__ bbox) {
    __shared__ float red[6][16];
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v = pts[3 * (size_t)i + c];
            mn[c] = fminf(mn[c], v);
            mx[c] = fmaxf(mx[c], v);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        for (int off = 32; off >= 1; off >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], off, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off, 64));
        }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { red[c][w] = mn[c]; red[3 + c][w] = mx[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[threadIdx.x][0];
        for (int k = 1; k < (int)(blockDim.x >> 6); k++) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][k]) : fmaxf(v, red[threadIdx.x][k]);
        bbox[threadIdx.x] = v;
    }
}

// N2: keys = (morton << 32) | index
__global__ void __launch_bounds__(256) k_knn_morton(int P, const float* __restrict__ pts, const float* __restrict__ bbox,
                                                    uint64_t* __restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float3 minn = make_float3(bbox[0], bbox[1], bbox[2]), maxx = make_float3(bbox[3], bbox[4], bbox[5]);
    const float3 c = make_float3(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]);
    const uint32_t x = prep_morton((uint32_t)(((c.x - minn.x) / (maxx.x - minn.x)) * ((1 << 10) - 1)));
    const uint32_t y = prep_morton((uint32_t)(((c.y - minn.y) / (maxx.y - minn.y)) * ((1 << 10) - 1)));
    const uint32_t z = prep_morton((uint32_t)(((c.z - minn.z) / (maxx.z - minn.z)) * ((1 << 10) - 1)));
    keys[i] = ((uint64_t)(x | (y << 1) | (z << 2)) << 32) | (uint32_t)i;
}

// N3: single-workgroup ascending bitonic sort ("flip + disperse" form: any n, no padding) on global memory
__global__ void __launch_bounds__(1024) k_knn_sort(uint32_t n, uint64_t* __restrict__ k, uint32_t* __restrict__ indices) {
    uint32_t n2 = 1;
    while (n2 < n) n2 <<= 1;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    for (uint32_t size = 2; size <= n2; size <<= 1) {
        for (uint32_t t = tid; t < n2 / 2; t += nt) {
            const uint32_t half = size >> 1, blk = t / half, j = t % half;
            const uint32_t lo = blk * size + j, hi = blk * size + size - 1 - j;
            if (hi < n) {
                const uint64_t a = k[lo], b = k[hi];
                if (a > b) { k[lo] = b; k[hi] = a; }
            }
        }
        __syncthreads();
        for (uint32_t d = size >> 2; d >= 1; d >>= 1) {
            for (uint32_t t = tid; t < n2 / 2; t += nt) {
                const uint32_t lo = 2 * d * (t / d) + (t % d), hi = lo + d;
                if (hi < n) {
                    const uint64_t a = k[lo], b = k[hi];
                    if (a > b) { k[lo] = b; k[hi] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < n; i += nt) indices[i] = (uint32_t)k[i];
}

// N4: AABB of every run of 1024 Morton-sorted points
__global__ void __launch_bounds__(KNN_BOX) k_knn_boxes(uint32_t P, const float* __restrict__ pts,
                                                       const uint32_t* __restrict__ indices, MinMax* __restrict__ boxes) {
    __shared__ float red[6][16];
    const uint32_t idx = blockIdx.x * KNN_BOX + threadIdx.x;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (idx < P) {
        const uint32_t id = indices[idx];
#pragma unroll
        for (int c = 0; c < 3; c++) mn[c] = mx[c] = pts[3 * (size_t)id + c];
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        for (int off = 32; off >= 1; off >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], off, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off, 64));
        }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { red[c][w] = mn[c]; red[3 + c][w] = mx[c]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        MinMax me;
        float v[6];
        for (int q = 0; q < 6; q++) {
            v[q] = red[q][0];
            for (int k = 1; k < 16; k++) v[q] = q < 3 ? fminf(v[q], red[q][k]) : fmaxf(v[q], red[q][k]);
        }
        me.minn = make_float3(v[0], v[1], v[2]);
        me.maxx = make_float3(v[3], v[4], v[5]);
        boxes[blockIdx.x] = me;
    }
}

__device__ __forceinline__ float dist_box_point(const MinMax& box, const float3& p) {  // simple_knn.cu:120-130
    float3 diff = make_float3(0, 0, 0);
    if (p.x < box.minn.x || p.x > box.maxx.x) diff.x = fminf(fabsf(p.x - box.minn.x), fabsf(p.x - box.maxx.x));
    if (p.y < box.minn.y || p.y > box.maxx.y) diff.y = fminf(fabsf(p.y - box.minn.y), fabsf(p.y - box.maxx.y));
    if (p.z < box.minn.z || p.z > box.maxx.z) diff.z = fminf(fabsf(p.z - box.minn.z), fabsf(p.z - box.maxx.z));
    return diff.x * diff.x + diff.y * diff.y + diff.z * diff.z;
}
__device__ __forceinline__ void update_3best(const float3& ref, const float3& point, float* knn) {  // :132-146
    const float3 d = make_float3(point.x - ref.x, point.y - ref.y, point.z - ref.z);
    float dist = d.x * d.x + d.y * d.y + d.z * d.z;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (knn[j] > dist) {
            const float t = knn[j];
            knn[j] = dist;
            dist = t;
        }
    }
}

// N5: exact 3-NN with box pruning (simple_knn.cu:148-184)
__global__ void __launch_bounds__(256) k_knn_mean_dist(uint32_t P, const float* __restrict__ pts,
                                                       const uint32_t* __restrict__ indices,
                                                       const MinMax* __restrict__ boxes, float* __restrict__ dists) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int)P) return;
    auto load = [&](uint32_t sorted_pos) {
        const uint32_t id = indices[sorted_pos];
        return make_float3(pts[3 * (size_t)id], pts[3 * (size_t)id + 1], pts[3 * (size_t)id + 2]);
    };
    const float3 point = load(idx);
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    for (int i = max(0, idx - 3); i <= min((int)P - 1, idx + 3); i++) {
        if (i == idx) continue;
        update_3best(point, load(i), best);
    }
    const float reject = best[2];
    best[0] = FLT_MAX; best[1] = FLT_MAX; best[2] = FLT_MAX;
    const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
    for (int b = 0; b < nboxes; b++) {
        const MinMax box = boxes[b];
        const float dist = dist_box_point(box, point);
        if (dist > reject || dist > best[2]) continue;
        for (int i = b * KNN_BOX; i < min((int)P, (b + 1) * KNN_BOX); i++) {
            if (i == idx) continue;
            update_3best(point, load(i), best);
        }
    }
    dists[indices[idx]] = (best[0] + best[1] + best[2]) / 3.0f;
}

size_t knn_workspace_bytes(int P) {
    const size_t n = (size_t)(P > 0 ? P : 1);
    const size_t nb = (n + KNN_BOX - 1) / KNN_BOX;
    return 128 + n * 8 + 128 + n * 4 + 128 + nb * sizeof(MinMax) + 128;
}

void launch_knn(hipStream_t s, int P, const float* pts, float* dists, void* workspace) {
    char* c = (char*)workspace;
    float* bbox; uint64_t* keys; uint32_t* indices; MinMax* boxes;
    carve(c, bbox, 8);
    carve(c, keys, (size_t)P);
    carve(c, indices, (size_t)P);
    const int nb = (P + KNN_BOX - 1) / KNN_BOX;
    carve(c, boxes, (size_t)nb);
    { ProfScope p("knn_bbox", s); hipLaunchKernelGGL(k_knn_bbox, dim3(1), dim3(1024), 0, s, P, pts, bbox); }
    { ProfScope p("knn_morton", s); hipLaunchKernelGGL(k_knn_morton, dim3((P + 255) / 256), dim3(256), 0, s, P, pts, bbox, keys); }
    { ProfScope p("knn_sort", s); hipLaunchKernelGGL(k_knn_sort, dim3(1), dim3(1024), 0, s, (uint32_t)P, keys, indices); }
    { ProfScope p("knn_boxes", s); hipLaunchKernelGGL(k_knn_boxes, dim3(nb), dim3(KNN_BOX), 0, s, (uint32_t)P, pts, indices, boxes); }
    { ProfScope p("knn_mean_dist", s); hipLaunchKernelGGL(k_knn_mean_dist, dim3((P + 255) / 256), dim3(256), 0, s, (uint32_t)P, pts, indices, boxes, dists); }
}

}  // namespace cgs
