// Tile compositing kernels: forward alpha-composite (reference K6, forward.cu:279-417) and its backward
// (reference K8, backward.cu:451-675).
//
// One workgroup (256 threads = 4 waves) per 16x16 tile.  Each wave owns an 8x8 pixel quadrant (lane l ->
// x = l & 7, y = l >> 3) so that a wave's pixels are spatially compact and whole-wave rejection of a splat
// (exec-mask skip) is likely.  Splat records (64 B, written by preprocess) are gathered once per 256-splat
// batch into LDS and broadcast-read by all lanes.
//
// Backward: per (wave, splat) the 64 per-pixel contributions are summed with DPP row reductions, the four
// row sums are added into per-batch LDS accumulators with ds_add_f32, and the batch is flushed with ONE
// global float atomic per (tile, splat, field) -- 64..256x fewer global atomics than the reference's
// per-pixel atomicAdd (backward.cu:613-672).
#include "kernels.h"

namespace cgs {

constexpr int BATCH = 256;

struct TileGeom {
    uint32_t tile, tx, ty;
    int px, py;       // this lane's pixel
    bool inside;
    uint32_t pix_id;
};
__device__ __forceinline__ TileGeom tile_geom(int W, int H, int grid_x) {
    TileGeom g;
    g.tile = blockIdx.x;
    g.tx = g.tile % grid_x;
    g.ty = g.tile / grid_x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lx = ((wave & 1) << 3) | (lane & 7);
    const int ly = ((wave >> 1) << 3) | (lane >> 3);
    g.px = g.tx * TILE + lx;
    g.py = g.ty * TILE + ly;
    g.inside = g.px < W && g.py < H;
    g.pix_id = (uint32_t)(W * g.py + g.px);
    return g;
}

// ------------------------------------------------------------------------------------------------ forward
template <bool GEO>
__global__ void __launch_bounds__(256) k_render_fwd(const uint2* __restrict__ ranges,
                                                    const uint32_t* __restrict__ point_list, int W, int H, int grid_x,
                                                    const SplatRec* __restrict__ rec, float* __restrict__ final_T,
                                                    uint32_t* __restrict__ n_contrib, const float* __restrict__ bg_color,
                                                    float* __restrict__ out_color, float* __restrict__ out_invdepth,
                                                    float* __restrict__ out_all_map) {
    __shared__ float4 s_a[BATCH];
    __shared__ float4 s_b[BATCH];
    __shared__ float4 s_c[GEO ? BATCH : 1];
    const TileGeom g = tile_geom(W, H, grid_x);
    const float pixfx = (float)g.px, pixfy = (float)g.py;
    const uint2 range = ranges[g.tile];
    const int total = (int)(range.y - range.x);
    const int rounds = (total + BATCH - 1) / BATCH;
    bool done = !g.inside;
    int toDo = total;
    float T = 1.0f;
    uint32_t contributor = 0, last_contributor = 0;
    float C = 0.f, Dacc = 0.f;
    float A0 = 0.f, A1 = 0.f, A2 = 0.f, A3 = 0.f;
    for (int i = 0; i < rounds; i++, toDo -= BATCH) {
        if (__syncthreads_count(done) == BATCH) break;
        const int progress = i * BATCH + threadIdx.x;
        if (progress < total) {
            const uint32_t id = point_list[range.x + progress];
            const SplatRec* r = rec + id;
            s_a[threadIdx.x] = r->a;
            s_b[threadIdx.x] = r->b;
            if (GEO) s_c[threadIdx.x] = r->c;
        }
        __syncthreads();
        const int nb = min(BATCH, toDo);
        for (int j = 0; !done && j < nb; j++) {
            contributor++;
            const float4 a = s_a[j];
            const float4 b = s_b[j];
            const float dx = a.x - pixfx, dy = a.y - pixfy;
            const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
            if (power > 0.0f) continue;
            const float alpha = fminf(0.99f, b.y * __expf(power));
            if (alpha < 1.0f / 255.0f) continue;
            const float test_T = T * (1.f - alpha);
            if (test_T < 0.0001f) {
                done = true;
                continue;
            }
            const float w = alpha * T;
            C += b.z * w;
            Dacc += b.w * w;
            if (GEO) {
                const float4 c = s_c[j];
                A0 += c.x * w; A1 += c.y * w; A2 += c.z * w; A3 += c.w * w;
            }
            T = test_T;
            last_contributor = contributor;
        }
    }
    if (g.inside) {
        const size_t HW = (size_t)H * W;
        final_T[g.pix_id] = T;
        n_contrib[g.pix_id] = last_contributor;
        out_color[g.pix_id] = C + T * bg_color[0];
        out_invdepth[g.pix_id] = Dacc;
        if (GEO) {
            out_all_map[g.pix_id] = A0;
            out_all_map[HW + g.pix_id] = A1;
            out_all_map[2 * HW + g.pix_id] = A2;
            out_all_map[3 * HW + g.pix_id] = A3;
        } else if (out_all_map) {
            out_all_map[g.pix_id] = 0.f;
            out_all_map[HW + g.pix_id] = 0.f;
            out_all_map[2 * HW + g.pix_id] = 0.f;
            out_all_map[3 * HW + g.pix_id] = 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// Accumulator slots per staged splat.
enum { ACC_MX = 0, ACC_MY, ACC_CA, ACC_CB, ACC_CC, ACC_OP, ACC_COL, ACC_INVD, ACC_M0, ACC_M1, ACC_M2, ACC_M3, ACC_N };

template <bool GEO, bool INVD>
__global__ void __launch_bounds__(256) k_render_bwd(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int grid_x,
    const float* __restrict__ bg_color, const SplatRec* __restrict__ rec, const float* __restrict__ final_Ts,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dout_invdepth, const float* __restrict__ dL_dout_all_map,
    float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic2D, float* __restrict__ dL_dopacity,
    float* __restrict__ dL_dcolors, float* __restrict__ dL_dinvdepths, float* __restrict__ dL_dall_map) {
    __shared__ float4 s_a[BATCH];
    __shared__ float4 s_b[BATCH];
    __shared__ float4 s_c[GEO ? BATCH : 1];
    __shared__ uint32_t s_id[BATCH];
    __shared__ float s_acc[ACC_N][BATCH];
    const TileGeom g = tile_geom(W, H, grid_x);
    const int lane = threadIdx.x & 63;
    const float pixfx = (float)g.px, pixfy = (float)g.py;
    const uint2 range = ranges[g.tile];
    const int total = (int)(range.y - range.x);
    if (total == 0) return;
    const int rounds = (total + BATCH - 1) / BATCH;
    const size_t HW = (size_t)H * W;

    const float T_final = g.inside ? final_Ts[g.pix_id] : 0.f;
    float T = T_final;
    uint32_t contributor = (uint32_t)total;
    const uint32_t last_contributor = g.inside ? n_contrib[g.pix_id] : 0u;
    float accum_rec = 0.f, accum_invd = 0.f, accum_m0 = 0.f, accum_m1 = 0.f, accum_m2 = 0.f, accum_m3 = 0.f;
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
    float last_alpha = 0.f, last_color = 0.f, last_invd = 0.f, lm0 = 0.f, lm1 = 0.f, lm2 = 0.f, lm3 = 0.f;
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);  // backward.cu:542-543
    const float bg_dot_dpixel = bg_color[0] * dL_dpixel;
    // A wave has nothing to do once every lane is past its last contributor == 0, i.e. for lanes whose pixel blended nothing.
    int toDo = total;
    for (int i = 0; i < rounds; i++, toDo -= BATCH) {
        __syncthreads();  // previous batch fully flushed / consumed
        const int progress = i * BATCH + threadIdx.x;
        if (progress < total) {
            const uint32_t id = point_list[range.y - progress - 1];  // back to front (backward.cu:554)
            const SplatRec* r = rec + id;
            s_id[threadIdx.x] = id;
            s_a[threadIdx.x] = r->a;
            s_b[threadIdx.x] = r->b;
            if (GEO) s_c[threadIdx.x] = r->c;
        }
#pragma unroll
        for (int k = 0; k < ACC_N; k++) s_acc[k][threadIdx.x] = 0.f;
        __syncthreads();
        const int nb = min(BATCH, toDo);
        for (int j = 0; j < nb; j++) {
            contributor--;
            const float4 a = s_a[j];
            const float4 b = s_b[j];
            const float dx = a.x - pixfx, dy = a.y - pixfy;
            const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
            const float G = __expf(power);
            const float alpha = fminf(0.99f, b.y * G);
            const bool active = (contributor < last_contributor) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
            if (__ballot(active) == 0ull) continue;  // wave-uniform: nobody in this 8x8 quadrant blended this splat
            float v_mx = 0.f, v_my = 0.f, v_ca = 0.f, v_cb = 0.f, v_cc = 0.f, v_op = 0.f, v_col = 0.f, v_invd = 0.f;
            float v_m0 = 0.f, v_m1 = 0.f, v_m2 = 0.f, v_m3 = 0.f;
            if (active) {
                T = T * __builtin_amdgcn_rcpf(1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha;
                {
                    const float c = b.z;
                    accum_rec = last_alpha * last_color + (1.f - last_alpha) * accum_rec;
                    last_color = c;
                    dL_dalpha = (c - accum_rec) * dL_dpixel;
                    v_col = dchannel_dcolor * dL_dpixel;
                }
                if (INVD) {
                    const float invd = b.w;
                    accum_invd = last_alpha * last_invd + (1.f - last_alpha) * accum_invd;
                    last_invd = invd;
                    dL_dalpha += (invd - accum_invd) * dL_invd;
                    v_invd = dchannel_dcolor * dL_invd;
                }
                if (GEO) {
                    const float4 c = s_c[j];
                    accum_m0 = last_alpha * lm0 + (1.f - last_alpha) * accum_m0; lm0 = c.x;
                    accum_m1 = last_alpha * lm1 + (1.f - last_alpha) * accum_m1; lm1 = c.y;
                    accum_m2 = last_alpha * lm2 + (1.f - last_alpha) * accum_m2; lm2 = c.z;
                    accum_m3 = last_alpha * lm3 + (1.f - last_alpha) * accum_m3; lm3 = c.w;
                    dL_dalpha += (c.x - accum_m0) * dm0;
                    dL_dalpha += (c.y - accum_m1) * dm1;
                    dL_dalpha += (c.z - accum_m2) * dm2;
                    dL_dalpha += (c.w - accum_m3) * dm3;
                    v_m0 = dchannel_dcolor * dm0; v_m1 = dchannel_dcolor * dm1;
                    v_m2 = dchannel_dcolor * dm2; v_m3 = dchannel_dcolor * dm3;
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final * __builtin_amdgcn_rcpf(1.f - alpha)) * bg_dot_dpixel;
                const float dL_dG = b.y * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * a.z - gdy * a.w;
                const float dG_ddely = -gdy * b.x - gdx * a.w;
                v_mx = dL_dG * dG_ddelx * ddelx_dx;
                v_my = dL_dG * dG_ddely * ddely_dy;
                v_ca = -0.5f * gdx * dx * dL_dG;
                v_cb = -0.5f * gdx * dy * dL_dG;
                v_cc = -0.5f * gdy * dy * dL_dG;
                v_op = G * dL_dalpha;
            }
            // 64 -> 4 partial sums per value (DPP), then one LDS atomic per 16-lane row
            v_mx = row16_sum(v_mx); v_my = row16_sum(v_my);
            v_ca = row16_sum(v_ca); v_cb = row16_sum(v_cb); v_cc = row16_sum(v_cc);
            v_op = row16_sum(v_op); v_col = row16_sum(v_col);
            if (INVD) v_invd = row16_sum(v_invd);
            if (GEO) { v_m0 = row16_sum(v_m0); v_m1 = row16_sum(v_m1); v_m2 = row16_sum(v_m2); v_m3 = row16_sum(v_m3); }
            if ((lane & 15) == 0) {
                atomicAdd(&s_acc[ACC_MX][j], v_mx); atomicAdd(&s_acc[ACC_MY][j], v_my);
                atomicAdd(&s_acc[ACC_CA][j], v_ca); atomicAdd(&s_acc[ACC_CB][j], v_cb);
                atomicAdd(&s_acc[ACC_CC][j], v_cc); atomicAdd(&s_acc[ACC_OP][j], v_op);
                atomicAdd(&s_acc[ACC_COL][j], v_col);
                if (INVD) atomicAdd(&s_acc[ACC_INVD][j], v_invd);
                if (GEO) {
                    atomicAdd(&s_acc[ACC_M0][j], v_m0); atomicAdd(&s_acc[ACC_M1][j], v_m1);
                    atomicAdd(&s_acc[ACC_M2][j], v_m2); atomicAdd(&s_acc[ACC_M3][j], v_m3);
                }
            }
        }
        __syncthreads();
        // flush: thread j owns staged splat j
        if ((int)threadIdx.x < nb) {
            const int j = threadIdx.x;
            const uint32_t id = s_id[j];
            const float mx = s_acc[ACC_MX][j], my = s_acc[ACC_MY][j];
            const float ca = s_acc[ACC_CA][j], cb = s_acc[ACC_CB][j], cc = s_acc[ACC_CC][j];
            const float op = s_acc[ACC_OP][j], col = s_acc[ACC_COL][j];
            // a splat no pixel of this tile blended contributes exact zeros: skip the atomics
            if (mx != 0.f || my != 0.f || ca != 0.f || cb != 0.f || cc != 0.f || op != 0.f || col != 0.f ||
                (INVD && s_acc[ACC_INVD][j] != 0.f) ||
                (GEO && (s_acc[ACC_M0][j] != 0.f || s_acc[ACC_M1][j] != 0.f || s_acc[ACC_M2][j] != 0.f || s_acc[ACC_M3][j] != 0.f))) {
                atomicAdd(&dL_dmean2D[3 * id + 0], mx);
                atomicAdd(&dL_dmean2D[3 * id + 1], my);
                atomicAdd(&dL_dconic2D[4 * id + 0], ca);
                atomicAdd(&dL_dconic2D[4 * id + 1], cb);
                atomicAdd(&dL_dconic2D[4 * id + 3], cc);
                atomicAdd(&dL_dopacity[id], op);
                atomicAdd(&dL_dcolors[id], col);
                if (INVD) atomicAdd(&dL_dinvdepths[id], s_acc[ACC_INVD][j]);
                if (GEO) {
                    atomicAdd(&dL_dall_map[4 * id + 0], s_acc[ACC_M0][j]);
                    atomicAdd(&dL_dall_map[4 * id + 1], s_acc[ACC_M1][j]);
                    atomicAdd(&dL_dall_map[4 * id + 2], s_acc[ACC_M2][j]);
                    atomicAdd(&dL_dall_map[4 * id + 3], s_acc[ACC_M3][j]);
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------ launchers
void launch_render_fwd(hipStream_t s, bool geo, int tiles, const uint2* ranges, const uint32_t* point_list, int W,
                       int H, int grid_x, const SplatRec* rec, float* final_T, uint32_t* n_contrib,
                       const float* bg_color, float* out_color, float* out_invdepth, float* out_all_map) {
    ProfScope p("render_fwd", s);
    if (geo)
        hipLaunchKernelGGL(k_render_fwd<true>, dim3(tiles), dim3(256), 0, s, ranges, point_list, W, H, grid_x, rec,
                           final_T, n_contrib, bg_color, out_color, out_invdepth, out_all_map);
    else
        hipLaunchKernelGGL(k_render_fwd<false>, dim3(tiles), dim3(256), 0, s, ranges, point_list, W, H, grid_x, rec,
                           final_T, n_contrib, bg_color, out_color, out_invdepth, out_all_map);
}
void launch_render_bwd(hipStream_t s, bool geo, bool invd, int tiles, const uint2* ranges,
                       const uint32_t* point_list, int W, int H, int grid_x, const float* bg_color,
                       const SplatRec* rec, const float* final_Ts, const uint32_t* n_contrib, const float* dL_dpixels,
                       const float* dL_dout_invdepth, const float* dL_dout_all_map, float* dL_dmean2D,
                       float* dL_dconic2D, float* dL_dopacity, float* dL_dcolors, float* dL_dinvdepths,
                       float* dL_dall_map) {
    ProfScope p("render_bwd", s);
#define CGS_BWD(G, I)                                                                                               \
    hipLaunchKernelGGL((k_render_bwd<G, I>), dim3(tiles), dim3(256), 0, s, ranges, point_list, W, H, grid_x, bg_color, \
                       rec, final_Ts, n_contrib, dL_dpixels, dL_dout_invdepth, dL_dout_all_map, dL_dmean2D,          \
                       dL_dconic2D, dL_dopacity, dL_dcolors, dL_dinvdepths, dL_dall_map)
    if (geo && invd) CGS_BWD(true, true);
    else if (geo) CGS_BWD(true, false);
    else if (invd) CGS_BWD(false, true);
    else CGS_BWD(false, false);
#undef CGS_BWD
}

}  // namespace cgs
