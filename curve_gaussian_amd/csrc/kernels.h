// Host-side launchers, one per kernel, defined next to the kernels (each .hip file is its own translation unit;
// no relocatable device code is needed).  api.hip sequences them on the caller's stream.
#pragma once
#include "common.h"

namespace cgs {

// preprocess.hip
void launch_preprocess_fwd(hipStream_t s, int P, int D, int M, const float* means3D, const float* scales,
                           float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                           uint8_t* clamped, const float* cov3D_precomp, const float* colors_precomp,
                           const float* all_map, const float* viewmatrix, const float* projmatrix,
                           const float* cam_pos, int W, int H, float tan_fovx, float tan_fovy, float focal_x,
                           float focal_y, int* radii, SplatRec* rec, float* rgb, int grid_x, int grid_y,
                           uint32_t* tile_count, int antialiasing, int cull, float* grad_acc, uint32_t* clear_words,
                           size_t n_clear);
void launch_mark_visible(hipStream_t s, int P, const float* means3D, const float* viewmatrix, uint8_t* present);
void launch_preprocess_bwd(hipStream_t s, int P, int D, int M, const float* means3D, const int* radii,
                           const float* shs, const uint8_t* clamped, const float* opacities, const float* scales,
                           const float* rotations, float scale_modifier, const float* cov3D_precomp,
                           const float* viewmatrix, const float* projmatrix, const float* cam_pos, float focal_x,
                           float focal_y, float tan_fovx, float tan_fovy, int W, int H, const SplatRec* rec,
                           float* grad_acc, float* dL_dmean2D, float* dL_dconic, float* dL_dinvdepth,
                           float* dL_dopacity, float* dL_dmean3D, float* dL_dcolor, float* dL_dall_map,
                           float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int antialiasing);

// binning.hip
void launch_scan_tiles(hipStream_t s, int tiles, const uint32_t* tile_count, uint2* ranges, uint32_t* total);
// `cap` = number of instances the binning buffer can hold: tiles whose range does not fit are skipped (speculative launch)
void launch_scatter(hipStream_t s, int P, const int* radii, const SplatRec* rec, int grid_x, int grid_y,
                    const uint2* ranges, uint32_t* tile_cursor, uint64_t* keys, uint32_t cap, int cull,
                    uint32_t* nonunit = nullptr);   // nonunit: word raised when a visible splat's colour / all_map[3] is not 1
// big_count: status word counting the splats whose rect exceeds the wave walk's comfort zone; big_queue (nullable,
// big_cap entries): where they are deferred to for a one-workgroup-per-splat second kernel
void launch_scatter_bucket(hipStream_t s, int P, const int* radii, const SplatRec* rec, int grid_x, int grid_y,
                           uint32_t* tile_count, uint64_t* keys, uint32_t cap, int cull, uint32_t* big_count,
                           uint32_t* big_queue, uint32_t big_cap, uint32_t* nonunit = nullptr,
                           int splats_per_wave = 12);   // 12 (one curve per wave) or 8 (sparse views: more, shorter waves)
void launch_tile_sort_bucket(hipStream_t s, int tiles, const uint32_t* tile_count, uint2* ranges, uint32_t* total,
                             uint64_t* keys, uint32_t* point_list, uint32_t cap);
uint32_t bucket_cap_limit();
void launch_tile_sort_small(hipStream_t s, int tiles, const uint2* ranges, uint64_t* keys, uint32_t* point_list,
                            uint32_t cap);
void launch_tile_sort_big(hipStream_t s, int tiles, const uint2* ranges, uint64_t* keys, uint32_t* point_list,
                          uint32_t max_count);

// render.hip
void launch_render_fwd(hipStream_t s, bool geo, int tiles, const uint2* ranges, const uint32_t* point_list, int W,
                       int H, int grid_x, const SplatRec* rec, float* final_T, uint32_t* n_contrib,
                       const float* bg_color, float* out_color, float* out_invdepth, float* out_all_map,
                       bool unit = false,    // unit: colour == 1 and all_map[3] == 1 for every splat (render.hip, UNIT)
                       bool tag = false);    // tag: staged list entries get the quadrant masks in their top bits (TAG)
bool render_fwd_can_sort(uint32_t cap);
void launch_render_fwd_sorting(hipStream_t s, bool geo, int tiles, const uint32_t* tile_count, const uint64_t* keys,
                               uint32_t cap, uint2* ranges, uint32_t* total, uint32_t* point_list, int W, int H,
                               int grid_x, const SplatRec* rec, float* final_T, uint32_t* n_contrib,
                               const float* bg_color, float* out_color, float* out_invdepth, float* out_all_map,
                               bool unit = false, bool tag = false,
                               // render()'s epilogue written by the same kernel (NULL: not wanted): the image clamped to
                               // [0, 1], the direction map in world space (wv = world_view_transform, device memory)
                               float* color_clamped = nullptr, float* dir_out = nullptr, const float* wv = nullptr);
void launch_render_bwd(hipStream_t s, bool geo, bool invd, bool colg, int tiles, const uint2* ranges,
                       const uint32_t* point_list, int W, int H, int grid_x, const float* bg_color,
                       const SplatRec* rec, const float* final_Ts, const uint32_t* n_contrib, const float* dL_dpixels,
                       const float* dL_dout_invdepth, const float* dL_dout_all_map, float* grad_acc,
                       int acc_stride = ACC_STRIDE,    // floats per accumulator record (ACC_STRIDE_VIEW on the view path)
                       uint32_t id_mask = 0xffffffffu,           // strips the forward's list tags (LIST_ID_MASK when it tagged)
                       const uint32_t* nonunit_gate = nullptr);  // device word: the kernel returns at once when it is zero


// render_unit_bwd.hip: the unit-colour training instance (colour == 1, only dL/dcolour flowing in), lane = (splat, quadrant)
// acc_stride: ACC_STRIDE_VIEW (view path) or ACC_STRIDE (operator API); nonunit_gate: device word, the kernel returns at
// once when it is NOT zero (the general training instance launched beside it takes over)
void launch_render_bwd_unit(hipStream_t s, int tiles, const uint2* ranges, const uint32_t* point_list, int W, int H,
                            int grid_x, const float* bg_color, const SplatRec* rec, const float* final_Ts,
                            const uint32_t* n_contrib, const float* dL_dpixels, float* grad_acc,
                            int acc_stride = ACC_STRIDE_VIEW, const uint32_t* nonunit_gate = nullptr,
                            // torch.clamp's gradient mask folded in: dL/dpixel counts only where 0 <= clamp_raw <= 1 (NULL: all)
                            const float* clamp_raw = nullptr);

// sampling.hip
int sample_norm_words();
void launch_sample_norms(hipStream_t s, int B, int m, const float* cp, const uint8_t* is_bezier, const void* coef, double* norms);
void launch_sample_backward_close(hipStream_t s, int B, int m, const float* cp, const float* width, const uint8_t* is_bezier,
                                  const void* coef, float eps, const double* norms, const float* part, float* g_cp, float* g_width,
                                  int accumulate);
// view.hip
void launch_view_forward(hipStream_t s, int B, int m, const float* cp, const float* width, const uint8_t* is_bezier,
                         const void* coef, float eps, const double* norms, const float* opacity_logit,
                         const float* mask_logit, float mask_thr, const float* colors_precomp, const float* campos,
                         const float* viewmatrix, const float* projmatrix, float tan_fovx, float tan_fovy, float focal_x,
                         float focal_y, int W, int H, int grid_x, int grid_y, float* xyz, float* rot, float* scl, int* radii,
                         SplatRec* rec, float* grad_acc, uint32_t* clear_words, size_t n_clear);
void launch_view_backward(hipStream_t s, int B, int m, const float* cp, const float* width, const uint8_t* is_bezier,
                          const void* coef, float eps, double* norms, const float* opacity_logit, const float* mask_logit,
                          float mask_thr, const float* campos, const float* viewmatrix, const float* projmatrix,
                          float tan_fovx, float tan_fovy, float focal_x, float focal_y, int W, int H, const int* radii,
                          const SplatRec* rec, float* grad_acc, const float* g_rot_raw_extra, float* dL_dmean2D,
                          float* g_opacity_logit, float* g_mask_logit, float* curve_part, int accumulate);
int sample_norm_fwd_words();
void launch_sample_forward(hipStream_t s, int B, int m, const float* cp, const float* width, const uint8_t* is_bezier,
                           const void* coef, float eps, double* norms, float* xyz, float* rot, float* scaling);
void launch_sample_backward(hipStream_t s, int B, int m, const float* cp, const float* width, const uint8_t* is_bezier,
                            const void* coef, float eps, double* norms, const float* g_xyz, const float* g_rot,
                            const float* g_scaling, float* g_cp, float* g_width, float* gv_cache);
void launch_attrs_forward(hipStream_t s, int B, int m, const float* rot_raw, const float* xyz, const float* opacity_logit,
                          const float* mask_logit, float mask_thr, const float* scaling, const float* campos,
                          const float* vm, float* rot_n, float* opac, float* scl_out, float* all_map);
void launch_attrs_backward(hipStream_t s, int B, int m, const float* rot_raw, const float* xyz,
                           const float* opacity_logit, const float* mask_logit, float mask_thr, const float* scaling,
                           const float* campos, const float* vm, const float* g_rot_n, const float* g_opac,
                           const float* g_scl_out, const float* g_all_map, float* g_rot_raw, float* g_opacity_logit,
                           float* g_mask_logit, float* g_scaling);


// ssim.hip
void launch_ssim_fwd(hipStream_t s, int planes, int H, int W, float C1, float C2, const float* img1, const float* img2,
                     float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12);
void launch_ssim_bwd(hipStream_t s, int planes, int H, int W, const float* img1, const float* img2,
                     const float* dL_dmap, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                     float* dL_dimg1);
// loss.hip  (scratch16: 16 zeroed bytes = {u32 n_pos, pad, f64 loss_sum})
size_t photometric_workspace_bytes(int H, int W);
void launch_photometric_loss(hipStream_t s, int H, int W, const float* image, const float* gt, const int* view_index,
                             float thr, const unsigned int* n_pos, float lambda_a, float lambda_b, int clamp,
                             void* workspace, float* grad, float* loss);
void launch_edge_count(hipStream_t s, int C, int HW, const float* gt, float thr, unsigned int* n_pos);
void launch_edge_aware_loss(hipStream_t s, int C, int H, int W, const float* image, const float* gt, float thr,
                            void* scratch16, float* grad);
size_t curve_reg_workspace_bytes();
size_t endpoint_connection_workspace_bytes(int B);
void launch_endpoint_connection(hipStream_t s, int B, const float* cp, float thr, float weight, void* workspace, float* loss,
                                float* dL_dcp, int accumulate);
void launch_curve_regularizers(hipStream_t s, int B, int m, const float* rot_raw, const float* opacity_logit,
                               const float* width, const int* radii, float w_op, const float* op_gate, float w_smo,
                               float w_width, float width_thr, void* workspace, float* loss, float* g_rot_raw,
                               float* g_opacity_logit, float* g_width);
int adam_max_segments();
size_t adam_state_bytes();
void launch_adam_flat_dev(hipStream_t s, long long n, float* p, float* g, float* m, float* v, const void* dev_state,
                          int nseg, float b1, float b2, float eps, int zero_grad, const unsigned int* skip_flag,
                          unsigned int* report_seq = nullptr, unsigned int* report_ring = nullptr, int report_len = 0);
void launch_adam_flat(hipStream_t s, long long n, float* p, float* g, float* m, float* v, const void* host_segs, int nseg,
                      float b1, float b2, float eps, float bc1, float sqrt_bc2, int zero_grad);
// knn.hip
size_t knn_workspace_bytes(int P);
void launch_knn(hipStream_t s, int P, const float* pts, float* dists, void* workspace);

}  // namespace cgs
