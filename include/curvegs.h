/*
 * curvegs.h -- C ABI of libcurvegs.so, the MI355X (gfx950) native curve-Gaussian hot path.
 *
 * Drop-in boundary: these entry points are what the reference's torch extension shims
 * (/root/reference/submodules/diff-cur-rasterization/rasterize_points.cu, submodules/fused-ssim/ssim.cu,
 * submodules/simple-knn/spatial.cu) call into, restated with plain pointers, sizes and a HIP stream.
 * No torch types cross this boundary.  All pointers are DEVICE pointers unless marked "host".
 * Every function returns 0 (CGS_OK) / a non-negative count on success and a negative cgs_status on
 * failure; cgs_last_error() returns a thread-local message for the last failure.
 *
 * Memory is caller-owned.  The rasterizer's scratch state lives in three byte buffers obtained through
 * caller-supplied allocation callbacks, exactly like the reference's std::function<char*(size_t)>
 * resize callbacks (rasterize_points.cu:27-33, cuda_rasterizer/rasterizer.h:24-60); the same three
 * buffers are handed back to cgs_rasterize_backward.
 */
#ifndef CURVEGS_H_INCLUDED
#define CURVEGS_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cgs_status {
    CGS_OK = 0,
    CGS_ERR_INVALID_ARGUMENT = -1,
    CGS_ERR_HIP = -2,          /* a HIP runtime call or kernel launch failed; see cgs_last_error() */
    CGS_ERR_ALLOC = -3,        /* an allocation callback returned NULL */
    CGS_ERR_NO_DEVICE = -4
} cgs_status;

/* Caller-supplied allocator: must return a device pointer to at least `bytes` bytes (any alignment >= 16;
 * the library aligns its carve-outs to 128 B itself), valid until the matching backward has run. */
typedef void* (*cgs_alloc_fn)(void* user, size_t bytes);

const char* cgs_last_error(void);
int cgs_version(void);
/* Name of the GPU ISA this library was compiled for ("gfx950"). */
const char* cgs_target_arch(void);

/* ------------------------------------------------------------------------------------------------
 * Rasterizer.  Replaces CudaRasterizer::Rasterizer::{forward,backward,markVisible}
 * (cuda_rasterizer/rasterizer.h:24-98; bodies cuda_rasterizer/rasterizer_impl.cu:198-347, :351-466, :141-153)
 * as called from RasterizeGaussiansCUDA / RasterizeGaussiansBackwardCUDA / markVisible
 * (rasterize_points.cu:35-130, :132-239, :241-260).
 *
 * Fixed by the reference's config.h: 1 colour channel, 4 "all_map" channels.
 * NULL for shs / colors_precomp / scales / rotations / cov3D_precomp / all_map plays the role of the
 * reference's empty tensors.  Exactly one of (shs, colors_precomp) and one of ((scales,rotations),
 * cov3D_precomp) must be non-NULL.  SH layout is the reference's single-channel [P, M] float layout
 * (forward.cu:32-33).
 *
 * cgs_rasterize_forward:
 *   out_color [1,H,W], out_invdepth [1,H,W], out_all_map [4,H,W] f32, radii [P] i32 are fully written
 *   (no pre-zeroing required).  Returns num_rendered (#(splat,tile) instances binned) >= 0, which the
 *   caller passes back as R.  Performs one stream synchronisation (to size the binning buffer), like
 *   the reference's blocking 4-byte D2H copy (rasterizer_impl.cu:287).
 * ------------------------------------------------------------------------------------------------ */
int64_t cgs_rasterize_forward(
    cgs_alloc_fn geometry_alloc, void* geometry_user,
    cgs_alloc_fn binning_alloc, void* binning_user,
    cgs_alloc_fn image_alloc, void* image_user,
    int P, int D, int M,
    const float* background,      /* [3]; only [0] is read (1 channel) */
    int width, int height,
    const float* means3D,         /* [P,3] */
    const float* shs,             /* [P,M] or NULL */
    const float* colors_precomp,  /* [P,1] or NULL */
    const float* opacities,       /* [P,1] */
    const float* scales,          /* [P,3] or NULL */
    float scale_modifier,
    const float* rotations,       /* [P,4] (w,x,y,z), used UN-normalised, or NULL */
    const float* cov3D_precomp,   /* [P,6] or NULL */
    const float* all_map,         /* [P,4] or NULL (required when render_geo) */
    const float* viewmatrix,      /* [16], column-major math matrix = row-major transposed torch tensor */
    const float* projmatrix,      /* [16] */
    const float* cam_pos,         /* [3] */
    float tan_fovx, float tan_fovy,
    int prefiltered,
    float* out_color, float* out_invdepth, float* out_all_map,
    int antialiasing, int render_geo,
    int* radii,
    int debug,
    void* stream /* hipStream_t */);

/* cgs_rasterize_backward:
 *   Every gradient output is fully WRITTEN for all P splats (zeros for culled ones); unlike the reference's shim
 *   (rasterize_points.cu:173-193) the caller does not have to zero-fill anything, except dL_dsh [P,M], which is
 *   written for visible splats only when shs != NULL (zero it on entry).  The per-splat accumulation scratch lives
 *   in the geometry buffer, which is therefore written by the backward: the forward's preprocess kernel zeroes it and
 *   the backward hands it back zeroed, so any number of backward calls may follow one forward (no fill launch); the
 *   buffer must be the one the forward of the same splats wrote.
 *   dL_dmean2D [P,3] (.z = 0, NDC-scaled, quirk 9), dL_dconic [P,4] (.x,.y,.w; scratch in the reference, may be
 *   NULL), dL_dopacity [P], dL_dcolor [P,1], dL_dinvdepth [P] (NULL together with dL_dout_invdepth),
 *   dL_dall_map [P,4], dL_dmean3D [P,3], dL_dcov3D [P,6], dL_dscale [P,3], dL_drot [P,4].
 *   dL_dout_all_map may be NULL (treated as zeros) -- lets the autograd wrapper skip materialising unused grads.
 *   dL_dcolor may be NULL when the caller does not need the colour gradient (legal only with shs == NULL and no
 *   depth / all_map gradients flowing in: the training configuration, where colours are a constant ones tensor).
 */
int cgs_rasterize_backward(
    int P, int D, int M, int64_t R,
    const float* background,
    int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* all_map,
    const float* opacities, const float* scales, float scale_modifier, const float* rotations,
    const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float tan_fovx, float tan_fovy,
    const int* radii,
    void* geometry_buffer, const void* binning_buffer, const void* image_buffer,
    const float* dL_dout_color,     /* [1,H,W] */
    const float* dL_dout_invdepth,  /* [1,H,W] or NULL */
    const float* dL_dout_all_map,   /* [4,H,W] or NULL */
    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dinvdepth,
    float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dall_map,
    int antialiasing, int render_geo, int debug,
    void* stream);

/* present[i] = (view-space z of means3D[i] > 0.2); rasterizer_impl.cu:54-66 */
int cgs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/* Byte sizes the allocation callbacks will be asked for (exposed for callers that pre-allocate). */
size_t cgs_geometry_bytes(int P);
size_t cgs_image_bytes(int width, int height);
size_t cgs_binning_bytes(int64_t R);

/* ------------------------------------------------------------------------------------------------
 * Fused per-view path of the training configuration (no counterpart in the reference, which runs ~45 PyTorch kernels and
 * three extension calls for the same work): curve parameters in, image out, and back to curve-parameter gradients.
 * Equivalent to  cgs_sample_curves_forward -> cgs_splat_attrs_forward -> cgs_rasterize_forward_static  (and their
 * backwards in reverse) with scales + rotations, precomputed colours (NULL = all ones), scale_modifier 1, no
 * antialiasing, render_geo on -- i.e. gaussian_renderer/__init__.py:18-157 as train.py calls it -- but the per-splat
 * tensors between those calls (xyz, rotation, scaling, opacity, all_map and their gradients) never leave registers:
 * 8 kernel launches per view and direction pair instead of 13.
 *   cgs_view_forward: arguments as the three calls it replaces; xyz / rotation / scaling may be NULL (all three) when the
 *     caller does not need the model's derived splat tensors.  out_invdepth and out_all_map may BOTH be NULL (only without
 *     colors_precomp): image-only forward for a training iteration, which reads `render` alone (train.py:98-107).  Status words as cgs_rasterize_forward_static.
 *     B * m < 2^28 (the tile-list entries of this path carry four tag bits for the matching cgs_view_backward; the binning
 *     buffer and the 32-byte gradient accumulator records it leaves in the geometry buffer are private to that pair of calls).
 *   cgs_view_backward: valid ONCE per cgs_view_forward (it consumes scratch sums the forward zeroed).  Only dL/dcolour
 *     flows in (train.py's loss reads `render` only); colors_precomp as given to the forward (NULL = unit colours: the
 *     compositors then use closed forms, sum w = 1 - T and dC/dalpha = (1 - bg) T_final / (1 - alpha)); dL_drotation_extra [P,4] or NULL is added to the gradient of the
 *     raw splat rotations before it is pulled back to the curves (the curve-smoothness regulariser enters there).
 *     Outputs: dL_dmeans2D [P,3] (NDC-scaled, feeds add_densification_stats), dL_dcurve_points [B,4,3], dL_dwidth [B,1],
 *     dL_dopacity_logit [B,1], dL_dmask_logit [P] (required iff mask_logit) -- overwritten, or added to when `flags`
 *     has CGS_VIEW_ACCUMULATE (several views summed into one gradient buffer without extra kernels).  scratch:
 *     cgs_view_backward_scratch_floats(B, m) floats (13 per curve are used: the part of dL/d{curve_points, width} that does not
 *     depend on the two grid-wide sums of the sampling backward, summed per curve inside the per-splat kernel; the closing
 *     pass adds the rest from the curve alone -- rounds 2-5 sent 15 floats per SPLAT through this buffer and back).
 *   cgs_view_forward_checked: the same forward for eager callers (the drop-in render(): gaussian_renderer/__init__.py:18-157
 *     as train.py:95-97 calls it).  Like the reference's forward it reports how much it binned -- but the host only waits for
 *     a 16-byte readback queued right behind the SCATTER, with the compositor already enqueued behind it (the reference blocks
 *     on num_rendered before it can even size its sort buffers, rasterizer_impl.cu:287).  Returns the longest tile list
 *     (>= 0; cgs_last_forward_stats has num_rendered): when it exceeds bucket_capacity the outputs are INVALID and the call
 *     must be repeated with a larger capacity; negative = cgs_status.  Updates the per-shape binning hints.
 *   cgs_bucket_capacity_hint: bucket capacity recommended for the next forward of this workload shape (1.25 x the longest
 *     list seen recently + 64, rounded up to 64), 0 when nothing is known about it yet.
 * ------------------------------------------------------------------------------------------------ */
int cgs_view_forward(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier, const float* coef,
                     float eps, double* norms, const float* opacity_logit, const float* mask_logit, float mask_thr,
                     const float* colors_precomp, void* geometry_buffer, void* binning_buffer, size_t binning_bytes,
                     void* image_buffer, uint32_t bucket_capacity, const float* background, int width_px, int height_px,
                     const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                     float* out_color, float* out_invdepth, float* out_all_map, int* radii, float* xyz, float* rotation,
                     float* scaling, void* stream);
int64_t cgs_view_forward_checked(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier, const float* coef,
                     float eps, double* norms, const float* opacity_logit, const float* mask_logit, float mask_thr,
                     const float* colors_precomp, void* geometry_buffer, void* binning_buffer, size_t binning_bytes,
                     void* image_buffer, uint32_t bucket_capacity, const float* background, int width_px, int height_px,
                     const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                     float* out_color, float* out_invdepth, float* out_all_map, int* radii, float* xyz, float* rotation,
                     float* scaling, void* stream);
/* The same in two halves, so that the caller can queue MORE work behind the forward before it blocks (render() queues its
 * clamp and direction-map kernels, then waits): cgs_view_forward_begin enqueues everything including the status readback and
 * returns a HANDLE (>= 0; negative: status code); cgs_view_forward_wait(handle, &n_visible) blocks on that readback, releases
 * the handle and returns what cgs_view_forward_checked returns (n_visible, optional: splats with radii > 0 -- sizes render()'s
 * visibility_filter = (radii > 0).nonzero(), gaussian_renderer/__init__.py:150, without a device-wide sync).  Any number of
 * forwards (threads, devices, streams, models) may be outstanding up to a pool of 64; a handle that will never be waited on
 * (an exception between the two halves) is returned with cgs_view_forward_abandon. */
int cgs_view_forward_begin(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier, const float* coef,
                     float eps, double* norms, const float* opacity_logit, const float* mask_logit, float mask_thr,
                     const float* colors_precomp, void* geometry_buffer, void* binning_buffer, size_t binning_bytes,
                     void* image_buffer, uint32_t bucket_capacity, const float* background, int width_px, int height_px,
                     const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                     float* out_color, float* out_invdepth, float* out_all_map, int* radii, float* xyz, float* rotation,
                     float* scaling, void* stream);
/* The view forward WITH render()'s epilogue (gaussian_renderer/__init__.py:138-145) written by the compositor itself instead of
 * a separate pass over the image (cgs_render_epilogue below): out_color_clamped [H,W] = clamp(out_color, 0, 1) beside the raw
 * out_color (torch.clamp's gradient rule needs the raw value: cgs_view_backward_render), out_rend_dir [3,H,W] = all_map[0:3]
 * taken from view to world space; either may be NULL.  Unit colours only (no colors_precomp); checked != 0: like
 * cgs_view_forward_begin (returns a handle for cgs_view_forward_wait), checked == 0: like cgs_view_forward (sync-free). */
int cgs_view_forward_render(int checked, int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier,
                            const float* coef, float eps, double* norms, const float* opacity_logit, const float* mask_logit,
                            float mask_thr, void* geometry_buffer, void* binning_buffer, size_t binning_bytes, void* image_buffer,
                            uint32_t bucket_capacity, const float* background, int width_px, int height_px, const float* viewmatrix,
                            const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, float* out_color,
                            float* out_invdepth, float* out_all_map, int* radii, float* out_color_clamped, float* out_rend_dir,
                            void* stream);
int64_t cgs_view_forward_wait(int handle, int64_t* n_visible);
void cgs_view_forward_abandon(int handle);
/* Epilogue of render() on the fused route, /root/reference/gaussian_renderer/__init__.py:138-145 in one launch: color_out [H,W] =
 * clamp ? clamp(color_raw, 0, 1) : color_raw (NULL: skipped); dir_out [3,H,W] = all_map[0:3] taken from view to world space,
 * out_i = sum_k all_map[k] * viewmatrix[4 i + k] (viewmatrix = world_view_transform, row-major 4x4; NULL: skipped).
 * cgs_clamp_backward: g_out = (0 <= raw <= 1) ? g_in : 0, torch.clamp's gradient rule. */
int cgs_render_epilogue(int height, int width, const float* color_raw, const float* all_map, const float* viewmatrix, int clamp,
                        float* color_out, float* dir_out, void* stream);
int cgs_clamp_backward(int64_t n, const float* raw, const float* g_in, float* g_out, void* stream);
uint32_t cgs_bucket_capacity_hint(int P, int width, int height);
/* Number of splats with radii > 0 in the calling thread's last cgs_view_forward_checked (-1: none yet); the two-halves form
 * hands it out through cgs_view_forward_wait. */
int64_t cgs_last_forward_visible(void);
/* (radii > 0).nonzero() (gaussian_renderer/__init__.py:150) in ONE launch and without a host sync, for the radii of a CHECKED
 * forward (cgs_view_forward_checked / _begin / _render with checked != 0, or cgs_rasterize_forward on its bucket path): that
 * forward left the visible count of every 1/64th of the splats in its image buffer, and n_visible = their sum came back with
 * its status readback.  out_indices [n_visible] int64, ascending (what torch's nonzero() returns, as a column). */
int cgs_visible_indices(int P, const int* radii, const void* image_buffer, int width, int height, int64_t* out_indices, void* stream);
/* Several views of ONE parameter state (a view batch between two optimizer steps; not the reference's one-view iteration):
 * cgs_view_forward_shared is cgs_view_forward without the grid-wide norm pass of prepare_scaling_rot, and cgs_view_backward
 * with CGS_VIEW_SHARED in its flags adds its per-curve partial gradients into `scratch` and skips the closing pass of the sampling
 * backward (linear in them); the caller brackets the batch with cgs_view_shared_begin (zeroes norms and scratch, computes the
 * norms once) and cgs_view_shared_end (that last pass, once: dL/dcurve_points, dL/dwidth written or added to).  Every view of
 * the batch must use the SAME norms and scratch buffers; opacity / mask gradients keep coming from cgs_view_backward
 * (CGS_VIEW_ACCUMULATE to sum them over the batch).  The mode is chosen per call: no process-wide state. */
int cgs_view_forward_shared(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier, const float* coef,
                     float eps, double* norms, const float* opacity_logit, const float* mask_logit, float mask_thr,
                     const float* colors_precomp, void* geometry_buffer, void* binning_buffer, size_t binning_bytes,
                     void* image_buffer, uint32_t bucket_capacity, const float* background, int width_px, int height_px,
                     const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                     float* out_color, float* out_invdepth, float* out_all_map, int* radii, float* xyz, float* rotation,
                     float* scaling, void* stream);
int cgs_view_shared_begin(int B, int m, const float* curve_points, const uint8_t* is_bezier, const float* coef, double* norms,
                          float* scratch, void* stream);
int cgs_view_shared_end(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier, const float* coef,
                        float eps, double* norms, float* scratch, float* dL_dcurve_points, float* dL_dwidth, int accumulate,
                        void* stream);
size_t cgs_view_backward_scratch_floats(int B, int m);
/* Layout of the `norms` buffer (f64 words): [0, *first) are the forward's grid-wide sums (written by the norm pass),
 * [*first, *first + *count) the two sums the sampling backward ACCUMULATES -- cleared by the forward's norm pass, so a caller that
 * runs a second backward over one forward (retain_graph) zeroes exactly this range in between.  Returns the buffer's size in words. */
int cgs_view_norms_backward_range(int* first, int* count);
/* flags of cgs_view_backward (a plain 0 / 1 keeps its old meaning: overwrite / accumulate) */
#define CGS_VIEW_ACCUMULATE 1 /* add to dL_dcurve_points, dL_dwidth, dL_dopacity_logit, dL_dmask_logit instead of writing them */
#define CGS_VIEW_SHARED 2     /* shared curve sampling of a view batch (above): per-splat gradients go to `scratch` only */
int cgs_view_backward(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier, const float* coef,
                      float eps, double* norms, const float* opacity_logit, const float* mask_logit, float mask_thr,
                      const float* colors_precomp, void* geometry_buffer, const void* binning_buffer, const void* image_buffer, const float* background,
                      int width_px, int height_px, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                      float tan_fovx, float tan_fovy, const int* radii, const float* dL_dout_color,
                      const float* dL_drotation_extra, float* dL_dmeans2D, float* dL_dcurve_points, float* dL_dwidth,
                      float* dL_dopacity_logit, float* dL_dmask_logit, float* scratch, int flags, void* stream);
/* cgs_view_backward for an image that went through render()'s clamp: dL_dout_color is the gradient of the CLAMPED image,
 * color_raw the forward's unclamped one; the gradient counts only where 0 <= color_raw <= 1 (torch.clamp's rule), applied
 * where the compositor loads the pixel's upstream gradient -- no separate cgs_clamp_backward pass.  Unit colours only. */
int cgs_view_backward_render(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier, const float* coef,
                      float eps, double* norms, const float* opacity_logit, const float* mask_logit, float mask_thr,
                      void* geometry_buffer, const void* binning_buffer, const void* image_buffer, const float* background,
                      int width_px, int height_px, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                      float tan_fovx, float tan_fovy, const int* radii, const float* dL_dout_color, const float* color_raw,
                      float* dL_dmeans2D, float* dL_dcurve_points, float* dL_dwidth,
                      float* dL_dopacity_logit, float* dL_dmask_logit, float* scratch, int flags, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Curve -> Gaussian sampling.  Replaces GaussianCurveModel.prepare_scaling_rot
 * (/root/reference/scene/gaussian_curve_model.py:180-198 with get_curve_gaussians :70-78, get_curve_tangent :80-89 and
 * rot_to_quat_batch, utils/general_utils.py:33-86) and its autograd backward.
 *   curve_points [B,4,3], width [B,1] (log), is_bezier [B] u8 or NULL (= all Bezier).  P = B*m, splat = b*m + i.
 *   coef [m,16] f32: per-sample weights computed by the host with the reference's float32 expressions
 *     {c0..c3 at t_i, c0..c3 at t_i-0.5/m, 3(1-t)^2, 6(1-t)t, 3t^2, (1-t), t, (1-t'), t', pad}.
 *   norms [384] f64 scratch: [0..191] = 64 partial sums each of three global sums written by the forward and needed
 *     by the backward (the two Frobenius norms of the reference's global normalisations and one cross term);
 *     [192..319] are backward scratch (cleared by the forward's norm pass; cgs_sample_curves_backward clears them itself
 *     on entry, the per-view backward of cgs_view_backward does not: ONE view backward per view forward); the rest is
 *     reserved.  The buffer needs no initialisation by the caller.
 *   outputs xyz [P,3], rotation [P,4] (w,x,y,z, un-normalised), scaling [P,3].
 * The backward accepts NULL for any upstream gradient (treated as zero).  m <= 32.
 * ------------------------------------------------------------------------------------------------ */
int cgs_sample_curves_forward(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier,
                              const float* coef, float eps, double* norms, float* xyz, float* rotation,
                              float* scaling, void* stream);
int cgs_sample_curves_backward(int B, int m, const float* curve_points, const float* width, const uint8_t* is_bezier,
                               const float* coef, float eps, double* norms, const float* dL_dxyz,
                               const float* dL_drotation, const float* dL_dscaling, float* dL_dcurve_points,
                               float* dL_dwidth, float* scratch /* [P,9] f32, required when dL_drotation != NULL */,
                               void* stream);

/* Per-view splat attributes fed to the rasterizer (one fused kernel each way instead of ~40 PyTorch kernels):
 *   rotation_n = F.normalize(rotation_raw)                      gaussian_curve_model.py:121-122
 *   opacity    = sigmoid(opacity_logit[b]) expanded to splats    :108-110   (* mask when mask_logit != NULL)
 *   scaling_out = scaling * mask (only when mask_logit != NULL; straight-through mask,
 *                 gaussian_renderer/__init__.py:72-76); pass scaling_out = NULL otherwise
 *   all_map    = [ R(rotation_n)[:,0] flipped toward the camera @ view[:3,:3], 1 ]   :99-105, renderer :98-104
 * opacity_logit [B,1], mask_logit [B,m,1] or NULL, campos [3], viewmatrix [16] (world_view_transform, row-major). */
int cgs_splat_attrs_forward(int B, int m, const float* rotation_raw, const float* xyz, const float* opacity_logit,
                            const float* mask_logit, float mask_thr, const float* scaling, const float* campos,
                            const float* viewmatrix, float* rotation_n, float* opacity, float* scaling_out,
                            float* all_map, void* stream);
int cgs_splat_attrs_backward(int B, int m, const float* rotation_raw, const float* xyz, const float* opacity_logit,
                             const float* mask_logit, float mask_thr, const float* scaling, const float* campos,
                             const float* viewmatrix, const float* dL_drotation_n, const float* dL_dopacity,
                             const float* dL_dscaling_out, const float* dL_dall_map, float* dL_drotation_raw,
                             float* dL_dopacity_logit, float* dL_dmask_logit, float* dL_dscaling, void* stream);

/* ------------------------------------------------------------------------------------------------
 * fused-ssim.  Replaces fusedssim / fusedssim_backward (/root/reference/submodules/fused-ssim/ssim.cu:368-404,
 * :406-444; kernels :187-286, :288-366; declared in ssim.h:7-26).  img [batch, channels, H, W] f32 contiguous, zero
 * padding ("same"); the "valid" crop is done by the Python wrapper like the reference (fused_ssim/__init__.py:13-14).
 * dm_* may be NULL (train == false).
 * ------------------------------------------------------------------------------------------------ */
int cgs_ssim_forward(int batch, int channels, int height, int width, float C1, float C2, const float* img1,
                     const float* img2, float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                     void* stream);
int cgs_ssim_backward(int batch, int channels, int height, int width, float C1, float C2, const float* img1,
                      const float* img2, const float* dL_dmap, const float* dm_dmu1, const float* dm_dsigma1_sq,
                      const float* dm_dsigma12, float* dL_dimg1, void* stream);

/* Fused class-balanced edge loss: edge_aware_loss(image, gt_image, threshold) of
 * /root/reference/utils/loss_utils.py:94-115 (train.py:101).  image, gt [C,H,W] f32.  scratch16: 16 bytes of device
 * scratch; after the call (stream order) the f64 at scratch16+8 holds SUM (image-gt)^2 * mask, so
 * loss = that / (C*H*W); the u32 at scratch16 holds the edge-pixel count.  dL_dimage (optional) = d loss / d image. */
int cgs_edge_aware_loss(int channels, int height, int width, const float* image, const float* gt, float threshold,
                        void* scratch16, float* dL_dimage, void* stream);

/* The photometric part of the training loss (train.py:101-107) for a 1-channel render, value and gradient in ONE
 * pass of three kernels (SSIM forward, SSIM backward + edge-loss/clamp epilogue, scalar finish):
 *   loss = lambda_edge * edge_aware_loss(x, gt, threshold) + lambda_ssim * (1 - fused_ssim(x, gt)),
 *   x = clamp_input ? clamp(image, 0, 1) : image          (render()'s clamp, gaussian_renderer/__init__.py:138)
 * with lambda_edge = lambda_mse (1 - lambda_dssim), lambda_ssim = lambda_mse lambda_dssim in train.py's notation.
 * n_pos: device u32 = #{gt > threshold}, from cgs_edge_count (depends on gt only: compute once per gt image).
 * workspace: cgs_photometric_workspace_bytes(H, W) bytes, zero-filled before its FIRST use, then reusable as is.
 * Outputs: dL_dimage [H*W] = d loss / d image (0 where the clamp is active), loss [1]. */
size_t cgs_photometric_workspace_bytes(int height, int width);
int cgs_edge_count(int channels, int height, int width, const float* gt, float threshold, uint32_t* n_pos, void* stream);
int cgs_photometric_loss(int height, int width, const float* image, const float* gt, float threshold,
                         const uint32_t* n_pos, float lambda_edge, float lambda_ssim, int clamp_input, void* workspace,
                         float* dL_dimage, float* loss, void* stream);
/* Same, with the target picked on the device: gt_stack [V,H,W], n_pos_table [V], *view_index (device int) selects the
 * entry.  For stream-captured training iterations (train.py:95 picks a random view per iteration): the captured launch
 * is the same for every view, no 4*H*W-byte copy of the step's edge map into a staging buffer. */
int cgs_photometric_loss_indexed(int height, int width, const float* image, const float* gt_stack, const int* view_index,
                                 float threshold, const uint32_t* n_pos_table, float lambda_edge, float lambda_ssim,
                                 int clamp_input, void* workspace, float* dL_dimage, float* loss, void* stream);

/* End-point connection loss of /root/reference/train.py:133-146 (active after opt.conn_from_iter): over the 2B curve end
 * points (first and last control point of every curve), loss = weight * mean distance of all ordered pairs of DIFFERENT
 * curves closer than distance_threshold (0.05 in the reference); 0 when there is no such pair.  The reference builds the
 * (2B)^2 cdist matrix; this is a neighbour search on a hashed uniform grid, O(B) memory and time.  curve_points [B,4,3].
 * dL_dcurve_points [B,4,3]: accumulate != 0 adds the gradient to the rows 0 and 3 (the other rows are untouched),
 * accumulate == 0 writes the whole tensor (rows 1, 2 zero).  workspace: cgs_endpoint_connection_workspace_bytes(B). */
size_t cgs_endpoint_connection_workspace_bytes(int B);
int cgs_endpoint_connection_loss(int B, const float* curve_points, float distance_threshold, float weight, void* workspace,
                                 float* loss, float* dL_dcurve_points, int accumulate, void* stream);

/* The per-iteration regularisers of /root/reference/train.py:113-131 (PyTorch ops over all P splats in the reference),
 * value and gradients in three launches:
 *   loss = w_opacity * gate * mean_{splats with radii > 0} log(1 + sigmoid(opacity_logit[b])^2 / 0.5)
 *        + w_smooth * [any radii > 0] * mean_{b, i < m-1} (1 - |cos(d_i, d_{i+1})|),
 *              d = column 0 of quaternion_to_matrix(normalize(rotation_raw))  (main axis of splat b*m + i)
 *        + w_width * mean_{curves with exp(width_log[b]) >= width_threshold} (exp(width_log[b]) - width_threshold)
 * Empty selections contribute 0 (the reference guards them with host-side ifs).  opacity_gate: device float or NULL
 * (= 1): train.py's `reset_timestep > 0` switch, kept on the device so a captured graph need not be re-captured.
 * rotation_raw [B*m,4] (16-byte aligned), opacity_logit [B], width_log [B], radii [B*m] int32.  workspace:
 * cgs_curve_regularizers_workspace_bytes() bytes, zero-filled before its FIRST use, then reusable as is.
 * Outputs: loss [1]; dL_drotation_raw [B*m,4], dL_dopacity_logit [B], dL_dwidth_log [B] (every element written). */
size_t cgs_curve_regularizers_workspace_bytes(void);
int cgs_curve_regularizers(int B, int m, const float* rotation_raw, const float* opacity_logit, const float* width_log,
                           const int* radii, float w_opacity, const float* opacity_gate, float w_smooth, float w_width,
                           float width_threshold, void* workspace, float* loss, float* dL_drotation_raw,
                           float* dL_dopacity_logit, float* dL_dwidth_log, void* stream);

/* One-launch Adam over a flat parameter buffer (torch.optim.Adam semantics: no weight decay, no amsgrad), replacing
 * the per-group foreach step of the reference (scene/gaussian_curve_model.py:200-213, train.py:235).
 * segments: HOST array of n_segments (<= 16) x {int64 begin; float lr; float pad}, sorted by begin,
 * segments[0].begin == 0; element i uses the lr of the last segment with begin <= i (the table is passed to the
 * kernel by value, so a per-iteration learning-rate change costs no copy).  step = 1-based step count (bias
 * correction).  zero_grads != 0 also clears grads (optimizer.zero_grad(), train.py:236) in the same pass. */
int cgs_adam_step_flat(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                       const void* segments, int n_segments, float beta1, float beta2, float eps, int step,
                       int zero_grads, void* stream);

/* Graph-replayable Adam: as cgs_adam_step_flat, but the per-step scalars live in DEVICE memory
 * (device_state: cgs_adam_state_bytes() bytes = 16 x {int64 begin; float lr; float pad} followed by
 * {float 1 - beta1^t; float sqrt(1 - beta2^t); float pad[2]}, refreshed by the caller with a stream-ordered copy), and
 * when skip_flag != NULL and *skip_flag != 0 (e.g. status word [2] of a cgs_rasterize_forward_static image buffer) the
 * parameters and moments are left untouched (gradients are still cleared if zero_grads). */
int cgs_adam_step_flat_dev(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                           const void* device_state, int n_segments, float beta1, float beta2, float eps, int zero_grads,
                           const uint32_t* skip_flag, void* stream);
/* The same step with a report for replayed iterations: report_seq (device u32, set by the caller once) counts the executions;
 * execution number n writes 1 (skipped) or 0 into report_ring[n % report_len].  report_ring may be pinned host memory mapped into
 * the device (hipHostMalloc / torch pin_memory): the host then learns about skipped iterations without a device-to-host copy
 * queued between one replay and the next -- it reads entry n once an event recorded behind replay n has completed. */
int cgs_adam_step_flat_dev_report(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                                  const void* device_state, int n_segments, float beta1, float beta2, float eps, int zero_grads,
                                  const uint32_t* skip_flag, uint32_t* report_seq, uint32_t* report_ring, int report_len,
                                  void* stream);
size_t cgs_adam_state_bytes(void);

/* ------------------------------------------------------------------------------------------------
 * simple-knn.  Replaces distCUDA2 -> SimpleKNN::knn (/root/reference/submodules/simple-knn/spatial.cu:15-26,
 * simple_knn.cu:186-222): mean_dist2[i] = mean of the 3 smallest SQUARED distances from point i to other points.
 * workspace: cgs_knn_workspace_bytes(P) bytes of device scratch.
 * ------------------------------------------------------------------------------------------------ */
size_t cgs_knn_workspace_bytes(int P);
int cgs_knn_mean_dist2(int P, const float* points /*[P,3]*/, float* mean_dist2 /*[P]*/, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-call options of the operator API.  The `debug` argument of cgs_rasterize_forward / cgs_rasterize_backward is a bit
 * set: bit 0 is the reference's debug flag (rasterize_points.cu:53: synchronise and check after every kernel), the bits
 * below select measurement / parity variants FOR THAT CALL ONLY -- nothing process-wide, nothing another thread's call
 * can see (rounds 1-5 had cgs_set_tile_culling / cgs_set_fused_tile_sort / cgs_set_operator_unit_route here).
 * ------------------------------------------------------------------------------------------------ */
#define CGS_OPT_DEBUG 0x1
#define CGS_OPT_NO_TILE_CULLING 0x100   /* cgs_rasterize_forward: bin like the reference (below) */
#define CGS_OPT_GENERAL_BACKWARD 0x200  /* cgs_rasterize_backward: never the unit-colour kernel (below) */
/* Tile-level culling (default on; CGS_OPT_NO_TILE_CULLING switches it off for one forward).  The reference bins a splat into EVERY tile of the
 * bounding square of radius ceil(3 sigma) (forward.cu:318-321 getRect, rasterizer_impl.cu:70-110) and lets the
 * compositor skip it per pixel when alpha < 1/255 (forward.cu:371-377).  With culling on, an instance
 * (splat, tile) is only created when the splat can reach alpha >= 1/255 at some pixel of that tile, so
 * num_rendered is smaller than the reference's while images, radii and every gradient are unchanged (the
 * dropped instances are exactly those the compositor would skip at all 256 pixels).  With culling off,
 * num_rendered and the per-tile lists are bit-identical to the reference's.
 *
 * Unit-colour route of the backward (default on; CGS_OPT_GENERAL_BACKWARD keeps the general instance for one backward): a
 * cgs_rasterize_backward that is asked for neither colour nor depth / all_map gradients (the training configuration of the
 * reference's own call, gaussian_renderer/__init__.py:96-129) lets the GPU choose between the pair-major unit-colour compositor
 * and the general one: the forward's scatter raises a word of the image buffer when some visible splat's colour or all_map[3]
 * is not exactly 1, and both kernels test it on entry (no host sync; the forward tags its tile-list entries with quadrant
 * masks whenever P < 2^28).
 *
 * The tile sort inside the forward compositor (sync-free forwards whose bucket capacity allows it) has a read-once environment
 * switch for A/B measurements: CGS_FUSED_TILE_SORT=0 -> separate per-tile sort launch.
 * ------------------------------------------------------------------------------------------------ */
/* Introspection of the calling thread's last cgs_rasterize_forward: num_rendered, the longest per-tile list and
 * which binning path produced it (0 = exact count/scan/scatter layout, 1 = single-pass fixed-capacity buckets). */
/* ------------------------------------------------------------------------------------------------
 * Sync-free forward for stream-ordered and hipGraph-captured pipelines (no counterpart in the reference, whose
 * forward blocks on a device-to-host copy of num_rendered, rasterizer_impl.cu:287).  Same inputs, outputs and saved
 * state as cgs_rasterize_forward, but:
 *   - the three buffers are allocated by the caller: cgs_geometry_bytes(P), cgs_image_bytes(W, H) and
 *     cgs_binning_bytes(bucket_capacity * tiles) bytes, tiles = ceil(W/16) * ceil(H/16);
 *   - binning is the single-pass bucket layout with `bucket_capacity` slots per tile (<= cgs_bucket_capacity_limit());
 *     a good value is 1.25-2 x the longest tile list reported by cgs_last_forward_stats after a normal forward;
 *   - nothing is read back: no num_rendered, no host wait -- the call only enqueues work on `stream`.
 * Status words (u32) at byte offset cgs_image_status_offset(W, H) of the image buffer, valid in stream order:
 *   [2] != 0: some tile list outgrew its bucket -> the image and every gradient of this forward are INVALID (redo it
 *       with cgs_rasterize_forward or a larger capacity); [4 + 2k], [5 + 2k], k < (cgs_status_words() - 4) / 2:
 *       partial sums / maxima of the tile list lengths (num_rendered = sum of the sums, longest list = max of maxima);
 *   [3]: number of splats whose tile rectangle exceeds 96 tiles (near-camera splats of room-scale scenes).  When a
 *       previous cgs_rasterize_forward of the same (P, width, height) saw any, such splats are binned by a second kernel, one workgroup
 *       each, instead of inside the wave that owns them (cgs_reset_binning_hints clears that memory too).
 *   word [cgs_status_words()] (right behind the status words) is a STICKY count of overflowed tile lists: the library only
 *       ever adds to it, so a caller that zeroes the image buffer once can replay many views and check one word at the
 *       end instead of one flag per view.
 * The backward is cgs_rasterize_backward with R = 1.
 * ------------------------------------------------------------------------------------------------ */
int cgs_rasterize_forward_static(void* geometry_buffer, void* binning_buffer, size_t binning_bytes, void* image_buffer,
                                 uint32_t bucket_capacity, int P, int D, int M, const float* background, int width,
                                 int height, const float* means3D, const float* shs, const float* colors_precomp,
                                 const float* opacities, const float* scales, float scale_modifier,
                                 const float* rotations, const float* cov3D_precomp, const float* all_map,
                                 const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                                 float tan_fovy, float* out_color, float* out_invdepth, float* out_all_map,
                                 int antialiasing, int render_geo, int* radii, void* stream);
size_t cgs_image_status_offset(int width, int height);
int cgs_status_words(void);
uint32_t cgs_bucket_capacity_limit(void);

/* cgs_rasterize_forward learns, per workload shape (P, width, height), how large the previous forward of that shape was
 * (num_rendered, longest tile list, oversized splats) and sizes its speculative buffers from it; shapes do not disturb
 * each other (a process may alternate train / test cameras, resolutions or models; the 32 most recent shapes are kept).
 * This call forgets everything learnt (the next forward of every shape takes the exact path and re-learns). */
void cgs_reset_binning_hints(void);
void cgs_last_forward_stats(int64_t* num_rendered, int64_t* longest_tile_list, int* binning_path);

/* ------------------------------------------------------------------------------------------------
 * Per-kernel timing hook used by bench.py: when enabled, every kernel launched by the library is
 * bracketed by hipEvents on the caller's stream; cgs_prof_collect synchronises and accumulates.
 * ------------------------------------------------------------------------------------------------ */
void cgs_prof_enable(int on);
void cgs_prof_reset(void);
/* Fills up to `cap` entries; returns the number of distinct kernels seen.  names[i] points to static storage. */
int cgs_prof_collect(const char** names, double* total_ms, int64_t* launches, int cap);

#ifdef __cplusplus
}
#endif
#endif /* CURVEGS_H_INCLUDED */
