/*
 * depthwarp.h -- C ABI of libdepthwarp.so (MI355X / gfx950).
 *
 * Hand-written HIP kernels for the view-synthesis + photometric-loss path of
 * self-supervised depth training, and for the one loss function the reference
 * holds (Train.compute_loss, /root/reference/train.py:272-309).
 *
 * Conventions (every entry point):
 *   - every data pointer is a DEVICE pointer owned by the caller; arrays of
 *     pointers ("const float* const* disp") are HOST arrays of device pointers;
 *   - tensors are contiguous NCHW float32 unless stated otherwise;
 *   - nothing is allocated, freed or synchronised inside a call: all launches
 *     are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream); scratch memory comes from the caller's `workspace`;
 *   - calls are stateless and re-entrant (two calls may overlap if their
 *     workspaces and outputs do not);
 *   - the return value is DW_OK (0) or a negative DW_ERR_* code; nothing is
 *     thrown across the boundary; dw_strerror() names a code.
 *
 * The reference has no FFI of its own: each declaration cites the Python
 * call it stands behind (reference file:line), or says that the stage is
 * absent from the reference.
 */
#ifndef DEPTHWARP_H
#define DEPTHWARP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(DW_BUILD)
#define DW_API __attribute__((visibility("default")))
#else
#define DW_API
#endif

#define DW_VERSION 2
#define DW_MAX_SCALES 4
#define DW_NUM_SRC 2

enum {
  DW_OK = 0,
  DW_ERR_NULL = -1,      /* a required pointer is NULL                         */
  DW_ERR_SHAPE = -2,     /* unsupported or inconsistent dimensions             */
  DW_ERR_WORKSPACE = -3, /* workspace too small or misaligned                  */
  DW_ERR_LAUNCH = -4,    /* hipLaunchKernel / hipGetLastError reported failure */
  DW_ERR_ARG = -5,       /* bad flag or enum value                             */
  DW_ERR_TOO_LARGE = -6  /* problem exceeds the fixed scratch of an allocation-free entry point */
};

/* Problem description of the photometric stack. Scale s has H>>s x W>>s
 * pixels; H and W must be divisible by 2^(S-1); F must be DW_NUM_SRC;
 * H * W <= 2^28 and B <= 65535. */
typedef struct dw_desc {
  int32_t B; /* samples (3-frame sequences)                 */
  int32_t H; /* full-resolution height                      */
  int32_t W; /* full-resolution width                       */
  int32_t S; /* scales, 1..DW_MAX_SCALES                    */
  int32_t F; /* source frames, = DW_NUM_SRC                 */
  int32_t flags; /* 0, or DW_FLAG_* for profiling runs       */
} dw_desc;

/* Profiling only: dw_photometric_fwd launches just its dominant kernel (the
 * per-tile warp/error/adjoint kernel); outputs other than the masks are then
 * not produced. Used to time that kernel alone with events.  The kernel reads
 * the depth maps a complete call leaves in the workspace: make one complete
 * call with the same inputs and workspace first. */
#define DW_FLAG_TILE_KERNEL_ONLY 1
/* Testing only: never split the last tiles of the launch into per-scale blocks
 * (the split is a scheduling choice; results are identical with and without). */
#define DW_FLAG_NO_TAIL_SPLIT 2
/* Also compute the gradient with respect to the two source images as the
 * warp samples them (the scatter adjoint of the bilinear warp; the identity
 * errors and the target image are not differentiated).  The workspace grows
 * (about 220 * B * H * W bytes at four scales: per-pixel adjoints of the warped
 * colours, per-tile patches, and a fixed-point image for the samples that leave
 * their tile's patch); read the result with dw_photometric_bwd_color.  Must be
 * set identically for workspace_bytes, fwd (with_grad != 0) and the bwd calls. */
#define DW_FLAG_COLOR_GRAD 4

DW_API int dw_version(void);
DW_API const char* dw_strerror(int code);

/* ---------------------------------------------------------------------------
 * Fused photometric stack (stages absent from the reference; they follow the
 * networks at /root/reference/train.py:111 and replace the loss call at
 * train.py:113 in a self-supervised set-up).
 *
 * Forward, per scale s: bilinear up-sample of disp[s] -> depth -> back-project
 * with inv_K -> project with (K @ T[f])[:3] -> bilinear border warp of
 * color[1+f] -> 0.85*SSIM(3x3, reflect) + 0.15*L1 against color[0]; identity
 * errors of the un-warped sources (+ tie_noise); per-pixel minimum over
 * [id0, id1, reproj0, reproj1]; mean; plus 1e-3/2^s * edge-aware smoothness of
 * the mean-normalised disp[s] against color_pyr[s]; averaged over scales.
 *
 *   disp[s]       (B,1,H>>s,W>>s)            s < S
 *   color[0..2]   (B,3,H,W)   target, source(-1), source(+1)
 *   color_pyr[s]  (B,3,H>>s,W>>s)  target colour per scale ([0] may alias color[0])
 *   K, inv_K      (B,4,4)  pixel-unit intrinsics
 *   T             (B,F,4,4)  target->source rigid transforms
 *   tie_noise     (B,F,H,W) or NULL
 *   mask[s]       (B,H,W) uint8 out: bits 0-1 arg-min index (0,1 identity;
 *                 2,3 re-projection), bit 2 / 3 validity of source 0 / 1.
 *                 An entry (or the array) may be NULL to skip the store.
 *   loss_out      1 + 2*S floats out: total, L_photo[0..S), L_sm[0..S)
 *   with_grad     non-zero: also stash in `workspace` what dw_photometric_bwd
 *                 needs (the pass is then forward + the heavy part of backward)
 * ------------------------------------------------------------------------- */
DW_API size_t dw_photometric_workspace_bytes(const dw_desc* desc);

DW_API int dw_photometric_fwd(const dw_desc* desc,
                       const float* const* disp,
                       const float* const* color,
                       const float* const* color_pyr,
                       const float* K, const float* inv_K, const float* T,
                       const float* tie_noise,
                       uint8_t* const* mask,
                       float* loss_out,
                       void* workspace, size_t workspace_bytes,
                       int with_grad, void* stream);

/* Backward of the call above (same desc, same workspace, untouched since).
 *   grad_loss     device floats, upstream gradient of loss_out: grad_loss_len = 1
 *                 (of the total only) or 1 + 2*S (of every entry: a loss built
 *                 from the per-scale terms L_photo[s], L_sm[s] is differentiated too)
 *   grad_disp[s]  (B,1,H>>s,W>>s) out (overwritten)
 *   grad_T        (B,F,4,4) out (overwritten); may be NULL
 */
DW_API int dw_photometric_bwd(const dw_desc* desc,
                       const float* const* disp,
                       const float* K,
                       const float* grad_loss, int32_t grad_loss_len,
                       const void* workspace, size_t workspace_bytes,
                       float* const* grad_disp,
                       float* grad_T,
                       void* stream);

/* Gradient of the loss with respect to the source images color[1], color[2] as
 * sampled by the warp (the optional image gradient of the view-synthesis
 * formulation; the reference, which has none of these stages, has no
 * counterpart).  Requires DW_FLAG_COLOR_GRAD in desc->flags for this call and
 * for the forward call that filled `workspace`.
 *   grad_loss     device scalar, upstream gradient of the loss
 *   grad_color    2 device pointers, (B,3,H,W) each, overwritten
 * The scatter is accumulated per tile in on-chip memory in 64-bit fixed point
 * (2^-32 units) and the tiles' patches are summed per pixel in a fixed order, so
 * the result does not depend on the order of execution; integer atomics in
 * global memory are used only by samples that fall outside their tile's patch. */
DW_API int dw_photometric_bwd_color(const dw_desc* desc,
                       const float* grad_loss,
                       const void* workspace, size_t workspace_bytes,
                       float* const* grad_color,
                       void* stream);

/* ---------------------------------------------------------------------------
 * Reference loss: Train.compute_loss(target, pred), train.py:272-309, forward
 * only as in the reference (its body runs under no_grad, train.py:274).
 * Batched: sample b of the result equals the reference's batch-1 call on
 * sample b (to float32 rounding: how the partial sums of an image are grouped
 * depends on the batch size; a given call is bitwise reproducible).
 *   target   (B,1,Ht,Wt)  ground truth, float32 (float64 with
 *            DW_CL_FLAG_TARGET_F64), resized to (H,W) inside (bilinear,
 *            align_corners=False; train.py:276); H, W >= 4 (reflect border of
 *            the 7x7 window), Wt >= 2 when it differs from (H,W), H*W and
 *            Ht*Wt <= 2^28, B <= 65535; otherwise DW_ERR_SHAPE
 *   pred     (B,1,H,W)
 *   out      (B,4) floats: loss, ssim_term, l1, smooth   (train.py:299-307)
 *   flags    0, or a combination of DW_CL_FLAG_TARGET_F64 and
 *            DW_CL_FLAG_ANTIALIAS: the resize of train.py:276 is
 *            torchvision's, whose default for tensors is a plain two-tap
 *            bilinear resize in some releases and the anti-aliased (area
 *            weighted triangle filter) one in others; the flag selects the latter
 *   workspace: dw_compute_loss_workspace_bytes(B,H,W) bytes (an upper bound; a
 *            call uses 32 bytes per sample, 128-column strip and row chunk)
 * ------------------------------------------------------------------------- */
#define DW_CL_FLAG_ANTIALIAS 1
/* `target` points to float64 values (what the reference's NumPy pipeline hands
 * over, load_dataset.py:69-77): each is rounded to float32 as it is read, which
 * is the `.float()` cast of the float32 path without a pass of its own. */
#define DW_CL_FLAG_TARGET_F64 2

DW_API size_t dw_compute_loss_workspace_bytes(int32_t B, int32_t H, int32_t W);

DW_API int dw_compute_loss_fwd(int32_t B, int32_t Ht, int32_t Wt, int32_t H, int32_t W,
                        const void* target, const float* pred, float* out, int32_t flags,
                        void* workspace, size_t workspace_bytes, void* stream);

/* The reference evaluates its loss twice per step, on the left and on the right
 * view, and adds the results (train.py:113, 177:
 * `compute_loss(l_depth, l_pred) + compute_loss(r_depth, r_pred)`).  This entry
 * point does both in one launch pair without the four maps having to be
 * stacked into batch tensors: sample 0 is (target_l, pred_l), sample 1
 * (target_r, pred_r), each a (1,1,Ht,Wt) / (1,1,H,W) map as in
 * dw_compute_loss_fwd.  The row chunks are those of a one-sample call, so
 * both sets of terms are bit-identical to two dw_compute_loss_fwd calls with
 * B = 1.
 *   out      9 floats: loss, ssim_term, l1, smooth of the left view; the same
 *            of the right view; out[8] = out[0] + out[4], the reference's sum
 *   workspace: dw_compute_loss_workspace_bytes(2,H,W) bytes */
DW_API int dw_compute_loss_pair_fwd(int32_t Ht, int32_t Wt, int32_t H, int32_t W,
                             const void* target_l, const float* pred_l,
                             const void* target_r, const float* pred_r,
                             float* out, int32_t flags,
                             void* workspace, size_t workspace_bytes, void* stream);

/* dw_compute_loss_fwd for callers that cannot allocate or bring a workspace
 * (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  No counterpart in the reference; same kernels and results as
 * dw_compute_loss_fwd.  The partial sums live in a fixed device array
 * inside the library (DW_STATIC_SCRATCH_BYTES per GPU the library runs on, held
 * until the process ends); the call allocates and frees nothing.
 *   - the problem must fit: the call keeps 32 bytes per sample, 128-column strip
 *     and chunk of rows (chunks are 2 to 64 rows tall, taller the more samples
 *     there are) and refuses with DW_ERR_TOO_LARGE what exceeds
 *     DW_STATIC_SCRATCH_BYTES (at 256 x 512: more than 2048 samples);
 *   - all calls on one GPU share that GPU's array: calls that overlap in time on
 *     one GPU (different streams or threads) race -- serialise them, or use
 *     dw_compute_loss_fwd with workspaces of their own;
 *   - the array's address is looked up on the first call on each device: make
 *     that first call outside stream capture. */
#define DW_STATIC_SCRATCH_BYTES (1u << 20)
DW_API int dw_compute_loss_fwd_static(int32_t B, int32_t Ht, int32_t Wt, int32_t H, int32_t W,
                               const void* target, const float* pred, float* out, int32_t flags,
                               void* stream);

/* The same pass, additionally producing the evaluation metrics the reference
 * computes from the same two tensors: Train.compute_errors /
 * compute_depth_errors, train.py:311-343 (rmse between the resized ground
 * truth and clamp(1/pred, clamp_min, clamp_max); the reference clamps to
 * [1e-3, Config.maxdisp]), and the extrema normalize_image needs
 * (train.py:250-251).
 *   out      (B,8) floats: loss, ssim_term, l1, smooth, rmse, pred_min, pred_max, 0
 */
DW_API int dw_compute_loss_metrics_fwd(int32_t B, int32_t Ht, int32_t Wt, int32_t H, int32_t W,
                                const void* target, const float* pred, float clamp_min,
                                float clamp_max, float* out, int32_t flags, void* workspace,
                                size_t workspace_bytes, void* stream);

/* dw_compute_loss_metrics_fwd and normalize_image of the prediction (train.py:248-253) in one call of two
 * launches: `pred_normalized` (B, H*W floats) receives (pred - min) / (max - min) per sample, with the
 * arithmetic of dw_normalize_image (1e5 as the divisor of a constant image), the extrema coming from the
 * same pass that produces the loss and the metrics.  Everything else as dw_compute_loss_metrics_fwd. */
DW_API int dw_compute_loss_metrics_norm_fwd(int32_t B, int32_t Ht, int32_t Wt, int32_t H, int32_t W,
                                     const void* target, const float* pred, float clamp_min,
                                     float clamp_max, float* out, float* pred_normalized, int32_t flags,
                                     void* workspace, size_t workspace_bytes, void* stream);

/* Train.normalize_image, train.py:248-253, per sample and without the two
 * host round trips of the reference: out = (x - min) / (max - min), with 1e5 as
 * the divisor when max == min.
 *   x, out   (B, n)
 *   minmax   NULL (extrema are computed here; needs the workspace), or device
 *            floats with min at [b*minmax_stride] and max at [b*minmax_stride+1]
 *            (e.g. out + 5 of the call above with stride 8; workspace unused)
 */
DW_API size_t dw_normalize_image_workspace_bytes(int32_t B, int64_t n);
DW_API int dw_normalize_image(int32_t B, int64_t n, const float* x, const float* minmax,
                              int32_t minmax_stride, float* out, void* workspace,
                              size_t workspace_bytes, void* stream);

/* Train.compute_depth_errors, train.py:338-343: out[b] = sqrt(mean((a - b)^2))
 * over the n elements of sample b.  Workspace: dw_normalize_image_workspace_bytes(B, n). */
DW_API int dw_rmse(int32_t B, int64_t n, const float* a, const float* b, float* out,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * Unfused building blocks (absent from the reference; the module interfaces
 * of the public view-synthesis formulation).  Same arithmetic as the fused
 * stack, one stage per call, for piecewise adoption and for testing.
 * ------------------------------------------------------------------------- */

/* scaled = min_disp + range * disp; depth = 1 / scaled   (n elements) */
DW_API int dw_disp_to_depth_fwd(int64_t n, const float* disp, float min_disp, float range,
                                float* scaled, float* depth, void* stream);
/* g_disp = g_scaled * range - g_depth * range * depth^2; either upstream may be NULL */
DW_API int dw_disp_to_depth_bwd(int64_t n, const float* depth, const float* g_scaled,
                                const float* g_depth, float range, float* g_disp, void* stream);

/* depth (B,1,H,W), inv_K (B,4,4) -> points (B,4,H*W) homogeneous camera points */
DW_API int dw_backproject_fwd(int32_t B, int32_t H, int32_t W, const float* depth,
                              const float* inv_K, float* points, void* stream);
DW_API int dw_backproject_bwd(int32_t B, int32_t H, int32_t W, const float* inv_K,
                              const float* g_points, float* g_depth, void* stream);

/* points (B,4,H*W), K, T (B,4,4) -> grid (B,H,W,2), (x,y) last, in [-1,1] */
DW_API int dw_project3d_fwd(int32_t B, int32_t H, int32_t W, const float* points, const float* K,
                            const float* T, float eps, float* grid, void* stream);
DW_API size_t dw_project3d_workspace_bytes(int32_t B, int32_t H, int32_t W);
/* g_points (B,4,H*W) out; g_T (B,4,4) out or NULL (then no workspace is needed) */
DW_API int dw_project3d_bwd(int32_t B, int32_t H, int32_t W, const float* points, const float* K,
                            const float* T, float eps, const float* g_grid, float* g_points,
                            float* g_T, void* workspace, size_t workspace_bytes, void* stream);

/* grid_sample(src (B,C,H,W), grid (B,Ho,Wo,2), bilinear, border, align_corners=true) */
DW_API int dw_warp_fwd(int32_t B, int32_t C, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                       const float* src, const float* grid, float* out, void* stream);
/* gradient w.r.t. the grid only (colour images carry no gradient on this path) */
DW_API int dw_warp_bwd(int32_t B, int32_t C, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                       const float* src, const float* grid, const float* g_out, float* g_grid,
                       void* stream);

/* clamp((1 - SSIM_3x3(x, y)) / 2, 0, 1) on `planes` = B*C independent H x W maps */
DW_API int dw_ssim_fwd(int32_t planes, int32_t H, int32_t W, const float* x, const float* y,
                       float* out, void* stream);
/* g_x and/or g_y out (either may be NULL) */
DW_API int dw_ssim_bwd(int32_t planes, int32_t H, int32_t W, const float* x, const float* y,
                       const float* g_out, float* g_x, float* g_y, void* stream);

/* edge-aware smoothness of disp (B,1,H,W) against img (B,3,H,W): out[0] scalar */
DW_API size_t dw_smooth_loss_workspace_bytes(int32_t B, int32_t H, int32_t W);
DW_API int dw_smooth_loss_fwd(int32_t B, int32_t H, int32_t W, const float* disp, const float* img,
                              float* out, void* workspace, size_t workspace_bytes, void* stream);
/* g_loss: device scalar; g_disp (B,1,H,W) out */
DW_API int dw_smooth_loss_bwd(int32_t B, int32_t H, int32_t W, const float* disp, const float* img,
                              const float* g_loss, float* g_disp, void* stream);

/* ---------------------------------------------------------------------------
 * Soft-argmin disparity regression: the tail of DepthBlock.forward,
 * /root/reference/model/model.py:454-456 and :417-429 --
 *   prob = softmax(-cost, D); prob2 = softmax(-prob, D); disp = sum_d d * prob2[d]
 * (the second soft-max of the normalised volume is the reference's own).
 *   cost    (B, D, N) float32, D <= 256 disparities, N = H*W pixels per plane
 *   disp    (B, N) out
 * Backward: g_cost (B, D, N) out from g_disp (B, N); nothing saved by forward.
 * ------------------------------------------------------------------------- */
DW_API int dw_soft_argmin_fwd(int32_t B, int32_t D, int64_t N, const float* cost, float* disp,
                              void* stream);
DW_API int dw_soft_argmin_bwd(int32_t B, int32_t D, int64_t N, const float* cost,
                              const float* g_disp, float* g_cost, void* stream);

/* ---------------------------------------------------------------------------
 * Cost-volume construction: DepthBlock.cost_volume,
 * /root/reference/model/model.py:460-508 (zero-fill shift + concatenate per
 * disparity, then stack), batch-generalised.
 *   imgl, imgr, carl, carr   (B, F, h, w)   feature maps (reference: F = 32)
 *   left, right              (B, D, 4F, h, w) out (reference: D = maxdisp/2 = 64)
 *     d == 0: left = [imgl, carl, imgr, carr]            right = [imgr, carr, imgl, carl]
 *     d  > 0: left = [imgl, imgr(x-d), carl, carr(x-d)]  right = [imgr, imgl(x+d), carr, carl(x+d)]
 *   (the differing channel order at d == 0 is the reference's own, model.py:471-472 vs 497-498)
 * Backward: gradients of the four maps from the gradients of the two volumes
 * (sums in ascending disparity order: reproducible run to run).
 * ------------------------------------------------------------------------- */
DW_API int dw_cost_volume_fwd(int32_t B, int32_t F, int32_t h, int32_t w, int32_t D,
                              const float* imgl, const float* imgr, const float* carl,
                              const float* carr, float* left, float* right, void* stream);
DW_API int dw_cost_volume_bwd(int32_t B, int32_t F, int32_t h, int32_t w, int32_t D,
                              const float* g_left, const float* g_right, float* g_imgl,
                              float* g_imgr, float* g_carl, float* g_carr, void* stream);

/* ---------------------------------------------------------------------------
 * Velodyne scan -> image plane: Dataset.generate_depth_map,
 * /root/reference/dataset/load_dataset.py:116-148 (projection, in-bounds
 * filter, nearest-depth map with the reference's duplicate handling), batched
 * over scans and evaluated in float64 as NumPy does.  The Delaunay hole filling
 * that follows in the reference (:151, :155-165) consumes `pts_im` on the host.
 *   points    (B, n_max, 4) float32 rows (forward, left, up, reflectance), 16-byte aligned
 *   n_points  (B) int32, device: rows in use per scan
 *   P         (B, 3, 4) float64: P_rect . R_cam2rect . velo2cam  (:111-114)
 *   vel_depth non-zero: z = forward instead of the projective depth (:125-126)
 *   pts_im    (B, n_max, 3) float64 out: kept points (x, y, z) in scan order;
 *             rows >= n_valid[b] are not written
 *   n_valid   (B) int32 out, device
 *   depth     (B, H, W) float64 out: the sparse map of :137-148
 * ------------------------------------------------------------------------- */
DW_API size_t dw_lidar_workspace_bytes(int32_t B, int32_t n_max, int32_t H, int32_t W);
DW_API int dw_lidar_project(int32_t B, int32_t n_max, int32_t H, int32_t W, const float* points,
                            const int32_t* n_points, const double* P, int32_t vel_depth,
                            double* pts_im, int32_t* n_valid, double* depth, void* workspace,
                            size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEPTHWARP_H */
