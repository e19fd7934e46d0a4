/*
 * fsgs.h -- C ABI of libfsgs_hip.so, the MI355X-native (gfx950) replacement of the
 * native operators on Free-SurGS's splat + pose-optimisation hot path.
 *
 * Plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 * unless marked [host].  Nothing is allocated, freed or kept between calls by the
 * library -- with one exception: a 16 KB pinned host "mailbox" (256 slots) created on
 * first use, through which the forward learns the pair count without synchronising the
 * stream.  All calls are asynchronous on `stream` except where noted and are
 * re-entrant (two Python threads -- trainer and viewer -- may interleave,
 * train.py:150,166-200,227-231).  Return value: 0 on success, negative FSGS_ERR_*.
 *
 * Reference interfaces replaced (file:line in /root/reference; "UPSTREAM" = the
 * un-vendored diff_gaussian_rasterization package, requirements.txt:26):
 *   fsgs_raster_forward / _backward   <- UPSTREAM _C.rasterize_gaussians[_backward],
 *                                        called via GaussianRasterizer at
 *                                        gaussian_renderer/__init__.py:68-69,131
 *   fsgs_knn_meandist2                <- simple_knn._C.distCUDA2,
 *                                        submodules/simple-knn/ext.cpp:15-17,
 *                                        spatial.cu:15-26, simple_knn.cu:185-221
 *   fsgs_render_forward / _backward   <- gaussian_renderer.render() body,
 *                                        gaussian_renderer/__init__.py:49-92 (fused:
 *                                        transform_to_frame + activations + eval_sh +
 *                                        both rasteriser passes in one binning)
 *   fsgs_photometric_loss_*           <- utils/loss_utils.py:41-96 (rgb_loss_func)
 *   fsgs_pearson_*                    <- utils/loss_utils.py:98-127
 *   fsgs_flow_pose_loss_*             <- scene/pose_optimizer.py:164-218
 *   fsgs_sampson_rigid_mask           <- train.py:157-165, scene/pose_optimizer.py:700-746
 *   fsgs_flow_targets_*               <- get_pointcloud + duplicate rejection, scene/pose_optimizer.py:42-73,171-181
 *   fsgs_adam_step                    <- torch.optim.Adam steps of train.py:194,272
 *   fsgs_render_backward_adam         <- loss.backward() + optimizer.step() of one single-view mapping iteration
 *                                        (train.py:265-272) in one pass
 *   fsgs_render_backward_compact /
 *   fsgs_adam_step_compact            <- the same for multi-view / multi-rank steps (summed 56 B/Gaussian gradient)
 *   fsgs_densify_plan / _apply        <- GaussianModel.densify_and_prune, scene/gaussian_model.py:523-676
 */
#ifndef FSGS_H
#define FSGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *fsgs_stream_t; /* hipStream_t */

enum {
  FSGS_OK = 0,
  FSGS_ERR_INVALID = -1,   /* bad argument (null pointer, channel count, size) */
  FSGS_ERR_CAPACITY = -2,  /* state/scratch buffer too small; *num_rendered holds a max_pairs that suffices */
  FSGS_ERR_HIP = -3,       /* a HIP runtime call or kernel launch failed */
  FSGS_ERR_STATE = -4      /* state buffer does not belong to a completed forward */
};

#define FSGS_MAX_CHANNELS 8
/* dispatch the blend workgroups XCD-banded (each XCD's L2 sees one band of the image: less fabric traffic)
 * instead of plain longest-list-first over the whole image (default, ~10 % faster on MI355X) */
#define FSGS_FLAG_XCD_BANDED_ORDER 1
/* fsgs_render_backward*: dL_ddepth_sil is a [1,H,W] gradient of the DEPTH plane only -- the silhouette and depth^2
 * planes get no gradient from any loss of the reference (train.py:166-272: rgb + Pearson(depth); the presence
 * mask and the uncertainty are detached), and the backward blend drops their terms */
#define FSGS_FLAG_DEPTH_GRAD_ONLY 2
/* fsgs_render_backward*: the caller has already zeroed the first P * 64 bytes of `scratch` (e.g. on another stream,
 * beside the loss kernels); the call does not enqueue its own fill in front of the backward blend */
#define FSGS_FLAG_SCRATCH_ZEROED 4
/* fsgs_render_backward*: the per-Gaussian backward stores zeros over every accumulator row once it has read it, so the
 * first P * 64 bytes of `scratch` are zero again when the call's kernels have run.  A caller that starts from a zeroed
 * scratch and passes SCRATCH_ZEROED | SCRATCH_SELF_CLEAN on every backward never clears it again (19 MB more stores inside a
 * bandwidth-bound kernel instead of a fill launch and the stream bookkeeping around it). */
#define FSGS_FLAG_SCRATCH_SELF_CLEAN 16
/* fsgs_render_forward: blend only the RGB planes and the depth plane -- out_depth_sil planes 1 (silhouette) and 2
 * (depth^2) are NOT written.  For the tracking iteration (train.py:166-200), which reads the image and `depth > 0` and
 * nothing else; the state it leaves serves the pose-only backward unchanged. */
#define FSGS_FLAG_RGB_DEPTH_ONLY 8
/* Flavour of the blend kernels (forward and backward of fsgs_raster_* and fsgs_render_*; results agree: the forward's are
 * bit-identical, the backward's differ by the order of float additions).  Default (neither bit): the forward blends with
 * four waves per 16x16 tile (one 8x8 quadrant per wave), the backward does on small tile grids, which one wave per tile
 * cannot fill the chip with (<= 3328 tiles, pose-only backward <= 4352; e.g. 640x512), and uses one wave per tile above.  The bits force one flavour for both directions (tests,
 * A/B measurements); both set = ONE_WAVE. */
#define FSGS_FLAG_BLEND_ONE_WAVE 32
#define FSGS_FLAG_BLEND_QUAD_WAVES 64
/* fsgs_raster_backward / fsgs_render_backward*: bit-reproducible gradients (tests; SURVEY.md s7 "deterministic mode").  The
 * product backward sums a Gaussian's per-tile totals, and dL/dw2c over the workgroups, with float atomics in arrival
 * order; with this flag every (tile, Gaussian) pair's totals are stored in a row of their own and summed per Gaussian in a
 * fixed order (tiles row by row), and dL/dw2c is summed over per-workgroup partials in workgroup order.  `scratch` must
 * then hold fsgs_deterministic_scratch_bytes(P, max_pairs) bytes (FSGS_ERR_CAPACITY otherwise).  Same results as the
 * product path up to the order of float additions; slower (a [max_pairs,16] float buffer is cleared and re-read). */
#define FSGS_FLAG_DETERMINISTIC 128

/* Mirror of GaussianRasterizationSettings (scene/pose_optimizer.py:619-632).
 * viewmatrix / projmatrix are in the reference's TRANSPOSED storage:
 * m[4*k + j] = M[j][k] (scene/pose_optimizer.py:604,617-618). */
typedef struct FsgsRasterCfg {
  int32_t image_height;
  int32_t image_width;
  int32_t channels;        /* colour channels of colors_precomp: 1, 3 (the reference) or 6 (both passes fused);
                            * any other count is FSGS_ERR_INVALID */
  int32_t flags;           /* FSGS_FLAG_* bits, 0 = defaults */
  float tanfovx;
  float tanfovy;
  float scale_modifier;
  float reserved0;
  float bg[FSGS_MAX_CHANNELS];
  float viewmatrix[16];
  float projmatrix[16];
} FsgsRasterCfg;

/* Library / build identification. */
/* Completion events without a marker packet.  An event recorded BETWEEN two kernels of a stream costs a packet of its own in
 * front of the second one (~6 us on MI355X); fsgs_forward_done_event(ev) instead makes the NEXT fsgs_render_forward* call of
 * this thread signal `ev` with the completion of its last kernel (the forward blend: hipExtLaunchKernelGGL's stop event), so
 * that another stream can wait for the forward's outputs while the calling stream goes on to its next kernel at once.
 * One-shot (call it before every forward that should signal; a call that fails before its blend launch records the event
 * the plain way); fsgs_stream_wait_event = hipStreamWaitEvent. */
typedef void *fsgs_event_t; /* hipEvent_t */
int fsgs_event_create(fsgs_event_t *event);
int fsgs_event_destroy(fsgs_event_t event);
int fsgs_stream_wait_event(fsgs_stream_t stream, fsgs_event_t event);
/* hipEventRecord: for a step driver that joins a side stream back into its main stream with an event it keeps (a host
 * language's own event objects cost a creation per step) */
int fsgs_event_record(fsgs_event_t event, fsgs_stream_t stream);
int fsgs_forward_done_event(fsgs_event_t event);
/* the same for the next fsgs_pose_adam_step of this thread: the event is signalled when the updated pose (w2c_next) exists */
int fsgs_pose_step_done_event(fsgs_event_t event);

const char *fsgs_version(void);
const char *fsgs_last_error(void); /* thread-local text of the last FSGS_ERR_HIP */

/* Self-test of the wave-level transposing reduction used by the backward blend:
 * in[lane*64 + i] (64 lanes x 64 values) -> out[l] = sum over lanes of in[lane*64 + l]. */
int fsgs_selftest_transpose_reduce(const float *in64x64, float *out64, fsgs_stream_t stream);
/* The narrower variants: width = 64, 32 or 16 values per lane (the first `width` columns of in); lane l
 * receives the total of column l / (64 / width).  width = 3212 / 1605: the sparse variants the backward blend uses
 * (two Gaussians x 16 slots of which 12 are reduced / x 8 of which 5; the other columns are taken as zero): out64 must
 * then hold 128 floats, out[l] = the lane's total and out[64 + l] = the column (slot) that lane ends up owning. */
int fsgs_selftest_transpose_reduce_n(const float *in64x64, float *out64, int width, fsgs_stream_t stream);
/* Self-test of the blend kernels' per-(pixel, Gaussian) evaluation -- pre-scaled exponent, one v_exp_f32, the
 * power > 0 and alpha < 1/255 skips -- on n independent samples: in8[i] = {mean2D x, y, conic A, B, C,
 * opacity, pixel x, y} -> out2[i] = {o e^power (unclamped; 0 when power > 0), 1 if the pair contributes else 0}.
 * tests/test_raster_gpu.py checks it against a float64 evaluation for bias around the 1/255 threshold. */
int fsgs_selftest_splat_alpha(int n, const float *in8, float *out2, fsgs_stream_t stream);

/* Optional per-kernel timing with HIP events recorded on the launching stream.
 * mask: bit i enables kernel id i (ids: fsgs_profile_name); 0 disables.  Enabling resets totals.
 * fsgs_profile_read synchronises the pending events of that id and returns the running totals. */
int fsgs_profile_enable(uint64_t mask);
/* Time only every stride-th launch of an enabled kernel (default 1 = every launch): each timed launch puts two
 * event packets into the stream, ~2-5 us of dispatch gap apiece, which a throughput measurement should not pay on
 * every step.  Set before fsgs_profile_enable. */
int fsgs_profile_stride(int stride);
int fsgs_profile_count(void);
const char *fsgs_profile_name(int id);
int fsgs_profile_read(int id, double *total_ms, int64_t *launches);

/* ---- rasteriser (drop-in operator boundary) -------------------------------- */

/* Bytes of the per-call buffers for P Gaussians, a W x H image and room for
 * `max_pairs` (tile, Gaussian) pairs.  `state` must stay alive and untouched from
 * forward to its backward (it plays the role of UPSTREAM's geomBuffer /
 * binningBuffer / imgBuffer); `scratch` is only used during forward. */
/* Which flavour of the blend kernels a call on a width x height image with these FsgsRasterCfg.flags takes on the CURRENT device:
 * 1 = one wave per 16x16 tile, 4 = four waves per tile (csrc/raster_kernels.h use_quad_waves: the forward always 4; the backward
 * up to 3.25 tiles per SIMD of the device -- 4.25 for the pose-only backward -- unless FSGS_FLAG_BLEND_ONE_WAVE / _QUAD_WAVES force
 * one).  The backward's float summation order differs between the flavours, so profiles, bench lines and parity logs state
 * which one ran.  No reference counterpart (UPSTREAM has one blend kernel). */
int fsgs_blend_waves_per_tile(int width, int height, int flags, int backward, int pose_only);

/* bytes of `scratch` a backward call with FSGS_FLAG_DETERMINISTIC needs (covers both fsgs_raster_backward and
 * fsgs_render_backward*): the per-Gaussian rows + one 64-byte row per pair slot + the per-workgroup dL/dw2c partials */
size_t fsgs_deterministic_scratch_bytes(int P, int64_t max_pairs);

int fsgs_raster_sizes(int P, int width, int height, int64_t max_pairs,
                      size_t *state_bytes, size_t *scratch_bytes);

/* Byte offsets of the sub-buffers inside `state` (for tests / debugging / a viewer):
 * [0] xy float2[P], [1] conic+opacity float4[P], [2] depth float[P], [3] tile ranges int2[tiles],
 * [4] final_T float[H*W], [5] n_contrib uint32[H*W], [6] sorted Gaussian ids uint32[max_pairs]. */
int fsgs_raster_state_layout(int P, int width, int height, int64_t max_pairs, size_t offsets[7]);

/* Forward: kernels R1-R6 of SURVEY.md s2.1.
 *  means3D[P,3] colors[P,C] opacities[P] scales[P,3] rotations[P,4] (r,x,y,z), fp32 row-major.
 *  out_color[C,H,W] planar, out_depth[H,W] (depth-fork third output), radii[P] int32.
 *  max_pairs: capacity for (tile, Gaussian) pairs, split evenly over (tiles x 8) fixed-capacity list segments
 *  (single-pass binning: no count pass, no scan; a segment holds max_pairs / (8 tiles) keys).
 *  num_rendered [host]: receives R = number of (tile, Gaussian) pairs.  The host learns R from a pinned
 *  mailbox word the last binning kernel writes (no stream synchronisation; the call returns while the blend is
 *  still queued).  If a segment overflowed the call returns FSGS_ERR_CAPACITY, *num_rendered holds a max_pairs
 *  that would have sufficed, the outputs are garbage (the blend ran on truncated, in-bounds lists) and the caller
 *  retries with bigger buffers. */
int fsgs_raster_forward(const FsgsRasterCfg *cfg, int P,
                        const float *means3D, const float *colors, const float *opacities,
                        const float *scales, const float *rotations,
                        float *out_color, float *out_depth, int32_t *radii,
                        void *state, size_t state_bytes, void *scratch, size_t scratch_bytes,
                        int64_t max_pairs, int64_t *num_rendered, fsgs_stream_t stream);

/* Backward: kernels R7-R9.  dL_dcolor[C,H,W].  Outputs are overwritten:
 *  dmeans2D[P,3] (NDC-scaled screen gradient, z = 0), dcolors[P,C], dopacities[P],
 *  dmeans3D[P,3], dscales[P,3], drotations[P,4].  `scratch` >= P*32 bytes (fsgs_raster_sizes covers it). */
int fsgs_raster_backward(const FsgsRasterCfg *cfg, int P,
                         const float *means3D, const float *colors,
                         const float *scales, const float *rotations, const int32_t *radii,
                         const void *state, size_t state_bytes,
                         int64_t max_pairs, int64_t num_rendered, /* as passed to / returned by forward */
                         const float *dL_dcolor,
                         float *dmeans2D, float *dcolors, float *dopacities,
                         float *dmeans3D, float *dscales, float *drotations,
                         void *scratch, size_t scratch_bytes, fsgs_stream_t stream);

/* ---- fused render (gaussian_renderer.render body) ------------------------------------------------- */

/* Raw GaussianModel.params tensors (scene/gaussian_model.py:350-357) + the current pose.  All DEVICE
 * pointers, fp32, contiguous.  w2c is row-major [4,4] (the tensor PoseModel.get_pose returns);
 * cam_center [3] is PoseModel.cam_center (scene/pose_optimizer.py:603). */
typedef struct FsgsRenderArgs {
  const float *xyz;           /* [P,3]   */
  const float *features_dc;   /* [P,1,3] */
  const float *features_rest; /* [P,(max_sh_degree+1)^2-1,3] */
  const float *opacity;       /* [P,1] raw (pre-sigmoid)     */
  const float *scaling;       /* [P,3] raw (log)             */
  const float *rotation;      /* [P,4] raw (un-normalised)   */
  const float *w2c;           /* [4,4] */
  const float *cam_center;    /* [3]   */
  int32_t active_sh_degree;   /* 0..3 */
  int32_t max_sh_degree;      /* 0..3 */
} FsgsRenderArgs;

/* Gradient outputs of fsgs_render_backward, same layouts as FsgsRenderArgs; overwritten.
 * means2D [P,3] = the RGB pass's NDC-scaled screen gradient (`viewspace_points.grad`); may be NULL in the pose-only call
 * (gs_grad = param_grads = 0, cam_grad = 1: the tracking step), whose per-Gaussian pass then reduces straight to dL/dw2c.
 * w2c [4,4]: rows 0..2 = dL/dw2c, row 3 = 0. */
typedef struct FsgsRenderGrads {
  float *xyz, *features_dc, *features_rest, *opacity, *scaling, *rotation, *means2D, *w2c;
} FsgsRenderGrads;

int fsgs_render_sizes(int P, int width, int height, int64_t max_pairs, size_t *state_bytes, size_t *scratch_bytes);
/* Byte offsets inside the fused render's `state` (tests / debugging / a viewer): [0..6] as fsgs_raster_state_layout,
 * [7] the packed per-Gaussian records float[P,16] the blend kernels gather, ONE 64-byte line per Gaussian:
 *     floats 0,1 mean2D (pixels) | 2,3,4 conic A,B,C | 5 opacity | 6 view depth | 7 - | 8..13 colours
 *     (r, g, b | z, 1, z^2) = clamp_min(eval_sh + 0.5, 0) and the depth / silhouette pseudo-colours
 *     (scene/gaussian_model.py:260-275,316-320) | 14,15 -,
 * [8] flag word uint32[P] (bit c = colour channel c was clamped at 0). */
int fsgs_render_state_layout(int P, int width, int height, int64_t max_pairs, size_t offsets[9]);

/* out_image [3,H,W] = the RGB pass; out_depth_sil [3,H,W] = (depth, silhouette, depth^2) pass of
 * gaussian_renderer/__init__.py:68-73; radii [P].  cfg->channels is ignored; cfg->bg[0..2] is used for
 * both passes (scene/pose_optimizer.py:624).  R reaches the host like in fsgs_raster_forward. */
int fsgs_render_forward(const FsgsRasterCfg *cfg, int P, const FsgsRenderArgs *args,
                        float *out_image, float *out_depth_sil, int32_t *radii,
                        void *state, size_t state_bytes, void *scratch, size_t scratch_bytes,
                        int64_t max_pairs, int64_t *num_rendered, fsgs_stream_t stream);
/* The same forward for a cloud whose parameters have NOT changed since an earlier forward `prev_state` (made with
 * prev_max_pairs; any pose): the per-Gaussian colours and clamp flags are copied from that state's packed records
 * instead of being evaluated from the 48 SH coefficients -- they depend on the parameters and the frame-0 camera centre
 * only (scene/gaussian_model.py:317-320).  For the 50 tracking iterations of a frame (train.py:166-200) and the second
 * view of a two-view mapping step (train.py:236-259).  prev_state must stay untouched until this call's kernels ran. */
int fsgs_render_forward_reuse_colors(const FsgsRasterCfg *cfg, int P, const FsgsRenderArgs *args,
                                     float *out_image, float *out_depth_sil, int32_t *radii,
                                     void *state, size_t state_bytes, void *scratch, size_t scratch_bytes,
                                     int64_t max_pairs, int64_t *num_rendered,
                                     const void *prev_state, size_t prev_state_bytes, int64_t prev_max_pairs,
                                     fsgs_stream_t stream);
/* A forward that takes the per-Gaussian colours from a colour cache colors4 [P,4] = (r, g, b, clamp flags as bits) written
 * by the Adam kernel of the PREVIOUS step right after it updated the parameters (FsgsFusedAdam.next_colors of
 * fsgs_render_backward_adam / fsgs_adam_step_compact[_sum]) -- the colours depend on the parameters and the frame-0
 * camera centre only (scene/gaussian_model.py:317-320).  Valid while neither has changed since that launch and
 * args->active_sh_degree is the one it ran with; the caller keeps track (fsgs_amd/fast_step.py checks identities
 * and version counters).  Bit-identical to fsgs_render_forward. */
int fsgs_render_forward_cached_colors(const FsgsRasterCfg *cfg, int P, const FsgsRenderArgs *args, float *out_image,
                                      float *out_depth_sil, int32_t *radii, void *state, size_t state_bytes,
                                      void *scratch, size_t scratch_bytes, int64_t max_pairs, int64_t *num_rendered,
                                      const float *colors4, fsgs_stream_t stream);

/* dL_dimage / dL_ddepth_sil [3,H,W] or NULL (= zero).  gs_grad / cam_grad as in render(...):
 * gs_grad routes the mean gradient to xyz, cam_grad reduces dL/dw2c.  param_grads = 0 skips the
 * gradients of features / opacity / scaling / rotation (pose-only backward of the tracking step,
 * observationally equivalent because train.py:220 discards them; SURVEY.md a1 note v).
 * scratch: >= P * 64 bytes (one 64-byte accumulator row per Gaussian), 16-byte aligned (FSGS_ERR_INVALID otherwise;
 * fsgs_render_sizes covers it). */
int fsgs_render_backward(const FsgsRasterCfg *cfg, int P, const FsgsRenderArgs *args, const int32_t *radii,
                         const void *state, size_t state_bytes, int64_t max_pairs, int64_t num_rendered,
                         const float *dL_dimage, const float *dL_ddepth_sil,
                         int gs_grad, int cam_grad, int param_grads, const FsgsRenderGrads *grads,
                         void *scratch, size_t scratch_bytes, fsgs_stream_t stream);

/* The same backward with gs_grad = 1, param_grads = 1, cam_grad = 0, FOLLOWED BY the Adam step of the six
 * GaussianModel groups (train.py:268-272), without materialising the gradients: every parameter element is updated
 * in place (args->xyz ... args->rotation are written!) together with its two moments.  Only valid when this is
 * the step's single contribution to the gradient (one view, one rank).  Arithmetic identical to
 * fsgs_render_backward + fsgs_adam_step.  Group order of the arrays: xyz, features_dc, features_rest, opacity,
 * scaling, rotation; step = the step count INCLUDING this step (>= 1).  means2D_grad [P,3] (the RGB pass's
 * dL/dmeans2D for the densification statistic) is still written, unless it is NULL: then nobody wants the
 * statistic (densification is over, iteration >= 15000 in train.py:305) and its per-pixel terms are skipped. */
typedef struct FsgsFusedAdam {
  float *exp_avg[6];
  float *exp_avg_sq[6];
  float lr[6];
  int32_t step[6];
  double beta1, beta2, eps;
  float *next_colors; /* optional [P,4]: filled with (r, g, b, clamp flags as bits) = clamp_min(eval_sh + 0.5, 0) of the
                       * UPDATED parameters, for fsgs_render_forward_cached_colors of the next step; NULL = not wanted */
} FsgsFusedAdam;
/* What train.py does between loss.backward() and optimizer.step() on the same iteration, folded into the same
 * launch (all optional; tail == NULL or NULL members = skipped):
 *  - add_densification_stats + the max_radii2D update (train.py:298-303, scene/gaussian_model.py:678-681) for the
 *    Gaussians with radii > 0:  max_radii2D = max(., radius), xyz_gradient_accum += ||dL/dmeans2D (RGB pass)||,
 *    denom += 1   (what fsgs_densify_stats does from the means2D_grad tensor in a launch of its own);
 *  - the scalar loss of the iteration, loss_total[0] = sum_k loss_terms[k] * loss_weights[k] (n_terms <= 16 device
 *    floats each, written by the loss kernels earlier on the stream): reporting only, nothing reads it back.
 *  With max_radii2D alone (xyz_gradient_accum = denom = NULL) only the radius maximum is raised: a further view of a
 *  multi-view step (render() updates max_radii2D for every view it renders, gaussian_renderer/__init__.py:79). */
typedef struct FsgsStepTail {
  float *max_radii2D;         /* [P] */
  float *xyz_gradient_accum;  /* [P] */
  float *denom;               /* [P] */
  const float *loss_terms;
  const float *loss_weights;
  int32_t n_terms;
  float *loss_total;
} FsgsStepTail;
int fsgs_render_backward_adam(const FsgsRasterCfg *cfg, int P, const FsgsRenderArgs *args, const int32_t *radii,
                              const void *state, size_t state_bytes, int64_t max_pairs, int64_t num_rendered,
                              const float *dL_dimage, const float *dL_ddepth_sil, const FsgsFusedAdam *adam,
                              float *means2D_grad, const FsgsStepTail *tail, void *scratch, size_t scratch_bytes,
                              fsgs_stream_t stream);

/* Compact gradient for steps that sum several views and / or several ranks: gcompact [P,14] =
 * [d xyz (3) | gcol (3) | d opacity | d scaling (3) | d rotation (4)] where gcol is the clamped dL/dcolour.  The
 * gradient of the 48 SH coefficients of a Gaussian is basis_k(direction) x gcol_c with a view-independent basis
 * (world position, frame-0 camera centre; scene/gaussian_model.py:317-318), so sums over views and ranks only
 * need gcol: 56 B per Gaussian cross xGMI instead of 236 B.  gs_grad = 1, param_grads = 1, cam_grad = 0. */
int fsgs_render_backward_compact(const FsgsRasterCfg *cfg, int P, const FsgsRenderArgs *args, const int32_t *radii,
                                 const void *state, size_t state_bytes, int64_t max_pairs, int64_t num_rendered,
                                 const float *dL_dimage, const float *dL_ddepth_sil, float *gcompact,
                                 float *means2D_grad, const FsgsStepTail *tail, void *scratch, size_t scratch_bytes,
                                 fsgs_stream_t stream);
/* The same in ROW CHUNKS, for producer-side pipelining of the multi-GPU step: the first call (first = 1) runs the blend
 * backward over the whole image and the per-Gaussian backward of Gaussians [row_lo, row_hi); every further call
 * (first = 0) only the per-Gaussian backward of its rows.  row_lo must be a multiple of 256.  The caller records an
 * event after each call and starts that chunk's all-reduce (56 B per Gaussian) on a second stream while the next
 * chunk is still being produced; fsgs_adam_step_compact then consumes chunk by chunk (fsgs_amd/dist.py). */
int fsgs_render_backward_compact_rows(const FsgsRasterCfg *cfg, int P, const FsgsRenderArgs *args, const int32_t *radii,
                                      const void *state, size_t state_bytes, int64_t max_pairs, int64_t num_rendered,
                                      const float *dL_dimage, const float *dL_ddepth_sil, float *gcompact,
                                      float *means2D_grad, const FsgsStepTail *tail, void *scratch,
                                      size_t scratch_bytes, int row_lo, int row_hi, int first, fsgs_stream_t stream);
/* Adam step of the six groups from the (summed / all-reduced) compact gradient: the SH outer products are formed
 * on the fly.  Updates args->xyz ... args->rotation and the moments in place; args->w2c is not read. */
int fsgs_adam_step_compact(int P, const FsgsRenderArgs *args, const float *gcompact, const FsgsFusedAdam *adam,
                           fsgs_stream_t stream);
/* The same from the SUM of two compact gradients (gcompact2 may be NULL): the two views of the reference's progressive
 * mapping iteration (train.py:236-259, summed loss) run their backwards on two streams into two buffers, and the
 * sum is formed here instead of by a launch of its own. */
int fsgs_adam_step_compact_sum(int P, const FsgsRenderArgs *args, const float *gcompact, const float *gcompact2,
                               const FsgsFusedAdam *adam, fsgs_stream_t stream);

/* ---- simple-knn -------------------------------------------------------------- */

/* Mean squared distance to the 3 nearest neighbours, exact (distCUDA2).
 * points[P,3] -> out[P].  Call with scratch == NULL to query *scratch_bytes. */
int fsgs_knn_meandist2(int P, const float *points, float *out,
                       void *scratch, size_t *scratch_bytes, fsgs_stream_t stream);

/* ---- photometric loss: 0.8 L1 + 0.2 (1 - SSIM)  (utils/loss_utils.py:41-96) ------------------- */

/* img, gt [C,H,W]; mask [H,W] or NULL (multiplies both images, utils/loss_utils.py:48-50); presence [H,W] or NULL:
 * the mask is additionally zero where presence <= 0 -- the tracking step's `render_dep > 0` factor (train.py:176-178)
 * read straight from the rendered depth plane instead of being materialised by an elementwise kernel.
 * maps [3,C,H,W] (kept for backward), scratch of fsgs_photometric_scratch_bytes(C,H,W) bytes (per-workgroup
 * partial sums), out3 float[3] = {loss, L1, SSIM} on the DEVICE (no host sync). */
size_t fsgs_photometric_scratch_bytes(int C, int H, int W);
int fsgs_photometric_loss_forward(int C, int H, int W, const float *img, const float *gt, const float *mask,
                                  const float *presence, float lambda_dssim, float *maps, void *scratch, float *out3,
                                  fsgs_stream_t stream);
/* dimg [C,H,W] = upstream[0] * dloss/dimg; upstream is a DEVICE scalar (NULL = 1). */
int fsgs_photometric_loss_backward(int C, int H, int W, const float *img, const float *gt, const float *mask,
                                   const float *presence, const float *maps, const float *upstream, float lambda_dssim,
                                   float *dimg, fsgs_stream_t stream);
/* Forward and backward of the same loss in TWO launches instead of three: the reduction of the per-workgroup partial
 * sums to out3 runs as one extra workgroup of the backward launch instead of as a launch of its own between the two.
 * Same results as the two calls above (for callers that need dimg whatever the loss value is: the step driver). */
int fsgs_photometric_loss_forward_backward(int C, int H, int W, const float *img, const float *gt, const float *mask,
                                           const float *presence, float lambda_dssim, float *maps, void *scratch,
                                           float *out3, const float *upstream, float *dimg, fsgs_stream_t stream);

/* ---- Pearson depth losses (utils/loss_utils.py:98-127) ------------------------------------------- */

/* Global loss over the whole [H,W] image and the mean loss over n_patches (<= 64) box x box patches
 * whose top-left corners (row, col) are int64 DEVICE arrays (what torch.randint produced).
 * scratch: fsgs_pearson_scratch_bytes(H, W, n_patches, box) bytes of per-workgroup partial sums (need not
 * be zeroed; the sums are reduced in a fixed order, so the result is run-to-run deterministic),
 * coef float[8*(n_patches+1)] (kept for backward), out2 float[2] = {global loss, mean patch loss} on the
 * device. */
size_t fsgs_pearson_scratch_bytes(int H, int W, int n_patches, int box);
int fsgs_pearson_forward(int H, int W, int n_patches, int box, const int64_t *patch_row0, const int64_t *patch_col0,
                         const float *src, const float *tgt, void *scratch, float *coef, float *out2,
                         fsgs_stream_t stream);
/* grad [H,W] = sum_r region_weight[r] * dloss_r/d(tgt or src); region_weight float[n_patches+1] on the
 * device (r = 0 global).  wrt_src = 0: gradient w.r.t. tgt (the rendered depth, train.py:256-257). */
int fsgs_pearson_backward(int H, int W, int n_patches, int box, const int64_t *patch_row0, const int64_t *patch_col0,
                          const float *src, const float *tgt, const float *coef, const float *region_weight,
                          int wrt_src, float *grad, fsgs_stream_t stream);

/* One view's whole loss stage (train.py:236-258: rgb loss of the rendered image + global and local Pearson losses of the
 * rendered depth against the mono-depth) in TWO launches on ONE stream: the Pearson statistics ride in the photometric
 * forward's launch and the Pearson gradient (whose waves finish the regions' sums themselves) in the backward's, as
 * extra workgroups at the front of each grid -- no second stream, no fork / join events.  Same results, bit for bit, as
 * fsgs_photometric_loss_forward_backward(img, gt, ...) + fsgs_pearson_forward(src, tgt, ...) +
 * fsgs_pearson_backward(..., wrt_src = 0, grad_tgt); arguments as there.  n_patches <= 63 and box <= 128 (at most
 * 4 partial sums per patch), else FSGS_ERR_INVALID: use the three calls. */
int fsgs_view_losses_forward_backward(int C, int H, int W, const float *img, const float *gt, const float *mask,
                                      const float *presence, float lambda_dssim, float *maps, void *photo_scratch,
                                      float *out3, const float *upstream, float *dimg, int n_patches, int box,
                                      const int64_t *patch_row0, const int64_t *patch_col0, const float *src,
                                      const float *tgt, void *pearson_scratch, float *coef, float *out2,
                                      const float *region_weight, float *grad_tgt, fsgs_stream_t stream);

/* ---- optical-flow reprojection loss of the tracking step (scene/pose_optimizer.py:164-218) ----------- */

/* pts_world [M,3] and pix_vu int64 [M,2] = (v, u): the back-projected valid pixels of the previous frame
 * (pose independent, prepared once per frame).  w2c [4,4] DEVICE row-major current pose; K9_host = row-major
 * 3x3 intrinsics on the HOST; flow_fw [2,H,W] (channel 0 = du, 1 = dv).  edge = 20 in the reference.
 * acc3 double[3] scratch kept for backward; out2 float[2] = {loss, #valid} on the device. */
int fsgs_flow_pose_loss_forward(int64_t M, const float *pts_world, const int64_t *pix_vu, const float *w2c,
                                const float *K9_host, const float *flow_fw, int W, int H, float edge,
                                double *acc3, float *out2, fsgs_stream_t stream);
/* dw2c [4,4] (rows 0..2 = upstream * dloss/dw2c, row 3 = 0); upstream DEVICE scalar or NULL. */
int fsgs_flow_pose_loss_backward(int64_t M, const float *pts_world, const int64_t *pix_vu, const float *w2c,
                                 const float *K9_host, const float *flow_fw, int W, int H, float edge,
                                 const double *acc3, const float *upstream, float *dw2c, fsgs_stream_t stream);

/* Flow targets of a tracked frame, once per frame (get_pointcloud + duplicate rejection,
 * scene/pose_optimizer.py:42-73,171-181), in three launches around a caller-side sort and scan:
 *  keys   : depth_prev [H,W], rigid uint8 [H,W] or NULL, K9 / c2w16 = inverse(w2c_prev) row-major on the HOST ->
 *           world [HW,3], rounded [HW,3] = |round(world, 4)|, keys int64 [HW] (hash of the rounded triple; -1 for
 *           pixels with depth * rigid <= 0);
 *  (caller: sort keys ascending carrying the pixel indices)
 *  flag   : keep int32 [HW] = 1 for valid points whose rounded triple is unique and not the origin;
 *  (caller: incl = inclusive prefix sum of keep; M = incl[HW-1])
 *  gather : pts [M,3], pix_vu int64 [M,2] = (v, u), in pixel order like the reference's boolean indexing. */
int fsgs_flow_targets_keys(int H, int W, const float *depth_prev, const uint8_t *rigid, const float *K9_host,
                           const float *c2w16_host, float *world, float *rounded, int64_t *keys,
                           fsgs_stream_t stream);
int fsgs_flow_targets_flag(int64_t HW, const int64_t *sorted_keys, const int64_t *sorted_idx, const float *rounded,
                           int32_t *keep, fsgs_stream_t stream);
int fsgs_flow_targets_gather(int H, int W, const int32_t *keep, const int32_t *incl, const float *world, float *pts,
                             int64_t *pix_vu, fsgs_stream_t stream);

/* Forward and backward in one streaming pass (the autograd-free tracking step): out2 = {loss, #valid} and
 * dw2c[4,4] = accumulate * dw2c + upstream * dloss/dw2c (accumulate = 0 overwrites without reading dw2c, so the
 * caller can fold the weighted sum with the rasteriser's pose gradient into this call).  upstream / accumulate
 * are HOST scalars (the loss weights).  scratch: fsgs_flow_scratch_bytes(M) bytes, need not be zeroed; partial
 * sums are added in a fixed order. */
size_t fsgs_flow_scratch_bytes(int64_t M);
int fsgs_flow_pose_loss_fused(int64_t M, const float *pts_world, const int64_t *pix_vu, const float *w2c,
                              const float *K9_host, const float *flow_fw, int W, int H, float edge, float upstream,
                              float accumulate, void *scratch, float *out2, float *dw2c, fsgs_stream_t stream);

/* Sampson-distance rigid mask of a tracked frame (train.py:157-165; scene/pose_optimizer.py:700-746;
 * utils/general_utils.py:96-116).  flow_fw [2,H,W] forward flow of the earlier frame, F9_host the row-major 3x3
 * fundamental matrix (HOST).  dist [H,W] squared Sampson distance; rigid uint8 [H,W] = (dist <= mean +
 * factor * std) & (dist < 1) -- the reference's `dist < adaptive_thresholding(dist)`; stats3 float[3] on the
 * device = {mean, std (unbiased), threshold}.  scratch: fsgs_sampson_scratch_bytes(H, W).  No host sync. */
size_t fsgs_sampson_scratch_bytes(int H, int W);
int fsgs_sampson_rigid_mask(int H, int W, const float *flow_fw, const float *F9_host, float factor, void *scratch,
                            float *dist, uint8_t *rigid, float *stats3, fsgs_stream_t stream);

/* ---- LearnPose.forward and its adjoint (scene/pose_optimizer.py:822-877) --------------------------------- */

/* r [1,4,N] quaternions (r,x,y,z), t [3,N] translations, cam_id -> w2c [4,4] row-major (all DEVICE). */
int fsgs_pose_forward(const float *r, const float *t, int num_cams, int cam_id, float *w2c, fsgs_stream_t stream);
/* dw2c [4,4] -> dr [1,4,N], dt [3,N] (overwritten; zero except column cam_id). */
int fsgs_pose_backward(const float *r, int num_cams, int cam_id, const float *dw2c, float *dr, float *dt,
                       fsgs_stream_t stream);
/* The tail of a tracking iteration in ONE launch (train.py:186-195): dW = weight_a * dw2c_a + dw2c_b (dw2c_b may be
 * NULL), the adjoint of LearnPose.forward for cam_id, torch.optim.Adam over all of r [1,4,N] and t [3,N] (the
 * gradient is zero outside column cam_id, as autograd would report it) with the moments of the two tensors, and
 * the updated pose of cam_id written to w2c_next [4,4] (may be NULL).  step_* = the step counts INCLUDING this
 * step; lr_* = the scheduled learning rates. */
int fsgs_pose_adam_step(float *r, float *t, int num_cams, int cam_id, const float *dw2c_a, float weight_a,
                        const float *dw2c_b, float *exp_avg_r, float *exp_avg_sq_r, float *exp_avg_t,
                        float *exp_avg_sq_t, float lr_r, float lr_t, int step_r, int step_t, double beta1, double beta2,
                        double eps, float *w2c_next, fsgs_stream_t stream);

/* The start of a tracked frame in ONE launch (train.py:322-331): PoseModel.initialize_pose for frame cam_id
 * (scene/pose_optimizer.py:498-516: with `extrapolate` and cam_id > 1 the constant-velocity rule
 * r_i = normalize(normalize(r_{i-1}) + (normalize(r_{i-1}) - normalize(r_{i-2}))), t_i = 2 t_{i-1} - t_{i-2}; otherwise a
 * copy of frame cam_id - 1; cam_id = 0 leaves the poses alone) and the state of the fresh Adam the reference builds per
 * frame (initialize_tracking_optimizer, :489-496): the four moment tensors are zeroed when given (NULL pairs are left
 * alone; the caller restarts its step counters).  All pointers DEVICE; r [1,4,N], t [3,N] as in fsgs_pose_forward. */
int fsgs_pose_frame_begin(float *r, float *t, int num_cams, int cam_id, int extrapolate, float *exp_avg_r,
                          float *exp_avg_sq_r, float *exp_avg_t, float *exp_avg_sq_t, fsgs_stream_t stream);

/* ---- optimiser step and densification statistics -------------------------------------------------- */

/* One parameter group of torch.optim.Adam (no weight decay, no amsgrad): all DEVICE pointers of n
 * fp32 elements; step = the 1-based step count AFTER this update (torch's state['step']). */
typedef struct FsgsAdamGroup {
  float *param;
  const float *grad;
  float *exp_avg;
  float *exp_avg_sq;
  int64_t n;
  float lr;
  int32_t step;
} FsgsAdamGroup;

/* groups: HOST array of <= 8 groups, updated by ONE kernel launch (train.py:194,272). */
int fsgs_adam_step(int ngroups, const FsgsAdamGroup *groups, double beta1, double beta2, double eps,
                   fsgs_stream_t stream);

/* For every Gaussian with radii > 0: max_radii2D = max(., radii); xyz_gradient_accum += ||viewspace_grad||;
 * denom += 1   (scene/gaussian_model.py:678-681, train.py:298-303).  viewspace_grad [P,3]. */
int fsgs_densify_stats(int P, const int32_t *radii, const float *viewspace_grad, float *max_radii2D,
                       float *xyz_gradient_accum, float *denom, fsgs_stream_t stream);

/* ---- device-side densify / prune with optimizer-state compaction (scene/gaussian_model.py:523-676) --------- */

/* Step 1: the per-Gaussian decisions of GaussianModel.densify_and_prune(max_grad, min_opacity, max_screen_size).
 * scaling [P,3] / opacity [P] are the RAW parameters (_scaling, _opacity).  small_extent = 0.01 * scene_radius
 * (clone vs split), big_extent = 0.1 * scene_radius, prune_big = (max_screen_size != 0).  The reference's
 * screen-size term compares max_radii2D AFTER densification_postfix has zeroed it, i.e. it never fires; it is
 * therefore not an input.  counts4 int32 [4,P]: original kept | clone kept | selected for the split | children
 * kept (per copy).  The caller forms their inclusive prefix sums (incl4, same shape) and reads the four totals. */
int fsgs_densify_plan(int P, const float *xyz_gradient_accum, const float *denom, const float *scaling,
                      const float *opacity, float max_grad, float min_opacity, float small_extent, float big_extent,
                      int prune_big, int32_t *counts4, fsgs_stream_t stream);

#define FSGS_DENSIFY_ROLE_PLAIN 0
#define FSGS_DENSIFY_ROLE_XYZ 1
#define FSGS_DENSIFY_ROLE_SCALING 2
#define FSGS_DENSIFY_ROLE_ROTATION 3
typedef struct FsgsDensifyGroup {
  const float *in_param;    /* [P, row] */
  const float *in_exp_avg;  /* Adam moments of the old tensor, or NULL (no optimizer state yet) */
  const float *in_exp_avg_sq;
  float *out_param;         /* [P', row], P' = totals[0] + totals[1] + 2 * totals[3] */
  float *out_exp_avg;       /* or NULL: kept rows keep their moments, new rows get zeros */
  float *out_exp_avg_sq;
  int32_t row;              /* floats per Gaussian */
  int32_t role;             /* FSGS_DENSIFY_ROLE_*: xyz / scaling rows of split children are recomputed */
} FsgsDensifyGroup;

/* Step 2: gather every group into its new buffers in the reference's output order
 * [kept originals | kept clones | kept children copy 1 | kept children copy 2].  totals4 (HOST) = last column of
 * incl4.  normals [2 * totals4[2], 3]: the N(0,1) draws, in the row order of the reference's torch.normal call
 * (all selected, copy-major).  src / aux: int32 [P'] scratch. */
int fsgs_densify_apply(int P, const int32_t *counts4, const int32_t *incl4, const int32_t totals4[4], int ngroups,
                       const FsgsDensifyGroup *groups, const float *normals, int32_t *src, int32_t *aux,
                       fsgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FSGS_H */
