// raster.hip -- tile-binned differentiable Gaussian-splat rasteriser for gfx950 (MI355X).
//
// Replaces the un-vendored CUDA extension `diff_gaussian_rasterization`
// (requirements.txt:26) that Free-SurGS calls at gaussian_renderer/__init__.py:68-69,131.
// Semantics = SURVEY.md s2.1 R1-R9 / Appendix A; the design is not the upstream one:
//
//   * binning: Gaussians are radix-sorted by depth ONCE (P 32-bit keys), pairs are emitted
//     in that order and a stable 13-bit radix sort by tile id finishes the job -- two passes
//     over R 8-byte pairs instead of six passes over R 12-byte (u64,u32) pairs.  The
//     resulting order (tile, depth, index) is the upstream order, ties included.
//   * blending: ONE 64-lane wavefront owns a 16x16 tile, 4 pixels per lane.  The tile's
//     Gaussian list is consumed 64 at a time: lane j gathers record j into registers and
//     the inner loop broadcasts it through SGPRs with v_readlane -- no LDS staging, no
//     barriers, and every evaluation runs with 4-way ILP.
//   * backward: per-Gaussian partial gradients are summed over the lane's 4 pixels in
//     registers, reduced across the wave with DPP row operations, parked in lane j with
//     v_writelane and flushed with one atomic per (tile, Gaussian, component) -- 256x
//     fewer atomics than one-per-pixel.
#include "raster_kernels.h"

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int fsgs_blend_waves_per_tile(int width, int height, int flags, int backward, int pose_only) {
  if (width <= 0 || height <= 0) return FSGS_ERR_INVALID;
  const int ntiles = ((width + FSGS_TILE - 1) / FSGS_TILE) * ((height + FSGS_TILE - 1) / FSGS_TILE);
  return use_quad_waves((uint32_t)flags, ntiles, backward != 0, pose_only != 0) ? 4 : 1;
}

size_t fsgs_deterministic_scratch_bytes(int P, int64_t max_pairs) {
  if (P < 0 || max_pairs < 0) return 0;
  return det_layout(P, max_pairs).total;
}

int fsgs_raster_sizes(int P, int width, int height, int64_t max_pairs, size_t *state_bytes, size_t *scratch_bytes) {
  if (P < 0 || width <= 0 || height <= 0 || max_pairs < 0 || !state_bytes || !scratch_bytes) return FSGS_ERR_INVALID;
  *state_bytes = state_layout(P, width, height, max_pairs).total;
  ScratchLayout sl;
  scratch_layout(P, width, height, max_pairs, sl);
  // backward needs P * 8 floats of accumulators; make one scratch size serve both directions
  size_t bwd = (size_t)(P > 0 ? P : 1) * kAccStride * sizeof(float) + 256;
  *scratch_bytes = sl.total_bytes > bwd ? sl.total_bytes : bwd;
  return FSGS_OK;
}

int fsgs_raster_state_layout(int P, int width, int height, int64_t max_pairs, size_t offsets[7]) {
  if (P < 0 || width <= 0 || height <= 0 || max_pairs < 0 || !offsets) return FSGS_ERR_INVALID;
  StateLayout L = state_layout(P, width, height, max_pairs);
  offsets[0] = L.xy; offsets[1] = L.conic_op; offsets[2] = L.depth; offsets[3] = L.ranges;
  offsets[4] = L.final_T; offsets[5] = L.n_contrib; offsets[6] = L.plist;
  return FSGS_OK;
}

int fsgs_raster_forward(const FsgsRasterCfg *cfg, int P, const float *means3D, const float *colors,
                        const float *opacities, const float *scales, const float *rotations, float *out_color,
                        float *out_depth, int32_t *radii, void *state, size_t state_bytes, void *scratch,
                        size_t scratch_bytes, int64_t max_pairs, int64_t *num_rendered, fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!cfg || P < 0 || !out_color || !out_depth || !state || !scratch || !num_rendered || max_pairs < 0)
    return FSGS_ERR_INVALID;
  if (P > 0 && (!means3D || !colors || !opacities || !scales || !rotations || !radii)) return FSGS_ERR_INVALID;
  const int C = cfg->channels, W = cfg->image_width, H = cfg->image_height;
  if (W <= 0 || H <= 0 || !(C == 1 || C == 3 || C == 6)) return FSGS_ERR_INVALID;
  CamParams cam = make_cam(cfg);
  const int ntiles = cam.gx * cam.gy;
  FwdBuffers B;
  int rc = bind_forward_buffers(P, W, H, max_pairs, 0, state, state_bytes, scratch, scratch_bytes, B);
  if (rc != FSGS_OK) return rc;
  if (P > 0) {
    ProfScope ps(PROF_PREPROCESS_FWD, stream);
    GeomOut g{B.xy, B.co, B.depth, B.rec, radii, B.tiles, B.rect, B.tile_count, cam.gx, binning_clear_words(ntiles)};
    switch (C) {  // the colours travel into the packed per-Gaussian record the blend kernels gather
      case 1: hipLaunchKernelGGL(preprocess_fwd_kernel<1>, dim3((P + 255) / 256), dim3(256), 0, stream, P, cam, means3D, colors, opacities, scales, rotations, g); break;
      case 3: hipLaunchKernelGGL(preprocess_fwd_kernel<3>, dim3((P + 255) / 256), dim3(256), 0, stream, P, cam, means3D, colors, opacities, scales, rotations, g); break;
      case 6: hipLaunchKernelGGL(preprocess_fwd_kernel<6>, dim3((P + 255) / 256), dim3(256), 0, stream, P, cam, means3D, colors, opacities, scales, rotations, g); break;
    }
  }
  FSGS_HIP(hipGetLastError());
  BinningTicket tk;
  rc = enqueue_binning(cam, P, B, max_pairs, tk, stream, /*cursors_cleared=*/true);  // by preprocess_fwd_kernel
  if (rc == FSGS_ERR_CAPACITY) *num_rendered = (int64_t)ntiles * BIN_SUBS * 32;  // not even one key per segment
  if (rc != FSGS_OK) return rc;
  const int2 *ranges = B.ranges;
  const uint32_t *order = B.order;
  const uint32_t *plist = B.plist;
  const float4 *rec = B.rec;
  float *final_T = B.final_T;
  uint32_t *n_contrib = B.n_contrib;
  ProfScope ps_blend(PROF_BLEND_FWD, stream);
  switch (C) {
    case 1: launch_blend_fwd<1>(cam, ntiles, order, ranges, plist, rec, final_T, n_contrib, out_color, out_color + 3 * (size_t)W * H, out_depth, stream); break;
    case 3: launch_blend_fwd<3>(cam, ntiles, order, ranges, plist, rec, final_T, n_contrib, out_color, out_color + 3 * (size_t)W * H, out_depth, stream); break;
    case 6: launch_blend_fwd<6>(cam, ntiles, order, ranges, plist, rec, final_T, n_contrib, out_color, out_color + 3 * (size_t)W * H, out_depth, stream); break;
  }
  FSGS_HIP(hipGetLastError());
  // only now does the host look at R (the blend is already queued behind the binning)
  return finish_binning(cam, B, max_pairs, tk, num_rendered, stream);
}

int fsgs_raster_backward(const FsgsRasterCfg *cfg, int P, const float *means3D, const float *colors,
                         const float *scales, const float *rotations, const int32_t *radii, const void *state,
                         size_t state_bytes, int64_t max_pairs, int64_t num_rendered, const float *dL_dcolor,
                         float *dmeans2D, float *dcolors,
                         float *dopacities, float *dmeans3D, float *dscales, float *drotations, void *scratch,
                         size_t scratch_bytes, fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!cfg || P < 0 || !state || !dL_dcolor) return FSGS_ERR_INVALID;
  if (P == 0) return FSGS_OK;
  if (!means3D || !colors || !scales || !rotations || !radii || !dmeans2D || !dcolors || !dopacities || !dmeans3D ||
      !dscales || !drotations || !scratch)
    return FSGS_ERR_INVALID;
  const int C = cfg->channels, W = cfg->image_width, H = cfg->image_height;
  if (!(C == 1 || C == 3 || C == 6)) return FSGS_ERR_INVALID;
  if (max_pairs < 0 || num_rendered < 0 || num_rendered > max_pairs) return FSGS_ERR_STATE;
  StateLayout SL = state_layout(P, W, H, max_pairs);
  if (state_bytes < SL.total) return FSGS_ERR_STATE;
  if (scratch_bytes < (size_t)P * kAccStride * sizeof(float)) return FSGS_ERR_CAPACITY;
  const bool deterministic = (cfg->flags & FSGS_FLAG_DETERMINISTIC) != 0;
  const DetLayout DL = det_layout(P, max_pairs);
  if (deterministic && (((uintptr_t)scratch) & 15)) return FSGS_ERR_INVALID;  // (as fsgs_render_backward*: a bad pointer, not a size)
  if (deterministic && scratch_bytes < DL.total) return FSGS_ERR_CAPACITY;
  CamParams cam = make_cam(cfg);
  const int ntiles = cam.gx * cam.gy;
  cam.bwd_prio_step = blend_bwd_prio_step(ntiles, num_rendered);
  const char *sb = (const char *)state;
  const float4 *co = (const float4 *)(sb + SL.conic_op);
  const float4 *rec = (const float4 *)(sb + SL.rec);  // colours as they were at the forward
  const int2 *ranges = (const int2 *)(sb + SL.ranges);
  const uint32_t *order = (ntiles <= ORDER_MAX_TILES && num_rendered > 0) ? (const uint32_t *)(sb + SL.order) : nullptr;
  const float *final_T = (const float *)(sb + SL.final_T);
  const uint32_t *n_contrib = (const uint32_t *)(sb + SL.n_contrib);
  const uint32_t *plist = (const uint32_t *)(sb + SL.plist);
  float *grad_acc = (float *)scratch;
  FSGS_HIP(hipMemsetAsync(grad_acc, 0, (size_t)P * kAccStride * sizeof(float), stream));
  FSGS_HIP(hipMemsetAsync(dcolors, 0, (size_t)P * C * sizeof(float), stream));
  if (num_rendered > 0) {
    ProfScope ps(PROF_BLEND_BWD, stream);
    const DetGather dg{(float *)((char *)scratch + DL.pair_rows), max_pairs, P, radii, (const float2 *)(sb + SL.xy),
                       (const float *)(sb + SL.depth)};
    const DetGather *det = deterministic ? &dg : nullptr;
    int rc = 0;
    switch (C) {
      case 1: rc = launch_blend_bwd<1>(cam, ntiles, order, ranges, plist, rec, final_T, n_contrib, dL_dcolor, dL_dcolor + 3 * (size_t)W * H, grad_acc, dcolors, stream, nullptr, det); break;
      case 3: rc = launch_blend_bwd<3>(cam, ntiles, order, ranges, plist, rec, final_T, n_contrib, dL_dcolor, dL_dcolor + 3 * (size_t)W * H, grad_acc, dcolors, stream, nullptr, det); break;
      case 6: rc = launch_blend_bwd<6>(cam, ntiles, order, ranges, plist, rec, final_T, n_contrib, dL_dcolor, dL_dcolor + 3 * (size_t)W * H, grad_acc, dcolors, stream, nullptr, det); break;
    }
    if (rc != 0) return rc;
    FSGS_HIP(hipGetLastError());
  }
  {
    ProfScope ps(PROF_PREPROCESS_BWD, stream);
    hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, cam, means3D, scales,
                       rotations, radii, co, grad_acc, dmeans2D, dopacities, dmeans3D, dscales, drotations);
  }
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

}  // extern "C"
