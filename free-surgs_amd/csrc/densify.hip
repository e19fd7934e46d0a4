// densify.hip -- device-side densify / prune with optimizer-state compaction for gfx950 (MI355X).
//
// Reference: GaussianModel.densify_and_prune and its helpers (scene/gaussian_model.py:523-676; called from
// train.py:297-316).  The reference runs clone -> cat -> split -> cat -> prune -> prune as ~12 boolean-index /
// cat reallocations of every parameter tensor and both Adam moments.  Every decision of that sequence is a
// function of ONE original Gaussian, so it collapses into
//   plan   : per Gaussian, the five decisions (cloned? split? original kept? clone kept? children kept?)
//   (scan) : exclusive prefix sums of four counters (done by the caller; one cumsum over a [4,P] tensor)
//   map    : every surviving output row j gets (source row, kind / normal-sample row)
//   apply  : ONE launch gathers all six parameter tensors and their two Adam moments into the new buffers,
//            computing the split children's position and scale on the way
// with the output ORDER of the reference: [kept originals | kept clones | kept children copy 1 | copy 2].
// The N(0,1) draws stay with torch (same generator, same count and order as the reference's torch.normal).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <cstring>

#include "../../include/fsgs.h"
#include "fsgs_device.h"
#include "fsgs_host.h"

using namespace fsgs;

namespace {

__device__ __forceinline__ float sigmoid_torch(float x) { return 1.0f / (1.0f + expf(-x)); }
// get_scaling / (0.8 * N), N = 2: torch divides a device tensor by a host scalar as a multiplication with the
// scalar's fp32 reciprocal
__device__ __forceinline__ float child_scale(float s) { return s * (1.0f / 1.6f); }

// counts[0][i] original kept, [1][i] clone kept, [2][i] selected for the split, [3][i] children kept (per copy)
__global__ __launch_bounds__(256) void densify_plan_kernel(int P, const float *__restrict__ accum,
                                                           const float *__restrict__ denom,
                                                           const float *__restrict__ scaling,
                                                           const float *__restrict__ opacity, float max_grad,
                                                           float min_opacity, float small_extent, float big_extent,
                                                           int prune_big, int32_t *__restrict__ counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float g = accum[i] / denom[i];  // NaN for never-seen Gaussians: both comparisons below are false
  const float s0 = expf(scaling[3 * i]), s1 = expf(scaling[3 * i + 1]), s2 = expf(scaling[3 * i + 2]);
  const float smax = fmaxf(fmaxf(s0, s1), s2);
  // densify_and_clone: torch.norm(grads, dim=-1) of a one-element row = sqrt(g*g)
  const bool clone_sel = (sqrtf(g * g) >= max_grad) && (smax <= small_extent);
  // densify_and_split: padded_grad >= thr (clones sit behind the originals with zero gradient)
  const bool split_sel = (g >= max_grad) && (smax > small_extent);
  // final prune (after both appends): opacity < min | world-space size too big.  max_radii2D was reset to zero
  // by densification_postfix before this test, so the reference's screen-size term is always false.
  const bool low_op = sigmoid_torch(opacity[i]) < min_opacity;
  const bool pruned_self = low_op || (prune_big && smax > big_extent);
  // a child's scale is exp(log(scale / 1.6)) when get_scaling is evaluated again
  const float c0 = expf(logf(child_scale(s0))), c1 = expf(logf(child_scale(s1))), c2 = expf(logf(child_scale(s2)));
  const bool pruned_child = low_op || (prune_big && fmaxf(fmaxf(c0, c1), c2) > big_extent);
  counts[i] = (!split_sel && !pruned_self) ? 1 : 0;
  counts[(size_t)P + i] = (clone_sel && !pruned_self) ? 1 : 0;
  counts[2 * (size_t)P + i] = split_sel ? 1 : 0;
  counts[3 * (size_t)P + i] = (split_sel && !pruned_child) ? 1 : 0;
}

// incl = inclusive prefix sums of counts ([4,P] int32); totals: K0 kept originals, K1 clones, NS split, K3 children/copy
__global__ __launch_bounds__(256) void densify_map_kernel(int P, const int32_t *__restrict__ counts,
                                                          const int32_t *__restrict__ incl, int K0, int K1, int NS,
                                                          int K3, int32_t *__restrict__ src,
                                                          int32_t *__restrict__ aux) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const size_t Ps = (size_t)P;
  if (counts[i]) {
    const int j = incl[i] - 1;
    src[j] = i;
    aux[j] = -1;  // kept original: moments travel with it
  }
  if (counts[Ps + i]) {
    const int j = K0 + incl[Ps + i] - 1;
    src[j] = i;
    aux[j] = -2;  // clone: parameters copied, fresh moments
  }
  if (counts[3 * Ps + i]) {
    const int rank_s = incl[2 * Ps + i] - 1;   // row among ALL selected (the order of the normal samples)
    const int rank_k = incl[3 * Ps + i] - 1;   // row among the surviving children of one copy
    for (int c = 0; c < 2; c++) {
      const int j = K0 + K1 + c * K3 + rank_k;
      src[j] = i;
      aux[j] = c * NS + rank_s;  // >= 0: split child, row of its N(0,1) sample
    }
  }
}

constexpr int DENSIFY_MAX_GROUPS = 8;
struct DensifyTable {
  const float *in_p[DENSIFY_MAX_GROUPS], *in_m[DENSIFY_MAX_GROUPS], *in_v[DENSIFY_MAX_GROUPS];
  float *out_p[DENSIFY_MAX_GROUPS], *out_m[DENSIFY_MAX_GROUPS], *out_v[DENSIFY_MAX_GROUPS];
  int row[DENSIFY_MAX_GROUPS], role[DENSIFY_MAX_GROUPS];
  long long elem_start[DENSIFY_MAX_GROUPS + 1];  // in units of output elements
  int ngroups;
  const float *xyz, *scaling, *rotation;  // OLD tensors, for the children
  const float *normals;                   // [2*NS, 3] N(0,1)
  const int32_t *src, *aux;
};

__global__ __launch_bounds__(256) void densify_apply_kernel(DensifyTable t, long long total) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int g = 0;
#pragma unroll
  for (int k = 1; k < DENSIFY_MAX_GROUPS; k++)
    if (k < t.ngroups && e >= t.elem_start[k]) g = k;
  const long long le = e - t.elem_start[g];
  const int L = t.row[g];
  const int j = (int)(le / L), c = (int)(le - (long long)j * L);
  const int s = t.src[j], a = t.aux[j];
  float val = t.in_p[g][(size_t)s * L + c];
  if (a >= 0 && t.role[g] == FSGS_DENSIFY_ROLE_SCALING) {
    val = logf(child_scale(expf(val)));  // scaling_inverse_activation(get_scaling / (0.8 N)), N = 2
  } else if (a >= 0 && t.role[g] == FSGS_DENSIFY_ROLE_XYZ) {
    // new_xyz = build_rotation(q) (z * scale) + xyz
    const float *q = t.rotation + 4 * (size_t)s;
    const float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float r = q[0] / nrm, x = q[1] / nrm, y = q[2] / nrm, z = q[3] / nrm;
    const float *sc = t.scaling + 3 * (size_t)s;
    const float *n3 = t.normals + 3 * (size_t)a;
    const float v0 = n3[0] * expf(sc[0]), v1 = n3[1] * expf(sc[1]), v2 = n3[2] * expf(sc[2]);
    float R0, R1, R2;
    if (c == 0) { R0 = 1.f - 2.f * (y * y + z * z); R1 = 2.f * (x * y - r * z); R2 = 2.f * (x * z + r * y); }
    else if (c == 1) { R0 = 2.f * (x * y + r * z); R1 = 1.f - 2.f * (x * x + z * z); R2 = 2.f * (y * z - r * x); }
    else { R0 = 2.f * (x * z - r * y); R1 = 2.f * (y * z + r * x); R2 = 1.f - 2.f * (x * x + y * y); }
    val = (R0 * v0 + R1 * v1 + R2 * v2) + val;
  }
  t.out_p[g][le] = val;
  if (t.out_m[g]) {
    const bool keep = a == -1 && t.in_m[g];
    t.out_m[g][le] = keep ? t.in_m[g][(size_t)s * L + c] : 0.f;
    t.out_v[g][le] = keep ? t.in_v[g][(size_t)s * L + c] : 0.f;
  }
}

}  // namespace

extern "C" {

int fsgs_densify_plan(int P, const float *xyz_gradient_accum, const float *denom, const float *scaling,
                      const float *opacity, float max_grad, float min_opacity, float small_extent, float big_extent,
                      int prune_big, int32_t *counts4, fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P < 0) return FSGS_ERR_INVALID;
  if (P == 0) return FSGS_OK;
  if (!xyz_gradient_accum || !denom || !scaling || !opacity || !counts4) return FSGS_ERR_INVALID;
  hipLaunchKernelGGL(densify_plan_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, xyz_gradient_accum, denom,
                     scaling, opacity, max_grad, min_opacity, small_extent, big_extent, prune_big, counts4);
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

int fsgs_densify_apply(int P, const int32_t *counts4, const int32_t *incl4, const int32_t totals4[4], int ngroups,
                       const FsgsDensifyGroup *groups, const float *normals, int32_t *src, int32_t *aux,
                       fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P < 0 || ngroups < 0 || ngroups > DENSIFY_MAX_GROUPS || !totals4) return FSGS_ERR_INVALID;
  const int K0 = totals4[0], K1 = totals4[1], NS = totals4[2], K3 = totals4[3];
  if (K0 < 0 || K1 < 0 || NS < 0 || K3 < 0 || K3 > NS) return FSGS_ERR_INVALID;
  const long long Pn = (long long)K0 + K1 + 2LL * K3;
  if (Pn == 0 || P == 0) return FSGS_OK;
  if (!counts4 || !incl4 || !groups || !src || !aux || (NS > 0 && !normals)) return FSGS_ERR_INVALID;
  hipLaunchKernelGGL(densify_map_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, counts4, incl4, K0, K1, NS, K3,
                     src, aux);
  DensifyTable t;
  std::memset(&t, 0, sizeof(t));
  long long total = 0;
  for (int i = 0; i < ngroups; i++) {
    const FsgsDensifyGroup &g = groups[i];
    if (!g.in_param || !g.out_param || g.row <= 0) return FSGS_ERR_INVALID;
    if ((g.out_exp_avg == nullptr) != (g.out_exp_avg_sq == nullptr)) return FSGS_ERR_INVALID;
    t.in_p[i] = g.in_param; t.in_m[i] = g.in_exp_avg; t.in_v[i] = g.in_exp_avg_sq;
    t.out_p[i] = g.out_param; t.out_m[i] = g.out_exp_avg; t.out_v[i] = g.out_exp_avg_sq;
    t.row[i] = g.row; t.role[i] = g.role;
    t.elem_start[i] = total;
    total += Pn * g.row;
    if (g.role == FSGS_DENSIFY_ROLE_XYZ) t.xyz = g.in_param;
    if (g.role == FSGS_DENSIFY_ROLE_SCALING) t.scaling = g.in_param;
    if (g.role == FSGS_DENSIFY_ROLE_ROTATION) t.rotation = g.in_param;
  }
  t.elem_start[ngroups] = total;
  t.ngroups = ngroups;
  if (NS > 0 && K3 > 0 && (!t.xyz || !t.scaling || !t.rotation)) return FSGS_ERR_INVALID;
  t.normals = normals; t.src = src; t.aux = aux;
  if (total > 0) {
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7FFFFFFFLL) return FSGS_ERR_INVALID;
    hipLaunchKernelGGL(densify_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, t, total);
  }
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

}  // extern "C"
