// optim.hip -- fused multi-tensor Adam and densification statistics for gfx950 (MI355X).
//
// Adam over the six GaussianModel parameter groups (scene/gaussian_model.py:396-409; steps at
// train.py:194,272) is the largest pure-HBM consumer of a mapping iteration: 59 floats/Gaussian x
// 7 accesses x 4 B = 1652 B/Gaussian.  torch.optim.Adam issues ~9 kernels per group;
// here ONE launch streams every group once with 16-byte accesses.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <cstring>

#include "../../include/fsgs.h"
#include "fsgs_device.h"
#include "fsgs_host.h"

using namespace fsgs;

namespace {

constexpr int ADAM_MAX_GROUPS = 8;
constexpr int ADAM_ELEMS_PER_BLOCK = 256 * 4 * 4;  // 256 threads x 4 float4

struct AdamTable {
  float *p[ADAM_MAX_GROUPS];
  const float *g[ADAM_MAX_GROUPS];
  float *m[ADAM_MAX_GROUPS];
  float *v[ADAM_MAX_GROUPS];
  int64_t n[ADAM_MAX_GROUPS];
  float step_size[ADAM_MAX_GROUPS];   // lr / (1 - beta1^t)
  float inv_bc2_sqrt[ADAM_MAX_GROUPS];  // 1 / sqrt(1 - beta2^t)
  int block_start[ADAM_MAX_GROUPS + 1];
  int ngroups;
  float beta1, beta2, one_minus_beta1, one_minus_beta2, eps;
};

__global__ __launch_bounds__(256) void adam_kernel(AdamTable t) {
  int gi = 0;
#pragma unroll
  for (int k = 1; k < ADAM_MAX_GROUPS; k++)
    if (k < t.ngroups && (int)blockIdx.x >= t.block_start[k]) gi = k;
  const int64_t n = t.n[gi];
  const int64_t base = (int64_t)(blockIdx.x - t.block_start[gi]) * ADAM_ELEMS_PER_BLOCK;
  float *p = t.p[gi], *m = t.m[gi], *v = t.v[gi];
  const float *g = t.g[gi];
  const float b1 = t.one_minus_beta1, b2 = t.beta2, ob2 = t.one_minus_beta2, eps = t.eps, ss = t.step_size[gi],
              ib = t.inv_bc2_sqrt[gi];
  const bool aligned = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
#pragma unroll
  for (int it = 0; it < 4; it++) {
    int64_t i = base + ((int64_t)it * 256 + threadIdx.x) * 4;
    if (i >= n) break;
    if (aligned && i + 4 <= n) {
      float4 pp = *(float4 *)(p + i), gg = *(const float4 *)(g + i), mm = *(float4 *)(m + i), vv = *(float4 *)(v + i);
      adam_one(pp.x, gg.x, mm.x, vv.x, b1, b2, ob2, eps, ss, ib);
      adam_one(pp.y, gg.y, mm.y, vv.y, b1, b2, ob2, eps, ss, ib);
      adam_one(pp.z, gg.z, mm.z, vv.z, b1, b2, ob2, eps, ss, ib);
      adam_one(pp.w, gg.w, mm.w, vv.w, b1, b2, ob2, eps, ss, ib);
      *(float4 *)(p + i) = pp; *(float4 *)(m + i) = mm; *(float4 *)(v + i) = vv;
    } else {
      for (int64_t j = i; j < n && j < i + 4; j++) {
        float pp = p[j], mm = m[j], vv = v[j];
        adam_one(pp, g[j], mm, vv, b1, b2, ob2, eps, ss, ib);
        p[j] = pp; m[j] = mm; v[j] = vv;
      }
    }
  }
}

// add_densification_stats + the max_radii2D update (scene/gaussian_model.py:678-681, train.py:298-303)
__global__ __launch_bounds__(256) void densify_stats_kernel(int P, const int32_t *__restrict__ radii,
                                                            const float *__restrict__ viewspace_grad,
                                                            float *__restrict__ max_radii2D,
                                                            float *__restrict__ grad_accum,
                                                            float *__restrict__ denom) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  int r = radii[i];
  if (r > 0) {
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
    float gx = viewspace_grad[3 * i], gy = viewspace_grad[3 * i + 1], gz = viewspace_grad[3 * i + 2];
    grad_accum[i] += sqrtf(gx * gx + gy * gy + gz * gz);
    denom[i] += 1.0f;
  }
}

}  // namespace

extern "C" {

int fsgs_adam_step(int ngroups, const FsgsAdamGroup *groups, double beta1, double beta2, double eps,
                   fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (ngroups < 0 || ngroups > ADAM_MAX_GROUPS || (ngroups > 0 && !groups)) return FSGS_ERR_INVALID;
  AdamTable t;
  std::memset(&t, 0, sizeof(t));
  t.beta1 = (float)beta1; t.beta2 = (float)beta2; t.eps = (float)eps;
  t.one_minus_beta1 = (float)(1.0 - beta1); t.one_minus_beta2 = (float)(1.0 - beta2);
  int blocks = 0, k = 0;
  for (int i = 0; i < ngroups; i++) {
    const FsgsAdamGroup &g = groups[i];
    if (g.n < 0 || g.step < 1) return FSGS_ERR_INVALID;
    if (g.n == 0) continue;
    if (!g.param || !g.grad || !g.exp_avg || !g.exp_avg_sq) return FSGS_ERR_INVALID;
    t.p[k] = g.param; t.g[k] = g.grad; t.m[k] = g.exp_avg; t.v[k] = g.exp_avg_sq; t.n[k] = g.n;
    double bc1 = 1.0 - pow(beta1, (double)g.step), bc2 = 1.0 - pow(beta2, (double)g.step);
    t.step_size[k] = (float)((double)g.lr / bc1);
    t.inv_bc2_sqrt[k] = (float)(1.0 / sqrt(bc2));
    t.block_start[k] = blocks;
    blocks += (int)((g.n + ADAM_ELEMS_PER_BLOCK - 1) / ADAM_ELEMS_PER_BLOCK);
    k++;
  }
  t.ngroups = k;
  t.block_start[k] = blocks;
  if (blocks == 0) return FSGS_OK;
  {
    ProfScope ps(PROF_ADAM, stream);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, t);
  }
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

int fsgs_densify_stats(int P, const int32_t *radii, const float *viewspace_grad, float *max_radii2D,
                       float *xyz_gradient_accum, float *denom, fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P < 0) return FSGS_ERR_INVALID;
  if (P == 0) return FSGS_OK;
  if (!radii || !viewspace_grad || !max_radii2D || !xyz_gradient_accum || !denom) return FSGS_ERR_INVALID;
  hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, radii, viewspace_grad,
                     max_radii2D, xyz_gradient_accum, denom);
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

}  // extern "C"
