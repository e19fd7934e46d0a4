// knn.hip -- exact 3-nearest-neighbour mean squared distance for gfx950 (MI355X).
//
// Replaces simple-knn's distCUDA2 (submodules/simple-knn/simple_knn.cu:185-221,
// spatial.cu:15-26, ext.cpp:15-17), used once at initialisation
// (scene/gaussian_model.py:217,346).  Same answer (the search is exact, self excluded by
// position, FLT_MAX terms when fewer than 3 neighbours exist), different machine mapping:
//   * 30-bit Morton codes + one radix sort put spatial neighbours next to each other;
//   * the sorted points are re-packed as float4 so every later access is a coalesced 16 B load;
//   * boxes are 256 points = one workgroup.  A workgroup answers the queries of ITS OWN box:
//     it walks the other boxes outward in Morton order, rejects a whole box with one
//     wave-uniform AABB-to-AABB bound against the workgroup's worst current 3rd-best
//     (scalar work only), and stages a surviving box in LDS where all 256 lanes scan it with
//     broadcast reads (same address in every lane -> conflict free).
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <cstring>
#include <rocprim/rocprim.hpp>

#include "../../include/fsgs.h"
#include "fsgs_device.h"
#include "fsgs_host.h"

using namespace fsgs;

namespace {

constexpr int KBOX = 256;

struct Box {
  float lo[3];
  float hi[3];
};

__device__ __forceinline__ uint32_t enc_f(float f) {
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f(uint32_t e) {
  uint32_t b = (e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e;
  return __uint_as_float(b);
}

__global__ void bbox_init_kernel(uint32_t *bbox) {
  // the reference seeds both reductions with (0,0,0) (simple_knn.cu:191): the origin is always inside
  if (threadIdx.x < 6) bbox[threadIdx.x] = enc_f(0.0f);
}

__global__ __launch_bounds__(256) void bbox_kernel(int P, const float *__restrict__ pts, uint32_t *bbox) {
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      float v = pts[3 * i + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
    for (int off = 32; off >= 1; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      atomicMin(&bbox[a], enc_f(lo[a]));
      atomicMax(&bbox[3 + a], enc_f(hi[a]));
    }
  }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {
  x = (x | (x << 16)) & 0x030000FF;
  x = (x | (x << 8)) & 0x0300F00F;
  x = (x | (x << 4)) & 0x030C30C3;
  x = (x | (x << 2)) & 0x09249249;
  return x;
}

__global__ __launch_bounds__(256) void morton_kernel(int P, const float *__restrict__ pts,
                                                     const uint32_t *__restrict__ bbox, uint32_t *codes,
                                                     uint32_t *index) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  uint32_t c = 0;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    float lo = dec_f(bbox[a]), hi = dec_f(bbox[3 + a]);
    float ext = hi - lo;
    float u = ext > 0.f ? (pts[3 * i + a] - lo) / ext : 0.f;
    uint32_t q = (uint32_t)fminf(fmaxf(u * 1023.0f, 0.f), 1023.0f);
    c |= spread10(q) << a;
  }
  codes[i] = c;
  index[i] = (uint32_t)i;
}

// sorted float4 copy of the points + per-box AABB
__global__ __launch_bounds__(KBOX) void pack_boxes_kernel(int P, const float *__restrict__ pts,
                                                          const uint32_t *__restrict__ order, float4 *sorted,
                                                          Box *boxes) {
  __shared__ float red[6][KBOX / 64];
  int i = blockIdx.x * KBOX + threadIdx.x;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (i < P) {
    uint32_t g = order[i];
    float4 p = make_float4(pts[3 * g], pts[3 * g + 1], pts[3 * g + 2], 0.f);
    sorted[i] = p;
    lo[0] = hi[0] = p.x; lo[1] = hi[1] = p.y; lo[2] = hi[2] = p.z;
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
    for (int off = 32; off >= 1; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
    if ((threadIdx.x & 63) == 0) {
      red[a][threadIdx.x >> 6] = lo[a];
      red[3 + a][threadIdx.x >> 6] = hi[a];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Box b;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      b.lo[a] = fminf(fminf(red[a][0], red[a][1]), fminf(red[a][2], red[a][3]));
      b.hi[a] = fmaxf(fmaxf(red[3 + a][0], red[3 + a][1]), fmaxf(red[3 + a][2], red[3 + a][3]));
    }
    boxes[blockIdx.x] = b;
  }
}

__device__ __forceinline__ void push3(float d, float &b0, float &b1, float &b2) {
  // keep the three smallest, sorted (simple_knn.cu:132-145)
  if (d < b2) {
    if (d < b0) { b2 = b1; b1 = b0; b0 = d; }
    else if (d < b1) { b2 = b1; b1 = d; }
    else b2 = d;
  }
}

__device__ __forceinline__ float box_box_dist2(const Box &a, const Box &b) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float d = fmaxf(0.f, fmaxf(a.lo[k] - b.hi[k], b.lo[k] - a.hi[k]));
    s = fmaf(d, d, s);
  }
  return s;
}
__device__ __forceinline__ float box_point_dist2(const Box &b, float x, float y, float z) {
  float dx = fmaxf(0.f, fmaxf(b.lo[0] - x, x - b.hi[0]));
  float dy = fmaxf(0.f, fmaxf(b.lo[1] - y, y - b.hi[1]));
  float dz = fmaxf(0.f, fmaxf(b.lo[2] - z, z - b.hi[2]));
  return dx * dx + dy * dy + dz * dz;
}

__global__ __launch_bounds__(KBOX) void knn_query_kernel(int P, int nboxes, const float4 *__restrict__ sorted,
                                                         const Box *__restrict__ boxes,
                                                         const uint32_t *__restrict__ order, float *out) {
  __shared__ float4 cand[KBOX];
  __shared__ float wmax[KBOX / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int i = b * KBOX + tid;
  const bool valid = i < P;
  float4 me = valid ? sorted[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
  const Box mybox = boxes[b];
  float blockmax = FLT_MAX;
  // visit order: own box, then b-1, b+1, b-2, b+2, ... (Morton neighbours first so the bound tightens early)
  for (int step = 0; step < 2 * nboxes; step++) {
    int cb;
    if (step == 0) cb = b;
    else {
      int k = (step + 1) >> 1;
      cb = (step & 1) ? b - k : b + k;
    }
    if (cb < 0 || cb >= nboxes) continue;
    const Box cbox = boxes[cb];
    // workgroup-uniform rejection: nothing in cbox can beat anybody's current 3rd best
    if (box_box_dist2(mybox, cbox) > blockmax) continue;
    __syncthreads();
    int ci = cb * KBOX + tid;
    cand[tid] = ci < P ? sorted[ci] : make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.f);
    __syncthreads();
    const int cn = min(KBOX, P - cb * KBOX);
    if (valid && box_point_dist2(cbox, me.x, me.y, me.z) <= b2) {
      for (int j = 0; j < cn; j++) {
        if (cb == b && j == tid) continue;  // self is excluded by position, not by value
        float4 c = cand[j];
        float dx = c.x - me.x, dy = c.y - me.y, dz = c.z - me.z;
        push3(dx * dx + dy * dy + dz * dz, b0, b1, b2);
      }
    }
    // refresh the workgroup bound (b2 only shrinks, so a stale bound is merely conservative)
    float m = valid ? b2 : 0.f;
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((tid & 63) == 0) wmax[tid >> 6] = m;
    __syncthreads();
    blockmax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  }
  if (valid) out[order[i]] = (b0 + b1 + b2) / 3.0f;
}

struct KnnLayout {
  size_t bbox, code_a, code_b, idx_a, idx_b, sorted, boxes, temp, temp_bytes, total;
};
int knn_layout(int P, KnnLayout &L) {
  Carver c;
  size_t Pn = (size_t)(P > 0 ? P : 1);
  size_t nb = (Pn + KBOX - 1) / KBOX;
  L.bbox = c.take(6 * 4);
  L.code_a = c.take(4 * Pn);
  L.code_b = c.take(4 * Pn);
  L.idx_a = c.take(4 * Pn);
  L.idx_b = c.take(4 * Pn);
  L.sorted = c.take(16 * Pn);
  L.boxes = c.take(sizeof(Box) * nb);
  size_t t = 0;
  uint32_t *nu = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, t, nu, nu, nu, nu, Pn, 0, 30, (hipStream_t)0) != hipSuccess) return -1;
  L.temp_bytes = t + 256;
  L.temp = c.take(L.temp_bytes);
  L.total = c.total();
  return 0;
}

}  // namespace

extern "C" int fsgs_knn_meandist2(int P, const float *points, float *out, void *scratch, size_t *scratch_bytes,
                                  fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P < 0 || !scratch_bytes) return FSGS_ERR_INVALID;
  KnnLayout L;
  if (knn_layout(P, L) != 0) return fsgs_fail("rocprim size query");
  if (!scratch) {
    *scratch_bytes = L.total;
    return FSGS_OK;
  }
  if (*scratch_bytes < L.total) return FSGS_ERR_CAPACITY;
  if (P == 0) return FSGS_OK;
  if (!points || !out) return FSGS_ERR_INVALID;
  char *xb = (char *)scratch;
  uint32_t *bbox = (uint32_t *)(xb + L.bbox);
  uint32_t *code_a = (uint32_t *)(xb + L.code_a), *code_b = (uint32_t *)(xb + L.code_b);
  uint32_t *idx_a = (uint32_t *)(xb + L.idx_a), *idx_b = (uint32_t *)(xb + L.idx_b);
  float4 *sorted = (float4 *)(xb + L.sorted);
  Box *boxes = (Box *)(xb + L.boxes);
  const int nboxes = (P + KBOX - 1) / KBOX;
  hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, stream, bbox);
  int rb = (P + 255) / 256;
  if (rb > 1024) rb = 1024;
  hipLaunchKernelGGL(bbox_kernel, dim3(rb), dim3(256), 0, stream, P, points, bbox);
  hipLaunchKernelGGL(morton_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, points, bbox, code_a, idx_a);
  FSGS_HIP(hipGetLastError());
  size_t tb = L.temp_bytes;
  FSGS_HIP(rocprim::radix_sort_pairs(xb + L.temp, tb, code_a, code_b, idx_a, idx_b, (size_t)P, 0, 30, stream));
  hipLaunchKernelGGL(pack_boxes_kernel, dim3(nboxes), dim3(KBOX), 0, stream, P, points, idx_b, sorted, boxes);
  hipLaunchKernelGGL(knn_query_kernel, dim3(nboxes), dim3(KBOX), 0, stream, P, nboxes, sorted, boxes, idx_b, out);
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}
