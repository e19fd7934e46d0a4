// flow.hip -- optical-flow reprojection loss of the tracking step for gfx950 (MI355X).
//
// Reference: projection_flow_loss (scene/pose_optimizer.py:164-218).  Its pose-INDEPENDENT half
// (back-projection of the previous depth, duplicate/origin rejection, scene/pose_optimizer.py:42-73) is
// constant over the 50 tracking iterations of a frame and is prepared once per frame on the host side
// (fsgs_amd/flow.py: FlowTargets).  These kernels are the per-iteration half: transform the M world
// points by the current pose, project with K, keep the border-safe points in front of the camera, L1
// against the forward flow -- one streaming pass forward, one backward that reduces dL/dw2c (12 floats)
// with DPP wave sums and one atomic per workgroup.  28 B/point read, HBM-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fsgs.h"
#include "fsgs_device.h"
#include "fsgs_host.h"

using namespace fsgs;

namespace {

struct FlowCam {
  float K[9];
  int W, H;
  float edge;
};

struct FlowPoint {
  bool valid;
  float eu, ev;        // residuals (projection flow - gt flow)
  float cx, cy, cz;    // camera-frame point
  float pz, u, v;      // K-projected depth (+1e-5) and pixel
};

__device__ __forceinline__ FlowPoint flow_point(const FlowCam &c, const float *__restrict__ w,
                                                const float *__restrict__ pts, const int64_t *__restrict__ pix_vu,
                                                const float *__restrict__ flow, size_t i) {
  FlowPoint o;
  float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  o.cx = w[0] * x + w[1] * y + w[2] * z + w[3];
  o.cy = w[4] * x + w[5] * y + w[6] * z + w[7];
  o.cz = w[8] * x + w[9] * y + w[10] * z + w[11];
  float p0 = c.K[0] * o.cx + c.K[1] * o.cy + c.K[2] * o.cz;
  float p1 = c.K[3] * o.cx + c.K[4] * o.cy + c.K[5] * o.cz;
  float p2 = c.K[6] * o.cx + c.K[7] * o.cy + c.K[8] * o.cz;
  o.pz = p2 + 1e-5f;
  o.u = p0 / o.pz;
  o.v = p1 / o.pz;
  o.valid = (o.u < (float)c.W - c.edge) && (o.u > c.edge) && (o.v < (float)c.H - c.edge) && (o.v > c.edge) &&
            (o.pz > 0.f);
  int64_t sv = pix_vu[2 * i], su = pix_vu[2 * i + 1];
  size_t at = (size_t)sv * c.W + (size_t)su;
  size_t plane = (size_t)c.H * c.W;
  o.eu = (o.u - (float)su) - flow[at];
  o.ev = (o.v - (float)sv) - flow[plane + at];
  return o;
}

// acc[0] += sum(|eu| + |ev|) over valid points, acc[1] += #valid, acc[2] += #NaN   (doubles)
__global__ __launch_bounds__(256) void flow_fwd_kernel(size_t M, FlowCam c, const float *__restrict__ w2c,
                                                       const float *__restrict__ pts,
                                                       const int64_t *__restrict__ pix_vu,
                                                       const float *__restrict__ flow, double *__restrict__ acc) {
  __shared__ float red[3][4];
  float s = 0.f, n = 0.f, bad = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (size_t)gridDim.x * 256) {
    FlowPoint p = flow_point(c, w2c, pts, pix_vu, flow, i);
    if (p.valid) {
      float e = fabsf(p.eu) + fabsf(p.ev);
      if (e != e) bad += 1.f; else s += e;
      n += 1.f;
    }
  }
  float ts = wave_sum(s), tn = wave_sum(n), tb = wave_sum(bad);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[0][wid] = ts; red[1][wid] = tn; red[2][wid] = tb; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float t = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    if (t != 0.f) atomicAdd(acc + threadIdx.x, (double)t);
  }
}

// loss = sum / (2 * count);  0 when nothing survives or a NaN was seen (scene/pose_optimizer.py:196-214)
__global__ void flow_finish_kernel(const double *acc, float *out) {
  double n = acc[1];
  out[0] = (n > 0 && acc[2] == 0) ? (float)(acc[0] / (2.0 * n)) : 0.f;
  out[1] = (float)n;
}

// dL/dw2c[r][c] += upstream/(2 n) * sum_i [sign(eu) du/dcam_r + sign(ev) dv/dcam_r] * [x;1]_c
__global__ __launch_bounds__(256) void flow_bwd_kernel(size_t M, FlowCam c, const float *__restrict__ w2c,
                                                       const float *__restrict__ pts,
                                                       const int64_t *__restrict__ pix_vu,
                                                       const float *__restrict__ flow,
                                                       const double *__restrict__ acc,
                                                       const float *__restrict__ upstream,
                                                       float *__restrict__ dw2c) {
  __shared__ float red[12][4];
  const double n = acc[1];
  const bool live = n > 0 && acc[2] == 0;
  const float scale = live ? (upstream ? upstream[0] : 1.f) / (2.f * (float)n) : 0.f;
  float g[12];
#pragma unroll
  for (int k = 0; k < 12; k++) g[k] = 0.f;
  if (live) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (size_t)gridDim.x * 256) {
      FlowPoint p = flow_point(c, w2c, pts, pix_vu, flow, i);
      if (!p.valid) continue;
      float su = p.eu > 0.f ? 1.f : (p.eu < 0.f ? -1.f : 0.f);
      float sv = p.ev > 0.f ? 1.f : (p.ev < 0.f ? -1.f : 0.f);
      float ipz = 1.0f / p.pz;
      // dL/dp (p = K cam): u = p0/pz, v = p1/pz
      float dp0 = su * ipz, dp1 = sv * ipz, dp2 = -(su * p.u + sv * p.v) * ipz;
      // dL/dcam = K^T dp
      float dc0 = c.K[0] * dp0 + c.K[3] * dp1 + c.K[6] * dp2;
      float dc1 = c.K[1] * dp0 + c.K[4] * dp1 + c.K[7] * dp2;
      float dc2 = c.K[2] * dp0 + c.K[5] * dp1 + c.K[8] * dp2;
      float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
      g[0] = fmaf(dc0, x, g[0]); g[1] = fmaf(dc0, y, g[1]); g[2] = fmaf(dc0, z, g[2]); g[3] += dc0;
      g[4] = fmaf(dc1, x, g[4]); g[5] = fmaf(dc1, y, g[5]); g[6] = fmaf(dc1, z, g[6]); g[7] += dc1;
      g[8] = fmaf(dc2, x, g[8]); g[9] = fmaf(dc2, y, g[9]); g[10] = fmaf(dc2, z, g[10]); g[11] += dc2;
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 12; k++) {
    float t = wave_sum(g[k]);
    if (lane == 0) red[k][wid] = t;
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    float t = (red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]) * scale;
    if (t != 0.f) atomicAdd(dw2c + threadIdx.x, t);
  }
}

// ---------------------------------------------------------------------------------------------------
// fused forward + backward (the autograd-free tracking step): ONE streaming pass produces, per workgroup,
// {sum |e|, #valid, #NaN, the 12 UNSCALED sums of dcam [x;1]^T}; the finish kernel adds the partials in a fixed
// order (doubles), forms the loss and applies 1/(2n) -- the only thing the backward needed from the forward.
// ---------------------------------------------------------------------------------------------------
constexpr int FLOW_PER_THREAD = 16;  // 4096 points per workgroup: ~320 partials at 1280x1024
constexpr int FLOW_VALS = 16;  // 15 used
__global__ __launch_bounds__(256) void flow_fused_kernel(size_t M, FlowCam c, const float *__restrict__ w2c,
                                                         const float *__restrict__ pts,
                                                         const int64_t *__restrict__ pix_vu,
                                                         const float *__restrict__ flow,
                                                         float *__restrict__ partials) {
  __shared__ float red[FLOW_VALS][4];
  float a[15];
#pragma unroll
  for (int k = 0; k < 15; k++) a[k] = 0.f;
  const size_t base = (size_t)blockIdx.x * (256 * FLOW_PER_THREAD) + threadIdx.x;
#pragma unroll
  for (int q = 0; q < FLOW_PER_THREAD; q++) {
    const size_t i = base + (size_t)q * 256;
    if (i >= M) break;
    FlowPoint p = flow_point(c, w2c, pts, pix_vu, flow, i);
    if (!p.valid) continue;
    const float e = fabsf(p.eu) + fabsf(p.ev);
    if (e != e) a[2] += 1.f; else a[0] += e;
    a[1] += 1.f;
    float su = p.eu > 0.f ? 1.f : (p.eu < 0.f ? -1.f : 0.f);
    float sv = p.ev > 0.f ? 1.f : (p.ev < 0.f ? -1.f : 0.f);
    float ipz = 1.0f / p.pz;
    float dp0 = su * ipz, dp1 = sv * ipz, dp2 = -(su * p.u + sv * p.v) * ipz;
    float dc0 = c.K[0] * dp0 + c.K[3] * dp1 + c.K[6] * dp2;
    float dc1 = c.K[1] * dp0 + c.K[4] * dp1 + c.K[7] * dp2;
    float dc2 = c.K[2] * dp0 + c.K[5] * dp1 + c.K[8] * dp2;
    float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    float *g = a + 3;
    g[0] = fmaf(dc0, x, g[0]); g[1] = fmaf(dc0, y, g[1]); g[2] = fmaf(dc0, z, g[2]); g[3] += dc0;
    g[4] = fmaf(dc1, x, g[4]); g[5] = fmaf(dc1, y, g[5]); g[6] = fmaf(dc1, z, g[6]); g[7] += dc1;
    g[8] = fmaf(dc2, x, g[8]); g[9] = fmaf(dc2, y, g[9]); g[10] = fmaf(dc2, z, g[10]); g[11] += dc2;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 15; k++) {
    float t = wave_sum(a[k]);
    if (lane == 0) red[k][wid] = t;
  }
  __syncthreads();
  if (threadIdx.x < FLOW_VALS)
    partials[(size_t)blockIdx.x * FLOW_VALS + threadIdx.x] =
        threadIdx.x < 15 ? red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3] : 0.f;
}

// out2 = {loss, #valid};  dw2c[r][c] = accumulate * dw2c[r][c] + upstream * dloss/dw2c[r][c]  (accumulate == 0
// overwrites without reading; row 3 of the flow term is zero)
__global__ __launch_bounds__(1024) void flow_fused_finish_kernel(const float *__restrict__ partials, int nblocks,
                                                                float upstream, float accumulate,
                                                                float *__restrict__ out2, float *__restrict__ dw2c) {
  __shared__ double red[64][17];
  const int v = threadIdx.x & 15, row = threadIdx.x >> 4;  // 16 values x 64 block stripes
  double s = 0.0;
  for (int b = row; b < nblocks; b += 64) s += (double)partials[(size_t)b * FLOW_VALS + v];
  red[row][v] = s;
  __syncthreads();
  if (threadIdx.x < 16) {
    double t = 0.0;
    for (int r = 0; r < 64; r++) t += red[r][threadIdx.x];
    red[0][threadIdx.x] = t;
  }
  __syncthreads();
  const double n = red[0][1];
  const bool live = n > 0 && red[0][2] == 0;
  if (threadIdx.x == 0) {
    out2[0] = live ? (float)(red[0][0] / (2.0 * n)) : 0.f;
    out2[1] = (float)n;
  }
  if (threadIdx.x < 16) {
    const float g = (live && threadIdx.x < 12) ? upstream / (2.f * (float)n) * (float)red[0][3 + threadIdx.x] : 0.f;
    dw2c[threadIdx.x] = accumulate != 0.f ? fmaf(accumulate, dw2c[threadIdx.x], g) : g;
  }
}

// ---------------------------------------------------------------------------------------------------
// Sampson-distance rigid mask, once per tracked frame (train.py:157-165; PoseModel.compute_epipolar_loss /
// get_matches, scene/pose_optimizer.py:700-746; adaptive_thresholding, utils/general_utils.py:96-116).
// Every pixel x1 = (u, v, 1) of frame t-2 is matched to x2 = x1 + flow; with l1 = F x1, l2 = F^T x2 the squared
// Sampson distance is (x2 . l1)^2 / (l1x^2 + l1y^2 + l2x^2 + l2y^2)  (the public definition the reference takes
// from kornia.geometry.epipolar.sampson_epipolar_distance, squared = True; kornia is not part of the reference
// tree: parity unpinned).  threshold = mean + factor * std (unbiased); the reference then evaluates
// `dist < (dist <= threshold)`, a float-vs-bool comparison, i.e. rigid = dist <= threshold AND dist < 1.
// ---------------------------------------------------------------------------------------------------
constexpr int SAMPSON_CHUNK = 4096;
struct Mat9 {
  float m[9];
};
__global__ __launch_bounds__(256) void sampson_kernel(int H, int W, Mat9 F, const float *__restrict__ flow,
                                                      float *__restrict__ dist, double *__restrict__ partials) {
  __shared__ double red[2][4];
  const size_t HW = (size_t)H * W;
  const size_t begin = (size_t)blockIdx.x * SAMPSON_CHUNK;
  double s1 = 0.0, s2 = 0.0;
  for (size_t p = begin + threadIdx.x; p < begin + SAMPSON_CHUNK && p < HW; p += 256) {
    const int v = (int)(p / W), u = (int)(p - (size_t)v * W);
    const float x1 = (float)u, y1 = (float)v;
    const float x2 = x1 + flow[p], y2 = y1 + flow[HW + p];
    const float* f = F.m;
    const float l1x = f[0] * x1 + f[1] * y1 + f[2], l1y = f[3] * x1 + f[4] * y1 + f[5], l1z = f[6] * x1 + f[7] * y1 + f[8];
    const float l2x = f[0] * x2 + f[3] * y2 + f[6], l2y = f[1] * x2 + f[4] * y2 + f[7];
    const float num = x2 * l1x + y2 * l1y + l1z;
    const float d = (num * num) / (l1x * l1x + l1y * l1y + l2x * l2x + l2y * l2y);
    dist[p] = d;
    s1 += (double)d;
    s2 += (double)d * (double)d;
  }
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[0][wid] = s1; red[1][wid] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * (size_t)blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partials[2 * (size_t)blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}
// every workgroup re-reduces the (few hundred, L2-resident) partials, then masks its own chunk
__global__ __launch_bounds__(256) void sampson_mask_kernel(int H, int W, int nparts, float factor,
                                                           const double *__restrict__ partials,
                                                           const float *__restrict__ dist,
                                                           uint8_t *__restrict__ rigid, float *__restrict__ stats3) {
  __shared__ double red[2][4];
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) {
    s1 += partials[2 * (size_t)i];
    s2 += partials[2 * (size_t)i + 1];
  }
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[0][wid] = s1; red[1][wid] = s2; }
  __syncthreads();
  const double N = (double)H * (double)W;
  const double S1 = red[0][0] + red[0][1] + red[0][2] + red[0][3], S2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  const double mean = S1 / N;
  double var = N > 1.0 ? (S2 - N * mean * mean) / (N - 1.0) : 0.0;
  var = var > 0.0 ? var : 0.0;
  // mean().item() and std().item() are fp32 values widened to python floats; the sum is then formed in double
  const float thr = (float)((double)(float)mean + (double)factor * (double)(float)sqrt(var));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    stats3[0] = (float)mean;
    stats3[1] = (float)sqrt(var);
    stats3[2] = thr;
  }
  const size_t HW = (size_t)H * W;
  const size_t begin = (size_t)blockIdx.x * SAMPSON_CHUNK;
  for (size_t p = begin + threadIdx.x; p < begin + SAMPSON_CHUNK && p < HW; p += 256) {
    const float d = dist[p];
    rigid[p] = (d <= thr && d < 1.0f) ? 1 : 0;  // dist < float(dist <= thr)
  }
}

// ---------------------------------------------------------------------------------------------------
// Flow targets of a tracked frame: the pose-INDEPENDENT half of projection_flow_loss, once per frame
// (get_pointcloud + the duplicate rejection, scene/pose_optimizer.py:42-73,171-181).  A valid pixel (depth * rigid
// > 0) is back-projected with K, moved to the world with inverse(w2c_prev), and dropped when |round(world, 4)|
// equals that of another point or the origin.  Three launches around a sort of 64-bit hashes of the rounded
// triples (the sort itself and the scan of the keep flags are the caller's: torch.sort / cumsum = rocPRIM):
//   keys   : world point + rounded triple + hash per pixel (invalid pixels get the all-ones key and sort last)
//   flag   : in sorted order, a point is a duplicate when a neighbour carries the same hash AND the same triple
//            (a colliding hash between two copies of a duplicate would hide it: probability ~ M / 2^64)
//   gather : the kept points in pixel order -> pts [M,3], pix_vu [M,2] = (v, u)
// torch.round(x, decimals=4) = nearbyint(x * 1e4f) / 1e4f in fp32, reproduced here; the world point itself comes
// from an fp32 FMA chain instead of torch's 4x4 GEMM, so a coordinate within an ulp of a rounding boundary may
// round the other way (tests allow <= 1e-5 of the points to differ in the keep decision).
// ---------------------------------------------------------------------------------------------------
struct Mat16 {
  float m[16];
};
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; x ^= x >> 31;
  return x;
}
__global__ __launch_bounds__(256) void flow_targets_keys_kernel(int H, int W, FlowCam c, Mat16 c2w,
                                                                const float *__restrict__ depth,
                                                                const uint8_t *__restrict__ rigid,
                                                                float *__restrict__ world, float *__restrict__ rounded,
                                                                long long *__restrict__ keys) {
  const size_t HW = (size_t)H * W;
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const float z = depth[p];
  const float dm = rigid ? z * (rigid[p] ? 1.0f : 0.0f) : z;
  if (!(dm > 0.f)) {
    keys[p] = -1;  // sorts after every valid key in an UNSIGNED view, first in a signed one: flagged by value either way
    return;
  }
  const int v = (int)(p / W), u = (int)(p - (size_t)v * W);
  const float xx = ((float)u - c.K[2]) / c.K[0], yy = ((float)v - c.K[5]) / c.K[4];
  const float cx = xx * z, cy = yy * z;
  const float *m = c2w.m;
  float r[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float wk = m[4 * k] * cx + m[4 * k + 1] * cy + m[4 * k + 2] * z + m[4 * k + 3];
    world[3 * p + k] = wk;
    r[k] = fabsf(nearbyintf(wk * 10000.0f) / 10000.0f);
    rounded[3 * p + k] = r[k];
  }
  unsigned long long h = mix64(((unsigned long long)__float_as_uint(r[0]) << 32) | __float_as_uint(r[1]));
  h = mix64(h ^ (0x9e3779b97f4a7c15ULL + __float_as_uint(r[2])));
  if (h == ~0ull) h = 0x1234567ULL;  // the all-ones pattern is reserved for invalid pixels
  keys[p] = (long long)h;
}
// sorted_keys / sorted_idx: the keys in ascending (signed) order with their pixel indices
__global__ __launch_bounds__(256) void flow_targets_flag_kernel(size_t HW, const long long *__restrict__ sorted_keys,
                                                                const long long *__restrict__ sorted_idx,
                                                                const float *__restrict__ rounded,
                                                                int32_t *__restrict__ keep) {
  const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= HW) return;
  const long long key = sorted_keys[s];
  const size_t p = (size_t)sorted_idx[s];
  if (key == -1) {
    keep[p] = 0;
    return;
  }
  const float a0 = rounded[3 * p], a1 = rounded[3 * p + 1], a2 = rounded[3 * p + 2];
  bool dup = (a0 == 0.f && a1 == 0.f && a2 == 0.f);  // the extra all-zero row of the reference's unique()
#pragma unroll
  for (int d = -1; d <= 1; d += 2) {
    const long long q = (long long)s + d;
    if (q < 0 || q >= (long long)HW) continue;
    if (sorted_keys[q] != key) continue;
    const size_t pq = (size_t)sorted_idx[q];
    dup = dup || (rounded[3 * pq] == a0 && rounded[3 * pq + 1] == a1 && rounded[3 * pq + 2] == a2);
  }
  keep[p] = dup ? 0 : 1;
}
// incl = inclusive prefix sum of keep (pixel order)
__global__ __launch_bounds__(256) void flow_targets_gather_kernel(int H, int W, const int32_t *__restrict__ keep,
                                                                  const int32_t *__restrict__ incl,
                                                                  const float *__restrict__ world,
                                                                  float *__restrict__ pts,
                                                                  long long *__restrict__ pix_vu) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (size_t)H * W || !keep[p]) return;
  const size_t j = (size_t)incl[p] - 1;
  pts[3 * j] = world[3 * p]; pts[3 * j + 1] = world[3 * p + 1]; pts[3 * j + 2] = world[3 * p + 2];
  const long long v = (long long)(p / W);
  pix_vu[2 * j] = v;
  pix_vu[2 * j + 1] = (long long)p - v * W;
}

// every workgroup ends with a handful of same-address atomics, which serialise (tens of ns each): with 1024
// workgroups that chain was longer than the streaming pass itself, so the grid is capped at one workgroup per CU
int flow_blocks(size_t M) {
  size_t b = (M + 255) / 256;
  return (int)(b > 256 ? 256 : (b ? b : 1));
}

FlowCam make_flow_cam(const float *K9, int W, int H, float edge) {
  FlowCam c;
  for (int i = 0; i < 9; i++) c.K[i] = K9[i];
  c.W = W; c.H = H; c.edge = edge;
  return c;
}

}  // namespace

extern "C" {

int fsgs_flow_targets_keys(int H, int W, const float *depth_prev, const uint8_t *rigid, const float *K9_host,
                           const float *c2w16_host, float *world, float *rounded, int64_t *keys,
                           fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (H <= 0 || W <= 0 || !depth_prev || !K9_host || !c2w16_host || !world || !rounded || !keys) return FSGS_ERR_INVALID;
  FlowCam c = make_flow_cam(K9_host, W, H, 0.f);
  Mat16 m;
  for (int i = 0; i < 16; i++) m.m[i] = c2w16_host[i];
  const size_t HW = (size_t)H * W;
  hipLaunchKernelGGL(flow_targets_keys_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, stream, H, W, c, m,
                     depth_prev, rigid, world, rounded, (long long *)keys);
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

int fsgs_flow_targets_flag(int64_t HW, const int64_t *sorted_keys, const int64_t *sorted_idx, const float *rounded,
                           int32_t *keep, fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (HW <= 0 || !sorted_keys || !sorted_idx || !rounded || !keep) return FSGS_ERR_INVALID;
  hipLaunchKernelGGL(flow_targets_flag_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, stream, (size_t)HW,
                     (const long long *)sorted_keys, (const long long *)sorted_idx, rounded, keep);
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

int fsgs_flow_targets_gather(int H, int W, const int32_t *keep, const int32_t *incl, const float *world, float *pts,
                             int64_t *pix_vu, fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (H <= 0 || W <= 0 || !keep || !incl || !world || !pts || !pix_vu) return FSGS_ERR_INVALID;
  const size_t HW = (size_t)H * W;
  hipLaunchKernelGGL(flow_targets_gather_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, stream, H, W, keep,
                     incl, world, pts, (long long *)pix_vu);
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

int fsgs_flow_pose_loss_forward(int64_t M, const float *pts_world, const int64_t *pix_vu, const float *w2c,
                                const float *K9_host, const float *flow_fw, int W, int H, float edge, double *acc3,
                                float *out2, fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M < 0 || !w2c || !K9_host || !flow_fw || !acc3 || !out2 || W <= 0 || H <= 0) return FSGS_ERR_INVALID;
  if (M > 0 && (!pts_world || !pix_vu)) return FSGS_ERR_INVALID;
  FSGS_HIP(hipMemsetAsync(acc3, 0, 3 * sizeof(double), stream));
  FlowCam c = make_flow_cam(K9_host, W, H, edge);
  {
    ProfScope ps(PROF_FLOW, stream);
    if (M > 0)
      hipLaunchKernelGGL(flow_fwd_kernel, dim3(flow_blocks((size_t)M)), dim3(256), 0, stream, (size_t)M, c, w2c,
                         pts_world, pix_vu, flow_fw, acc3);
    hipLaunchKernelGGL(flow_finish_kernel, dim3(1), dim3(1), 0, stream, acc3, out2);
  }
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

size_t fsgs_flow_scratch_bytes(int64_t M) {
  if (M < 0) return 0;
  const size_t nb = ((size_t)M + 256 * FLOW_PER_THREAD - 1) / (256 * FLOW_PER_THREAD);
  return (nb ? nb : 1) * FLOW_VALS * sizeof(float) + 64;
}

int fsgs_flow_pose_loss_fused(int64_t M, const float *pts_world, const int64_t *pix_vu, const float *w2c,
                              const float *K9_host, const float *flow_fw, int W, int H, float edge, float upstream,
                              float accumulate, void *scratch, float *out2, float *dw2c, fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M < 0 || !w2c || !K9_host || !flow_fw || !scratch || !out2 || !dw2c || W <= 0 || H <= 0) return FSGS_ERR_INVALID;
  if (M > 0 && (!pts_world || !pix_vu)) return FSGS_ERR_INVALID;
  FlowCam c = make_flow_cam(K9_host, W, H, edge);
  const int nb = (int)(((size_t)M + 256 * FLOW_PER_THREAD - 1) / (256 * FLOW_PER_THREAD));
  {
    ProfScope ps(PROF_FLOW, stream);
    if (nb > 0)
      hipLaunchKernelGGL(flow_fused_kernel, dim3(nb), dim3(256), 0, stream, (size_t)M, c, w2c, pts_world, pix_vu,
                         flow_fw, (float *)scratch);
    hipLaunchKernelGGL(flow_fused_finish_kernel, dim3(1), dim3(1024), 0, stream, (const float *)scratch, nb, upstream,
                       accumulate, out2, dw2c);
  }
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

size_t fsgs_sampson_scratch_bytes(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  const size_t nb = ((size_t)H * W + SAMPSON_CHUNK - 1) / SAMPSON_CHUNK;
  return nb * 2 * sizeof(double) + 64;
}

int fsgs_sampson_rigid_mask(int H, int W, const float *flow_fw, const float *F9_host, float factor, void *scratch,
                            float *dist, uint8_t *rigid, float *stats3, fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (H <= 0 || W <= 0 || !flow_fw || !F9_host || !scratch || !dist || !rigid || !stats3) return FSGS_ERR_INVALID;
  Mat9 F;
  for (int i = 0; i < 9; i++) F.m[i] = F9_host[i];
  const int nb = (int)(((size_t)H * W + SAMPSON_CHUNK - 1) / SAMPSON_CHUNK);
  hipLaunchKernelGGL(sampson_kernel, dim3(nb), dim3(256), 0, stream, H, W, F, flow_fw, dist, (double *)scratch);
  hipLaunchKernelGGL(sampson_mask_kernel, dim3(nb), dim3(256), 0, stream, H, W, nb, factor, (const double *)scratch,
                     dist, rigid, stats3);
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

int fsgs_flow_pose_loss_backward(int64_t M, const float *pts_world, const int64_t *pix_vu, const float *w2c,
                                 const float *K9_host, const float *flow_fw, int W, int H, float edge,
                                 const double *acc3, const float *upstream, float *dw2c, fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M < 0 || !w2c || !K9_host || !flow_fw || !acc3 || !dw2c || W <= 0 || H <= 0) return FSGS_ERR_INVALID;
  FSGS_HIP(hipMemsetAsync(dw2c, 0, 16 * sizeof(float), stream));
  if (M == 0) return FSGS_OK;
  if (!pts_world || !pix_vu) return FSGS_ERR_INVALID;
  FlowCam c = make_flow_cam(K9_host, W, H, edge);
  {
    ProfScope ps(PROF_FLOW, stream);
    hipLaunchKernelGGL(flow_bwd_kernel, dim3(flow_blocks((size_t)M)), dim3(256), 0, stream, (size_t)M, c, w2c,
                       pts_world, pix_vu, flow_fw, acc3, upstream, dw2c);
  }
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

}  // extern "C"
