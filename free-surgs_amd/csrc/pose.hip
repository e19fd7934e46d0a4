// pose.hip -- LearnPose.forward and its adjoint as single launches (scene/pose_optimizer.py:822-877).
//
// w2c = [[R(q2), t], [0 0 0 1]] with q1 = r / max(|r|, 1e-12) (F.normalize) and q2 = q1 / |q1| (q2rot
// normalises again).  In PyTorch this is ~15 tiny kernels forward and ~30 backward per tracking iteration;
// 7 floats in, 16 out.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <math.h>

#include "../../include/fsgs.h"
#include "fsgs_device.h"
#include "fsgs_host.h"

namespace {

struct PoseQ {
  float q1[4], q2[4], n0, n1;
};
__device__ __forceinline__ PoseQ pose_quat(const float *r, int N, int id) {
  PoseQ o;
  float a[4];
#pragma unroll
  for (int k = 0; k < 4; k++) a[k] = r[k * N + id];  // r is [1,4,N]
  o.n0 = fmaxf(sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3]), 1e-12f);
#pragma unroll
  for (int k = 0; k < 4; k++) o.q1[k] = a[k] / o.n0;
  o.n1 = sqrtf(o.q1[0] * o.q1[0] + o.q1[1] * o.q1[1] + o.q1[2] * o.q1[2] + o.q1[3] * o.q1[3]);
#pragma unroll
  for (int k = 0; k < 4; k++) o.q2[k] = o.q1[k] / o.n1;
  return o;
}

__global__ void pose_fwd_kernel(const float *r, const float *t, int N, int id, float *w2c) {
  if (threadIdx.x != 0) return;
  PoseQ p = pose_quat(r, N, id);
  const float qr = p.q2[0], x = p.q2[1], y = p.q2[2], z = p.q2[3];
  w2c[0] = 1.f - 2.f * (y * y + z * z); w2c[1] = 2.f * (x * y - qr * z);       w2c[2] = 2.f * (x * z + qr * y);
  w2c[4] = 2.f * (x * y + qr * z);       w2c[5] = 1.f - 2.f * (x * x + z * z); w2c[6] = 2.f * (y * z - qr * x);
  w2c[8] = 2.f * (x * z - qr * y);       w2c[9] = 2.f * (y * z + qr * x);       w2c[10] = 1.f - 2.f * (x * x + y * y);
  w2c[3] = t[0 * N + id]; w2c[7] = t[1 * N + id]; w2c[11] = t[2 * N + id];  // t is [3,N]
  w2c[12] = 0.f; w2c[13] = 0.f; w2c[14] = 0.f; w2c[15] = 1.f;
}

// dr [1,4,N], dt [3,N]: overwritten (zero except column id)
__global__ void pose_bwd_kernel(const float *r, int N, int id, const float *dW, float *dr, float *dt) {
  for (int i = threadIdx.x; i < 4 * N; i += blockDim.x) dr[i] = 0.f;
  for (int i = threadIdx.x; i < 3 * N; i += blockDim.x) dt[i] = 0.f;
  __syncthreads();
  if (threadIdx.x != 0) return;
  PoseQ p = pose_quat(r, N, id);
  const float qr = p.q2[0], x = p.q2[1], y = p.q2[2], z = p.q2[3];
  const float d0 = dW[0], d1 = dW[1], d2 = dW[2], d3 = dW[4], d4 = dW[5], d5 = dW[6], d6 = dW[8], d7 = dW[9],
              d8 = dW[10];
  float g[4];
  g[0] = 2.f * (-z * d1 + y * d2 + z * d3 - x * d5 - y * d6 + x * d7);
  g[1] = 2.f * (y * d1 + z * d2 + y * d3 - 2.f * x * d4 - qr * d5 + z * d6 + qr * d7 - 2.f * x * d8);
  g[2] = 2.f * (-2.f * y * d0 + x * d1 + qr * d2 + x * d3 + z * d5 - qr * d6 + z * d7 - 2.f * y * d8);
  g[3] = 2.f * (-2.f * z * d0 - qr * d1 + x * d2 + qr * d3 - 2.f * z * d4 + y * d5 + x * d6 + y * d7);
  // q2 = q1 / |q1|
  float dot = p.q2[0] * g[0] + p.q2[1] * g[1] + p.q2[2] * g[2] + p.q2[3] * g[3];
  float h[4];
#pragma unroll
  for (int k = 0; k < 4; k++) h[k] = (g[k] - p.q2[k] * dot) / p.n1;
  // q1 = r / max(|r|, eps)
  dot = p.q1[0] * h[0] + p.q1[1] * h[1] + p.q1[2] * h[2] + p.q1[3] * h[3];
#pragma unroll
  for (int k = 0; k < 4; k++) dr[k * N + id] = (h[k] - p.q1[k] * dot) / p.n0;
  dt[0 * N + id] = dW[3]; dt[1 * N + id] = dW[7]; dt[2 * N + id] = dW[11];
}

// One launch for the tail of a tracking iteration (train.py:186-195): dW = w_a dW_a + dW_b, the adjoint of
// LearnPose.forward for camera `id`, torch.optim.Adam over ALL of r [1,4,N] and t [3,N] (zero gradient outside
// column id, exactly what autograd would hand the optimizer), and the new w2c of camera id for the next
// iteration.  Replaces: add, pose backward, Adam, pose forward (4-6 launches of ~5 us each).
__global__ __launch_bounds__(256) void pose_adam_kernel(float *r, float *t, int N, int id, const float *dWa, float wa,
                                                        const float *dWb, float *m_r, float *v_r, float *m_t,
                                                        float *v_t, float ss_r, float ib_r, float ss_t, float ib_t,
                                                        float omb1, float b2, float omb2, float eps, float *w2c) {
  __shared__ float g7[7];
  if (threadIdx.x == 0) {
    float dW[12];
#pragma unroll
    for (int k = 0; k < 12; k++) dW[k] = wa * dWa[k] + (dWb ? dWb[k] : 0.f);
    PoseQ p = pose_quat(r, N, id);
    const float qr = p.q2[0], x = p.q2[1], y = p.q2[2], z = p.q2[3];
    const float d0 = dW[0], d1 = dW[1], d2 = dW[2], d3 = dW[4], d4 = dW[5], d5 = dW[6], d6 = dW[8], d7 = dW[9],
                d8 = dW[10];
    float g[4];
    g[0] = 2.f * (-z * d1 + y * d2 + z * d3 - x * d5 - y * d6 + x * d7);
    g[1] = 2.f * (y * d1 + z * d2 + y * d3 - 2.f * x * d4 - qr * d5 + z * d6 + qr * d7 - 2.f * x * d8);
    g[2] = 2.f * (-2.f * y * d0 + x * d1 + qr * d2 + x * d3 + z * d5 - qr * d6 + z * d7 - 2.f * y * d8);
    g[3] = 2.f * (-2.f * z * d0 - qr * d1 + x * d2 + qr * d3 - 2.f * z * d4 + y * d5 + x * d6 + y * d7);
    float dot = p.q2[0] * g[0] + p.q2[1] * g[1] + p.q2[2] * g[2] + p.q2[3] * g[3];
    float h[4];
#pragma unroll
    for (int k = 0; k < 4; k++) h[k] = (g[k] - p.q2[k] * dot) / p.n1;
    dot = p.q1[0] * h[0] + p.q1[1] * h[1] + p.q1[2] * h[2] + p.q1[3] * h[3];
#pragma unroll
    for (int k = 0; k < 4; k++) g7[k] = (h[k] - p.q1[k] * dot) / p.n0;
    g7[4] = dW[3]; g7[5] = dW[7]; g7[6] = dW[11];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 7 * N; e += blockDim.x) {
    const bool is_r = e < 4 * N;
    const int le = is_r ? e : e - 4 * N;
    const int row = le / N, col = le - row * N;
    const float g = col == id ? g7[is_r ? row : 4 + row] : 0.f;
    float *pp = is_r ? r + le : t + le, *mp = is_r ? m_r + le : m_t + le, *vp = is_r ? v_r + le : v_t + le;
    float pv = *pp, mv = *mp, vv = *vp;
    fsgs::adam_one(pv, g, mv, vv, omb1, b2, omb2, eps, is_r ? ss_r : ss_t, is_r ? ib_r : ib_t);
    *pp = pv; *mp = mv; *vp = vv;
  }
  __syncthreads();
  if (threadIdx.x == 0 && w2c) {
    PoseQ p = pose_quat(r, N, id);  // the updated quaternion
    const float qr = p.q2[0], x = p.q2[1], y = p.q2[2], z = p.q2[3];
    w2c[0] = 1.f - 2.f * (y * y + z * z); w2c[1] = 2.f * (x * y - qr * z);       w2c[2] = 2.f * (x * z + qr * y);
    w2c[4] = 2.f * (x * y + qr * z);       w2c[5] = 1.f - 2.f * (x * x + z * z); w2c[6] = 2.f * (y * z - qr * x);
    w2c[8] = 2.f * (x * z - qr * y);       w2c[9] = 2.f * (y * z + qr * x);       w2c[10] = 1.f - 2.f * (x * x + y * y);
    w2c[3] = t[0 * N + id]; w2c[7] = t[1 * N + id]; w2c[11] = t[2 * N + id];
    w2c[12] = 0.f; w2c[13] = 0.f; w2c[14] = 0.f; w2c[15] = 1.f;
  }
}

// The start of a tracked frame (train.py:322-331) in one launch: PoseModel.initialize_pose (scene/pose_optimizer.py:498-516;
// constant-velocity extrapolation of frames id-1, id-2 for id > 1, a copy of frame id-1 otherwise) and the moments of the
// FRESH Adam that initialize_tracking_optimizer builds for every frame (:489-496): zero.  In torch: ~20 small kernels for
// the pose + a new optimizer + scheduler object per frame (2.3 ms of host time per frame at C2, round 4).
__global__ __launch_bounds__(256) void pose_frame_begin_kernel(float *r, float *t, int N, int id, int extrapolate, float *m_r,
                                                               float *v_r, float *m_t, float *v_t) {
  if (m_r)
    for (int e = threadIdx.x; e < 4 * N; e += blockDim.x) m_r[e] = v_r[e] = 0.f;
  if (m_t)
    for (int e = threadIdx.x; e < 3 * N; e += blockDim.x) m_t[e] = v_t[e] = 0.f;
  if (threadIdx.x != 0 || id < 1) return;
  if (extrapolate && id > 1) {
    // F.normalize(x) = x / max(|x|_2, 1e-12), three times (both previous rotations, then their extrapolation)
    float a[4], b[4], c[4];
    float na = 0.f, nb = 0.f, nc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) { a[k] = r[k * N + id - 1]; b[k] = r[k * N + id - 2]; na += a[k] * a[k]; nb += b[k] * b[k]; }
    na = fmaxf(sqrtf(na), 1e-12f); nb = fmaxf(sqrtf(nb), 1e-12f);
#pragma unroll
    for (int k = 0; k < 4; k++) { a[k] /= na; b[k] /= nb; c[k] = a[k] + (a[k] - b[k]); nc += c[k] * c[k]; }
    nc = fmaxf(sqrtf(nc), 1e-12f);
#pragma unroll
    for (int k = 0; k < 4; k++) r[k * N + id] = c[k] / nc;
#pragma unroll
    for (int k = 0; k < 3; k++) { const float t1 = t[k * N + id - 1]; t[k * N + id] = t1 + (t1 - t[k * N + id - 2]); }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) r[k * N + id] = r[k * N + id - 1];
#pragma unroll
    for (int k = 0; k < 3; k++) t[k * N + id] = t[k * N + id - 1];
  }
}

}  // namespace

extern "C" {

int fsgs_pose_frame_begin(float *r, float *t, int num_cams, int cam_id, int extrapolate, float *exp_avg_r,
                          float *exp_avg_sq_r, float *exp_avg_t, float *exp_avg_sq_t, fsgs_stream_t stream) {
  if (!r || !t || num_cams <= 0 || cam_id < 0 || cam_id >= num_cams) return FSGS_ERR_INVALID;
  if ((exp_avg_r == nullptr) != (exp_avg_sq_r == nullptr) || (exp_avg_t == nullptr) != (exp_avg_sq_t == nullptr))
    return FSGS_ERR_INVALID;
  hipLaunchKernelGGL(pose_frame_begin_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, r, t, num_cams, cam_id,
                     extrapolate, exp_avg_r, exp_avg_sq_r, exp_avg_t, exp_avg_sq_t);
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

int fsgs_pose_adam_step(float *r, float *t, int num_cams, int cam_id, const float *dw2c_a, float weight_a,
                        const float *dw2c_b, float *exp_avg_r, float *exp_avg_sq_r, float *exp_avg_t,
                        float *exp_avg_sq_t, float lr_r, float lr_t, int step_r, int step_t, double beta1, double beta2,
                        double eps, float *w2c_next, fsgs_stream_t stream) {
  if (!r || !t || !dw2c_a || !exp_avg_r || !exp_avg_sq_r || !exp_avg_t || !exp_avg_sq_t || num_cams <= 0 ||
      cam_id < 0 || cam_id >= num_cams || step_r < 1 || step_t < 1) {
    if (hipEvent_t done = fsgs::take_pose_step_done_event()) (void)hipEventRecord(done, (hipStream_t)stream);
    return FSGS_ERR_INVALID;
  }
  // the host arithmetic of fsgs_adam_step
  const float ss_r = (float)((double)lr_r / (1.0 - pow(beta1, (double)step_r)));
  const float ib_r = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step_r)));
  const float ss_t = (float)((double)lr_t / (1.0 - pow(beta1, (double)step_t)));
  const float ib_t = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step_t)));
  // fsgs_pose_step_done_event: the update's own completion signals the event (no marker packet behind the launch), so that
  // the next iteration's second stream can wait for the new pose while this stream goes straight on
  if (hipEvent_t done = fsgs::take_pose_step_done_event())
    hipExtLaunchKernelGGL(pose_adam_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, nullptr, done, 0, r, t, num_cams,
                          cam_id, dw2c_a, weight_a, dw2c_b, exp_avg_r, exp_avg_sq_r, exp_avg_t, exp_avg_sq_t, ss_r, ib_r, ss_t,
                          ib_t, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, w2c_next);
  else
    hipLaunchKernelGGL(pose_adam_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, r, t, num_cams, cam_id, dw2c_a,
                       weight_a, dw2c_b, exp_avg_r, exp_avg_sq_r, exp_avg_t, exp_avg_sq_t, ss_r, ib_r, ss_t, ib_t,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, w2c_next);
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}


int fsgs_pose_forward(const float *r, const float *t, int num_cams, int cam_id, float *w2c, fsgs_stream_t stream) {
  if (!r || !t || !w2c || num_cams <= 0 || cam_id < 0 || cam_id >= num_cams) return FSGS_ERR_INVALID;
  hipLaunchKernelGGL(pose_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, r, t, num_cams, cam_id, w2c);
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

int fsgs_pose_backward(const float *r, int num_cams, int cam_id, const float *dw2c, float *dr, float *dt,
                       fsgs_stream_t stream) {
  if (!r || !dw2c || !dr || !dt || num_cams <= 0 || cam_id < 0 || cam_id >= num_cams) return FSGS_ERR_INVALID;
  hipLaunchKernelGGL(pose_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, r, num_cams, cam_id, dw2c, dr, dt);
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

}  // extern "C"
