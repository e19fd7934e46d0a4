// mfma_reduce.hip -- the 64-lane x 24-value transposing reduction of blend_bwd: permlane swaps (shipping) against the
// matrix-pipe version (wave_mfma_reduce32_12of16).  Checks the lane -> slot mapping of both against plain sums, then times
// RESULT (MI355X, profiles/r02_mfma_reduce_ubench.txt): correct, but 2x slower than the swaps on its own and strictly
// additive to the other waves' VALU work -- f32 MFMA shares the vector ALUs' issue, nothing overlaps.  Not used.
// a loop of  [F independent FMAs per lane + one reduction]  at 4 waves per SIMD (what the backward runs at), F = 0 and
// F = 160 (the VALU work of two pairs' quadrant bodies), to see whether the MFMAs overlap the other waves' VALU work.
//   hipcc --offload-arch=gfx950 -O3 -I free-surgs_amd/csrc -o scripts/ubench/mfma_reduce.bin scripts/ubench/mfma_reduce.hip
#include "fsgs_device.h"
#include "selftest_reductions.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using namespace fsgs;

// ---- the same 12-of-16 reduction with the first two halving steps on the MATRIX pipe ---------------------------------
// v_mfma_f32_16x16x4_f32 contracts over the four 16-lane groups of the wave: D[i][j] = sum_k A[i][k] B[k][j] with
// A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k and D[i][j] in lane j + 16 (i / 4), register i % 4.  With B = one
// of the values and A = the 0/1 indicator "row i belongs to lane group q", the product leaves, in every lane of group q,
// the sum of that value over the lanes j, j+16, j+32, j+48 -- and zeros elsewhere, so FOUR values (q = 0..3) accumulate
// into one D and each lane group ends up with a different one.  That is the work of the permlane32 / permlane16 swap
// steps (20 swaps + 20 adds, the swaps at 2.5x the issue cost of a plain VALU instruction) done by 24 MFMAs on a pipe
// the blend kernels otherwise leave idle; the remaining three lane bits fold with DPP as before.
// out: lane l holds the total of v[reduce_mfma_slot32(l)] (lanes l and l ^ 1 the same value); slots 12..15 and 28..31
// must be zero on entry and are not reduced.
typedef float fsgs_f32x4 __attribute__((ext_vector_type(4)));
__host__ __device__ inline int reduce_mfma_slot32(int lane) {
  return 16 * ((lane >> 3) & 1) + 4 * ((lane >> 1) & 3) + (lane >> 4);
}
__device__ __forceinline__ float wave_mfma_reduce32_12of16(const float (&v)[32], int lane) {
  const int q = (lane & 15) >> 2;
  const float m[4] = {q == 0 ? 1.f : 0.f, q == 1 ? 1.f : 0.f, q == 2 ? 1.f : 0.f, q == 3 ? 1.f : 0.f};
  float P[8];
#pragma unroll
  for (int t = 0; t < 8; t++) {
    if ((t & 3) == 3) { P[t] = 0.f; continue; }  // slots 12..15 of either Gaussian
    fsgs_f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; k++) d = __builtin_amdgcn_mfma_f32_16x16x4f32(m[k], v[4 * t + k], d, 0, 0, 0);
    P[t] = d[0];  // all four registers hold the same row sums
  }
  const float w0 = fold_bit3(P[0], P[4]), w1 = fold_bit3(P[1], P[5]), w2 = fold_bit3(P[2], P[6]);  // lane bit 3: Gaussian
  const float y0 = fold_bit2(w0, w2), y1 = fold_bit2(w1, 0.f);                                      // lane bit 2: slot bit 3
  const float z = fold_dpp<0x4E>(y0, y1, (lane & 2) != 0);                                          // lane bit 1: slot bit 2
  return dpp_add<0xB1>(z);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ void check(const float *in, float *out) {  // in[lane][32]
  float v[32];
  for (int i = 0; i < 32; i++) v[i] = ((i & 15) < 12) ? in[threadIdx.x * 32 + i] : 0.f;
  out[threadIdx.x] = MODE ? wave_mfma_reduce32_12of16(v, threadIdx.x) : wave_transpose_reduce32_12of16(v, threadIdx.x);
}

template <int MODE, int F>
__global__ __launch_bounds__(256) void loop(float *out, int iters, float b, float c) {
  float v[32], f[8];
  const int lane = threadIdx.x & 63;
  for (int i = 0; i < 32; i++) v[i] = ((i & 15) < 12) ? 0.001f * (threadIdx.x + i) : 0.f;
  for (int i = 0; i < 8; i++) f[i] = 0.002f * (threadIdx.x + i);
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < F / 8; k++)
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(b), "v"(c));
#pragma unroll
    for (int i = 0; i < 32; i++)
      if ((i & 15) < 12) v[i] = v[i] + f[i & 7];  // the accumulators change every round (24 adds, both variants)
    acc += MODE ? wave_mfma_reduce32_12of16(v, lane) : wave_transpose_reduce32_12of16(v, lane);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + f[0];
}

template <int MODE, int F>
static float time_loop(float *d_out, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, iters = 2000;  // 256 CUs x 4 SIMDs; 4 waves per block
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(e0));
    loop<MODE, F><<<blocks, 256>>>(d_out, iters, 0.999f, 0.001f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) best = fminf(best, ms);
  }
  return best * 1e6f / iters / waves_per_simd;  // ns per (iteration of one wave) per SIMD
}

int main() {
  std::vector<float> in(64 * 32), sums(32, 0.f);
  srand(3);
  for (auto &x : in) x = (float)rand() / RAND_MAX - 0.5f;
  for (int l = 0; l < 64; l++)
    for (int i = 0; i < 32; i++) if ((i & 15) < 12) sums[i] += in[l * 32 + i];
  float *d_in, *d_out;
  CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_out, 4 * 256 * 256 * 8));
  CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
  for (int mode = 0; mode < 2; mode++) {
    if (mode) check<1><<<1, 64>>>(d_in, d_out); else check<0><<<1, 64>>>(d_in, d_out);
    CK(hipDeviceSynchronize());
    float out[64];
    CK(hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost));
    float worst = 0.f;
    for (int l = 0; l < 64; l++) {
      const int slot = mode ? reduce_mfma_slot32(l) : (l >> 1);
      const float want = (slot & 15) < 12 ? sums[slot] : 0.f;
      worst = fmaxf(worst, fabsf(out[l] - want));
    }
    printf("%s: max |lane total - plain sum| = %.3g %s\n", mode ? "mfma reduction   " : "permlane reduction", worst, worst < 1e-5f ? "(ok)" : "(MISMATCH)");
  }
  for (int w : {4, 5}) {
    printf("%d waves/SIMD, ns per round and SIMD:  reduction only: permlane %.1f  mfma %.1f   |  + 160 FMAs: permlane %.1f  mfma %.1f   (160 FMAs alone: %.1f)\n", w,
           time_loop<0, 0>(d_out, w), time_loop<1, 0>(d_out, w), time_loop<0, 160>(d_out, w), time_loop<1, 160>(d_out, w),
           time_loop<0, 160>(d_out, w) - time_loop<0, 0>(d_out, w));
  }
  return 0;
}
