// splat_parallel_bwd.hip -- prototype of the SPLAT-PARALLEL blend backward, built to be measured against
// the pixel-parallel kernel that ships (csrc/raster_kernels.h blend_bwd_kernel):
//   one wave per 16x16 tile; the tile's depth-sorted list is walked BACK TO FRONT in batches of 64 Gaussians; lane j of a
//   batch OWNS one Gaussian (its record and its 12 gradient sums live in registers for the whole batch), the wave loops
//   over the tile's 256 pixels, whose state (dL/dpixel[6], transmittance behind the batch, the scalar "colour behind"
//   S = sum_i alpha_i T_i (dL . c_i) + T_final (dL . bg)) sits in LDS and is read as a broadcast.  Per pixel step the
//   transmittance in front of each Gaussian is an inclusive PRODUCT SCAN over the lanes and the colour behind it a SUM
//   SCAN (lanes hold the batch back-most first, so both are DPP prefix scans: row_shr 1,2,4,8 + row_bcast 15,31).
//   No 64-lane transposing reduction, one accumulator row of atomics per (tile, Gaussian) -- exactly the design asked for.
// The host checks one tile against a double-precision loop, then times 5120 tiles x L pairs (C2: 1280x1024, R = 1.03 M
// pairs => L = 202) with HIP events.  Same 64-byte records, same 64-byte accumulator rows as the shipping kernel.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/splat_parallel_bwd.bin scripts/ubench/splat_parallel_bwd.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// inclusive prefix scans over the 64 lanes for TWO pixels at once, DPP modifiers fused into the arithmetic (lanes without a
// source keep their value: bound_ctrl off); the two pixels' steps alternate so that one s_nop covers the DPP read hazard
#define SCAN2(OP)                                                                                                   \
  asm("s_nop 1\n\t" OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" OP " %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
      "s_nop 0\n\t" OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" OP " %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
      "s_nop 0\n\t" OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" OP " %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
      "s_nop 0\n\t" OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" OP " %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
      "s_nop 0\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" OP " %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
      "s_nop 0\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" OP " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" \
      : "+v"(u), "+v"(v))
__device__ __forceinline__ void scan2_mul(float &u, float &v) { SCAN2("v_mul_f32_dpp"); }
__device__ __forceinline__ void scan2_add(float &u, float &v) { SCAN2("v_add_f32_dpp"); }

// pix: per tile 256 x 2 float4 = {dL0..3 | dL4, dL5, final_T, S0 = T_final (dL . bg)}; ncon: per tile 256 ints (last contributor)
__global__ __launch_bounds__(64) void splat_bwd(int L, const int *__restrict__ plist, const float4 *__restrict__ rec,
                                                const float4 *__restrict__ pix, const int *__restrict__ ncon,
                                                float *__restrict__ acc, int tiles_x) {
  __shared__ float4 px[256][2];
  __shared__ int nc[256];
  const int lane = threadIdx.x, tile = blockIdx.x;
  for (int p = lane; p < 256; p += 64) {
    px[p][0] = pix[(size_t)tile * 512 + 2 * p];
    px[p][1] = pix[(size_t)tile * 512 + 2 * p + 1];
    nc[p] = ncon[(size_t)tile * 256 + p];
  }
  __syncthreads();
  const float x0 = (float)((tile % tiles_x) * 16), y0 = (float)((tile / tiles_x) * 16);
  const int nb = (L + 63) / 64;
  for (int b = nb - 1; b >= 0; --b) {
    const int idx = b * 64 + (63 - lane);  // lane 0 = back-most Gaussian of the batch
    const bool live = idx < L;
    const int id = live ? plist[(size_t)tile * L + idx] : 0;
    const float4 r0 = rec[(size_t)id * 4], r1 = rec[(size_t)id * 4 + 1], r2 = rec[(size_t)id * 4 + 2], r3 = rec[(size_t)id * 4 + 3];
    const float gx = r0.x - x0, gy = r0.y - y0, ca = r0.z, cb = r0.w, cc = r1.x, op = live ? r1.y : 0.f;
    const float col[6] = {r2.x, r2.y, r2.z, r2.w, r3.x, r3.y};
    float gcol[6] = {0, 0, 0, 0, 0, 0}, m[6] = {0, 0, 0, 0, 0, 0};
    for (int p = 0; p < 256; p += 2) {  // two pixels of one row per step
      float4 q0[2], q1[2];
      float dx[2], dy, a[2], alpha[2], om[2], prod[2], Tj[2], dc[2], v[2], incl[2];
      dy = gy - (float)(p >> 4);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        q0[k] = px[p + k][0]; q1[k] = px[p + k][1];
        dx[k] = gx - (float)((p + k) & 15);
        const float pw = fmaf(cc * dy, dy, fmaf(cb, dy, ca * dx[k]) * dx[k]);  // prescaled: exp2 directly
        const float a0 = op * __builtin_amdgcn_exp2f(pw);
        const bool ok = (int)!(pw > 0.f) & (int)(a0 >= (1.f / 255.f)) & (int)(idx <= nc[p + k]);
        a[k] = ok ? a0 : 0.f;
        alpha[k] = fminf(0.99f, a[k]);
        prod[k] = om[k] = 1.f - alpha[k];
      }
      scan2_mul(prod[0], prod[1]);  // product over this Gaussian and everything behind it in the batch
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        Tj[k] = q1[k].z * __builtin_amdgcn_rcpf(prod[k]);  // transmittance in front of Gaussian j
        dc[k] = fmaf(q1[k].y, col[5], fmaf(q1[k].x, col[4], fmaf(q0[k].w, col[3], fmaf(q0[k].z, col[2], fmaf(q0[k].y, col[1], q0[k].x * col[0])))));
        v[k] = alpha[k] * Tj[k];
        incl[k] = v[k] * dc[k];
      }
      const float w0 = incl[0], w1 = incl[1];
      scan2_add(incl[0], incl[1]);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float w = k ? w1 : w0;
        const float behind = q1[k].w + (incl[k] - w);
        const float dLda = fmaf(Tj[k], dc[k], -behind * __builtin_amdgcn_rcpf(om[k]));
        const float wp = (a[k] < 0.99f ? a[k] : 0.f) * dLda;  // alpha's derivative is zero where it was clamped
        gcol[0] = fmaf(v[k], q0[k].x, gcol[0]); gcol[1] = fmaf(v[k], q0[k].y, gcol[1]); gcol[2] = fmaf(v[k], q0[k].z, gcol[2]);
        gcol[3] = fmaf(v[k], q0[k].w, gcol[3]); gcol[4] = fmaf(v[k], q1[k].x, gcol[4]); gcol[5] = fmaf(v[k], q1[k].y, gcol[5]);
        const float wx = wp * dx[k], wy = wp * dy;
        m[0] += wp; m[1] += wx; m[2] += wy;
        m[3] = fmaf(wx, dx[k], m[3]); m[4] = fmaf(wx, dy, m[4]); m[5] = fmaf(wy, dy, m[5]);
        if (lane == 63) {  // front-most lane: transmittance and colour-behind seen by the next batch (the one in front)
          px[p + k][1].z = Tj[k];
          px[p + k][1].w = q1[k].w + incl[k];
        }
      }
    }
    if (live) {
      float *row = acc + (size_t)id * 16;
#pragma unroll
      for (int k = 0; k < 6; ++k) atomicAdd(row + k, m[k]);
#pragma unroll
      for (int k = 0; k < 6; ++k) atomicAdd(row + 8 + k, gcol[k]);
    }
    __syncthreads();
  }
}

static float frand() { return (float)rand() / (float)RAND_MAX; }

int main(int argc, char **argv) {
  const int tiles_x = 80, tiles = 80 * 64, P = 80 * 64 * 202;  // one record per pair (ids permuted: a gather)
  const float LOG2E = 1.4426950408889634f;
  srand(1);
  std::vector<float> rec((size_t)P * 16);
  for (int i = 0; i < P; ++i) {
    float *r = &rec[(size_t)i * 16];
    float s = 3.f + 9.f * frand();
    r[0] = 0; r[1] = 0;  // centre is set per use below (relative to the tile that lists it)
    r[2] = -LOG2E * 0.5f / (s * s); r[3] = -LOG2E * 0.3f / (s * s) * (frand() - 0.5f); r[4] = -LOG2E * 0.5f / (s * s);
    r[5] = 0.02f + 0.2f * frand();
    for (int k = 0; k < 6; ++k) r[8 + k] = frand();
  }
  for (int run = 0; run < 2; ++run) {
    const int L = run == 0 ? 202 : 128;
    std::vector<int> plist((size_t)tiles * L);
    for (int t = 0; t < tiles; ++t)
      for (int i = 0; i < L; ++i) {
        int id = (int)((((size_t)t * L + i) * 7919u) % P);
        plist[(size_t)t * L + i] = id;
        rec[(size_t)id * 16 + 0] = (t % tiles_x) * 16 + 24.f * frand() - 4.f;
        rec[(size_t)id * 16 + 1] = (t / tiles_x) * 16 + 24.f * frand() - 4.f;
      }
    std::vector<float> pix((size_t)tiles * 2048);
    std::vector<int> ncon((size_t)tiles * 256);
    for (size_t i = 0; i < (size_t)tiles * 256; ++i) {
      for (int k = 0; k < 6; ++k) pix[i * 8 + k] = frand() - 0.5f;
      ncon[i] = L - 1 - (rand() % 8);
    }
    // forward on the host for final_T (so that T / prod reproduces the forward's transmittances)
    for (int t = 0; t < tiles; ++t)
      for (int p = 0; p < 256; ++p) {
        double T = 1;
        float xx = (t % tiles_x) * 16 + (p & 15), yy = (t / tiles_x) * 16 + (p >> 4);
        for (int i = 0; i <= ncon[(size_t)t * 256 + p]; ++i) {
          const float *r = &rec[(size_t)plist[(size_t)t * L + i] * 16];
          float dx = r[0] - xx, dy = r[1] - yy, pw = r[4] * dy * dy + (r[3] * dy + r[2] * dx) * dx;
          float a = r[5] * exp2f(pw);
          if (pw > 0 || a < 1.f / 255.f) continue;
          T *= 1.0 - fmin(0.99f, a);
        }
        pix[((size_t)t * 256 + p) * 8 + 6] = (float)T;
        pix[((size_t)t * 256 + p) * 8 + 7] = (float)T * 0.1f * (pix[((size_t)t * 256 + p) * 8] + pix[((size_t)t * 256 + p) * 8 + 1]);
      }
    int *d_plist, *d_ncon; float4 *d_rec, *d_pix; float *d_acc;
    CK(hipMalloc(&d_plist, plist.size() * 4)); CK(hipMalloc(&d_ncon, ncon.size() * 4)); CK(hipMalloc(&d_rec, rec.size() * 4));
    CK(hipMalloc(&d_pix, pix.size() * 4)); CK(hipMalloc(&d_acc, (size_t)P * 64));
    CK(hipMemcpy(d_plist, plist.data(), plist.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ncon, ncon.data(), ncon.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pix, pix.data(), pix.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_acc, 0, (size_t)P * 64));
    splat_bwd<<<tiles, 64>>>(L, d_plist, d_rec, d_pix, d_ncon, d_acc, tiles_x);
    CK(hipDeviceSynchronize());
    // check tile 7 against a double-precision back-to-front loop
    {
      const int t = 7;
      std::vector<float> acc((size_t)P * 16);
      CK(hipMemcpy(acc.data(), d_acc, acc.size() * 4, hipMemcpyDeviceToHost));
      std::vector<double> want((size_t)L * 12, 0.0);
      for (int p = 0; p < 256; ++p) {
        const float *q = &pix[((size_t)t * 256 + p) * 8];
        double T = q[6], S = q[7];
        float xx = (t % tiles_x) * 16 + (p & 15), yy = (t / tiles_x) * 16 + (p >> 4);
        for (int i = ncon[(size_t)t * 256 + p]; i >= 0; --i) {
          const float *r = &rec[(size_t)plist[(size_t)t * L + i] * 16];
          float dx = r[0] - xx, dy = r[1] - yy, pw = r[4] * dy * dy + (r[3] * dy + r[2] * dx) * dx;
          float a = r[5] * exp2f(pw);
          if (pw > 0 || a < 1.f / 255.f) continue;
          double al = fmin(0.99f, a);
          T /= (1.0 - al);
          double dc = 0;
          for (int k = 0; k < 6; ++k) dc += (double)q[k] * r[8 + k];
          double dLda = T * dc - S / (1.0 - al);
          double wp = (a < 0.99f ? al : 0.0) * dLda;
          double *w = &want[(size_t)i * 12];
          w[0] += wp; w[1] += wp * dx; w[2] += wp * dy; w[3] += wp * dx * dx; w[4] += wp * dx * dy; w[5] += wp * dy * dy;
          for (int k = 0; k < 6; ++k) w[6 + k] += al * T * q[k];
          S += al * T * dc;
        }
      }
      double worst = 0, scale = 0;
      for (int i = 0; i < L; ++i) {
        const float *g = &acc[(size_t)plist[(size_t)t * L + i] * 16];
        for (int k = 0; k < 12; ++k) {
          double got = k < 6 ? g[k] : g[8 + k - 6];
          worst = fmax(worst, fabs(got - want[(size_t)i * 12 + k]));
          scale = fmax(scale, fabs(want[(size_t)i * 12 + k]));
        }
      }
      printf("L=%d  check tile %d: max |got - want| = %.3g on a scale of %.3g  (%s)\n", L, t, worst, scale,
             worst <= 2e-4 * scale ? "ok" : "MISMATCH");
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      splat_bwd<<<tiles, 64>>>(L, d_plist, d_rec, d_pix, d_ncon, d_acc, tiles_x);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = fminf(best, ms);
    }
    printf("L=%d  pairs=%.3f M  splat-parallel backward: %.1f us  (%.0f SIMD-ns per 64-pair batch x pixel step on 1024 SIMDs; %.2f ns per pair)\n", L,
           tiles * (double)L * 1e-6, best * 1e3, 1024.0 * best * 1e6 / (tiles * (double)((L + 63) / 64) * 256), best * 1e6 / (tiles * (double)L));
    hipFree(d_plist); hipFree(d_ncon); hipFree(d_rec); hipFree(d_pix); hipFree(d_acc);
  }
  return 0;
}
