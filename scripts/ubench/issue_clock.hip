// issue_clock.hip -- (1) what shader clock does an MI355X hold while every SIMD issues VALU work back to back, and (2) how
// many wave64 VALU instructions per shader cycle does a SIMD retire at W = 1..8 resident waves?  (VERDICT r5 #4 a, b: the
// bench line's `valu_frac` used an ASSUMED 2.4 GHz.)
// Every wave stamps s_memtime (shader clock, the guide: "tick = shader cycle") and s_memrealtime (constant 100 MHz) around its
// loop; the ratio of the two deltas is the clock the wave ran at, with no assumption.  The loop bodies:
//   fma   : 8 independent v_fma_f32 chains per lane (the issue-rate ceiling)
//   blend : the instruction MIX of the backward blend body per 40 instructions -- 1 v_exp_f32, 1 v_rcp_f32, 3 v_cmp/v_cndmask
//           pairs, 3 broadcast ds_read_b128 per two bodies, the rest FMA / mul / add -- what `valu_frac_of_achievable` is
//           quoted against
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/issue_clock.bin scripts/ubench/issue_clock.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp {
  unsigned long long cyc, rt;
};
__device__ __forceinline__ Stamp stamp() {
  Stamp s;
  s.cyc = __builtin_readcyclecounter();  // s_memtime
  s.rt = wall_clock64();                 // s_memrealtime, 100 MHz
  return s;
}

// `gate`: every workgroup checks in and spins (bounded) until all of the launch have -- so the timed loops of ALL waves overlap,
// W per SIMD, or the run says that they did not (round 6's first version timed W = 8 launches whose waves lasted half the
// kernel: the launch ramp / placement had put them in two generations, and "8 waves" meant ~4.3 resident).
template <int MIX>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *st, int iters, float b, float c, unsigned *gate) {
  __shared__ float4 lds[64];
  float a[8];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
  if (threadIdx.x < 64) lds[threadIdx.x] = make_float4(a[0], a[1], b, c);
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(gate, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
    if (spins >= (1 << 22)) __hip_atomic_fetch_add(gate + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // gave up
  }
  __syncthreads();
  const Stamp s0 = stamp();
  for (int it = 0; it < iters; it++) {
    if (MIX == 0) {
#pragma unroll
      for (int r = 0; r < 5; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    } else {
      // 40 VALU: 1 exp, 1 rcp, 3 x (cmp + cndmask), 32 FMA-class; + 1.5 broadcast LDS reads
      const float4 r0 = lds[it & 63];
      float4 r1 = r0;
      if (it & 1) r1 = lds[(it + 7) & 63];
      asm volatile("v_exp_f32 %0, %0" : "+v"(a[0]));
      asm volatile("v_rcp_f32 %0, %0" : "+v"(a[1]));
#pragma unroll
      for (int i = 0; i < 3; i++) {
        asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[2 + i]) : "v"(b), "v"(c) : "vcc");
      }
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(r0.z), "v"(r1.w));
    }
  }
  const Stamp s1 = stamp();
  float acc = 0.f;
  for (int i = 0; i < 8; i++) acc += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    st[2 * w + 0] = s1.cyc - s0.cyc;
    st[2 * w + 1] = s1.rt - s0.rt;
  }
}

template <int MIX>
static void run(int W, float *d_out, unsigned long long *d_st, unsigned *d_gate) {
  const int blocks = 256 * W, iters = 20000, per_iter = 40;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipMemsetAsync(d_gate, 0, 8, 0));
    CK(hipEventRecord(e0));
    k<MIX><<<blocks, 256>>>(d_out, d_st, iters, 0.999f, 0.001f, d_gate);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = fminf(best, ms);
  }
  std::vector<unsigned long long> st(2 * (size_t)blocks * 4);
  CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, rt = 0;
  for (size_t w = 0; w < (size_t)blocks * 4; w++) { cyc += (double)st[2 * w]; rt += (double)st[2 * w + 1]; }
  unsigned gate[2];
  CK(hipMemcpy(gate, d_gate, 8, hipMemcpyDeviceToHost));
  const double mhz = cyc / rt * 100.0;                       // shader cycles per 100 MHz tick, over every wave's own loop
  const double n_inst = (double)iters * per_iter * W;        // wave-instructions per SIMD
  const double wave_ms = rt / ((double)blocks * 4) * 1e-5;   // MEAN duration of a wave's timed loop
  // The rate is taken from the KERNEL's duration: a SIMD arbitrates VALU issue by age, so its W waves do not advance together --
  // the oldest finishes first, the youngest last -- and the mean wave duration is well below the time the SIMD needs for all of
  // them (W = 8: 3.9 ms against 6.6 ms; the first version of this file divided by the wave mean and "measured" 0.74 instructions
  // per cycle, above the 32-lanes-per-cycle datapath).
  const double ns = best * 1e6 / n_inst;
  printf("%-5s W=%d waves/SIMD: kernel %7.3f ms (mean wave loop %7.3f ms; %s)  shader clock %6.0f MHz  %6.3f ns/inst/SIMD = %5.2f cycles/inst/SIMD "
         "(%4.2f inst/cycle/SIMD)\n",
         MIX ? "blend" : "fma", W, best, wave_ms, gate[1] ? "NOT all resident at once" : "all waves resident at once",
         mhz, ns, ns * mhz * 1e-3, 1.0 / (ns * mhz * 1e-3));
}

int main() {
  float *d_out; unsigned long long *d_st; unsigned *d_gate;
  CK(hipMalloc(&d_out, 4 * 256 * 256 * 8)); CK(hipMalloc(&d_st, 16 * 256 * 8 * 4)); CK(hipMalloc(&d_gate, 8));
  for (int W : {1, 2, 3, 4, 5, 6, 8}) run<0>(W, d_out, d_st, d_gate);
  for (int W : {1, 2, 3, 4, 5, 6, 8}) run<1>(W, d_out, d_st, d_gate);
  return 0;
}
