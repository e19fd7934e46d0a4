// mfma_valu_overlap.hip -- is the f32 matrix pipe a SECOND pipe beside the VALU when the two kinds of work come from
// DIFFERENT waves of one SIMD?  (VERDICT r5 #1 c: "the reduction on the idle matrix pipe from a different wave than the one
// doing VALU work"; round 2's scripts/ubench/mfma_reduce.hip had every wave do both.)
// One 512-thread workgroup per CU (100 KB of dynamic LDS keeps a second one out): eight waves, two per SIMD.  Waves 0..3 run
// a VALU loop (8 independent v_fma_f32 chains), waves 4..7 an MFMA loop (4 independent accumulators), each for a fixed number
// of instructions, each stamping s_memrealtime around its own loop.  Three launches per MFMA kind: VALU waves alone, MFMA waves
// alone, both.  If the pipes are independent, "both" lasts as long as the longer of the two alone; if the MFMA runs on the
// VALU's multipliers, "both" lasts as long as their sum.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/mfma_valu_overlap.bin scripts/ubench/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// KIND 0: v_mfma_f32_16x16x4_f32   KIND 1: v_mfma_f32_32x32x2_f32   KIND 2: v_mfma_f32_16x16x16_bf16
template <int KIND>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *st, int valu_iters, int mfma_iters, int run_valu,
                                         int run_mfma, float b, float c) {
  extern __shared__ float pad[];
  const int wave = threadIdx.x >> 6;
  const bool is_valu = wave < 4;
  if (threadIdx.x == 0) pad[0] = b;
  float acc = 0.f;
  unsigned long long t0 = 0, t1 = 0;
  if (is_valu && run_valu) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
    t0 = wall_clock64();
    for (int it = 0; it < valu_iters; it++) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    t1 = wall_clock64();
    for (int i = 0; i < 8; i++) acc += a[i];
  } else if (!is_valu && run_mfma) {
    t0 = wall_clock64();
    if constexpr (KIND == 0) {
      f32x4 d[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      const float x = threadIdx.x * 0.001f, y = 0.5f;
      for (int it = 0; it < mfma_iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) d[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, d[i], 0, 0, 0);
      }
      t1 = wall_clock64();
      for (int i = 0; i < 4; i++) acc += d[i][0] + d[i][1] + d[i][2] + d[i][3];
    } else if constexpr (KIND == 1) {
      typedef float f32x16 __attribute__((ext_vector_type(16)));
      f32x16 d[2] = {};
      const float x = threadIdx.x * 0.001f, y = 0.5f;
      for (int it = 0; it < mfma_iters; it++) {
#pragma unroll
        for (int i = 0; i < 2; i++) d[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, d[i], 0, 0, 0);
      }
      t1 = wall_clock64();
      for (int i = 0; i < 2; i++) acc += d[i][0] + d[i][5] + d[i][15];
    } else {
      f32x4 d[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      const s16x4 x = {(short)0x3f80, (short)0x3f00, (short)0x3e80, (short)(threadIdx.x & 0x3f00)}, y = {(short)0x3f80, 0, 0, 0};
      for (int it = 0; it < mfma_iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) d[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, d[i], 0, 0, 0);
      }
      t1 = wall_clock64();
      for (int i = 0; i < 4; i++) acc += d[i][0] + d[i][1] + d[i][2] + d[i][3];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) st[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int mfma_per_iter, float *d_out, unsigned long long *d_st) {
  // both loops last ~1 ms alone (one wave per SIMD: v_fma ~3.8 ns, 16x16x4 f32 ~13 ns, 32x32x2 f32 ~27 ns, 16x16x16 bf16 ~7 ns)
  const int blocks = 256, valu_iters = 8000, mfma_iters = KIND == 2 ? 40000 : 20000;
  double v_alone = 0, m_alone = 0, v_both = 0, m_both = 0;
  for (int mode = 0; mode < 3; mode++) {  // 0: VALU waves alone, 1: MFMA waves alone, 2: both
    const int rv = mode != 1, rm = mode != 0;
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(512), 100 * 1024, 0, d_out, d_st, valu_iters, mfma_iters, rv, rm, 0.999f,
                         0.001f);
      CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> st((size_t)blocks * 8);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    double tv = 0, tm = 0;
    for (int b = 0; b < blocks; b++)
      for (int w = 0; w < 8; w++) (w < 4 ? tv : tm) += (double)st[(size_t)b * 8 + w] * 10.0 / (blocks * 4);  // ns (100 MHz ticks)
    if (mode == 0) v_alone = tv;
    if (mode == 1) m_alone = tm;
    if (mode == 2) { v_both = tv; m_both = tm; }
  }
  const double nv = (double)valu_iters * 32, nm = (double)mfma_iters * mfma_per_iter;
  printf("%-26s VALU wave alone %7.1f us (%5.2f ns/v_fma)  MFMA wave alone %7.1f us (%6.2f ns/mfma)  |  together: VALU wave %7.1f us, MFMA wave %7.1f us"
         "  => longer-alone %.1f, sum %.1f, measured %.1f us: %s\n",
         name, v_alone * 1e-3, v_alone / nv, m_alone * 1e-3, m_alone / nm, v_both * 1e-3, m_both * 1e-3,
         fmax(v_alone, m_alone) * 1e-3, (v_alone + m_alone) * 1e-3, fmax(v_both, m_both) * 1e-3,
         fmax(v_both, m_both) < 0.5 * (fmax(v_alone, m_alone) + v_alone + m_alone) ? "OVERLAP (closer to the longer one)" : "NO overlap (closer to the sum)");
}

int main() {
  CK(hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  CK(hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  CK(hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  float *d_out; unsigned long long *d_st;
  CK(hipMalloc(&d_out, 4 * 512 * 256)); CK(hipMalloc(&d_st, 8 * 256 * 8));
  run<0>("v_mfma_f32_16x16x4_f32", 4, d_out, d_st);
  run<1>("v_mfma_f32_32x32x2_f32", 2, d_out, d_st);
  run<2>("v_mfma_f32_16x16x16_bf16", 4, d_out, d_st);
  return 0;
}
