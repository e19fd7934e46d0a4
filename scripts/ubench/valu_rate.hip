// VALU issue-rate microbenchmark: how many cycles does a wave64 v_mul / v_add / v_fma / v_cndmask / v_exp /
// v_readlane cost on gfx950?  One workgroup of 64 threads per SIMD slot, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int OP>
__global__ __launch_bounds__(64) void k(float *out, int iters) {
  float a[8];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
  float b = 1.0001f, c = 0.5f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) a[i] = __fmul_rn(a[i], b);
      if (OP == 1) a[i] = __fadd_rn(a[i], c);
      if (OP == 2) a[i] = __fmaf_rn(a[i], b, c);
      if (OP == 3) a[i] = a[i] > c ? b : a[i] + 0.f;  // cmp + cndmask
      if (OP == 4) a[i] = __expf(a[i]) ;
      if (OP == 5) a[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[i]), it & 63)) + a[i];
      if (OP == 6) { int t = __builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0xB1, 0xF, 0xF, true); a[i] = a[i] + __int_as_float(t); }
      if (OP == 7) a[i] = fminf(a[i], b);
    }
  }
  float s = 0; for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int OP> void run(const char *name, int waves_per_simd) {
  int blocks = 256 * 4 * waves_per_simd, iters = 20000;
  float *out; hipMalloc(&out, blocks * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 64>>>(out, 100);
  hipEventRecord(e0); k<OP><<<blocks, 64>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts = (double)blocks * iters * 8;  // wave-instructions of the op itself
  double per_simd_per_s = insts / (ms * 1e-3) / 1024.0;
  printf("%-14s waves/SIMD %d : %.2f G wave-inst/s/SIMD -> %.2f cycles/inst at 2.4 GHz (%.3f ms)\n", name, waves_per_simd,
         per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, ms);
  hipFree(out);
}
int main() {
  for (int w : {1, 4, 8}) {
    run<0>("v_mul_f32", w); run<1>("v_add_f32", w); run<2>("v_fma_f32", w); run<3>("cmp+cndmask", w);
    run<4>("mul+v_exp", w); run<5>("readlane+add", w); run<6>("add_dpp", w); run<7>("v_min_f32", w);
  }
  return 0;
}
