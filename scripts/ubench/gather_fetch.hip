// gather_fetch.hip -- calibration of rocprofv3's FETCH_SIZE for the blend kernels' access pattern: every lane reads ONE
// whole 64-byte record (4 x 16 B) at a random index, each record exactly once.  Known bytes = N * 64 (+ 4 N of coalesced
// index reads).  MI355X_MICROARCH.md (HBM): FETCH_SIZE tallies wide COALESCED streaming reads at half their bytes
// (128-byte requests counted as 64); "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern".  Kernel `stream_read` is the coalesced control (N * 64 bytes, 16 B per lane, consecutive).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/gather_fetch.bin scripts/ubench/gather_fetch.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/gf -o gf --output-format csv -- scripts/ubench/gather_fetch.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

__global__ __launch_bounds__(64) void gather64(const float4 *rec, const uint32_t *idx, float *out, int n) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const float4 *p = rec + (size_t)idx[i] * 4;
  float4 a = p[0], b = p[1], c = p[2], d = p[3];
  out[i] = a.x + b.y + c.z + d.w;
}
__global__ __launch_bounds__(256) void stream_read(const float4 *rec, float *out, size_t n4) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  for (; i < n4; i += (size_t)gridDim.x * 256) s += rec[i].x;
  if (s == 12345.f) out[0] = s;
}
int main() {
  const int N = 4 << 20;  // 4 Mi records = 256 MiB (the Infinity Cache holds 256 MiB: indices are unique, nothing is re-read)
  float4 *rec; uint32_t *idx; float *out;
  hipMalloc(&rec, (size_t)N * 64); hipMalloc(&idx, (size_t)N * 4); hipMalloc(&out, (size_t)N * 4);
  hipMemset(rec, 0, (size_t)N * 64);
  std::vector<uint32_t> h(N); std::iota(h.begin(), h.end(), 0u);
  std::mt19937 g(1); std::shuffle(h.begin(), h.end(), g);
  hipMemcpy(idx, h.data(), (size_t)N * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(gather64, dim3(N / 64), dim3(64), 0, 0, rec, idx, out, N);
    hipLaunchKernelGGL(stream_read, dim3(4096), dim3(256), 0, 0, rec, out, (size_t)N * 4);
  }
  hipDeviceSynchronize();
  printf("known bytes per launch: gather64 %.1f MB (+ %.1f MB of indices), stream_read %.1f MB\n", N * 64 / 1e6, N * 4 / 1e6, N * 64 / 1e6);
  return 0;
}
