// dispatch_map.hip -- where do the workgroups of a one-wave-per-workgroup launch land?  (MI355X, gfx950)
// The blend kernels launch one 64-lane workgroup per tile in longest-list-first order; whether a SIMD's tiles balance depends
// on the (undocumented, speed-only) block -> SIMD placement.  Every block records HW_ID / XCC_ID and its start time, then
// keeps its slot busy for ~60 us so that the whole launch is resident at once.  VGPRS = registers the kernel pretends to need
// (96 -> 5 waves / SIMD like blend_bwd, 80 -> 6 like blend_fwd).
//   hipcc --offload-arch=gfx950 -O3 -o dispatch_map.bin dispatch_map.hip && ./dispatch_map.bin [nblocks]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>

template <int VG>
__global__ __launch_bounds__(64) void probe(uint32_t *hw, uint32_t *xcc, unsigned long long *t0, int spin_us) {
  if (VG >= 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  else asm volatile("v_mov_b32 v79, 0" ::: "v79");
  const unsigned long long start = wall_clock64();
  if (threadIdx.x == 0) {
    hw[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    t0[blockIdx.x] = start;
  }
  while (wall_clock64() - start < (unsigned long long)spin_us * 100ull) __builtin_amdgcn_s_sleep(8);
}

template <int VG>
static void run(int n) {
  uint32_t *hw, *xcc;
  unsigned long long *t0;
  hipMalloc(&hw, 4 * n); hipMalloc(&xcc, 4 * n); hipMalloc(&t0, 8 * n);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(probe<VG>, dim3(n), dim3(64), 0, 0, hw, xcc, t0, 60);
    hipDeviceSynchronize();
  }
  std::vector<uint32_t> h(n), x(n);
  std::vector<unsigned long long> t(n);
  hipMemcpy(h.data(), hw, 4 * n, hipMemcpyDeviceToHost);
  hipMemcpy(x.data(), xcc, 4 * n, hipMemcpyDeviceToHost);
  hipMemcpy(t.data(), t0, 8 * n, hipMemcpyDeviceToHost);
  auto key = [&](int b) {  // one id per SIMD: xcc | se | sh | cu | simd
    const uint32_t v = h[b];
    return ((x[b] & 15u) << 16) | (((v >> 13) & 7u) << 12) | (((v >> 12) & 1u) << 11) | (((v >> 8) & 15u) << 4) | ((v >> 4) & 3u);
  };
  unsigned long long tmin = ~0ull;
  for (int b = 0; b < n; b++) tmin = t[b] < tmin ? t[b] : tmin;
  printf("== %d blocks, kernel sized for %d waves per SIMD\n", n, VG >= 96 ? 5 : 6);
  printf("block: xcc se sh cu simd wave  start(us)\n");
  for (int b = 0; b < n; b += (b < 48 ? 1 : (b < 1200 ? 97 : 509)))
    printf("%5d: %2u %2u %u %2u %u %2u  %7.2f\n", b, x[b] & 15u, (h[b] >> 13) & 7u, (h[b] >> 12) & 1u, (h[b] >> 8) & 15u,
           (h[b] >> 4) & 3u, h[b] & 15u, (double)(t[b] - tmin) / 100.0);
  std::map<uint32_t, std::vector<int>> per;
  for (int b = 0; b < n; b++) per[key(b)].push_back(b);
  size_t lo = n, hi = 0;
  for (auto &kv : per) { lo = kv.second.size() < lo ? kv.second.size() : lo; hi = kv.second.size() > hi ? kv.second.size() : hi; }
  printf("distinct SIMDs used: %zu; blocks per SIMD: min %zu max %zu\n", per.size(), lo, hi);
  // do the blocks of a SIMD follow b, b + P, b + 2P ... ?  histogram of the differences between consecutive blocks of one SIMD
  std::map<int, int> diffs;
  for (auto &kv : per)
    for (size_t i = 1; i < kv.second.size(); i++) diffs[kv.second[i] - kv.second[i - 1]]++;
  printf("differences between consecutive blocks of the same SIMD (value: count), most frequent first:\n");
  std::vector<std::pair<int, int>> d;
  for (auto &kv : diffs) d.push_back({kv.second, kv.first});
  std::sort(d.rbegin(), d.rend());
  for (size_t i = 0; i < d.size() && i < 12; i++) printf("   %6d: %d\n", d[i].second, d[i].first);
  // the imbalance a strictly descending work order would suffer: work(b) = n - b
  double wmin = 1e30, wmax = 0, wsum = 0;
  for (auto &kv : per) {
    double w = 0;
    for (int b : kv.second) w += (double)(n - b);
    wmin = w < wmin ? w : wmin; wmax = w > wmax ? w : wmax; wsum += w;
  }
  printf("if block b carried work n - b: per-SIMD work min %.0f mean %.0f max %.0f (max / mean %.3f)\n", wmin, wsum / per.size(),
         wmax, wmax / (wsum / per.size()));
  // the same with the order folded back and forth with period 1024 (snake)
  wmin = 1e30; wmax = 0; wsum = 0;
  for (auto &kv : per) {
    double w = 0;
    for (int b : kv.second) {
      const int round = b / 1024, idx = b % 1024;
      const int rank = round * 1024 + ((round & 1) ? 1023 - idx : idx);  // the rank dispatched at position b
      w += (double)(n - rank);
    }
    wmin = w < wmin ? w : wmin; wmax = w > wmax ? w : wmax; wsum += w;
  }
  printf("with the order folded (period 1024):      per-SIMD work min %.0f mean %.0f max %.0f (max / mean %.3f)\n", wmin,
         wsum / per.size(), wmax, wmax / (wsum / per.size()));
  hipFree(hw); hipFree(xcc); hipFree(t0);
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 5120;
  run<96>(n);
  run<80>(n);
  return 0;
}
