// Atomic-throughput microbenchmark for the binning passes: 1 M returning atomicAdds onto 5120 tile counters,
// (a) device scope on one shared array, (b) workgroup scope (executed in the issuing XCD's L2) on a per-XCD
// private copy selected by the hardware XCC_ID.  Checks that (b) loses no updates.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__device__ __forceinline__ uint32_t xcc_id() {
  // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
  return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xF;
}
__device__ __forceinline__ uint32_t hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *cnt, uint32_t *out, uint32_t *xcd_seen, int n, int ntiles, int hot) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t h = hash(i);
  uint32_t t = (h % 100u) < (uint32_t)hot ? (h >> 8) % 16u : (h >> 8) % (uint32_t)ntiles;  // `hot` % of traffic on 16 tiles
  uint32_t r;
  if (MODE == 0) r = atomicAdd(&cnt[t], 1u);
  if (MODE == 1) {
    uint32_t x = xcc_id();
    if (threadIdx.x == 0) xcd_seen[blockIdx.x] = x;
    r = __hip_atomic_fetch_add(&cnt[x * ntiles + t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (MODE == 2) {  // device scope but per-XCD copies (separates contention from scope effects)
    uint32_t x = xcc_id();
    r = atomicAdd(&cnt[x * ntiles + t], 1u);
  }
  if (MODE == 3) {  // non-returning device scope
    atomicAdd(&cnt[t], 1u); r = 0;
  }
  out[i] = r;
}
template <int MODE> void run(const char *name, int hot) {
  const int n = 1 << 20, ntiles = 5120;
  uint32_t *cnt, *out, *seen; hipMalloc(&cnt, 16 * ntiles * 4); hipMalloc(&out, n * 4); hipMalloc(&seen, (n / 256) * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 5; rep++) {
    hipMemset(cnt, 0, 16 * ntiles * 4);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<n / 256, 256>>>(cnt, out, seen, n, ntiles, hot); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  std::vector<uint32_t> h(16 * ntiles); hipMemcpy(h.data(), cnt, 16 * ntiles * 4, hipMemcpyDeviceToHost);
  uint64_t tot = 0; for (auto v : h) tot += v;
  std::vector<uint32_t> hs(n / 256); hipMemcpy(hs.data(), seen, (n / 256) * 4, hipMemcpyDeviceToHost);
  int hist[16] = {0}; if (MODE == 1) for (auto v : hs) hist[v & 15]++;
  printf("%-28s hot %2d%%: %7.1f us  (%.1f G atomics/s)  total %llu / %d %s", name, hot, best * 1e3, n / (best * 1e-3) / 1e9,
         (unsigned long long)tot, n, tot == (uint64_t)n ? "OK" : "LOST UPDATES");
  if (MODE == 1) { printf("  xcc hist:"); for (int i = 0; i < 16; i++) if (hist[i]) printf(" %d:%d", i, hist[i]); }
  printf("\n");
  hipFree(cnt); hipFree(out); hipFree(seen);
}
int main() {
  for (int hot : {0, 10, 30}) {
    run<0>("device scope, shared", hot);
    run<3>("device scope, no return", hot);
    run<2>("device scope, per-XCD copy", hot);
    run<1>("workgroup scope, per-XCD copy", hot);
  }
  return 0;
}
