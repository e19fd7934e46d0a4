// common.hip -- version string and error plumbing of libfsgs_hip.so
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "fsgs_device.h"
#include "selftest_reductions.h"  // reductions only the selftest below calls
#include "fsgs_host.h"

namespace fsgs {
static thread_local char g_err[512] = "";
char *last_error_buffer() { return g_err; }
int fsgs_fail(const char *what) {
  snprintf(g_err, sizeof(g_err), "%s", what);
  return FSGS_ERR_HIP;
}
int fsgs_fail_hip(hipError_t e, const char *expr, const char *file, int line) {
  snprintf(g_err, sizeof(g_err), "%s -> %s (%s:%d)", expr, hipGetErrorString(e), file, line);
  (void)hipGetLastError();
  return FSGS_ERR_HIP;
}

// ---- completion event of the next forward (fsgs_forward_done_event) ----
static thread_local hipEvent_t g_forward_done = nullptr;
hipEvent_t take_forward_done_event() {
  hipEvent_t e = g_forward_done;
  g_forward_done = nullptr;
  return e;
}

static thread_local hipEvent_t g_pose_step_done = nullptr;
hipEvent_t take_pose_step_done_event() {
  hipEvent_t e = g_pose_step_done;
  g_pose_step_done = nullptr;
  return e;
}

// ---- mailbox ----------------------------------------------------------------------------------
static std::once_flag g_mail_once;
static uint32_t *g_mail_base = nullptr;
static std::atomic<uint32_t> g_mail_next{0};
constexpr int kMailSlots = 256, kMailStride = 16;  // one 64-byte line per slot
volatile uint32_t *mailbox_acquire() {
  std::call_once(g_mail_once, [] {
    void *p = nullptr;
    if (hipHostMalloc(&p, sizeof(uint32_t) * kMailSlots * kMailStride,
                      hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess)
      g_mail_base = (uint32_t *)p;
    else
      (void)hipGetLastError();
  });
  if (!g_mail_base) return nullptr;
  volatile uint32_t *slot = g_mail_base + (size_t)(g_mail_next.fetch_add(1) % kMailSlots) * kMailStride;
  *slot = kMailboxEmpty;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  return slot;
}
bool mailbox_wait(volatile uint32_t *slot, hipStream_t stream, uint32_t *value) {
  for (unsigned spins = 0;; spins++) {
    const uint32_t v = *slot;
    if (v != kMailboxEmpty) {
      std::atomic_thread_fence(std::memory_order_acquire);
      *value = v;
      return true;
    }
    if ((spins & 0x3FFF) == 0x3FFF) {  // every few tens of microseconds: is the stream still working on it?
      const hipError_t q = hipStreamQuery(stream);
      if (q != hipErrorNotReady) {
        (void)hipGetLastError();
        const uint32_t v2 = *slot;  // finished (or failed) -- the store, if any, has landed
        if (v2 != kMailboxEmpty) {
          *value = v2;
          return true;
        }
        return false;
      }
    }
    __builtin_ia32_pause();
  }
}

// ---- profiler ---------------------------------------------------------------------------------
static const char *kProfNames[PROF_COUNT] = {
    "preprocess_fwd", "sort_depth", "scan_tiles", "emit_pairs", "sort_tile", "tile_ranges", "blend_fwd",
    "blend_bwd", "preprocess_bwd", "knn", "loss_rgb_fwd", "loss_rgb_bwd", "pearson", "adam", "render_pre_fwd",
    "render_pre_bwd", "flow_loss"};
struct ProfRec {
  int id;
  hipEvent_t a, b;
  bool closed;
};
static std::mutex g_prof_mu;
static uint64_t g_prof_mask = 0;
static int g_prof_stride = 1;            // time every g_prof_stride-th scope of an enabled id
static long long g_prof_seen[PROF_COUNT];
static std::vector<ProfRec> g_prof_recs;
static double g_prof_ms[PROF_COUNT];
static long long g_prof_n[PROF_COUNT];

bool prof_enabled(int id) {
  if (!((g_prof_mask >> id) & 1ull)) return false;
  if (g_prof_stride <= 1) return true;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  return (g_prof_seen[id]++ % g_prof_stride) == 0;
}
void prof_record(int id, hipStream_t s, bool begin) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (begin) {
    ProfRec r;
    r.id = id;
    r.closed = false;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    g_prof_recs.push_back(r);
  } else {
    for (size_t i = g_prof_recs.size(); i-- > 0;)
      if (g_prof_recs[i].id == id && !g_prof_recs[i].closed) {
        (void)hipEventRecord(g_prof_recs[i].b, s);
        g_prof_recs[i].closed = true;
        break;
      }
  }
}
static void prof_drain() {
  for (auto &r : g_prof_recs) {
    if (r.closed && hipEventSynchronize(r.b) == hipSuccess) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
        g_prof_ms[r.id] += ms;
        g_prof_n[r.id] += 1;
      }
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof_recs.clear();
}
}  // namespace fsgs

namespace {
__global__ void selftest_transpose_reduce_kernel(const float *in, float *out) {
  const int lane = threadIdx.x & 63;
  float v[64];
#pragma unroll
  for (int i = 0; i < 64; i++) v[i] = in[lane * 64 + i];
  out[lane] = fsgs::wave_transpose_reduce64(v, lane);
}
template <int N>
__global__ void selftest_transpose_reduce_n_kernel(const float *in, float *out) {
  const int lane = threadIdx.x & 63;
  float v[N];
#pragma unroll
  for (int i = 0; i < N; i++) v[i] = in[lane * 64 + i];
  if (N == 32) out[lane] = fsgs::wave_transpose_reduce32(reinterpret_cast<const float(&)[32]>(v), lane);
  if (N == 16) out[lane] = fsgs::wave_transpose_reduce16(reinterpret_cast<const float(&)[16]>(v), lane);
}
// the reductions blend_bwd actually uses: slots 12..15 of either Gaussian (width 3212) / 5..7 (width 1605) are ZERO on
// entry and are not reduced; out[64 + lane] = the slot this lane ends up owning
__global__ void selftest_transpose_reduce_sparse_kernel(const float *in, float *out, int width) {
  const int lane = threadIdx.x & 63;
  if (width == 3212) {
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] = (i & 15) < 12 ? in[lane * 64 + i] : 0.f;
    out[lane] = fsgs::wave_transpose_reduce32_12of16_cheap_first(v, lane);
    out[64 + lane] = (float)fsgs::transpose12_slot(lane);
  } else {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = (i & 7) < 5 ? in[lane * 64 + i] : 0.f;
    out[lane] = fsgs::wave_transpose_reduce16_5of8_cheap_first(v, lane);
    out[64 + lane] = (float)fsgs::transpose5_slot(lane);
  }
}
// the blend kernels' one definition of "does Gaussian g reach pixel p" (fsgs_device.h: splat_coef + splat_alpha, the
// pre-scaled exponent and the single v_exp_f32), isolated: in[i] = {gx, gy, A, B, C, o, px, py}
__global__ void selftest_splat_alpha_kernel(int n, const float *__restrict__ in, float *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *r = in + (size_t)i * 8;
  const fsgs::SplatCoef k = fsgs::splat_coef(r[2], r[3], r[4]);
  fsgs::SplatEval e;
  e.a = 0.f;
  const bool ok = fsgs::splat_alpha(__fsub_rn(r[0], r[6]), __fsub_rn(r[1], r[7]), k.a, k.b, k.c, r[5], e);
  out[2 * i] = e.a;
  out[2 * i + 1] = ok ? 1.f : 0.f;
}
}  // namespace

extern "C" {
int fsgs_event_create(fsgs_event_t *event) {
  if (!event) return FSGS_ERR_INVALID;
  hipEvent_t e = nullptr;
  FSGS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *event = (fsgs_event_t)e;
  return FSGS_OK;
}
int fsgs_event_destroy(fsgs_event_t event) {
  if (!event) return FSGS_OK;
  FSGS_HIP(hipEventDestroy((hipEvent_t)event));
  return FSGS_OK;
}
int fsgs_stream_wait_event(fsgs_stream_t stream, fsgs_event_t event) {
  if (!event) return FSGS_ERR_INVALID;
  FSGS_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return FSGS_OK;
}
int fsgs_event_record(fsgs_event_t event, fsgs_stream_t stream) {
  if (!event) return FSGS_ERR_INVALID;
  FSGS_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return FSGS_OK;
}
int fsgs_forward_done_event(fsgs_event_t event) {
  fsgs::g_forward_done = (hipEvent_t)event;
  return FSGS_OK;
}
int fsgs_pose_step_done_event(fsgs_event_t event) {
  fsgs::g_pose_step_done = (hipEvent_t)event;
  return FSGS_OK;
}
int fsgs_selftest_splat_alpha(int n, const float *in8, float *out2, fsgs_stream_t stream) {
  if (n < 0 || (n > 0 && (!in8 || !out2))) return FSGS_ERR_INVALID;
  if (n == 0) return FSGS_OK;
  hipLaunchKernelGGL(selftest_splat_alpha_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, in8, out2);
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}
int fsgs_selftest_transpose_reduce_n(const float *in64x64, float *out64, int width, fsgs_stream_t stream) {
  if (!in64x64 || !out64) return FSGS_ERR_INVALID;
  if (width == 64) return fsgs_selftest_transpose_reduce(in64x64, out64, stream);
  if (width == 32)
    hipLaunchKernelGGL(selftest_transpose_reduce_n_kernel<32>, dim3(1), dim3(64), 0, (hipStream_t)stream, in64x64, out64);
  else if (width == 16)
    hipLaunchKernelGGL(selftest_transpose_reduce_n_kernel<16>, dim3(1), dim3(64), 0, (hipStream_t)stream, in64x64, out64);
  else if (width == 3212 || width == 1605)  // the sparse variants of the backward blend: out64 must hold 128 floats
    hipLaunchKernelGGL(selftest_transpose_reduce_sparse_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in64x64, out64, width);
  else
    return FSGS_ERR_INVALID;
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}
int fsgs_selftest_transpose_reduce(const float *in64x64, float *out64, fsgs_stream_t stream) {
  if (!in64x64 || !out64) return FSGS_ERR_INVALID;
  hipLaunchKernelGGL(selftest_transpose_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in64x64, out64);
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}
int fsgs_profile_enable(uint64_t mask) {
  std::lock_guard<std::mutex> lk(fsgs::g_prof_mu);
  fsgs::prof_drain();
  for (int i = 0; i < fsgs::PROF_COUNT; i++) { fsgs::g_prof_ms[i] = 0; fsgs::g_prof_n[i] = 0; fsgs::g_prof_seen[i] = 0; }
  fsgs::g_prof_mask = mask;
  return FSGS_OK;
}
int fsgs_profile_stride(int stride) {
  std::lock_guard<std::mutex> lk(fsgs::g_prof_mu);
  fsgs::g_prof_stride = stride < 1 ? 1 : stride;
  return FSGS_OK;
}
int fsgs_profile_count(void) { return fsgs::PROF_COUNT; }
const char *fsgs_profile_name(int id) { return (id >= 0 && id < fsgs::PROF_COUNT) ? fsgs::kProfNames[id] : ""; }
int fsgs_profile_read(int id, double *total_ms, int64_t *launches) {
  if (id < 0 || id >= fsgs::PROF_COUNT || !total_ms || !launches) return FSGS_ERR_INVALID;
  std::lock_guard<std::mutex> lk(fsgs::g_prof_mu);
  fsgs::prof_drain();
  *total_ms = fsgs::g_prof_ms[id];
  *launches = fsgs::g_prof_n[id];
  return FSGS_OK;
}
const char *fsgs_version(void) { return "fsgs-hip 0.1 (gfx950)"; }
const char *fsgs_last_error(void) { return fsgs::last_error_buffer(); }
}
