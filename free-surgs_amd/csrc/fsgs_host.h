// fsgs_host.h -- host-side helpers of libfsgs_hip.so (error plumbing, buffer carving).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdio.h>

#include "../../include/fsgs.h"

namespace fsgs {

// thread-local text of the last HIP failure, exposed through fsgs_last_error()
char *last_error_buffer();
int fsgs_fail(const char *what);
int fsgs_fail_hip(hipError_t e, const char *expr, const char *file, int line);

// carve 256-byte aligned sub-buffers out of one caller-owned allocation
struct Carver {
  size_t off = 0;
  size_t take(size_t bytes) {
    size_t at = off;
    off += (bytes + 255) & ~(size_t)255;
    return at;
  }
  size_t total() const { return off + 256; }
};

// ---- device -> host mailbox ---------------------------------------------------------------------
// The forward needs ONE number back from the device per call (R, the number of tile pairs).  A D2H copy plus
// hipStreamSynchronize leaves the GPU idle for ~40 us (interrupt wake-up + the launches that follow).  Instead
// the scan kernel stores R with system scope into a pinned, coherent host word; the host enqueues every
// R-independent launch first and then polls that word.  Slots are handed out round-robin so concurrent calls
// (trainer + viewer thread) never share one.  acquire() returns nullptr if pinned memory is unavailable, and
// the caller falls back to copy + synchronize.
constexpr uint32_t kMailboxEmpty = 0xFFFFFFFFu;
volatile uint32_t *mailbox_acquire();
// the event the next fused forward of this thread signals with its blend launch (fsgs_forward_done_event); cleared by the take
hipEvent_t take_forward_done_event();
hipEvent_t take_pose_step_done_event();  // the same for the next fsgs_pose_adam_step (fsgs_pose_step_done_event)
// spins until *slot != kMailboxEmpty; gives up (returns false) when the stream reports an error or is idle
// without the slot having been written
bool mailbox_wait(volatile uint32_t *slot, hipStream_t stream, uint32_t *value);

// ---- optional per-kernel timing with HIP events on the launching stream (bench.py roofline) ----
enum ProfId {
  PROF_PREPROCESS_FWD = 0,
  PROF_SORT_DEPTH,
  PROF_SCAN,
  PROF_EMIT,
  PROF_SORT_TILE,
  PROF_RANGES,
  PROF_BLEND_FWD,
  PROF_BLEND_BWD,
  PROF_PREPROCESS_BWD,
  PROF_KNN,
  PROF_LOSS_RGB_FWD,
  PROF_LOSS_RGB_BWD,
  PROF_PEARSON,
  PROF_ADAM,
  PROF_RENDER_PRE_FWD,
  PROF_RENDER_PRE_BWD,
  PROF_FLOW,
  PROF_COUNT
};
bool prof_enabled(int id);
void prof_record(int id, hipStream_t s, bool begin);
struct ProfScope {
  int id;
  hipStream_t s;
  bool on;
  ProfScope(int id_, hipStream_t s_) : id(id_), s(s_), on(prof_enabled(id_)) {
    if (on) prof_record(id, s, true);
  }
  ~ProfScope() {
    if (on) prof_record(id, s, false);
  }
};

}  // namespace fsgs

#define FSGS_HIP(expr)                                                         \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) return fsgs::fsgs_fail_hip(_e, #expr, __FILE__, __LINE__); \
  } while (0)
