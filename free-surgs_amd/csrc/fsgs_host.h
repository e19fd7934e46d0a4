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

}  // namespace fsgs

#define FSGS_HIP(expr)                                                         \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) return fsgs::fsgs_fail_hip(_e, #expr, __FILE__, __LINE__); \
  } while (0)
