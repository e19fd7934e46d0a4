// common.hip -- version string and error plumbing of libfsgs_hip.so
#include <string.h>

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
}  // namespace fsgs

extern "C" {
const char *fsgs_version(void) { return "fsgs-hip 0.1 (gfx950)"; }
const char *fsgs_last_error(void) { return fsgs::last_error_buffer(); }
}
