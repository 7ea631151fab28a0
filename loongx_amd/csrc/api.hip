// api.hip -- error plumbing + version/device queries of the C ABI (include/lx.h).
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void lx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int lx_version(void) { return LX_VERSION; }
extern "C" const char* lx_last_error(void) { return g_err; }
extern "C" int lx_device_arch(char* name, size_t n) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { lx_set_error("no HIP device"); return LX_ERR_NO_DEVICE; }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) { lx_set_error("hipGetDeviceProperties failed"); return LX_ERR_NO_DEVICE; }
  if (name && n) { strncpy(name, p.gcnArchName, n - 1); name[n - 1] = 0; }
  return LX_OK;
}
