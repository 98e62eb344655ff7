// Error channel and ABI version of libwcmc_hip.so.
#include "common.h"

namespace wcmc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return WCMC_OK;
  set_error("%s: launch failed: %s", what, hipGetErrorString(e));
  return WCMC_ERR_LAUNCH;
}

}  // namespace wcmc

extern "C" int wcmc_abi_version(void) { return WCMC_ABI_VERSION; }
extern "C" const char* wcmc_last_error(void) { return wcmc::g_err; }
