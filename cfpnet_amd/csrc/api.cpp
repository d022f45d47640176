// Error plumbing and version of the C ABI (no exception crosses the boundary).
#include <hip/hip_runtime.h>
#include <string>

#include "../../include/cfpnet_hip.h"

static thread_local std::string g_last_error;

void cfp_set_error(const std::string& msg) { g_last_error = msg; }

int cfp_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return CFP_EHIP;
  }
  return CFP_OK;
}

extern "C" int cfp_version(void) { return 100; }
extern "C" const char* cfp_last_error(void) { return g_last_error.c_str(); }
