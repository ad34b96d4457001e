// be_api.hip — library-level entry points of the C ABI (version, error string, device count).
#include "be_common.h"
#include "../../include/brainevent_amd.h"

static thread_local std::string g_last_error;

void be_set_error(const std::string& msg) { g_last_error = msg; }

extern "C" {

int be_version(void) { return 100; }  // 0.1.0

const char* be_last_error(void) { return g_last_error.c_str(); }

const char* be_build_arch(void) { return "gfx950"; }

int be_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    be_set_error(std::string("be_device_count: ") + hipGetErrorString(e));
    return BE_ERR_HIP;
  }
  return n;
}

}  // extern "C"
