// Error channel + version of the C ABI (include/csts_hip.h).
#include "common.h"

static thread_local std::string g_last_error;

void csts_set_error(const std::string& s) { g_last_error = s; }

extern "C" const char* csts_last_error(void) { return g_last_error.c_str(); }
extern "C" int csts_abi_version(void) { return CSTS_ABI_VERSION; }
// 0: the 16-bit type of this library is bfloat16 (libcsts_hip.so); 1: IEEE half (libcsts_hip_f16.so)
extern "C" int csts_half_kind(void) {
#ifdef CSTS_HALF_F16
  return 1;
#else
  return 0;
#endif
}
