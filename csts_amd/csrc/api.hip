// Error channel + version of the C ABI (include/csts_hip.h).
#include "common.h"

static thread_local std::string g_last_error;

void csts_set_error(const std::string& s) { g_last_error = s; }

extern "C" const char* csts_last_error(void) { return g_last_error.c_str(); }
extern "C" int csts_abi_version(void) { return CSTS_ABI_VERSION; }
