// Error reporting and ABI version of libmadm_hip.
#include "common.hpp"
#include <cstdarg>
#include <cstdio>

namespace {
thread_local char g_err[512] = "";
}

void madm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int madm_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        madm_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return MADM_ERR_LAUNCH;
    }
    return MADM_OK;
}

extern "C" {
int madm_abi_version(void) { return MADM_ABI_VERSION; }
const char* madm_last_error(void) { return g_err; }
}
