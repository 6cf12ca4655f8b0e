// Error reporting for the C ABI: thread-local message buffer, no exceptions, no allocation.
#include <stdarg.h>
#include <string.h>
#include "common.h"

namespace ptt {

static thread_local char g_err[512] = "";

char* last_error_buf() { return g_err; }

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return PTT_OK;
    return fail(PTT_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
}

}  // namespace ptt

extern "C" int ptt_version(void) { return PTT_ABI_VERSION; }

extern "C" const char* ptt_last_error_string(void) { return ptt::last_error_buf(); }

extern "C" const char* ptt_error_name(int code) {
    switch (code) {
        case PTT_OK: return "PTT_OK";
        case PTT_EINVAL: return "PTT_EINVAL";
        case PTT_EUNSUPPORTED: return "PTT_EUNSUPPORTED";
        case PTT_ELAUNCH: return "PTT_ELAUNCH";
        case PTT_EWORKSPACE: return "PTT_EWORKSPACE";
        default: return "PTT_E?";
    }
}
