// Error reporting for the C ABI: thread-local message buffer, no exceptions, no allocation.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <utility>
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

#ifdef PTT_DEV
static DevSwitches read_switches() {
    DevSwitches d;
    if (const char* e = getenv("PTT_LINEAR_TILE")) { d.linear_rt = (e[0] == '2') ? 2 : 1; d.linear_ct = (e[0] && e[1] == '2') ? 2 : 1; }
    if (getenv("PTT_SA_GATHER1")) d.sa_gather1 = 1;
    if (const char* e = getenv("PTT_SA_STAGGER")) d.sa_stagger = atoi(e);
    if (const char* e = getenv("PTT_SA_WAVE")) d.sa_wave = atoi(e) != 0;
    if (const char* e = getenv("PTT_SA_RT")) d.sa_rt = (atoi(e) == 1) ? 1 : 2;
    if (const char* e = getenv("PTT_SA_STREAM")) d.sa_stream = atoi(e) != 0;
    if (const char* e = getenv("PTT_SA_LDS")) d.sa_lds = atoi(e) != 0;
    if (const char* e = getenv("PTT_SA_CHUNK")) d.sa_chunk = atoi(e);
    if (const char* e = getenv("PTT_PAIR_STAGGER")) d.pair_stagger = atoi(e);
    if (const char* e = getenv("PTT_PAIR_LDS_PAD")) d.pair_lds_pad = atoi(e);
    if (const char* e = getenv("PTT_LINEAR_SMALL")) d.linear_small = atoi(e);
    if (const char* e = getenv("PTT_BALL_CPW")) d.ball_cpw = atoi(e);
    if (const char* e = getenv("PTT_FPS_PLAIN")) d.fps_plain = atoi(e);
    if (const char* e = getenv("PTT_SA_LDS_CHUNK")) d.sa_lds_chunk = atoi(e);
    if (const char* e = getenv("PTT_FPS_T")) d.fps_t = atoi(e);
    if (getenv("PTT_GROUP_GRAD_GLOBAL")) d.group_grad_global = 1;
    if (const char* e = getenv("PTT_DEBUG_STAMPS")) d.stamps = reinterpret_cast<long long*>(strtoull(e, nullptr, 16));
    return d;
}
const DevSwitches& dev_switches() {
    static thread_local DevSwitches d;                  // developer build: re-read per call so that the sweep scripts
    d = read_switches();                                // (scripts/kernel_bench.py, *_phases.py) can flip a switch
    return d;                                           // between launches
}
#else
const DevSwitches& dev_switches() {
    static const DevSwitches d;                         // the measured-best defaults; nothing is read from the environment
    return d;
}
#endif

int set_lds_limit(const void* fn, int bytes) {
    if (bytes <= 48 * 1024) return PTT_OK;
    // (kernel, device) -> the largest limit set so far. A map, not a fixed table: a process that drives eight devices, or one
    // that runs every workload of bench.py, registers a few hundred pairs — and a pair that does not fit would fall back to
    // hipFuncSetAttribute on EVERY launch, which inside a stream capture is an error, not a slowdown.
    static std::map<std::pair<const void*, int>, int> table;
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return check_launch("hipGetDevice");
    std::lock_guard<std::mutex> g(mu);
    const auto key = std::make_pair(fn, dev);
    const auto it = table.find(key);
    if (it != table.end() && it->second >= bytes) return PTT_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
        return check_launch("hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    table[key] = bytes;
    return PTT_OK;
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
