// Error plumbing, version and device query of libgaext (include/gaext.h).
#include <stdarg.h>
#include "common.h"

namespace {
thread_local char g_err[512] = "";
}

void ga_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ga_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ga_set_error("%s: %s", what, hipGetErrorString(e));
        return GA_ERR_HIP;
    }
    return GA_OK;
}

extern "C" int ga_version(void) { return 101; }

// ------------------------------------------------------------------------------------------------------------------
// Tuning knobs.  Every kernel-selection switch of the library lives in ONE table: a knob takes its value from the
// environment variable GAEXT_<NAME> exactly once -- when the knob is first looked up -- or from ga_set_knob(); the
// dispatch code holds a pointer to the slot, so a launch never calls getenv.  ga_config_string() lists every knob that
// is not at its default, so that a bench / parity record shows what actually ran.
// ------------------------------------------------------------------------------------------------------------------
#include <atomic>
#include <mutex>
namespace {
struct Knob {
    char name[32];
    int dflt;
    std::atomic<int> value;
    int source;   // 0 default, 1 environment, 2 ga_set_knob
};
constexpr int kMaxKnobs = 64;
Knob g_knobs[kMaxKnobs];
int g_nknobs = 0;
std::mutex g_knob_mu;

Knob* knob_find_locked(const char* name) {
    for (int i = 0; i < g_nknobs; ++i)
        if (!strcmp(g_knobs[i].name, name)) return &g_knobs[i];
    return nullptr;
}
}  // namespace

const std::atomic<int>* ga_knob_slot(const char* name, int dflt) {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    Knob* k = knob_find_locked(name);
    if (!k) {
        if (g_nknobs >= kMaxKnobs) {   // cannot happen with the fixed set of call sites; fall back to slot 0's storage
            static std::atomic<int> overflow;
            overflow = dflt;
            return &overflow;
        }
        k = &g_knobs[g_nknobs++];
        strncpy(k->name, name, sizeof(k->name) - 1);
        k->dflt = dflt;
        k->value = dflt;
        k->source = 0;
        char env[64];
        snprintf(env, sizeof(env), "GAEXT_%s", name);
        if (const char* e = getenv(env)) {
            k->value = atoi(e);
            k->source = 1;
        }
    } else if (k->source == 3) {       // registered by ga_set_knob before the first dispatch: adopt the real default
        k->dflt = dflt;
        k->source = 2;
    }
    return &k->value;
}

extern "C" int ga_set_knob(const char* name, int value) {
    if (!name || !*name || strlen(name) >= sizeof(g_knobs[0].name)) {
        ga_set_error("ga_set_knob: bad name");
        return GA_ERR_BAD_ARG;
    }
    std::lock_guard<std::mutex> lk(g_knob_mu);
    Knob* k = knob_find_locked(name);
    if (!k) {
        if (g_nknobs >= kMaxKnobs) {
            ga_set_error("ga_set_knob: table full");
            return GA_ERR_BAD_ARG;
        }
        k = &g_knobs[g_nknobs++];
        strncpy(k->name, name, sizeof(k->name) - 1);
        k->dflt = value;
        k->source = 3;                 // default not known yet (no dispatch has asked for this knob)
        k->value = value;
        return GA_OK;
    }
    k->value = value;
    if (k->source != 3) k->source = 2;
    return GA_OK;
}

extern "C" int ga_unset_knob(const char* name) {
    if (!name) return GA_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_knob_mu);
    Knob* k = knob_find_locked(name);
    if (k && k->source != 3) {
        k->value = k->dflt;
        k->source = 0;
    } else if (k) {                    // never looked up by a dispatch: forget the override
        *k->name = 0;
    }
    return GA_OK;
}

extern "C" int ga_config_string(char* buf, size_t n) {
    char tmp[1024];
    int len = snprintf(tmp, sizeof(tmp), "libgaext %d gfx950%s", ga_version(),
#ifdef GAEXT_DEBUG
                       " DEBUG-BUILD"
#else
                       ""
#endif
    );
    {
        std::lock_guard<std::mutex> lk(g_knob_mu);
        for (int i = 0; i < g_nknobs && len < (int)sizeof(tmp) - 64; ++i) {
            const Knob& k = g_knobs[i];
            if (!*k.name) continue;
            const int v = k.value.load();
            if (k.source == 0 || (k.source != 3 && v == k.dflt)) continue;
            len += snprintf(tmp + len, sizeof(tmp) - len, " %s=%d(%s)", k.name, v, k.source == 1 ? "env" : "api");
        }
    }
    if (buf && n) {
        strncpy(buf, tmp, n - 1);
        buf[n - 1] = 0;
    }
    return len;
}

extern "C" int ga_last_error(char* buf, size_t n) {
    if (buf && n) {
        strncpy(buf, g_err, n - 1);
        buf[n - 1] = 0;
    }
    return (int)strlen(g_err);
}

extern "C" int ga_device_info(int* num_cu, int* lds_bytes, int* wave_size) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        ga_set_error("ga_device_info: no HIP device");
        return GA_ERR_HIP;
    }
    if (num_cu) *num_cu = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
    if (wave_size) *wave_size = p.warpSize;
    return GA_OK;
}

extern "C" int ga_memset(void* p, int value, size_t bytes, ga_stream_t stream) {
    if (!p || !bytes) {
        ga_set_error("ga_memset: null/empty");
        return GA_ERR_BAD_ARG;
    }
    const hipError_t e = hipMemsetAsync(p, value, bytes, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        ga_set_error("ga_memset: %s", hipGetErrorString(e));
        return GA_ERR_HIP;
    }
    return GA_OK;
}

