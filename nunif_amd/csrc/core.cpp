// Error reporting and the optional HIP-event kernel-class profiler of libnunif_hip.so.
#include <cstdlib>

#include "common.h"

#include <cstring>
#include <mutex>
#include <vector>

namespace nunif {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

// ---- profiler ------------------------------------------------------------------------------------------------
// bench.py needs the dominant kernel's average launch duration measured with HIP events on the stream the
// kernel is launched on.  When enabled, every ProfScope records a start/stop event pair around its launch;
// nunif_hip_profile_read() synchronises and sums them per name.  Disabled (default) it costs one branch.
struct ProfEvent { int slot; hipEvent_t a, b; };
struct ProfSlot { std::string name; double ms = 0; int64_t launches = 0; double flops = 0, bytes = 0; };
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::vector<ProfSlot> g_slots;
static std::vector<ProfEvent> g_events;
static std::vector<hipEvent_t> g_free_events;

bool profiling_enabled() { return g_prof_on; }
bool profile_tags_enabled() {
    static const bool on = getenv("NUNIF_PROF_TAGS") != nullptr;
    return on;
}

static hipEvent_t get_event() {
    if (!g_free_events.empty()) {
        hipEvent_t e = g_free_events.back();
        g_free_events.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

ProfScope::ProfScope(const char *name, hipStream_t s, double flops, double bytes) : slot(-1), stream(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (size_t i = 0; i < g_slots.size(); ++i)
        if (g_slots[i].name == name) slot = (int)i;
    if (slot < 0) {
        g_slots.push_back(ProfSlot());
        g_slots.back().name = name;
        slot = (int)g_slots.size() - 1;
    }
    g_slots[slot].flops += flops;
    g_slots[slot].bytes += bytes;
    ProfEvent ev;
    ev.slot = slot;
    ev.a = get_event();
    ev.b = get_event();
    (void)hipEventRecord(ev.a, s);
    g_events.push_back(ev);
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    // the matching start event is the last one pushed for this scope (scopes nest strictly per thread)
    for (size_t i = g_events.size(); i-- > 0;) {
        if (g_events[i].slot == slot) {
            (void)hipEventRecord(g_events[i].b, stream);
            break;
        }
    }
}

}  // namespace nunif

using namespace nunif;

extern "C" int nunif_hip_abi_version(void) { return NUNIF_HIP_ABI_VERSION; }

extern "C" const char *nunif_hip_last_error(void) { return g_last_error.c_str(); }

extern "C" int nunif_hip_profile_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_profile_read(nunif_prof_record *out, int32_t cap, int32_t reset) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &ev : g_events) {
        (void)hipEventSynchronize(ev.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
            g_slots[ev.slot].ms += ms;
            g_slots[ev.slot].launches += 1;
        }
        g_free_events.push_back(ev.a);
        g_free_events.push_back(ev.b);
    }
    g_events.clear();
    int n = 0;
    for (auto &s : g_slots) {
        if (n >= cap) break;
        memset(&out[n], 0, sizeof(out[n]));
        strncpy(out[n].name, s.name.c_str(), sizeof(out[n].name) - 1);
        out[n].total_ms = s.ms;
        out[n].launches = s.launches;
        out[n].flops = s.flops;
        out[n].bytes = s.bytes;
        ++n;
    }
    if (reset) g_slots.clear();
    return n;
}
