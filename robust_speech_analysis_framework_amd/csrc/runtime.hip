// librsaf.so runtime: error string, ABI version, per-kernel event profiling.
#include <mutex>
#include <vector>

#include "rsaf_common.h"

namespace rsaf {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

struct ProfEntry {
    std::string name;
    int64_t launches = 0;
    double flops = 0, bytes = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
};

static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfEntry> g_prof;
// Events are pooled: created in rsaf_prof_begin (outside any timed region) or on first need, returned by rsaf_prof_end, so
// that a profiled launch costs two hipEventRecord calls and no hipEventCreate (~1 000 creations per step before).
static std::vector<hipEvent_t> g_event_pool;
constexpr size_t EVENT_POOL_PREFILL = 8192;

static bool take_event(hipEvent_t* e) {
    if (!g_event_pool.empty()) {
        *e = g_event_pool.back();
        g_event_pool.pop_back();
        return true;
    }
    return hipEventCreate(e) == hipSuccess;
}

ProfScope::ProfScope(const char* name, hipStream_t s, double flops, double bytes)
    : slot(-1), pair(-1), stream(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on) return;
    for (size_t i = 0; i < g_prof.size(); ++i)
        if (g_prof[i].name == name) slot = (int)i;
    if (slot < 0) {
        g_prof.emplace_back();
        g_prof.back().name = name;
        slot = (int)g_prof.size() - 1;
    }
    ProfEntry& e = g_prof[slot];
    e.launches += 1;
    e.flops += flops;
    e.bytes += bytes;
    hipEvent_t a, b;
    if (!take_event(&a)) { slot = -1; return; }
    if (!take_event(&b)) { g_event_pool.push_back(a); slot = -1; return; }
    e.events.emplace_back(a, b);
    pair = (int)e.events.size() - 1;          // this scope's own event pair: other threads may open the same family meanwhile
    (void)hipEventRecord(a, stream);
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (slot >= (int)g_prof.size() || pair < 0 || pair >= (int)g_prof[slot].events.size()) return;
    (void)hipEventRecord(g_prof[slot].events[pair].second, stream);
}

}  // namespace rsaf

using namespace rsaf;

extern "C" {

int rsaf_abi_version(void) { return RSAF_ABI_VERSION; }

const char* rsaf_last_error(void) { return g_last_error.c_str(); }

int rsaf_prof_begin(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_prof)
        for (auto& p : e.events) {
            g_event_pool.push_back(p.first);
            g_event_pool.push_back(p.second);
        }
    g_prof.clear();
    while (g_event_pool.size() < EVENT_POOL_PREFILL) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) break;
        g_event_pool.push_back(e);
    }
    g_prof_on = true;
    return RSAF_OK;
}

int rsaf_prof_end(rsaf_prof_record* records_host, int cap, int* n_records_host) {
    RSAF_CHECK_ARG(n_records_host != nullptr, "n_records_host is NULL");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = false;
    int n = 0;
    for (auto& e : g_prof) {
        double ms = 0;
        for (auto& p : e.events) {
            RSAF_CHECK_HIP(hipEventSynchronize(p.second));
            float t = 0;
            RSAF_CHECK_HIP(hipEventElapsedTime(&t, p.first, p.second));
            ms += t;
            g_event_pool.push_back(p.first);
            g_event_pool.push_back(p.second);
        }
        e.events.clear();
        if (records_host && n < cap) {
            rsaf_prof_record& r = records_host[n];
            std::memset(&r, 0, sizeof(r));
            std::strncpy(r.name, e.name.c_str(), sizeof(r.name) - 1);
            r.launches = e.launches;
            r.ms = ms;
            r.flops = e.flops;
            r.bytes = e.bytes;
        }
        ++n;
    }
    g_prof.clear();
    *n_records_host = n;
    return RSAF_OK;
}

}  // extern "C"
