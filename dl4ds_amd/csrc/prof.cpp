#include "prof.h"
#include <map>
#include <sstream>

static Profiler g_prof;
Profiler& prof() { return g_prof; }
static ProfScope* g_cur = nullptr;
ProfScope*& prof_current() { return g_cur; }

hipEvent_t Profiler::get_event() {
    hipEvent_t e;
    if (!pool.empty()) { e = pool.back(); pool.pop_back(); return e; }
    // hipEventDisableSystemFence: a default event performs a system-scope fence when it is recorded -- the L2 is written back and
    // invalidated around every profiled launch, so a bandwidth-bound kernel whose input the previous kernel left in the L2 / MALL
    // was timed cold and with its predecessor's write-back in its window (maxpool2_fwd in cfg5: 111 us per launch under the
    // profiler, 17 us in rocprofv3's kernel trace).  Agent-scope ordering is all the timing needs.
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) (void)hipEventCreate(&e);
    return e;
}

void Profiler::reset() {
    for (auto& e : entries) {
        pool.push_back(e.e0); pool.push_back(e.e1);
        for (hipEvent_t k : e.k) pool.push_back(k);
    }
    entries.clear();
}

std::string Profiler::report_json(hipStream_t s) {
    (void)hipStreamSynchronize(s);
    struct Agg { int n = 0; double ms = 0, flops = 0, bytes = 0, direct = 0; };
    std::map<std::string, Agg> agg;
    for (auto& e : entries) {
        float ms = 0.f;
        if (!e.k.empty()) {                     // kernel time: the sum over the scope's launches
            bool ok = true;
            for (size_t i = 0; i + 1 < e.k.size(); i += 2) {
                float t = 0.f;
                if (hipEventElapsedTime(&t, e.k[i], e.k[i + 1]) != hipSuccess) { ok = false; break; }
                ms += t;
            }
            if (!ok) continue;
        } else if (hipEventElapsedTime(&ms, e.e0, e.e1) != hipSuccess) continue;
        Agg& a = agg[e.tag];
        a.n += 1; a.ms += ms; a.flops += e.flops; a.bytes += e.bytes; a.direct += e.direct_flops;
    }
    std::ostringstream o;
    o.precision(9);
    o << "{";
    bool first = true;
    for (auto& kv : agg) {
        if (!first) o << ",";
        first = false;
        o << "\"" << kv.first << "\":{\"n\":" << kv.second.n << ",\"ms\":" << kv.second.ms << ",\"flops\":" << kv.second.flops
          << ",\"bytes\":" << kv.second.bytes << ",\"direct_flops\":" << kv.second.direct << "}";
    }
    o << "}";
    reset();
    return o.str();
}
