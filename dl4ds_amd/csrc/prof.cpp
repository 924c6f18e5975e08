#include "prof.h"
#include <map>
#include <sstream>

static Profiler g_prof;
Profiler& prof() { return g_prof; }

hipEvent_t Profiler::get_event() {
    hipEvent_t e;
    if (!pool.empty()) { e = pool.back(); pool.pop_back(); return e; }
    (void)hipEventCreate(&e);
    return e;
}

void Profiler::reset() {
    for (auto& e : entries) { pool.push_back(e.e0); pool.push_back(e.e1); }
    entries.clear();
}

std::string Profiler::report_json(hipStream_t s) {
    (void)hipStreamSynchronize(s);
    struct Agg { int n = 0; double ms = 0, flops = 0, bytes = 0, direct = 0; };
    std::map<std::string, Agg> agg;
    for (auto& e : entries) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.e0, e.e1) != hipSuccess) continue;
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
