// Optional per-launch timing with HIP events on the launch stream (off by default; used by bench.py to
// measure the dominant kernel's duration live and to print a per-kernel breakdown).
#pragma once
#include "common.h"
#include <string>
#include <vector>

struct ProfEntry {
    std::string tag;
    double flops, bytes;
    double direct_flops;      // launches that issue fewer multiply-adds than the layer's direct form (Winograd): the direct-form count
    hipEvent_t e0, e1;
};

struct Profiler {
    bool on = false;
    std::string filter;       // non-empty: only launches whose tag starts with it are timed (bench.py: the dominant
                              // kernel, measured inside the timed region at negligible cost)
    std::vector<ProfEntry> entries;
    std::vector<hipEvent_t> pool;
    hipEvent_t get_event();
    void reset();
    std::string report_json(hipStream_t s);     // synchronises the stream
};
Profiler& prof();

struct ProfScope {
    hipStream_t s;
    int idx = -1;
    ProfScope(hipStream_t stream, const std::string& tag, double flops, double bytes, double direct_flops = 0.0) : s(stream) {
        Profiler& p = prof();
        if (!p.on) return;
        if (!p.filter.empty() && tag.compare(0, p.filter.size(), p.filter) != 0) return;
        ProfEntry e;
        e.tag = tag; e.flops = flops; e.bytes = bytes; e.direct_flops = direct_flops > 0.0 ? direct_flops : flops;
        e.e0 = p.get_event();
        e.e1 = p.get_event();
        (void)hipEventRecord(e.e0, s);
        p.entries.push_back(e);
        idx = (int)p.entries.size() - 1;
    }
    ~ProfScope() {
        if (idx >= 0) (void)hipEventRecord(prof().entries[idx].e1, s);
    }
};
