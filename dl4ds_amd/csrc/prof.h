// Optional per-launch timing with HIP events on the launch stream (off by default; used by bench.py to
// measure the dominant kernel's duration live and to print a per-kernel breakdown).
//
// A tag's time is KERNEL time: every launch made through DL4DS_LAUNCH inside a ProfScope carries its own start / stop events
// (hipExtLaunchKernelGGL: the dispatch packet's begin and end timestamps), and the tag's figure is the sum over its launches.
// (Round 3 recorded two events AROUND the scope on the stream: with every launch instrumented the host falls behind the
// device, and the window of a short kernel then held the host's launch latency -- maxpool2_fwd 0.67 ms per step of cfg5
// against 0.10 ms in rocprofv3's kernel trace.)  A scope that launches nothing through the macro (memcpys) falls back to
// the pair of events around it.
#pragma once
#include "common.h"
#include <hip/hip_ext.h>
#include <string>
#include <vector>

struct ProfEntry {
    std::string tag;
    double flops, bytes;
    double direct_flops;      // launches that issue fewer multiply-adds than the layer's direct form (Winograd): the direct-form count
    hipEvent_t e0, e1;        // around the scope (fallback)
    std::vector<hipEvent_t> k;      // start, stop of every kernel launched inside the scope
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

struct ProfScope;
ProfScope*& prof_current();       // innermost live scope.  SINGLE HOST THREAD: the profiler, this pointer, convlstm_seq's epoch
                                  // counter and the lazily created device error word are process-wide statics without locks -- the
                                  // library is driven from one host thread per process (one process per GPU), as the C header states

struct ProfScope {
    hipStream_t s;
    int idx = -1;
    ProfScope* outer = nullptr;
    bool linked = false;
    ProfScope(hipStream_t stream, const std::string& tag, double flops, double bytes, double direct_flops = 0.0) : s(stream) {
        Profiler& p = prof();
        if (!p.on) return;
        outer = prof_current();
        prof_current() = this;
        linked = true;
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
        if (linked) prof_current() = outer;
    }
};

// start / stop events for the next kernel launch if it happens inside a timed scope, else (nullptr, nullptr)
inline bool prof_launch_events(hipEvent_t& a, hipEvent_t& b) {
    ProfScope* sc = prof().on ? prof_current() : nullptr;
    while (sc && sc->idx < 0) sc = sc->outer;          // a nested scope the filter rejected must not hide a matching outer one
    if (!sc) { a = b = nullptr; return false; }
    Profiler& p = prof();
    a = p.get_event();
    b = p.get_event();
    p.entries[sc->idx].k.push_back(a);
    p.entries[sc->idx].k.push_back(b);
    return true;
}

// every kernel launch of the library goes through this: a plain launch unless the profiler wants this launch's own timestamps
#define DL4DS_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                          \
    do {                                                                                                               \
        hipEvent_t pe0_, pe1_;                                                                                         \
        if (prof_launch_events(pe0_, pe1_)) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, pe0_, pe1_, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                      \
    } while (0)
