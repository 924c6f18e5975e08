// Library-wide runtime state: device, the single compute stream, the side (comm) stream, error slot.
#pragma once
#include "common.h"

struct Runtime {
    int device = -1;
    hipStream_t stream = nullptr;        // all kernels
    hipStream_t comm_stream = nullptr;   // RCCL collectives overlapped with backward
    hipStream_t aux_stream = nullptr;    // weight-gradient kernels, concurrent with the dgrad chain
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool inited = false;
};
Runtime& rt();
void rt_ensure_init();
void set_last_error(const std::string& s);
