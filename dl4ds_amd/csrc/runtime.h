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

// Sticky device-side error word (host-pinned, device-mapped): a persistent kernel that gives up a bounded spin stores a code
// here instead of continuing silently; every host-side wait (dist_stream_sync) checks it after the stream has drained and
// throws.  device_error_word(): the DEVICE pointer kernels store through; device_error_check(what): throws if set.
enum : unsigned { DEV_ERR_CONVLSTM_SEQ_TIMEOUT = 1u };
unsigned* device_error_word();
void device_error_check(const char* what);
