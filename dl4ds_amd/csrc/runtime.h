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
// The error is FATAL FOR THE STEP, not for the optimiser state: the trainers' Adam kernel reads the word and leaves parameters and
// moments untouched when it is set (the gradients of that step are invalid), so a caller that catches the exception continues
// from the state before the failed step.  In a multi-rank job the raising rank throws out of its step and the launcher tears the
// job down (bench.py / parallel.py propagate a rank's failure); peers blocked in a collective end at the communicator watchdog.
enum : unsigned { DEV_ERR_CONVLSTM_SEQ_TIMEOUT = 1u };
unsigned* device_error_word();
unsigned* device_error_word_if_any();     // nullptr until some op has asked for the word (Adam skips its update when it is set)
void device_error_check(const char* what);
