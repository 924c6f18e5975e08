// Data parallelism over RCCL/xGMI (one process per GPU) -- replaces Horovod (SURVEY.md section 2 rows 18,19).
#pragma once
#include "common.h"
void dist_unique_id(char id128[128]);
void dist_init(int rank, int world, const char id128[128]);
void dist_world(int& rank, int& world);
// sum-all-reduce `n` floats in place.  Enqueued on `stream` order: the collective runs on the side (comm)
// stream behind an event recorded on `stream`, and `stream` waits for its completion event.
void dist_allreduce_grads(float* buf, size_t n, hipStream_t stream);
// bucketed form: launch the collective for one bucket behind `stream`'s current position (returns immediately, the
// compute stream does NOT wait) ... and one wait for all buckets launched since the last wait
void dist_allreduce_bucket_async(float* buf, size_t n, hipStream_t stream, hipStream_t aux = nullptr);
void dist_allreduce_wait(hipStream_t stream);
bool dist_active();
void dist_broadcast(float* buf, size_t n, int root, hipStream_t stream);
void dist_finalize();
