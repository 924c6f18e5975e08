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
// hipStreamSynchronize with a watchdog when more than one rank takes part (a stranded peer must fail loudly, not hang)
void dist_stream_sync(hipStream_t stream, const char* what);
void dist_broadcast(float* buf, size_t n, int root, hipStream_t stream);
void dist_finalize();
// --- fail-safe bring-up (the reference calls hvd.init() itself, training/base.py:97-107): the launcher's WORLD_SIZE is
// the number of ranks the job was started with; a train step in a process that was launched as one of several ranks but has
// no communicator (or one of another size) must fail instead of silently training an unsynchronised replica.
// DL4DS_ALLOW_UNSYNCED=1 opts out (independent replicas on purpose).
int dist_expected_world();
void dist_require_ready(const char* what);
// small host-side reductions across ranks through RCCL (validation loss, early-stopping decisions, max-over-ranks
// timing): op 0 sum, 1 max, 2 min; in place on `host`; synchronous; identity without a communicator
void dist_allreduce_host(float* host, int n, int op);
void dist_barrier();
// what RCCL itself reports for the communicator (0 ranks when there is none)
void dist_comm_info(int& nranks, int& rank, int& device);
void dist_broadcast_i64(long* host_value, int root);     // optimizer.iterations
