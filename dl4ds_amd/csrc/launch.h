// Grid sizing for persistent kernels: blocks that are all co-resident on the device (CUs x occupancy).  Blocks of a
// persistent kernel do equal work, so a grid slightly larger than one residency round costs a whole extra round.
#pragma once
#include "common.h"

template <auto Kern>
inline int resident_blocks(int threads, size_t dyn_lds = 0) {
    static int cached = 0;
    if (cached == 0) {
        int dev = 0, ncu = 0, per_cu = 0;
        HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, Kern, threads, dyn_lds));
        cached = ncu * (per_cu < 1 ? 1 : per_cu);
    }
    return cached;
}
