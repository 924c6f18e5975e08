// dl4ds_amd -- conv_wino_kernel<2, 2, *>: 32 input channels per pass, 32 output channels per workgroup (see conv_wino_kernel.h)
#include "conv_wino_kernel.h"

void launch_wino_22(hipStream_t s, WinoParams& wp, int SX, int epi) { wino::launch_shape<2, 2>(s, wp, SX, epi); }
