// dl4ds_amd -- conv_wino_kernel<2, 3, *>: 32 input channels per pass, 48 output channels per workgroup (see conv_wino_kernel.h)
#include "conv_wino_kernel.h"

void launch_wino_23(hipStream_t s, WinoParams& wp, int SX, int epi) { wino::launch_shape<2, 3>(s, wp, SX, epi); }
