// dl4ds_amd -- conv_wino_kernel<3, 3, *>: 48 input channels per pass, 48 output channels per workgroup (see conv_wino_kernel.h)
#include "conv_wino_kernel.h"

void launch_wino_33(hipStream_t s, WinoParams& wp, int SX, int epi) { wino::launch_shape<3, 3>(s, wp, SX, epi); }
