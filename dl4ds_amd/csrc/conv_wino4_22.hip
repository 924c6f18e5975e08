// dl4ds_amd -- conv_wino4_kernel<2, 2, *>: F(4x4, 3x3), 32 input channels per pass, 32 output channels per workgroup (conv_wino4_kernel.h)
#include "conv_wino4_kernel.h"

void launch_wino4_22(hipStream_t s, WinoParams& wp, int SX, int epi) { wino4::launch_shape<2, 2>(s, wp, SX, epi); }
