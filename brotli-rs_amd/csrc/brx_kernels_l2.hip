// brx_kernels_l2.hip -- level-2 instance of the decode kernel (brx_device.h, "Four instances of the kernel"): the same
// source with 20 480 B of LDS per wave, 8 waves per CU, for the streams the level below lists because their
// meta-block tables spill its LDS table memory (BrxKernelArgs::defer).
#define BRX_LEVEL 2
#include "brx_kernels.hip"
