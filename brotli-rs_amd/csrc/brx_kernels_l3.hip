// brx_kernels_l3.hip -- level-3 instance of the decode kernel (brx_device.h, "Four instances of the kernel"): the same
// source with 40 960 B of LDS per wave, 4 waves per CU, for the streams the level below lists because their
// meta-block tables spill its LDS table memory (BrxKernelArgs::defer).
#define BRX_LEVEL 3
#include "brx_kernels.hip"
