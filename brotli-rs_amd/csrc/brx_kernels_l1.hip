// brx_kernels_l1.hip -- level-1 instance of the decode kernel (brx_device.h, "Four instances of the kernel"): the same
// source with 12 800 B of LDS per wave, 12 waves per CU, for the streams the level below lists because their
// meta-block tables spill its LDS table memory (BrxKernelArgs::defer).
#define BRX_LEVEL 1
#include "brx_kernels.hip"
