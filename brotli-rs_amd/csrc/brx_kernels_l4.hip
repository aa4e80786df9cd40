// brx_kernels_l4.hip -- level-4 instance of the decode kernel (brx_device.h): the same source with 150 KiB of LDS per wave, ONE wave
// per CU, for the streams level 3 hands on because a meta-block's tables spill even its 37.6 KiB (BrxKernelArgs::handup2).
#define BRX_LEVEL 4
#include "brx_kernels.hip"
