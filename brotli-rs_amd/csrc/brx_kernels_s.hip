// the lean instance of the decode kernel (brx_device.h, brx_small.h): 5 KiB of LDS per wave, <= 64 VGPRs, 32 waves per CU
#define BRX_SMALL 1
#include "brx_kernels.hip"
