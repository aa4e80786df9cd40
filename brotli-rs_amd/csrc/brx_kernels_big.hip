// brx_kernels_big.hip -- the wide-LDS instance of the decode kernel (see brx_device.h, "Two variants of the kernel"):
// the same source with 20 KiB of LDS per wave (17 152 B of table memory), 8 waves per CU, for the streams the regular
// kernel defers because their meta-block tables spill its 6 912 B (BrxKernelArgs::defer).
#define BRX_BIG 1
#include "brx_kernels.hip"
