// bf16 instantiations of the pipelined conv-GEMM (see conv_fast.inc)
#define LDC_FAST_T __bf16
// #define LDC_FAST_TALL 1   // 256 x 64 tiles (conv_fast.inc: launch_fast_bn): built and measured SLOWER (256 VGPRs, two workgroups per CU: +1.6 % on the
                             // largest grids alone, +10 % everywhere); the instantiations cost 3 min of build time, so they are compiled out
#define LDC_FAST_NS fast_bf16
#define LDC_FAST_ENTRY launch_conv_fast_bf16
#define LDC_FAST_PAIR_ENTRY launch_conv_fast_pair_bf16
#define LDC_FAST_PREP_ENTRY prep_conv_fast_bf16
#include "conv_fast.inc"
