// bf16 instantiations of the pipelined conv-GEMM (see conv_fast.inc)
#define LDC_FAST_T __bf16
#define LDC_FAST_NS fast_bf16
#define LDC_FAST_ENTRY launch_conv_fast_bf16
#define LDC_FAST_PAIR_ENTRY launch_conv_fast_pair_bf16
#define LDC_FAST_PREP_ENTRY prep_conv_fast_bf16
#include "conv_fast.inc"
