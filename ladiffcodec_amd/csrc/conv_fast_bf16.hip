// bf16 instantiations of the pipelined conv-GEMM (conv_fast.inc: conv_fast_kernel; conv_lean.inc: conv_lean_kernel)
#define LDC_FAST_T __bf16
#define LDC_FAST_NS fast_bf16
#define LDC_FAST_ENTRY launch_conv_fast_bf16
#define LDC_FAST_RESIDENCY conv_wgs_per_cu_bf16
#include "conv_fast.inc"
