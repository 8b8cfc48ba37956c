// bf16 activations x fp8 (e4m3) weights instantiations of the pipelined conv-GEMM (see conv_fast.inc, LDC_FAST_W8)
#define LDC_FAST_T __bf16
#define LDC_FAST_NS fast_bf16w8
#define LDC_FAST_ENTRY launch_conv_fast_bf16w8
#define LDC_FAST_W8 1
#define LDC_FAST_RESIDENCY conv_wgs_per_cu_bf16w8
#include "conv_fast.inc"
