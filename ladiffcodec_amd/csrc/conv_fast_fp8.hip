// fp8 (e4m3) activations x fp8 (e4m3) weights on the block-scaled fp8 MFMA: instantiations of the pipelined conv-GEMM
// (see conv_fast.inc, LDC_FAST_FP8).  Outputs are bf16.
#define LDC_FAST_FP8 1
#define LDC_FAST_T fast_fp8::fp8_tag
#define LDC_FAST_NS fast_fp8
#define LDC_FAST_ENTRY launch_conv_fast_fp8
#define LDC_FAST_RESIDENCY conv_wgs_per_cu_fp8
#include "conv_fast.inc"
