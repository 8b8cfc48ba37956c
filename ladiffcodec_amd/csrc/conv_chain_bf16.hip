// the XCD-team chain kernel (bf16): the body of the pipelined conv-GEMM inside one persistent, ticket-driven launch (see conv_fast.inc)
#define LDC_FAST_T __bf16
#define LDC_FAST_NS chain_bf16
#define LDC_FAST_CHAIN_ONLY 1
#define LDC_OPAQUE_TID 1
#include "conv_fast.inc"
