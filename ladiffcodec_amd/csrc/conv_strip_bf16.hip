// bf16 instantiations of the strip conv (see conv_strip.inc)
#define LDC_STRIP_T __bf16
#define LDC_STRIP_NS strip_bf16
#define LDC_STRIP_GEOM strip_geom_bf16
#define LDC_STRIP_ENTRY launch_conv_strip_bf16
#include "conv_strip.inc"
#include "conv_strip_entry.inc"
