// f32 (exact-fp32 MFMA) instantiations of the pipelined conv-GEMM (see conv_fast.inc) + the dtype dispatch
#define LDC_FAST_T float
#define LDC_FAST_NS fast_f32
#define LDC_FAST_ENTRY launch_conv_fast_f32
#include "conv_fast.inc"

namespace ldc {

unsigned long long* g_conv_stamps = nullptr;

hipError_t launch_conv_fast_bf16(const ConvLayer& ly, const ConvKArgs& a_in, int M, int span_rows, hipStream_t s, bool* launched);

bool conv_fast_eligible(const ConvLayer& ly) {
  return ly.pad_mode == PAD_ZERO && ly.pre_act == ACT_NONE && ly.tr_stride == 0 && ly.taps <= 8;
}

hipError_t launch_conv_fast(const ConvLayer& ly, const ConvKArgs& a, int M, int span_rows, hipStream_t s, bool* launched) {
  if (ly.dt == DT_F32) return launch_conv_fast_f32(ly, a, M, span_rows, s, launched);
  if (ly.dt == DT_FP8) return launch_conv_fast_fp8(ly, a, M, span_rows, s, launched);
  return ly.w8 ? launch_conv_fast_bf16w8(ly, a, M, span_rows, s, launched) : launch_conv_fast_bf16(ly, a, M, span_rows, s, launched);
}

}  // namespace ldc
