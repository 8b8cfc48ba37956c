// f32 (exact-fp32 MFMA) instantiations of the pipelined conv-GEMM (see conv_fast.inc) + the dtype dispatch
#define LDC_FAST_T float
#define LDC_FAST_NS fast_f32
#define LDC_FAST_ENTRY launch_conv_fast_f32
#define LDC_FAST_PAIR_ENTRY launch_conv_fast_pair_f32
#include "conv_fast.inc"

namespace ldc {

unsigned long long* g_conv_stamps = nullptr;

hipError_t launch_conv_fast_bf16(const ConvLayer& ly, const ConvKArgs& a_in, int M, int span_rows, hipStream_t s, bool* launched);

hipError_t launch_conv_fast_pair_bf16(const ConvLayer& ly0, const ConvKArgs& a0, int M0, int span0, const ConvLayer& ly1, const ConvKArgs& a1, int M1,
                                      int span1, unsigned* pair_done, int pair_done_cap, hipStream_t s, bool* launched);

hipError_t launch_conv_fast_pair(const ConvLayer& ly0, const ConvKArgs& a0, int M0, int span0, const ConvLayer& ly1, const ConvKArgs& a1, int M1,
                                 int span1, unsigned* pair_done, int pair_done_cap, hipStream_t s, bool* launched) {
  *launched = false;
  if (ly0.dt != ly1.dt || ly0.w8 || ly1.w8 || !conv_fast_eligible(ly0) || !conv_fast_eligible(ly1)) return hipSuccess;
  if (ly0.dt == DT_F32) return launch_conv_fast_pair_f32(ly0, a0, M0, span0, ly1, a1, M1, span1, pair_done, pair_done_cap, s, launched);
  if (ly0.dt == DT_BF16) return launch_conv_fast_pair_bf16(ly0, a0, M0, span0, ly1, a1, M1, span1, pair_done, pair_done_cap, s, launched);
  return hipSuccess;
}

bool conv_fast_eligible(const ConvLayer& ly) {
  return ly.pad_mode == PAD_ZERO && ly.pre_act == ACT_NONE && ly.tr_stride == 0 && ly.taps <= 8;
}

hipError_t launch_conv_fast(const ConvLayer& ly, const ConvKArgs& a, int M, int span_rows, hipStream_t s, bool* launched) {
  if (ly.dt == DT_F32) return launch_conv_fast_f32(ly, a, M, span_rows, s, launched);
  if (ly.dt == DT_FP8) return launch_conv_fast_fp8(ly, a, M, span_rows, s, launched);
  return ly.w8 ? launch_conv_fast_bf16w8(ly, a, M, span_rows, s, launched) : launch_conv_fast_bf16(ly, a, M, span_rows, s, launched);
}

}  // namespace ldc
