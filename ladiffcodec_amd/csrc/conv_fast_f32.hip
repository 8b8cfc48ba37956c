// f32 (exact-fp32 MFMA) instantiations of the pipelined conv-GEMM (see conv_fast.inc) + the dtype dispatch
#define LDC_FAST_T float
#define LDC_FAST_NS fast_f32
#define LDC_FAST_ENTRY launch_conv_fast_f32
#define LDC_FAST_RESIDENCY conv_wgs_per_cu_f32
#include "conv_fast.inc"

namespace ldc {

unsigned long long* g_conv_stamps = nullptr;

hipError_t launch_conv_fast_bf16(const ConvLayer& ly, const ConvKArgs& a_in, int M, int span_rows, hipStream_t s, bool* launched);
int conv_wgs_per_cu_bf16();
int conv_wgs_per_cu_bf16w8();
int conv_wgs_per_cu_fp8();

// workgroups per CU of the kernels a fused-GroupNorm conv of this engine lands on (asked once per kind)
int conv_fused_gn_wgs_per_cu(int dt, bool w8) {
  static int cache[4] = {-1, -1, -1, -1};
  const int k = dt == DT_F32 ? 0 : (dt == DT_FP8 ? 1 : (w8 ? 2 : 3));
  if (cache[k] < 0) cache[k] = k == 0 ? conv_wgs_per_cu_f32() : (k == 1 ? conv_wgs_per_cu_fp8() : (k == 2 ? conv_wgs_per_cu_bf16w8() : conv_wgs_per_cu_bf16()));
  return cache[k];
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
