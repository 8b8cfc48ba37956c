// ldc_math.h -- activation math shared by the kernels.  v_exp_f32 / v_rcp_f32 forms (1 ulp each): the IEEE
// division + libm expm1f/tanhf forms cost ~25 VALU instructions per element and made the GroupNorm-apply
// kernel VALU-bound (7 of its 15 us); absolute error of these forms is < 2e-7.
#pragma once
#include <hip/hip_runtime.h>

namespace ldc {

// f32 -> bf16, round to nearest even, NaN kept: gfx950 has it in hardware (v_cvt_pk_bf16_f32 converts a pair)
__device__ __forceinline__ unsigned short hw_bf16(float f) {
  const __bf16 h = (__bf16)f;
  unsigned short u;
  __builtin_memcpy(&u, &h, 2);
  return u;
}
__device__ __forceinline__ unsigned hw_bf16x2(float lo, float hi) {
  typedef __bf16 v2 __attribute__((ext_vector_type(2)));
  v2 v;
  v[0] = (__bf16)lo; v[1] = (__bf16)hi;
  unsigned u;
  __builtin_memcpy(&u, &v, 4);
  return u;
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sigmoid(float v) { return fast_rcp(1.0f + __expf(-v)); }
__device__ __forceinline__ float fast_silu(float v) { return v * fast_sigmoid(v); }
__device__ __forceinline__ float fast_elu(float v) { return v > 0.0f ? v : __expf(v) - 1.0f; }
__device__ __forceinline__ float fast_tanh(float v) {
  const float e = __expf(-2.0f * fabsf(v));            // in (0, 1]: no overflow
  const float t = (1.0f - e) * fast_rcp(1.0f + e);
  return copysignf(t, v);
}

}  // namespace ldc
