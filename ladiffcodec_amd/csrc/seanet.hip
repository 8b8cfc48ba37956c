// seanet.hip -- the two SEANet pieces that are not a channel GEMM: the Cin=1 input conv and the LSTM
// time recurrence.
//
//   conv_cin1 : SEANetEncoder.model[0] = SConv1d(1, n_filters, 7, causal, reflect)  (reference
//               srcs/modules/seanet.py:108-111, conv.py:217-232).  HBM-bound: reads T floats, writes T*Cout.
//   lstm      : SLSTM.forward (lstm.py:22-28) = nn.LSTM (gate order i,f,g,o) + skip.  The input
//               projection for all T runs as one GEMM on the conv kernel; what is left is the strictly
//               sequential h_{t-1} -> h_t chain, latency-bound: one workgroup per utterance keeps W_hh in
//               registers (H <= 128: one gate row per thread) or streams a k-major copy from L2 (H = 512).
#include "ldc_kernels.h"

namespace ldc {

__device__ __forceinline__ float sbf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short sf2bf(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
template <typename T>
__device__ __forceinline__ float sld(const void* p, size_t i);
template <>
__device__ __forceinline__ float sld<float>(const void* p, size_t i) { return reinterpret_cast<const float*>(p)[i]; }
template <>
__device__ __forceinline__ float sld<__bf16>(const void* p, size_t i) {
  return sbf2f(reinterpret_cast<const unsigned short*>(p)[i]);
}
template <typename T>
__device__ __forceinline__ void sst(void* p, size_t i, float v);
template <>
__device__ __forceinline__ void sst<float>(void* p, size_t i, float v) { reinterpret_cast<float*>(p)[i] = v; }
template <>
__device__ __forceinline__ void sst<__bf16>(void* p, size_t i, float v) {
  reinterpret_cast<unsigned short*>(p)[i] = sf2bf(v);
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_cin1_kernel(const float* x, void* y, const float* w, const float* bias,
                                                        int L, int Cout, int k) {
  extern __shared__ float sw[];   // [Cout][k] + [Cout]
  for (int i = threadIdx.x; i < Cout * k; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < Cout; i += 256) sw[Cout * k + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int b = blockIdx.y;
  const int pad = k - 1;
  const size_t total = (size_t)L * Cout;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int l = (int)(idx / Cout), co = (int)(idx % Cout);
    float acc = sw[Cout * k + co];
    for (int t = 0; t < k; ++t) {
      int u = l + t - pad;
      if (u < 0) u = -u;            // causal reflect (left only; L > pad)
      if (u >= L) u = 2 * (L - 1) - u;
      acc += sw[co * k + t] * x[(size_t)b * L + u];
    }
    sst<T>(y, ((size_t)b * L + l) * Cout + co, acc);
  }
}

hipError_t launch_conv_cin1(int dt, const float* x, void* y, const float* w, const float* bias, int B, int L, int Cout,
                            int k, hipStream_t s) {
  if (L <= k - 1) return hipErrorInvalidValue;
  const size_t lds = (size_t)(Cout * k + Cout) * sizeof(float);
  int bx = (int)std::min<size_t>(((size_t)L * Cout + 255) / 256, 512);
  if (dt == DT_F32)
    hipLaunchKernelGGL(conv_cin1_kernel<float>, dim3(bx, B), dim3(256), lds, s, x, y, w, bias, L, Cout, k);
  else
    hipLaunchKernelGGL(conv_cin1_kernel<__bf16>, dim3(bx, B), dim3(256), lds, s, x, y, w, bias, L, Cout, k);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_acc(float v) { return 1.0f / (1.0f + expf(-v)); }

// One thread per gate row (4H threads), W_hh row in registers, h broadcast from LDS.
template <typename T, int H>
__global__ __launch_bounds__(4 * H) void lstm_reg_kernel(const void* pre, const float* w_hh, void* out, const void* skip,
                                                         int T_len) {
  __shared__ __attribute__((aligned(16))) float sh[H];
  __shared__ float sg[4 * H];
  const int b = blockIdx.x, row = threadIdx.x;
  float w[H];
#pragma unroll
  for (int k = 0; k < H; k += 4) {
    const float4 v = *reinterpret_cast<const float4*>(w_hh + (size_t)row * H + k);
    w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
  }
  if (row < H) sh[row] = 0.f;
  float c = 0.f;
  float p_next = sld<T>(pre, ((size_t)b * T_len) * (4 * H) + row);
  __syncthreads();
  for (int t = 0; t < T_len; ++t) {
    float g = p_next;
    if (t + 1 < T_len) p_next = sld<T>(pre, ((size_t)b * T_len + t + 1) * (4 * H) + row);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < H; k += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(&sh[k]);
      a0 += w[k] * hv.x; a1 += w[k + 1] * hv.y; a2 += w[k + 2] * hv.z; a3 += w[k + 3] * hv.w;
    }
    g += (a0 + a1) + (a2 + a3);
    sg[row] = g;
    __syncthreads();
    if (row < H) {
      const float ig = sigmoid_acc(sg[row]), fg = sigmoid_acc(sg[H + row]);
      const float gg = tanhf(sg[2 * H + row]), og = sigmoid_acc(sg[3 * H + row]);
      c = fg * c + ig * gg;
      const float h = og * tanhf(c);
      sh[row] = h;
      const size_t o = ((size_t)b * T_len + t) * H + row;
      sst<T>(out, o, skip ? h + sld<T>(skip, o) : h);
    }
    __syncthreads();
  }
}

// Generic H: 1024 threads, thread owns gate rows {tid, tid+1024, ...}; W_hh pre-transposed k-major
// [H/4][4H][4] so that a wavefront reads 1 KiB contiguous per instruction from L2.
template <typename T>
__global__ __launch_bounds__(1024) void lstm_stream_kernel(const void* pre, const float* w_km, void* out, const void* skip,
                                                           int T_len, int H) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sh = sm;            // [H]
  float* sg = sm + H;        // [4H]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int G = 4 * H;
  for (int i = tid; i < H; i += 1024) sh[i] = 0.f;
  // cell state lives with the threads that own rows < H
  float c_state[4] = {0.f, 0.f, 0.f, 0.f};   // up to H = 4096
  __syncthreads();
  for (int t = 0; t < T_len; ++t) {
    for (int row = tid; row < G; row += 1024) {
      float acc0 = 0.f, acc1 = 0.f;
      const float4* wp = reinterpret_cast<const float4*>(w_km) + row;
      for (int k4 = 0; k4 < H / 4; k4 += 2) {
        const float4 w0 = wp[(size_t)k4 * G];
        const float4 w1 = wp[(size_t)(k4 + 1) * G];
        const float4 h0 = *reinterpret_cast<const float4*>(&sh[4 * k4]);
        const float4 h1 = *reinterpret_cast<const float4*>(&sh[4 * k4 + 4]);
        acc0 += w0.x * h0.x + w0.y * h0.y + w0.z * h0.z + w0.w * h0.w;
        acc1 += w1.x * h1.x + w1.y * h1.y + w1.z * h1.z + w1.w * h1.w;
      }
      sg[row] = acc0 + acc1 + sld<T>(pre, ((size_t)b * T_len + t) * G + row);
    }
    __syncthreads();
    int ci = 0;
    for (int j = tid; j < H; j += 1024, ++ci) {
      const float ig = sigmoid_acc(sg[j]), fg = sigmoid_acc(sg[H + j]);
      const float gg = tanhf(sg[2 * H + j]), og = sigmoid_acc(sg[3 * H + j]);
      const float c = fg * c_state[ci] + ig * gg;
      c_state[ci] = c;
      const float h = og * tanhf(c);
      sh[j] = h;
      const size_t o = ((size_t)b * T_len + t) * H + j;
      sst<T>(out, o, skip ? h + sld<T>(skip, o) : h);
    }
    __syncthreads();
  }
}

// w_hh points at: [4H][H] row-major for the register variants (H = 64, 128), k-major [H/4][4H][4] otherwise.
hipError_t launch_lstm_layer(int dt, const void* pre, const float* w_hh, void* out, const void* skip, int B, int T,
                             int H, hipStream_t s) {
  if (H % 8 || H > 4096) return hipErrorInvalidValue;
  if (H == 64) {
    if (dt == DT_F32) hipLaunchKernelGGL((lstm_reg_kernel<float, 64>), dim3(B), dim3(256), 0, s, pre, w_hh, out, skip, T);
    else hipLaunchKernelGGL((lstm_reg_kernel<__bf16, 64>), dim3(B), dim3(256), 0, s, pre, w_hh, out, skip, T);
  } else if (H == 128) {
    if (dt == DT_F32) hipLaunchKernelGGL((lstm_reg_kernel<float, 128>), dim3(B), dim3(512), 0, s, pre, w_hh, out, skip, T);
    else hipLaunchKernelGGL((lstm_reg_kernel<__bf16, 128>), dim3(B), dim3(512), 0, s, pre, w_hh, out, skip, T);
  } else {
    const size_t lds = (size_t)5 * H * sizeof(float);
    if (dt == DT_F32) hipLaunchKernelGGL(lstm_stream_kernel<float>, dim3(B), dim3(1024), lds, s, pre, w_hh, out, skip, T, H);
    else hipLaunchKernelGGL(lstm_stream_kernel<__bf16>, dim3(B), dim3(1024), lds, s, pre, w_hh, out, skip, T, H);
  }
  return hipGetLastError();
}

}  // namespace ldc
