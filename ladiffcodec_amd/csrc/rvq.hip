// rvq.hip -- residual vector quantisation, nearest neighbour (eval mode).
//
// ResidualVectorQuantization.forward (reference srcs/quantization/core_vq.py:324-342) over
// EuclideanCodebook.quantize (:174-182) / dequantize (:187-189):
//   per stage: idx = argmax_j -( |x|^2 - 2 x.e_j + |e_j|^2 ) (first maximum wins) ; q = E[idx] ;
//   residual -= q ; out += q.
// Integer output (codes) must be bit-exact, so everything is fp32 FMA with the reference's expansion.
// A workgroup owns 16 rows for ALL stages (rows are independent), streaming each 512 KB codebook from
// L2: latency/L2-bound, 0.19 GFLOP per utterance.
#include "ldc_kernels.h"

namespace ldc {

template <int D, int ROWS>
__global__ __launch_bounds__(256) void rvq_kernel(const float* z, int rows, const float* cb, const float* cbn, int bins,
                                                  int n_q, int64_t* codes, float* qout) {
  __shared__ __attribute__((aligned(16))) float sres[ROWS][D];     // residual rows
  __shared__ float sxx[ROWS];
  __shared__ float sbest[ROWS][4];
  __shared__ int sidx[ROWS][4];
  __shared__ int scode[32][ROWS];   // chosen index per stage (n_q <= 32)
  const int r0 = blockIdx.x * ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < ROWS * D; i += 256) {
    const int r = i / D, c = i % D;
    sres[r][c] = (r0 + r < rows) ? z[(size_t)(r0 + r) * D + c] : 0.f;
  }
  __syncthreads();
  for (int q = 0; q < n_q; ++q) {
    const float* E = cb + (size_t)q * bins * D;
    const float* En = cbn + (size_t)q * bins;
    if (tid < ROWS) {
      float s = 0.f;
      for (int c = 0; c < D; ++c) s += sres[tid][c] * sres[tid][c];
      sxx[tid] = s;
    }
    __syncthreads();
    // thread -> codes j = tid, tid+256, ...; all ROWS rows at once (code vector read once from L2)
    float best[ROWS];
    int bidx[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) { best[r] = -INFINITY; bidx[r] = 0; }
    for (int j = tid; j < bins; j += 256) {
      float dot[ROWS];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) dot[r] = 0.f;
      const float4* ej = reinterpret_cast<const float4*>(E + (size_t)j * D);
      for (int c4 = 0; c4 < D / 4; ++c4) {
        const float4 e = ej[c4];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          const float4 x = *reinterpret_cast<const float4*>(&sres[r][4 * c4]);
          dot[r] = fmaf(x.w, e.w, fmaf(x.z, e.z, fmaf(x.y, e.y, fmaf(x.x, e.x, dot[r]))));
        }
      }
      const float ee = En[j];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const float d = -((sxx[r] - 2.0f * dot[r]) + ee);
        if (d > best[r]) { best[r] = d; bidx[r] = j; }   // j ascending per thread: first max kept
      }
    }
    // reduce over the 256 threads: max value, ties -> smallest index
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      float v = best[r];
      int ix = bidx[r];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(ix, o);
        if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
      }
      if (lane == 0) { sbest[r][wave] = v; sidx[r][wave] = ix; }
    }
    __syncthreads();
    if (tid < ROWS) {
      float v = sbest[tid][0];
      int ix = sidx[tid][0];
      for (int wv = 1; wv < 4; ++wv) {
        const float ov = sbest[tid][wv];
        const int oi = sidx[tid][wv];
        if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
      }
      sidx[tid][0] = ix;
      scode[q][tid] = ix;
      if (codes && r0 + tid < rows) codes[(size_t)q * rows + r0 + tid] = (int64_t)ix;
    }
    __syncthreads();
    for (int i = tid; i < ROWS * D; i += 256) {
      const int r = i / D, c = i % D;
      sres[r][c] -= E[(size_t)sidx[r][0] * D + c];
    }
    __syncthreads();
  }
  // quantized = z - final residual is NOT what the reference computes (it sums the code vectors);
  // accumulate the sum explicitly to keep the same rounding: out = ((q0 + q1) + q2) ...
  if (qout) {
    for (int i = tid; i < ROWS * D; i += 256) {
      const int r = i / D, c = i % D;
      if (r0 + r >= rows) continue;
      float acc = 0.f;
      for (int q = 0; q < n_q; ++q) {
        acc += cb[((size_t)q * bins + scode[q][r]) * D + c];
      }
      qout[(size_t)(r0 + r) * D + c] = acc;
    }
  }
}

__global__ __launch_bounds__(256) void rvq_decode_kernel(const int64_t* codes, int rows, int D, const float* cb, int bins,
                                                         int n_q, float* qout) {
  const size_t total = (size_t)rows * D;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / D;
    const int c = (int)(i % D);
    float acc = 0.f;
    for (int q = 0; q < n_q; ++q) acc += cb[((size_t)q * bins + codes[(size_t)q * rows + r]) * D + c];
    qout[i] = acc;
  }
}

__global__ __launch_bounds__(256) void sqnorm_rows_kernel(const float* x, int rows, int D, float* out) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  // same summation order as torch's `embed.pow(2).sum(0)` is not reproducible bit-for-bit anyway;
  // sequential fp32 sum
  float s = 0.f;
  for (int c = 0; c < D; ++c) s += x[(size_t)r * D + c] * x[(size_t)r * D + c];
  out[r] = s;
}

hipError_t launch_rvq(const float* z_rows, int rows, int D, const float* codebooks, const float* cb_sqnorm, int bins,
                      int n_q, int64_t* codes, float* quantized_rows, hipStream_t s) {
  if (D != 128) return hipErrorInvalidValue;
  if (n_q > 32) return hipErrorInvalidValue;
  constexpr int ROWS = 8;
  hipLaunchKernelGGL((rvq_kernel<128, ROWS>), dim3((rows + ROWS - 1) / ROWS), dim3(256), 0, s, z_rows, rows, codebooks,
                     cb_sqnorm, bins, n_q, codes, quantized_rows);
  return hipGetLastError();
}

hipError_t launch_rvq_decode(const int64_t* codes, int rows, int D, const float* codebooks, int bins, int n_q,
                             float* quantized_rows, hipStream_t s) {
  int blocks = (int)std::min<size_t>(((size_t)rows * D + 255) / 256, 2048);
  hipLaunchKernelGGL(rvq_decode_kernel, dim3(blocks), dim3(256), 0, s, codes, rows, D, codebooks, bins, n_q, quantized_rows);
  return hipGetLastError();
}

hipError_t launch_sqnorm_rows(const float* x, int rows, int D, float* out, hipStream_t s) {
  hipLaunchKernelGGL(sqnorm_rows_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, x, rows, D, out);
  return hipGetLastError();
}

}  // namespace ldc
