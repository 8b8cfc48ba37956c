// rvq.hip -- residual vector quantisation, nearest neighbour (eval mode).
//
// ResidualVectorQuantization.forward (reference srcs/quantization/core_vq.py:324-342) over
// EuclideanCodebook.quantize (:174-182) / dequantize (:187-189):
//   per stage: idx = argmax_j -( |x|^2 - 2 x.e_j + |e_j|^2 ) (first maximum wins) ; q = E[idx] ;
//   residual -= q ; out += q.
// Integer output (codes) must be bit-exact, so everything is fp32 FMA with the reference's expansion.
// Two kernels with the SAME arithmetic per (row, code) -- one fp32 FMA chain over the 128 dimensions in ascending order,
// the same distance expression, first maximum wins -- so their codes are bit-identical (tests/test_gpu_parity.py):
//   rvq_tiled_kernel (round 5, the default for 128-aligned codebooks): a workgroup of four waves owns 4 R rows for ALL stages;
//     every stage's codebook streams through LDS in 128-code tiles (coalesced 16-byte loads, double-buffered, one barrier per
//     tile); a lane owns two codes of the tile, a wave = R rows whose residual lives lane-distributed in two VGPRs per row and
//     reaches the FMAs as v_readlane scalars.  VALU-bound: 1.5 issue slots per (row, code, dimension), 2 R chains in flight.
//   rvq_kernel (round 1): thread = code, 8 rows per workgroup, each lane walks its own 512-byte code vector in global memory
//     (64 cache lines per load instruction): latency-bound, 250..270 us at any batch.
#include <hip/hip_runtime.h>

#include <utility>

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
      for (int c = 0; c < D; ++c) s = fmaf(sres[tid][c], sres[tid][c], s);
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

// Dimensions 4 C4 .. 4 C4 + 3 of the FMA chains of R rows x the lane's two codes: the rows' four residual values come out of their
// lanes as scalars and feed the 8 R FMAs at once, the 2 R independent chains interleaved (a dependent v_fmac issues ~16 cycles after
// its predecessor, an independent one after 4).  One asm block, because left to the scheduler all the v_readlane of a tile move to
// its top and spill the SGPR file through v_writelane (3 instructions per value instead of 1); v_fmac_f32 is the fused multiply-add
// the compiler emits for fmaf.  (at least four instructions between a v_readlane and the v_fmac reading its SGPR: no hazard to pad)
template <int C4>
__device__ __forceinline__ void rvq_dot4(float (&dot)[1][2], const float (&xr)[1][2], const float4& e0, const float4& e1) {
  constexpr int L = (4 * C4) & 63;
  const float xs = (4 * C4) < 64 ? xr[0][0] : xr[0][1];
  int s0, s1, s2, s3;
  asm("v_readlane_b32 %2, %6, %15\n\t"
      "v_readlane_b32 %3, %6, %16\n\t"
      "v_readlane_b32 %4, %6, %17\n\t"
      "v_readlane_b32 %5, %6, %18\n\t"
      "v_fmac_f32 %0, %2, %7\n\t"
      "v_fmac_f32 %1, %2, %11\n\t"
      "v_fmac_f32 %0, %3, %8\n\t"
      "v_fmac_f32 %1, %3, %12\n\t"
      "v_fmac_f32 %0, %4, %9\n\t"
      "v_fmac_f32 %1, %4, %13\n\t"
      "v_fmac_f32 %0, %5, %10\n\t"
      "v_fmac_f32 %1, %5, %14"
      : "+v"(dot[0][0]), "+v"(dot[0][1]), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3)
      : "v"(xs), "v"(e0.x), "v"(e0.y), "v"(e0.z), "v"(e0.w), "v"(e1.x), "v"(e1.y), "v"(e1.z), "v"(e1.w), "n"(L), "n"(L + 1), "n"(L + 2),
        "n"(L + 3));
}
template <int C4>
__device__ __forceinline__ void rvq_dot4(float (&dot)[2][2], const float (&xr)[2][2], const float4& e0, const float4& e1) {
  constexpr int L = (4 * C4) & 63;
  const float xa = (4 * C4) < 64 ? xr[0][0] : xr[0][1];
  const float xb = (4 * C4) < 64 ? xr[1][0] : xr[1][1];
  int s0, s1, s2, s3, t0, t1, t2, t3;
  asm("v_readlane_b32 %4, %12, %22\n\t"
      "v_readlane_b32 %8, %13, %22\n\t"
      "v_readlane_b32 %5, %12, %23\n\t"
      "v_readlane_b32 %9, %13, %23\n\t"
      "v_readlane_b32 %6, %12, %24\n\t"
      "v_readlane_b32 %10, %13, %24\n\t"
      "v_readlane_b32 %7, %12, %25\n\t"
      "v_readlane_b32 %11, %13, %25\n\t"
      "v_fmac_f32 %0, %4, %14\n\t"
      "v_fmac_f32 %1, %4, %18\n\t"
      "v_fmac_f32 %2, %8, %14\n\t"
      "v_fmac_f32 %3, %8, %18\n\t"
      "v_fmac_f32 %0, %5, %15\n\t"
      "v_fmac_f32 %1, %5, %19\n\t"
      "v_fmac_f32 %2, %9, %15\n\t"
      "v_fmac_f32 %3, %9, %19\n\t"
      "v_fmac_f32 %0, %6, %16\n\t"
      "v_fmac_f32 %1, %6, %20\n\t"
      "v_fmac_f32 %2, %10, %16\n\t"
      "v_fmac_f32 %3, %10, %20\n\t"
      "v_fmac_f32 %0, %7, %17\n\t"
      "v_fmac_f32 %1, %7, %21\n\t"
      "v_fmac_f32 %2, %11, %17\n\t"
      "v_fmac_f32 %3, %11, %21"
      : "+v"(dot[0][0]), "+v"(dot[0][1]), "+v"(dot[1][0]), "+v"(dot[1][1]), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "=&s"(t0), "=&s"(t1),
        "=&s"(t2), "=&s"(t3)
      : "v"(xa), "v"(xb), "v"(e0.x), "v"(e0.y), "v"(e0.z), "v"(e0.w), "v"(e1.x), "v"(e1.y), "v"(e1.z), "v"(e1.w), "n"(L), "n"(L + 1),
        "n"(L + 2), "n"(L + 3));
}
template <int R, int C0, int... I>
__device__ __forceinline__ void rvq_dot_half(float (&dot)[R][2], const float (&xr)[R][2], const float4 (&e0)[16], const float4 (&e1)[16],
                                             std::integer_sequence<int, I...>) {
  (rvq_dot4<C0 + I>(dot, xr, e0[I], e1[I]), ...);
}

template <int R>
__global__ __launch_bounds__(256) void rvq_tiled_kernel(const float* __restrict__ z, int rows, const float* __restrict__ cb,
                                                        const float* __restrict__ cbn, int bins, int n_q, int64_t* __restrict__ codes,
                                                        float* __restrict__ qout) {
  constexpr int D = 128, CT = 128, P = D + 4;   // tile: 128 codes at a pitch of 132 floats (the 16 lanes of a ds_read_b128 group start on 16 distinct 4-bank slots)
  constexpr int NPF = CT * D / 4 / 256;         // 16-byte pieces of a tile per thread
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [2][CT][P]
  __shared__ int scode[32][4 * R];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * 4 * R + wave * R;
  float xr[R][2];   // residual row r: dimension lane in [0], dimension 64 + lane in [1]
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool ok = row0 + r < rows;
    xr[r][0] = ok ? z[(size_t)(row0 + r) * D + lane] : 0.f;
    xr[r][1] = ok ? z[(size_t)(row0 + r) * D + 64 + lane] : 0.f;
  }
  const int ntiles = bins / CT;
  for (int q = 0; q < n_q; ++q) {
    const float* E = cb + (size_t)q * bins * D;
    const float* En = cbn + (size_t)q * bins;
    {  // tile 0 (buffer 0 was last read two barriers ago)
      const float4* src = reinterpret_cast<const float4*>(E);
      float4 p0[NPF];
#pragma unroll
      for (int k = 0; k < NPF; ++k) p0[k] = src[k * 256 + tid];
#pragma unroll
      for (int k = 0; k < NPF; ++k) {
        const int f = k * 256 + tid;
        *reinterpret_cast<float4*>(tile + (f >> 5) * P + (f & 31) * 4) = p0[k];
      }
    }
    float sxx[R], best[R];
    int bidx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float sacc = 0.f;
#pragma unroll
      for (int c = 0; c < D; ++c) {
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xr[r][c >> 6]), c & 63));
        sacc = fmaf(v, v, sacc);
      }
      sxx[r] = sacc;
      best[r] = -INFINITY;
      bidx[r] = 0;
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
      const bool more = t + 1 < ntiles;
      float4 pf[NPF];   // the next tile, in flight while this one is searched (the last iteration re-reads its own tile: no branch around the loads)
      {
        const float4* src = reinterpret_cast<const float4*>(E + (size_t)min(t + 1, ntiles - 1) * CT * D);
#pragma unroll
        for (int k = 0; k < NPF; ++k) pf[k] = src[k * 256 + tid];
      }
      const float* mine = tile + (t & 1) * CT * P + lane * P;   // codes lane and 64 + lane of the tile
      float dot[R][2];
#pragma unroll
      for (int r = 0; r < R; ++r) dot[r][0] = dot[r][1] = 0.f;
      {
        float4 e0[16], e1[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          e0[i] = *reinterpret_cast<const float4*>(mine + 4 * i);
          e1[i] = *reinterpret_cast<const float4*>(mine + 64 * P + 4 * i);
        }
        rvq_dot_half<R, 0>(dot, xr, e0, e1, std::make_integer_sequence<int, 16>());
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          e0[i] = *reinterpret_cast<const float4*>(mine + 64 + 4 * i);
          e1[i] = *reinterpret_cast<const float4*>(mine + 64 * P + 64 + 4 * i);
        }
        rvq_dot_half<R, 16>(dot, xr, e0, e1, std::make_integer_sequence<int, 16>());
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {   // j ascending per lane (64 h + lane inside the tile, tiles in order): first max kept
        const int j = t * CT + h * 64 + lane;
        const float ee = En[j];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float d = -((sxx[r] - 2.0f * dot[r][h]) + ee);
          if (d > best[r]) { best[r] = d; bidx[r] = j; }
        }
      }
      if (more) {
        float* dst = tile + ((t + 1) & 1) * CT * P;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
          const int f = k * 256 + tid;
          *reinterpret_cast<float4*>(dst + (f >> 5) * P + (f & 31) * 4) = pf[k];
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float v = best[r];
      int ix = bidx[r];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(ix, o);
        if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
      }
      ix = __builtin_amdgcn_readfirstlane(ix);
      if (lane == 0) {
        scode[q][wave * R + r] = ix;
        if (codes && row0 + r < rows) codes[(size_t)q * rows + row0 + r] = (int64_t)ix;
      }
      xr[r][0] -= E[(size_t)ix * D + lane];
      xr[r][1] -= E[(size_t)ix * D + 64 + lane];
    }
  }
  if (qout) {   // the reference sums the chosen code vectors in stage order (core_vq.py:338-340), not z - residual
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (row0 + r >= rows) continue;
      float a0 = 0.f, a1 = 0.f;
      for (int q = 0; q < n_q; ++q) {
        const float* ev = cb + ((size_t)q * bins + scode[q][wave * R + r]) * D;
        a0 += ev[lane];
        a1 += ev[64 + lane];
      }
      qout[(size_t)(row0 + r) * D + lane] = a0;
      qout[(size_t)(row0 + r) * D + 64 + lane] = a1;
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

template <int R>
static hipError_t launch_rvq_tiled(const float* z_rows, int rows, const float* codebooks, const float* cb_sqnorm, int bins, int n_q,
                                   int64_t* codes, float* quantized_rows, hipStream_t s) {
  constexpr size_t lds = 2 * 128 * 132 * sizeof(float);
  auto kern = rvq_tiled_kernel<R>;
  static bool lds_opt_in = false;   // one-off, outside any stream capture (the codec stages are first called eagerly)
  if (!lds_opt_in) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    lds_opt_in = true;
  }
  hipLaunchKernelGGL(kern, dim3((rows + 4 * R - 1) / (4 * R)), dim3(256), lds, s, z_rows, rows, codebooks, cb_sqnorm, bins, n_q, codes,
                     quantized_rows);
  return hipGetLastError();
}

hipError_t launch_rvq(const float* z_rows, int rows, int D, const float* codebooks, const float* cb_sqnorm, int bins,
                      int n_q, int64_t* codes, float* quantized_rows, hipStream_t s, int variant) {
  if (D != 128) return hipErrorInvalidValue;
  if (n_q > 32) return hipErrorInvalidValue;
  if (variant != 0 && bins % 128 == 0 && bins >= 128) {
    // up to 1024 rows: one row per wave (the launch is latency-bound: as many workgroups as the rows give); beyond: two, which
    // halves the codebook traffic out of L2
    if (rows <= 1024) return launch_rvq_tiled<1>(z_rows, rows, codebooks, cb_sqnorm, bins, n_q, codes, quantized_rows, s);
    return launch_rvq_tiled<2>(z_rows, rows, codebooks, cb_sqnorm, bins, n_q, codes, quantized_rows, s);
  }
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
