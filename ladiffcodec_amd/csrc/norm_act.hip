// norm_act.hip -- GroupNorm(+timestep scale/shift)+SiLU, channel LayerNorm, elementwise activations.
//
// Replaces nn.GroupNorm + `x*(scale+1)+shift` + nn.SiLU of Block.forward (reference
// srcs/modules/unet.py:145-154), the residual add of ResnetBlock.forward (:192), the gain-only
// channel LayerNorm (:82-91) used by PreNorm (:93-101) and LinearAttention.to_out (:203-206), and
// the tanh before final_conv (:467).  All are HBM-bound streaming kernels over channels-last rows:
// algorithmic bytes = (reads + writes) * rows * C * sizeof(dtype).
#include <stdlib.h>

#include "ldc_kernels.h"
#include "ldc_math.h"

namespace ldc {

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) { return hw_bf16(f); }

// 8 consecutive channels per thread: one 16 B (bf16) or two 16 B (f32) accesses
template <typename T>
struct Vec8;
template <>
struct Vec8<float> {
  static __device__ __forceinline__ void load(const void* p, size_t idx, float (&v)[8]) {
    const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + idx);
    float4 a = q[0], b = q[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void store(void* p, size_t idx, const float (&v)[8]) {
    float4* q = reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + idx);
    q[0] = make_float4(v[0], v[1], v[2], v[3]);
    q[1] = make_float4(v[4], v[5], v[6], v[7]);
  }
  static __device__ __forceinline__ void store_nt(void* p, size_t idx, const float (&v)[8]) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4* q = reinterpret_cast<f4*>(reinterpret_cast<float*>(p) + idx);
    f4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    __builtin_nontemporal_store(a, q);
    __builtin_nontemporal_store(b, q + 1);
  }
};
template <>
struct Vec8<__bf16> {
  static __device__ __forceinline__ void load(const void* p, size_t idx, float (&v)[8]) {
    uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p) + idx);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = bf2f((unsigned short)(w[i] & 0xffffu));
      v[2 * i + 1] = bf2f((unsigned short)(w[i] >> 16));
    }
  }
  static __device__ __forceinline__ void store(void* p, size_t idx, const float (&v)[8]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = hw_bf16x2(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p) + idx) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  static __device__ __forceinline__ void store_nt(void* p, size_t idx, const float (&v)[8]) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = hw_bf16x2(v[2 * i], v[2 * i + 1]);
    __builtin_nontemporal_store(w, reinterpret_cast<u4*>(reinterpret_cast<unsigned short*>(p) + idx));
  }
};

__device__ __forceinline__ float act_f(float v, int act) {
  switch (act) {
    case ACT_SILU: return fast_silu(v);
    case ACT_ELU: return fast_elu(v);
    case ACT_TANH: return fast_tanh(v);
    case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    default: return v;
  }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics.  grid (chunks, B); a block walks rows [r0, r1) of one item, thread -> fixed
// 8-channel slice (so a fixed group), then LDS-reduces per group and issues one atomic pair per group.
// ---------------------------------------------------------------------------------------------
// 8 values -> 8 OCP e4m3 bytes at byte offset `off` (saturating at +-448: v_cvt_pk_fp8_f32 rounds to nearest even)
__device__ __forceinline__ void store8_fp8(void* y, size_t off, const float* o) {
  float c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = fminf(fmaxf(o[i], -448.0f), 448.0f);
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  *reinterpret_cast<int2*>(reinterpret_cast<char*>(y) + off) = make_int2(lo, hi);
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const void* x, int L, int C, int groups, int rows_per_block,
                                                       float* stats) {
  __shared__ float red[2][64];   // up to 64 groups
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(L, r0 + rows_per_block);
  const int vec_per_row = C / 8;
  const int tid = threadIdx.x;
  if (tid < 64) { red[0][tid] = 0.f; red[1][tid] = 0.f; }
  __syncthreads();
  const int cpg = C / groups;
  // thread -> (row phase, vec): iterate flat over (rows x vec_per_row)
  const int total = (r1 - r0) * vec_per_row;
  float s = 0.f, ss = 0.f;
  int cur_g = -1;
  for (int idx = tid; idx < total; idx += 256) {
    const int r = idx / vec_per_row, v = idx - r * vec_per_row;
    const int g = (v * 8) / cpg;
    float f[8];
    Vec8<T>::load(x, ((size_t)(b * L + r0 + r)) * C + v * 8, f);
    if (cpg >= 8) {
      if (g != cur_g) {
        if (cur_g >= 0) { atomicAdd(&red[0][cur_g], s); atomicAdd(&red[1][cur_g], ss); }
        s = 0.f; ss = 0.f; cur_g = g;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { s += f[i]; ss += f[i] * f[i]; }
    } else {
      // groups narrower than the vector (tiny test widths): per-element
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int gi = (v * 8 + i) / cpg;
        atomicAdd(&red[0][gi], f[i]);
        atomicAdd(&red[1][gi], f[i] * f[i]);
      }
    }
  }
  if (cur_g >= 0) { atomicAdd(&red[0][cur_g], s); atomicAdd(&red[1][cur_g], ss); }
  __syncthreads();
  if (tid < groups) {
    atomicAdd(&stats[((size_t)b * groups + tid) * kGnPad + 0], red[0][tid]);
    atomicAdd(&stats[((size_t)b * groups + tid) * kGnPad + 1], red[1][tid]);
  }
}

hipError_t launch_gn_stats(int dt, const void* x, int B, int L, int C, int groups, float* stats, hipStream_t s) {
  if (groups > 64 || C % 8 || C % groups) return hipErrorInvalidValue;
  // ~64 KB of data per block
  int rows_per_block = (int)std::max<size_t>(1, (64 * 1024) / ((size_t)C * dt_size(dt)));
  // keep the 256-thread flat walk aligned so that a thread stays in one vec column when possible
  dim3 grid((L + rows_per_block - 1) / rows_per_block, B);
  if (dt == DT_F32)
    hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), 0, s, x, L, C, groups, rows_per_block, stats);
  else
    hipLaunchKernelGGL(gn_stats_kernel<__bf16>, grid, dim3(256), 0, s, x, L, C, groups, rows_per_block, stats);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// GroupNorm apply: y = act( ((x-mean)*rstd*gamma+beta) * (scale+1) + shift ) (+ residual)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const void* x, void* y, const void* residual, int B, int L, int C,
                                                       int groups, const float* stats, const float* gamma,
                                                       const float* beta, const float* ss_table, int ss_stride,
                                                       const int* t_ptr, int act) {
  const int vec_per_row = C / 8;
  const size_t total = (size_t)B * L * vec_per_row;
  const int cpg = C / groups;
  const float inv_n = 1.0f / ((float)L * (float)cpg);
  const float* ss = nullptr;
  if (ss_table) ss = ss_table + (size_t)(t_ptr ? *t_ptr : 0) * ss_stride;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const size_t row = idx / vec_per_row;
    const int v = (int)(idx - row * vec_per_row);
    const int b = (int)(row / L);
    float f[8], o[8];
    Vec8<T>::load(x, row * C + v * 8, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = v * 8 + i;
      const int g = c / cpg;
      const float sum = stats[((size_t)b * groups + g) * kGnPad], sq = stats[((size_t)b * groups + g) * kGnPad + 1];
      const float mean = sum * inv_n;
      const float var = fmaxf(sq * inv_n - mean * mean, 0.0f);
      const float rstd = rsqrtf(var + 1e-5f);
      float val = (f[i] - mean) * rstd * gamma[c] + beta[c];
      if (ss) val = val * (ss[c] + 1.0f) + ss[C + c];
      o[i] = act_f(val, act);
    }
    if (residual) {
      float r[8];
      Vec8<T>::load(residual, row * C + v * 8, r);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] += r[i];
    }
    Vec8<T>::store(y, row * C + v * 8, o);
  }
}

// Fast variant: grid (row chunks, B).  The per-channel affine of this item (mean, rstd, gamma, beta and the
// timestep scale/shift folded into a*x+b) is built once per workgroup in LDS; a thread then keeps ONE 8-channel
// column and streams U rows with all loads in flight before the first use.  Needs C/8 to divide 256, C <= 2048.
// LN: additionally write the channel LayerNorm (unet.py:82-101, the PreNorm of the attention block that follows
// a ResnetBlock) of the freshly produced rows to `y_ln`: a row's channels sit in C/8 consecutive threads, so the
// two-pass mean/variance is a shuffle reduction (plus one LDS hop when a row spans two wavefronts, C = 1024).
template <typename T, int ACT, bool LN, int VPR = 0, int U = 8>   // ACT >= 0: compile-time activation; -1: runtime `act`; VPR = C/8 (LN only); U rows per thread
__global__ __launch_bounds__(256) void gn_apply_cols_kernel(const void* x, void* y, const void* residual, int L, int C,
                                                            int groups, int rows_per_block, const float* stats,
                                                            const float* gamma, const float* beta, const float* ss_table,
                                                            int ss_stride, const int* t_ptr, int act, int dbg,
                                                            void* y_ln, const float* ln_g, int out8) {
  __shared__ __attribute__((aligned(16))) float s_a[2048];
  __shared__ __attribute__((aligned(16))) float s_b[2048];
  __shared__ float s_red[LN ? 2 * 4 * 8 : 1];   // [sum | sumsq][wave][q]
  const int vpr = C / 8;
  const int b = blockIdx.y;
  const int v = threadIdx.x % vpr, rph = threadIdx.x / vpr, nph = 256 / vpr;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(L, r0 + rows_per_block);   // rows_per_block == nph * U
  const int rb = r0 + rph;
  // 1. the activation rows do not depend on the statistics: get them moving first
  float f[U][8], rr[U][8];
#pragma unroll
  for (int q = 0; q < U; ++q) {
    const int r = rb + q * nph;
    if (r < r1) {
      const size_t off = ((size_t)b * L + r) * C + v * 8;
      Vec8<T>::load(x, off, f[q]);
      if (residual) Vec8<T>::load(residual, off, rr[q]);
    }
  }
  // 2. per-channel affine of this item, once per workgroup
  const int cpg = C / groups;
  const float inv_n = 1.0f / ((float)L * (float)cpg);
  const float* ss = nullptr;
  if (ss_table) ss = ss_table + (size_t)(t_ptr ? *t_ptr : 0) * ss_stride;
  for (int c = threadIdx.x; c < C; c += 256) {
    if (dbg & 1) { s_a[c] = 1.0f; s_b[c] = 0.5f; continue; }
    const int g = c / cpg;
    const float mean = stats[((size_t)b * groups + g) * kGnPad] * inv_n;
    const float var = fmaxf(stats[((size_t)b * groups + g) * kGnPad + 1] * inv_n - mean * mean, 0.0f);
    const float rstd = rsqrtf(var + 1e-5f);
    float a1 = rstd * gamma[c];
    float b1 = beta[c] - mean * a1;
    if (ss) {
      const float sc = ss[c] + 1.0f;
      a1 *= sc;
      b1 = b1 * sc + ss[C + c];
    }
    s_a[c] = a1;
    s_b[c] = b1;
  }
  __syncthreads();
  float ca[8], cb[8];
  {
    const float4 a0 = *reinterpret_cast<const float4*>(&s_a[v * 8]), a1 = *reinterpret_cast<const float4*>(&s_a[v * 8 + 4]);
    const float4 b0 = *reinterpret_cast<const float4*>(&s_b[v * 8]), b1 = *reinterpret_cast<const float4*>(&s_b[v * 8 + 4]);
    ca[0] = a0.x; ca[1] = a0.y; ca[2] = a0.z; ca[3] = a0.w; ca[4] = a1.x; ca[5] = a1.y; ca[6] = a1.z; ca[7] = a1.w;
    cb[0] = b0.x; cb[1] = b0.y; cb[2] = b0.z; cb[3] = b0.w; cb[4] = b1.x; cb[5] = b1.y; cb[6] = b1.z; cb[7] = b1.w;
  }
  // 3. apply and store
  float lsum[LN ? U : 1];
#pragma unroll
  for (int q = 0; q < U; ++q) {
    const int r = rb + q * nph;
    if (LN) lsum[q] = 0.f;
    if (r < r1) {
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float lin = fmaf(f[q][i], ca[i], cb[i]);
        o[i] = (dbg & 2) ? lin : (ACT == ACT_SILU ? fast_silu(lin) : act_f(lin, act));
      }
      if (residual) {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] += rr[q][i];
      }
      if (out8 & 4) {   // tanh of the block's output (unet.py:467: the final ResnetBlock feeds torch.tanh alone)
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = act_f(o[i], ACT_TANH);
      }
      if (out8 & 1) store8_fp8(y, ((size_t)b * L + r) * C + v * 8, o);
      else if (dbg & 8) Vec8<T>::store_nt(y, ((size_t)b * L + r) * C + v * 8, o);
      else if (!(dbg & 4)) Vec8<T>::store(y, ((size_t)b * L + r) * C + v * 8, o);
      else if (o[0] == 12345.678f) Vec8<T>::store(y, 0, o);
      if (LN) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { f[q][i] = o[i]; lsum[q] += o[i]; }
      }
    }
  }
  if (LN) {
    // row-wise mean and (biased) variance in one sweep: a row = VPR consecutive threads, VPR compile-time
    constexpr int SPAN = VPR < 64 ? VPR : 64;
    const int wave = threadIdx.x >> 6;
    const float inv_c = 1.0f / (float)(VPR * 8);
    float lsq[U], mean[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = fmaf(f[q][i], f[q][i], acc);
      lsq[q] = acc;
    }
#pragma unroll
    for (int o = SPAN >> 1; o > 0; o >>= 1) {
      float t0[U], t1[U];
#pragma unroll
      for (int q = 0; q < U; ++q) { t0[q] = __shfl_xor(lsum[q], o); t1[q] = __shfl_xor(lsq[q], o); }
#pragma unroll
      for (int q = 0; q < U; ++q) { lsum[q] += t0[q]; lsq[q] += t1[q]; }
    }
    if (VPR > 64) {   // a row spans VPR/64 wavefronts
      if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < U; ++q) { s_red[wave * 8 + q] = lsum[q]; s_red[(4 + wave) * 8 + q] = lsq[q]; }
      }
      __syncthreads();
      constexpr int WPR = VPR > 64 ? VPR / 64 : 1;
      const int w0 = (wave / WPR) * WPR;
#pragma unroll
      for (int q = 0; q < U; ++q) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int w = 0; w < WPR; ++w) { a0 += s_red[(w0 + w) * 8 + q]; a1 += s_red[(4 + w0 + w) * 8 + q]; }
        lsum[q] = a0; lsq[q] = a1;
      }
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      mean[q] = lsum[q] * inv_c;
      lsum[q] = fmaxf(lsq[q] * inv_c - mean[q] * mean[q], 0.0f);   // variance
    }
    float g8[8];
    {
      const float4 g0 = *reinterpret_cast<const float4*>(ln_g + v * 8), g1 = *reinterpret_cast<const float4*>(ln_g + v * 8 + 4);
      g8[0] = g0.x; g8[1] = g0.y; g8[2] = g0.z; g8[3] = g0.w; g8[4] = g1.x; g8[5] = g1.y; g8[6] = g1.z; g8[7] = g1.w;
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int r = rb + q * nph;
      if (r < r1) {
        const float rstd = rsqrtf(lsum[q] + 1e-5f);
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (f[q][i] - mean[q]) * rstd * g8[i];
        if (out8 & 2) store8_fp8(y_ln, ((size_t)b * L + r) * C + v * 8, o);
        else Vec8<T>::store(y_ln, ((size_t)b * L + r) * C + v * 8, o);
      }
    }
  }
}

bool gn_apply_ln_fusable(int C) { return C == 256 || C == 512 || C == 1024; }

// Rows per thread: these kernels run as ONE round of workgroups, so a thread's serial instruction stream (64
// elements at U = 8: ~1500 VALU instructions with bf16 packing and SiLU) IS the kernel's duration.  Measured at the
// half-batch level shapes: U = 8 8.3-10.4 us, U = 2 4.6-7.1 us, U = 1 4.0-7.6 us.  Pick the largest U that still
// gives >= 4 workgroups per CU.
static int gn_pick_u(int B, int L, int vpr) {
  const int nph = 256 / vpr;
  for (int u = 8; u > 1; u >>= 1)
    if ((long)B * ((L + nph * u - 1) / (nph * u)) >= 1024) return u;
  return 1;
}

template <typename T, int ACT, bool LN, int VPR>
static void gn_launch_u(int U, dim3 grid, hipStream_t s, const void* x, void* y, const void* residual, int L, int C, int groups, int rpb,
                        const float* stats, const float* gamma, const float* beta, const float* ss_table, int ss_stride,
                        const int* t_ptr, int act, int dbg, void* y_ln, const float* ln_g, int out8) {
#define LDC_GN_GO(UU)                                                                                                     \
  hipLaunchKernelGGL((gn_apply_cols_kernel<T, ACT, LN, VPR, UU>), grid, dim3(256), 0, s, x, y, residual, L, C, groups, rpb, \
                     stats, gamma, beta, ss_table, ss_stride, t_ptr, act, dbg, y_ln, ln_g, out8)
  if (U == 8) LDC_GN_GO(8); else if (U == 4) LDC_GN_GO(4); else if (U == 2) LDC_GN_GO(2); else LDC_GN_GO(1);
#undef LDC_GN_GO
}

hipError_t launch_gn_apply(int dt, const void* x, void* y, const void* residual, int B, int L, int C, int groups,
                           const float* stats, const float* gamma, const float* beta, const float* ss_table,
                           int ss_stride, const int* t_ptr, int act, hipStream_t s, void* y_ln, const float* ln_g, int out8) {
  const int vpr = C / 8;
  if (y_ln && (!gn_apply_ln_fusable(C) || act != ACT_SILU)) return hipErrorInvalidValue;
  if (C % 8 == 0 && vpr <= 256 && 256 % vpr == 0 && C <= 2048) {
    const int U = gn_pick_u(B, L, vpr);
    const int rpb = (256 / vpr) * U;   // one U-row trip per thread
    const int dbg = 0;
    dim3 grid((L + rpb - 1) / rpb, B);
#define LDC_GN_ARGS grid, s, x, y, residual, L, C, groups, rpb, stats, gamma, beta, ss_table, ss_stride, t_ptr, act, dbg, y_ln, ln_g, out8
    if (y_ln) {
      if (dt == DT_F32) {
        if (vpr == 32) gn_launch_u<float, ACT_SILU, true, 32>(U, LDC_GN_ARGS);
        else if (vpr == 64) gn_launch_u<float, ACT_SILU, true, 64>(U, LDC_GN_ARGS);
        else gn_launch_u<float, ACT_SILU, true, 128>(U, LDC_GN_ARGS);
      } else {
        if (vpr == 32) gn_launch_u<__bf16, ACT_SILU, true, 32>(U, LDC_GN_ARGS);
        else if (vpr == 64) gn_launch_u<__bf16, ACT_SILU, true, 64>(U, LDC_GN_ARGS);
        else gn_launch_u<__bf16, ACT_SILU, true, 128>(U, LDC_GN_ARGS);
      }
    } else if (dt == DT_F32 && act == ACT_SILU) gn_launch_u<float, ACT_SILU, false, 0>(U, LDC_GN_ARGS);
    else if (dt == DT_F32) gn_launch_u<float, -1, false, 0>(U, LDC_GN_ARGS);
    else if (act == ACT_SILU) gn_launch_u<__bf16, ACT_SILU, false, 0>(U, LDC_GN_ARGS);
    else gn_launch_u<__bf16, -1, false, 0>(U, LDC_GN_ARGS);
#undef LDC_GN_ARGS
    return hipGetLastError();
  }
  if (out8) return hipErrorInvalidValue;   // the generic form has no fp8 / tanh output (the planner asks for it only where the column form applies)
  const size_t total = (size_t)B * L * (C / 8);
  int blocks = (int)std::min<size_t>((total + 255) / 256, 256 * 8);
  if (blocks < 1) blocks = 1;
  if (dt == DT_F32)
    hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(blocks), dim3(256), 0, s, x, y, residual, B, L, C, groups, stats,
                       gamma, beta, ss_table, ss_stride, t_ptr, act);
  else
    hipLaunchKernelGGL(gn_apply_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, x, y, residual, B, L, C, groups, stats,
                       gamma, beta, ss_table, ss_stride, t_ptr, act);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// channel LayerNorm over each row (one wavefront per row)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ln_rows_kernel(const void* x, void* y, const void* residual, const float* g,
                                                      int rows, int C) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * 256) >> 6;
  const int vec_per_row = C / 8;
  for (int row = wave; row < rows; row += nwaves) {
    float s = 0.f;
    // pass 1: mean
    for (int v = lane; v < vec_per_row; v += 64) {
      float f[8];
      Vec8<T>::load(x, (size_t)row * C + v * 8, f);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += f[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    // pass 2: biased variance about the mean (row is L1/L2 resident)
    float ss = 0.f;
    for (int v = lane; v < vec_per_row; v += 64) {
      float f[8];
      Vec8<T>::load(x, (size_t)row * C + v * 8, f);
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = f[i] - mean; ss += d * d; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float rstd = rsqrtf(ss / (float)C + 1e-5f);
    for (int v = lane; v < vec_per_row; v += 64) {
      float f[8], o8[8];
      Vec8<T>::load(x, (size_t)row * C + v * 8, f);
#pragma unroll
      for (int i = 0; i < 8; ++i) o8[i] = (f[i] - mean) * rstd * g[v * 8 + i];
      if (residual) {
        float r[8];
        Vec8<T>::load(residual, (size_t)row * C + v * 8, r);
#pragma unroll
        for (int i = 0; i < 8; ++i) o8[i] += r[i];
      }
      Vec8<T>::store(y, (size_t)row * C + v * 8, o8);
    }
  }
}

// Row held in registers (C <= 512 * NV): one global read, two-pass statistics on the registers, one write.
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_rows_reg_kernel(const void* x, void* y, const void* residual, const float* g,
                                                          int rows, int C, int out8) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= rows) return;
  const int vec_per_row = C / 8;
  float f[NV][8], r[NV][8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = lane + 64 * k;
    if (v < vec_per_row) {
      Vec8<T>::load(x, (size_t)row * C + v * 8, f[k]);
      if (residual) Vec8<T>::load(residual, (size_t)row * C + v * 8, r[k]);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += f[k][i];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (lane + 64 * k < vec_per_row) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = f[k][i] - mean; ss += d * d; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float rstd = rsqrtf(ss / (float)C + 1e-5f);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = lane + 64 * k;
    if (v < vec_per_row) {
      float o8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o8[i] = (f[k][i] - mean) * rstd * g[v * 8 + i];
      if (residual) {
#pragma unroll
        for (int i = 0; i < 8; ++i) o8[i] += r[k][i];
      }
      if (out8) store8_fp8(y, (size_t)row * C + v * 8, o8);
      else Vec8<T>::store(y, (size_t)row * C + v * 8, o8);
    }
  }
}

hipError_t launch_ln_rows(int dt, const void* x, void* y, const void* residual, const float* g, int rows, int C,
                          hipStream_t s, int out8) {
  if (C % 8 || (out8 && C > 1024)) return hipErrorInvalidValue;
  if (C <= 1024) {
    const int nb = (rows + 3) / 4;
    if (dt == DT_F32) {
      if (C <= 512) hipLaunchKernelGGL((ln_rows_reg_kernel<float, 1>), dim3(nb), dim3(256), 0, s, x, y, residual, g, rows, C, out8);
      else hipLaunchKernelGGL((ln_rows_reg_kernel<float, 2>), dim3(nb), dim3(256), 0, s, x, y, residual, g, rows, C, out8);
    } else {
      if (C <= 512) hipLaunchKernelGGL((ln_rows_reg_kernel<__bf16, 1>), dim3(nb), dim3(256), 0, s, x, y, residual, g, rows, C, out8);
      else hipLaunchKernelGGL((ln_rows_reg_kernel<__bf16, 2>), dim3(nb), dim3(256), 0, s, x, y, residual, g, rows, C, out8);
    }
    return hipGetLastError();
  }
  int blocks = std::min((rows + 3) / 4, 256 * 8);
  if (blocks < 1) blocks = 1;
  if (dt == DT_F32)
    hipLaunchKernelGGL(ln_rows_kernel<float>, dim3(blocks), dim3(256), 0, s, x, y, residual, g, rows, C);
  else
    hipLaunchKernelGGL(ln_rows_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, x, y, residual, g, rows, C);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void act_kernel(const void* x, void* y, size_t nvec, int act, int out8) {
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < nvec; idx += (size_t)gridDim.x * 256) {
    float f[8];
    Vec8<T>::load(x, idx * 8, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = act_f(f[i], act);
    if (out8) store8_fp8(y, idx * 8, f);
    else Vec8<T>::store(y, idx * 8, f);
  }
}

hipError_t launch_act(int dt, const void* x, void* y, int64_t n, int act, hipStream_t s, int out8) {
  if (n % 8) return hipErrorInvalidValue;
  const size_t nvec = (size_t)n / 8;
  int blocks = (int)std::min<size_t>((nvec + 255) / 256, 256 * 8);
  if (blocks < 1) blocks = 1;
  if (dt == DT_F32)
    hipLaunchKernelGGL(act_kernel<float>, dim3(blocks), dim3(256), 0, s, x, y, nvec, act, out8);
  else
    hipLaunchKernelGGL(act_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, x, y, nvec, act, out8);
  return hipGetLastError();
}

}  // namespace ldc
