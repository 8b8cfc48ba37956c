// diffusion.hip -- layout changes at the ABI boundary, max-abs scaling, the fused p_sample update,
// the device-resident step counter, Philox noise and the output normalisation.
//
//   p_sample arithmetic  : reference srcs/losses/ddpm_loss.py:175-179 (x0 from eps), :237-238 (clamp),
//                          :199-206 (posterior mean), :249-250 (noise unless t == 0)
//   max-abs scaling      : unet.py:401-403 (per item, +1e-20) and sample.py:129 (whole tensor, +1e-8)
//   output normalisation : sample.py:133-134
// All kernels here are HBM-bound: p_sample_update moves 4 fp32 reads/writes + 2 dtype accesses per
// element of [B,128,L].
#include <algorithm>

#include "ldc_kernels.h"
#include "ldc_math.h"

namespace ldc {

__device__ __forceinline__ float dbf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short df2bf(float f) { return hw_bf16(f); }
template <typename T>
__device__ __forceinline__ float dld(const void* p, size_t i);
template <>
__device__ __forceinline__ float dld<float>(const void* p, size_t i) { return reinterpret_cast<const float*>(p)[i]; }
template <>
__device__ __forceinline__ float dld<__bf16>(const void* p, size_t i) {
  return dbf2f(reinterpret_cast<const unsigned short*>(p)[i]);
}
template <typename T>
__device__ __forceinline__ void dst(void* p, size_t i, float v);
template <>
__device__ __forceinline__ void dst<float>(void* p, size_t i, float v) { reinterpret_cast<float*>(p)[i] = v; }
template <>
__device__ __forceinline__ void dst<__bf16>(void* p, size_t i, float v) {
  reinterpret_cast<unsigned short*>(p)[i] = df2bf(v);
}

// ---------------------------------------------------------------------------------------------
// [B][C][L] f32  <->  [B][L][C] dt through a 32x32 LDS tile (both sides coalesced)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void to_cl_kernel(const float* x, void* y, int C, int L, const float* maxabs,
                                                    int per_item, float eps) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, l0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  float den = 1.0f;
  if (maxabs) den = maxabs[per_item ? b : 0] + eps;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, l = l0 + tx;
    tile[i][tx] = (c < C && l < L) ? x[((size_t)b * C + c) * L + l] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int l = l0 + i, c = c0 + tx;
    if (l < L && c < C) {
      float v = tile[tx][i];
      if (maxabs) v = v / den;
      dst<T>(y, ((size_t)b * L + l) * C + c, v);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void from_cl_kernel(const void* x, float* y, int C, int L, const float* maxabs,
                                                      int per_item, float eps) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, l0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  float den = 1.0f;
  if (maxabs) den = maxabs[per_item ? b : 0] + eps;
  for (int i = ty; i < 32; i += 8) {
    const int l = l0 + i, c = c0 + tx;
    tile[i][tx] = (c < C && l < L) ? dld<T>(x, ((size_t)b * L + l) * C + c) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, l = l0 + tx;
    if (l < L && c < C) {
      float v = tile[tx][i];
      if (maxabs) v = v / den;
      y[((size_t)b * C + c) * L + l] = v;
    }
  }
}

hipError_t launch_to_cl(int dt, const float* x, void* y, int B, int C, int L, const float* maxabs, int per_item,
                        float eps, hipStream_t s) {
  dim3 grid((L + 31) / 32, (C + 31) / 32, B);
  if (dt == DT_F32)
    hipLaunchKernelGGL(to_cl_kernel<float>, grid, dim3(256), 0, s, x, y, C, L, maxabs, per_item, eps);
  else
    hipLaunchKernelGGL(to_cl_kernel<__bf16>, grid, dim3(256), 0, s, x, y, C, L, maxabs, per_item, eps);
  return hipGetLastError();
}

hipError_t launch_from_cl(int dt, const void* x, float* y, int B, int C, int L, const float* maxabs, int per_item,
                          float eps, hipStream_t s) {
  dim3 grid((L + 31) / 32, (C + 31) / 32, B);
  if (dt == DT_F32)
    hipLaunchKernelGGL(from_cl_kernel<float>, grid, dim3(256), 0, s, x, y, C, L, maxabs, per_item, eps);
  else
    hipLaunchKernelGGL(from_cl_kernel<__bf16>, grid, dim3(256), 0, s, x, y, C, L, maxabs, per_item, eps);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// max |x| per item (or global).  |x| >= 0, so the raw float bits order like unsigned ints.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxabs_kernel(const void* x, int64_t n_per_item, int per_item, float* maxabs) {
  const int b = blockIdx.y;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per_item; i += (int64_t)gridDim.x * 256)
    m = fmaxf(m, fabsf(dld<T>(x, (size_t)b * n_per_item + i)));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(reinterpret_cast<unsigned*>(maxabs) + (per_item ? b : 0), __float_as_uint(m));
  }
}

hipError_t launch_maxabs(int dt, const void* x, int B, int64_t n_per_item, int per_item, float* maxabs, hipStream_t s) {
  int bx = (int)std::min<int64_t>((n_per_item + 255) / 256, 64);
  if (bx < 1) bx = 1;
  if (dt == DT_F32)
    hipLaunchKernelGGL(maxabs_kernel<float>, dim3(bx, B), dim3(256), 0, s, x, n_per_item, per_item, maxabs);
  else
    hipLaunchKernelGGL(maxabs_kernel<__bf16>, dim3(bx, B), dim3(256), 0, s, x, n_per_item, per_item, maxabs);
  return hipGetLastError();
}

template <typename T>
__global__ __launch_bounds__(256) void scale_by_maxabs_kernel(void* x, int64_t n_per_item, const float* maxabs,
                                                              int per_item, float eps) {
  const int b = blockIdx.y;
  const float den = maxabs[per_item ? b : 0] + eps;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per_item; i += (int64_t)gridDim.x * 256) {
    const size_t idx = (size_t)b * n_per_item + i;
    dst<T>(x, idx, dld<T>(x, idx) / den);
  }
}

hipError_t launch_scale_by_maxabs(int dt, void* x, int B, int64_t n_per_item, const float* maxabs, int per_item,
                                  float eps, hipStream_t s) {
  int bx = (int)std::min<int64_t>((n_per_item + 255) / 256, 256);
  if (bx < 1) bx = 1;
  if (dt == DT_F32)
    hipLaunchKernelGGL(scale_by_maxabs_kernel<float>, dim3(bx, B), dim3(256), 0, s, x, n_per_item, maxabs, per_item, eps);
  else
    hipLaunchKernelGGL(scale_by_maxabs_kernel<__bf16>, dim3(bx, B), dim3(256), 0, s, x, n_per_item, maxabs, per_item, eps);
  return hipGetLastError();
}

// y = x / (maxabs[b] + eps), out of place (--unet_scale_x: the UNet input scaled per item, unet.py:432-433)
template <typename T>
__global__ __launch_bounds__(256) void scale_copy_kernel(const void* x, void* y, int64_t n_per_item, const float* maxabs, float eps) {
  const int b = blockIdx.y;
  const float den = maxabs[b] + eps;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per_item; i += (int64_t)gridDim.x * 256) {
    const size_t idx = (size_t)b * n_per_item + i;
    dst<T>(y, idx, dld<T>(x, idx) / den);
  }
}
hipError_t launch_scale_copy(int dt, const void* x, void* y, int B, int64_t n_per_item, const float* maxabs, float eps, hipStream_t s) {
  int bx = (int)std::min<int64_t>((n_per_item + 255) / 256, 256);
  if (bx < 1) bx = 1;
  if (dt == DT_F32) hipLaunchKernelGGL(scale_copy_kernel<float>, dim3(bx, B), dim3(256), 0, s, x, y, n_per_item, maxabs, eps);
  else hipLaunchKernelGGL(scale_copy_kernel<__bf16>, dim3(bx, B), dim3(256), 0, s, x, y, n_per_item, maxabs, eps);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller: one normal per (seed, step, element)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned (&k)[2]) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
  const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
  const unsigned c0 = (unsigned)(p1 >> 32) ^ c[1] ^ k[0];
  const unsigned c2 = (unsigned)(p0 >> 32) ^ c[3] ^ k[1];
  c[1] = (unsigned)p1; c[3] = (unsigned)p0; c[0] = c0; c[2] = c2;
  k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}
__device__ __forceinline__ float philox_normal(uint64_t seed, unsigned step, uint64_t elem) {
  unsigned c[4] = {(unsigned)elem, (unsigned)(elem >> 32), step, 0x4c444321u};
  unsigned k[2] = {(unsigned)seed, (unsigned)(seed >> 32)};
#pragma unroll
  for (int i = 0; i < 10; ++i) philox_round(c, k);
  const float u1 = ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0,1]
  const float u2 = (float)(c[1] >> 8) * (1.0f / 16777216.0f);            // [0,1)
  return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
}

// four normals from one Philox block: both Box-Muller branches of two (u1, u2) pairs
__device__ __forceinline__ void philox_normal4(uint64_t seed, unsigned step, uint64_t group, float (&z)[4]) {
  unsigned c[4] = {(unsigned)group, (unsigned)(group >> 32), step, 0x4c444323u};
  unsigned k[2] = {(unsigned)seed, (unsigned)(seed >> 32)};
#pragma unroll
  for (int i = 0; i < 10; ++i) philox_round(c, k);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)(c[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0,1]
    const float u2 = (float)(c[2 * h + 1] >> 8) * (1.0f / 16777216.0f);        // [0,1)
    const float r = sqrtf(-2.0f * __logf(u1));
    float sn, cs;
    __sincosf(6.28318530717958647692f * u2, &sn, &cs);
    z[2 * h] = r * cs;
    z[2 * h + 1] = r * sn;
  }
}

// ---------------------------------------------------------------------------------------------
// p_sample update on [B][C][L] fp32 state, eps arriving channels-last; also emits the channels-last
// copy of the new state for the next UNet call.  Tile: 32 positions x 32 channels.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void p_sample_update_kernel(float* x, const void* eps_cl, const float* noise,
                                                              int64_t noise_step_stride, void* x_cl, int C, int L,
                                                              StepTables tb, const int* st, uint64_t elem_base) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, l0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  // the tensor loads do not depend on the step: issue them before the (dependent) step-counter -> schedule-table chain
  // eps tile: read channels-last (coalesced over c), hand over transposed
  float ev[4], xin[4];
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    const int i = ty + ii * 8;
    const int l = l0 + i, c = c0 + tx;
    ev[ii] = (l < L && c < C) ? dld<T>(eps_cl, ((size_t)b * L + l) * C + c) : 0.f;
    const int cc = c0 + i, ll = l0 + tx;
    xin[ii] = (cc < C && ll < L) ? x[((size_t)b * C + cc) * L + ll] : 0.f;
  }
  const int t = st[0], j = st[1];
  const uint64_t seed = ((uint64_t)(unsigned)st[3] << 32) | (uint64_t)(unsigned)st[2];
  const float recip = tb.sqrt_recip_alphas_cumprod[t], recipm1 = tb.sqrt_recipm1_alphas_cumprod[t];
  const float c1 = tb.posterior_mean_coef1[t], c2 = tb.posterior_mean_coef2[t];
  const float sigma = expf(0.5f * tb.posterior_log_variance_clipped[t]);
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) tile[ty + ii * 8][tx] = ev[ii];
  __syncthreads();
  float newv[4];
  // the thread's four elements (channels c0+ty+{0,8,16,24}, one position) share one Philox block; the block index is
  // the global index of the first of them, so the draws do not depend on how the batch is split
  float zz[4] = {0.f, 0.f, 0.f, 0.f};
  if (t > 0 && !noise) philox_normal4(seed, (unsigned)j, elem_base + ((size_t)b * C + c0 + ty) * L + l0 + tx, zz);
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    const int i = ty + ii * 8;
    const int c = c0 + i, l = l0 + tx;
    newv[ii] = 0.f;
    if (c < C && l < L) {
      const size_t idx = ((size_t)b * C + c) * L + l;
      const float xv = xin[ii];
      const float e = tile[tx][i];
      float x0 = recip * xv - recipm1 * e;
      x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
      float v = c1 * x0 + c2 * xv;
      if (t > 0) {
        const float z = noise ? noise[(size_t)j * noise_step_stride + idx] : zz[ii];
        v += sigma * z;
      }
      x[idx] = v;
      newv[ii] = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) tile[ty + ii * 8][tx] = newv[ii];   // tile[c][l]
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int l = l0 + i, c = c0 + tx;
    if (l < L && c < C) dst<T>(x_cl, ((size_t)b * L + l) * C + c, tile[tx][i]);
  }
}

// start images of the alternative samplers: standard normal (p_sample_loop, ddpm_loss.py:257) or uniform [0,1)
// (infilling, ddpm_loss.py:336) from the same counter-based generator; stream `step` keeps them apart from the
// per-step noise draws
__global__ __launch_bounds__(256) void random_fill_kernel(float* x, int64_t n, int uniform, uint64_t seed, unsigned step) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (uniform) {
      unsigned c[4] = {(unsigned)i, (unsigned)((uint64_t)i >> 32), step, 0x4c444322u};
      unsigned k[2] = {(unsigned)seed, (unsigned)(seed >> 32)};
#pragma unroll
      for (int r = 0; r < 10; ++r) philox_round(c, k);
      x[i] = (float)(c[0] >> 8) * (1.0f / 16777216.0f);
    } else {
      x[i] = philox_normal(seed, step, (uint64_t)i);
    }
  }
}
hipError_t launch_random_fill(float* x, int64_t n, int uniform, uint64_t seed, unsigned step, hipStream_t s) {
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(random_fill_kernel, dim3(std::max(1, blocks)), dim3(256), 0, s, x, n, uniform, seed, step);
  return hipGetLastError();
}

// x = a*x + b*y (the blends of infilling, ddpm_loss.py:357,361)
__global__ __launch_bounds__(256) void axpby_kernel(float* x, const float* y, float a, float b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = a * x[i] + b * y[i];
}
hipError_t launch_axpby(float* x, const float* y, float a, float b, int64_t n, hipStream_t s) {
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(axpby_kernel, dim3(std::max(1, blocks)), dim3(256), 0, s, x, y, a, b, n);
  return hipGetLastError();
}

hipError_t launch_p_sample_update(int dt, float* x, const void* eps_cl, const float* noise, int64_t noise_step_stride,
                                  void* x_cl, int B, int C, int L, StepTables tb, const int* st,
                                  uint64_t elem_base, hipStream_t s) {
  dim3 grid((L + 31) / 32, (C + 31) / 32, B);
  if (dt == DT_F32)
    hipLaunchKernelGGL(p_sample_update_kernel<float>, grid, dim3(256), 0, s, x, eps_cl, noise, noise_step_stride, x_cl, C, L,
                       tb, st, elem_base);
  else
    hipLaunchKernelGGL(p_sample_update_kernel<__bf16>, grid, dim3(256), 0, s, x, eps_cl, noise, noise_step_stride, x_cl, C,
                       L, tb, st, elem_base);
  return hipGetLastError();
}

// tl (optional): device-side timeline of the timed (graph-replayed, multi-stream) mode: constant-rate clock (100 MHz,
// s_memrealtime) at the begin and the end of every step of this batch part, slot = the step's iteration index
__global__ void step_advance_kernel(int* st, unsigned long long* tl) {
  if (tl) tl[2 * (st[1] & 2047) + 1] = wall_clock64();
  st[0] -= 1;
  st[1] += 1;
}
__global__ void step_set_kernel(int* st, int t, int j, unsigned key_lo, unsigned key_hi) {
  st[0] = t;
  st[1] = j;
  st[2] = (int)key_lo;
  st[3] = (int)key_hi;
}
// First kernel of a denoise step.  Workgroup 0 owns the part's step state: with `advance` it first moves on from the previous step
// (t - 1, iteration + 1: until round 5 a one-thread launch of its own behind p_sample_update; the timeline's end stamp of that step
// is taken here), counts the UNet-pass epoch up (st[4]: the XCD-team chains tag their tile flags with it, never cleared, never 0)
// and copies the timestep's (scale | shift) row; every other workgroup clears the step's accumulator region (GroupNorm sums,
// split-K counters, k-max keys: until round 3 a memset node of its own): zero_n16 16-byte pieces starting at `zero`.
__global__ __launch_bounds__(1024) void step_begin_kernel(const float* table, int stride, int* st, float* cur,
                                                          unsigned long long* tl, uint4* zero, long long zero_n16, int advance) {
  constexpr int nthr = 1024;
  if (blockIdx.x == 0) {
    __shared__ int sh_t;
    if (threadIdx.x == 0) {
      int t = st[0], j = st[1];
      const int epoch = st[4];
      const unsigned long long now = tl ? wall_clock64() : 0ull;
      if (advance) {
        if (tl && j >= 0) tl[2 * (j & 2047) + 1] = now;
        t -= 1;
        j += 1;
        st[0] = t;
        st[1] = j;
      }
      if (tl) tl[2 * (j & 2047)] = now;
      st[4] = epoch + 1;
      sh_t = t;
    }
    __syncthreads();
    const float* row = table + (size_t)max(sh_t, 0) * stride;
    const int n4 = stride >> 2;
    if ((stride & 3) == 0 && n4 <= 8 * nthr) {   // eight 16-byte loads in flight per thread: one memory latency for a 100 KB row, not one per element
      const float4* r4 = reinterpret_cast<const float4*>(row);
      float4* c4 = reinterpret_cast<float4*>(cur);
      const int i0 = threadIdx.x;
      const float4 v0 = r4[min(i0, n4 - 1)], v1 = r4[min(i0 + nthr, n4 - 1)], v2 = r4[min(i0 + 2 * nthr, n4 - 1)], v3 = r4[min(i0 + 3 * nthr, n4 - 1)];
      const float4 v4 = r4[min(i0 + 4 * nthr, n4 - 1)], v5 = r4[min(i0 + 5 * nthr, n4 - 1)], v6 = r4[min(i0 + 6 * nthr, n4 - 1)], v7 = r4[min(i0 + 7 * nthr, n4 - 1)];
      if (i0 < n4) c4[i0] = v0;
      if (i0 + nthr < n4) c4[i0 + nthr] = v1;
      if (i0 + 2 * nthr < n4) c4[i0 + 2 * nthr] = v2;
      if (i0 + 3 * nthr < n4) c4[i0 + 3 * nthr] = v3;
      if (i0 + 4 * nthr < n4) c4[i0 + 4 * nthr] = v4;
      if (i0 + 5 * nthr < n4) c4[i0 + 5 * nthr] = v5;
      if (i0 + 6 * nthr < n4) c4[i0 + 6 * nthr] = v6;
      if (i0 + 7 * nthr < n4) c4[i0 + 7 * nthr] = v7;
    } else {
      for (int i = threadIdx.x; i < stride; i += nthr) cur[i] = row[i];
    }
    if (gridDim.x > 1) return;
  }
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  const long long nb = gridDim.x > 1 ? gridDim.x - 1 : 1, bi = gridDim.x > 1 ? blockIdx.x - 1 : 0;
  for (long long i = bi * nthr + threadIdx.x; i < zero_n16; i += nb * nthr) zero[i] = z;
}
hipError_t launch_step_begin(const float* table, int stride, int* st, float* cur, unsigned long long* tl, hipStream_t s,
                             void* zero, size_t zero_bytes, int advance) {
  const long long n16 = (long long)((zero_bytes + 15) / 16);
  const long long want = n16 > 0 ? 1 + (n16 + 4095) / 4096 : 1;
  hipLaunchKernelGGL(step_begin_kernel, dim3((unsigned)std::min<long long>(256, want)), dim3(1024), 0, s, table, stride, st, cur, tl,
                     reinterpret_cast<uint4*>(zero), n16, advance);
  return hipGetLastError();
}
// one workgroup that holds its CU slot for `us` microseconds of the 100 MHz wall clock (bounded): the stream-overlap calibration of ldc_api.cpp
__global__ void spin_us_kernel(unsigned us) {
  const unsigned long long t0 = wall_clock64();
  int guard = 0;
  while (wall_clock64() - t0 < (unsigned long long)us * 100ull && guard < (1 << 20)) ++guard;
}
hipError_t launch_spin_us(unsigned us, hipStream_t s) {
  hipLaunchKernelGGL(spin_us_kernel, dim3(1), dim3(64), 0, s, us);
  return hipGetLastError();
}
__global__ void clock_sample_kernel(unsigned long long* out) {
  out[0] = wall_clock64();
  out[1] = __builtin_amdgcn_s_memtime();
}
hipError_t launch_clock_sample(unsigned long long* out2, hipStream_t s) {
  hipLaunchKernelGGL(clock_sample_kernel, dim3(1), dim3(1), 0, s, out2);
  return hipGetLastError();
}
hipError_t launch_step_advance(int* st, unsigned long long* tl, hipStream_t s) {
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, st, tl);
  return hipGetLastError();
}
hipError_t launch_step_set(int* st, int t, int j, uint64_t noise_key, hipStream_t s) {
  hipLaunchKernelGGL(step_set_kernel, dim3(1), dim3(1), 0, s, st, t, j, (unsigned)noise_key, (unsigned)(noise_key >> 32));
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// output normalisation: x /= std(x)+1e-8 (unbiased) ; x /= max|x|+1e-8     (sample.py:133-134)
// ws per slot: double sum, double sumsq, then float maxabs array after the doubles
// ---------------------------------------------------------------------------------------------
size_t output_normalise_ws_bytes(int B) { return (size_t)B * (2 * sizeof(double) + sizeof(float)) + 16; }

__global__ __launch_bounds__(256) void outnorm_reduce_kernel(const float* x, int64_t n_per_item, int per_item,
                                                             double* sums, float* maxabs) {
  const int b = blockIdx.y;
  double s = 0.0, ss = 0.0;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per_item; i += (int64_t)gridDim.x * 256) {
    const float v = x[(size_t)b * n_per_item + i];
    s += v; ss += (double)v * v; m = fmaxf(m, fabsf(v));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); m = fmaxf(m, __shfl_xor(m, o));
  }
  __shared__ double rs[4], rss[4];
  __shared__ float rm[4];
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rss[threadIdx.x >> 6] = ss; rm[threadIdx.x >> 6] = m; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int slot = per_item ? b : 0;
    atomicAdd(&sums[2 * slot], rs[0] + rs[1] + rs[2] + rs[3]);
    atomicAdd(&sums[2 * slot + 1], rss[0] + rss[1] + rss[2] + rss[3]);
    atomicMax(reinterpret_cast<unsigned*>(maxabs) + slot, __float_as_uint(fmaxf(fmaxf(rm[0], rm[1]), fmaxf(rm[2], rm[3]))));
  }
}

__global__ __launch_bounds__(256) void outnorm_apply_kernel(float* x, int64_t n_per_item, int per_item, int B,
                                                            const double* sums, const float* maxabs) {
  const int b = blockIdx.y;
  const int slot = per_item ? b : 0;
  const double n = per_item ? (double)n_per_item : (double)n_per_item * B;
  const double mean = sums[2 * slot] / n;
  double var = (sums[2 * slot + 1] - n * mean * mean) / (n - 1.0);
  if (var < 0.0) var = 0.0;
  const float sd = (float)sqrt(var) + 1e-8f;
  const float mx = maxabs[slot] / sd + 1e-8f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per_item; i += (int64_t)gridDim.x * 256) {
    const size_t idx = (size_t)b * n_per_item + i;
    x[idx] = (x[idx] / sd) / mx;
  }
}

hipError_t launch_output_normalise(float* x, int B, int64_t n_per_item, int per_item, void* ws, hipStream_t s) {
  hipError_t e = hipMemsetAsync(ws, 0, output_normalise_ws_bytes(B), s);
  if (e != hipSuccess) return e;
  double* sums = reinterpret_cast<double*>(ws);
  float* maxabs = reinterpret_cast<float*>(sums + 2 * B);
  int bx = (int)std::min<int64_t>((n_per_item + 255) / 256, 64);
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(outnorm_reduce_kernel, dim3(bx, B), dim3(256), 0, s, x, n_per_item, per_item, sums, maxabs);
  hipLaunchKernelGGL(outnorm_apply_kernel, dim3(bx, B), dim3(256), 0, s, x, n_per_item, per_item, B, sums, maxabs);
  return hipGetLastError();
}

}  // namespace ldc

// ---------------------------------------------------------------------------------------------
// Front end: torchaudio.functional.resample (sinc_interp_hann), the call at srcs/sample.py:84.  Polyphase form: output sample
// i * new + p = sum_k kernel[p][k] * padded[i * orig + k] with the [new][2 width + orig] filter bank built on the host in
// float64 (torchaudio 0.13 _get_sinc_resample_kernel) -- one thread per output sample, HBM/L2-bound streaming.
// ---------------------------------------------------------------------------------------------
namespace ldc {
__global__ __launch_bounds__(256) void resample_kernel(const float* wav, int64_t T, const float* bank, int orig, int nnew, int width,
                                                       int K, int64_t target, float* out) {
  const int c = blockIdx.y;
  const float* x = wav + (size_t)c * T;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < target; j += (int64_t)gridDim.x * 256) {
    const int64_t i = j / nnew;
    const int p = (int)(j - i * nnew);
    const float* kr = bank + (size_t)p * K;
    const int64_t s0 = i * orig - width;          // index into the unpadded signal of tap 0
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      const int64_t s = s0 + k;
      if (s >= 0 && s < T) acc = fmaf(x[s], kr[k], acc);
    }
    out[(size_t)c * target + j] = acc;
  }
}
hipError_t launch_resample(const float* wav, int C, int64_t T, const float* bank, int orig, int nnew, int width, int64_t target, float* out,
                           hipStream_t s) {
  if (target <= 0 || C <= 0) return hipSuccess;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)std::min<int64_t>((target + 255) / 256, 2048), C), dim3(256), 0, s, wav, T, bank, orig,
                     nnew, width, 2 * width + orig, target, out);
  return hipGetLastError();
}
}  // namespace ldc
