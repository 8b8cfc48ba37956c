// attention.hip -- LinearAttention core and bottleneck softmax Attention core (4 heads x 32).
//
// LinearAttention.forward (reference srcs/modules/unet.py:208-222), between to_qkv and to_out:
//   q = softmax_d(q) * 32^-1/2 ; k = softmax_n(k) ; context[d,e] = sum_n k[d,n] v[e,n] ;
//   out[e,n] = sum_d context[d,e] q[d,n]
// Attention.forward (:234-246): out = softmax_j((q*scale)^T k) v.
// The 1x1 convs either side run on the conv-GEMM kernel.  These cores are HBM/L2-bound streaming
// passes over the qkv rows ([rows][384], channels-last): linattn reads qkv twice (k statistics,
// then k,v) plus q once and writes [rows][128]; algorithmic bytes ~ rows * (384*2 + 128) * sizeof(dtype).
#include <algorithm>

#include "ldc_kernels.h"
#include "ldc_math.h"

namespace ldc {

__device__ __forceinline__ float abf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short af2bf(float f) { return hw_bf16(f); }
template <typename T>
__device__ __forceinline__ float ld1(const void* p, size_t i);
template <>
__device__ __forceinline__ float ld1<float>(const void* p, size_t i) { return reinterpret_cast<const float*>(p)[i]; }
template <>
__device__ __forceinline__ float ld1<__bf16>(const void* p, size_t i) {
  return abf2f(reinterpret_cast<const unsigned short*>(p)[i]);
}
template <typename T>
__device__ __forceinline__ void st1(void* p, size_t i, float v);
template <>
__device__ __forceinline__ void st1<float>(void* p, size_t i, float v) { reinterpret_cast<float*>(p)[i] = v; }
template <>
__device__ __forceinline__ void st1<__bf16>(void* p, size_t i, float v) {
  reinterpret_cast<unsigned short*>(p)[i] = af2bf(v);
}

// order-preserving float <-> uint key so that atomicMax(unsigned) implements a float max from a zeroed buffer
__device__ __forceinline__ unsigned fkey(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
  const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(b);
}

// ws layout per item b (floats): kmax_key[HD] | ksum[HD] | ctx[H][D][D]
__host__ __device__ inline size_t linattn_ws_per_item(int H, int D) { return (size_t)2 * H * D + (size_t)H * D * D; }

// ---- pass 1: column max of k over positions ----
template <typename T>
__global__ __launch_bounds__(256) void linattn_kmax_kernel(const void* qkv, float* ws, int L, int HD, int rows_per_block,
                                                           size_t ws_stride) {
  __shared__ float red[256];
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(L, r0 + rows_per_block);
  const int col = threadIdx.x % HD, ph = threadIdx.x / HD, nph = 256 / HD;
  float m = -INFINITY;
  for (int r = r0 + ph; r < r1; r += nph) m = fmaxf(m, ld1<T>(qkv, ((size_t)(b * L + r)) * (3 * HD) + HD + col));
  red[threadIdx.x] = m;
  __syncthreads();
  if (ph == 0) {
    for (int p = 1; p < nph; ++p) m = fmaxf(m, red[p * HD + col]);
    atomicMax(reinterpret_cast<unsigned*>(ws + (size_t)b * ws_stride) + col, fkey(m));
  }
}

// ---- pass 2: unnormalised context and column sums over a chunk of rows, atomically merged ----
template <typename T, int D>
__global__ __launch_bounds__(256) void linattn_ctx_kernel(const void* qkv, float* ws, int L, int H, int rows_per_block,
                                                          size_t ws_stride) {
  constexpr int TR = 64;
  __shared__ float sp[TR][D + 1];
  __shared__ float sv[TR][D];
  __shared__ float skmax[D];
  const int HD = H * D;
  const int b = blockIdx.y / H, h = blockIdx.y % H;
  float* wsb = ws + (size_t)b * ws_stride;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(L, r0 + rows_per_block);
  const int tid = threadIdx.x;
  if (tid < D) skmax[tid] = fkey_inv(reinterpret_cast<const unsigned*>(wsb)[h * D + tid]);
  __syncthreads();
  const int d = tid / 8, e0 = (tid % 8) * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float ssum = 0.f;
  for (int t0 = r0; t0 < r1; t0 += TR) {
    const int nr = min(TR, r1 - t0);
    for (int idx = tid; idx < TR * D; idx += 256) {
      const int r = idx / D, c = idx % D;
      float p = 0.f, v = 0.f;
      if (r < nr) {
        const size_t base = ((size_t)(b * L + t0 + r)) * (3 * HD);
        p = __expf(ld1<T>(qkv, base + HD + h * D + c) - skmax[c]);
        v = ld1<T>(qkv, base + 2 * HD + h * D + c);
      }
      sp[r][c] = p;
      sv[r][c] = v;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < TR; ++r) {
      const float p = sp[r][d];
      const float4 v4 = *reinterpret_cast<const float4*>(&sv[r][e0]);
      acc[0] += p * v4.x; acc[1] += p * v4.y; acc[2] += p * v4.z; acc[3] += p * v4.w;
      ssum += p;
    }
    __syncthreads();
  }
  float* ctx = wsb + 2 * HD + (size_t)h * D * D;
#pragma unroll
  for (int j = 0; j < 4; ++j) atomicAdd(&ctx[d * D + e0 + j], acc[j]);
  if ((tid % 8) == 0) atomicAdd(&wsb[HD + h * D + d], ssum);
}

// ---- pass 3: out[n, h*D+e] = sum_d (ctx[d][e]/ksum[d]) * softmax_d(q[n,h,:])[d] * scale ----
template <typename T, int D>
__global__ __launch_bounds__(256) void linattn_out_kernel(const void* qkv, void* out, const float* ws, int L, int H,
                                                          size_t ws_stride, float scale) {
  constexpr int HS = D * D + 8;   // padded head stride: the 4 heads of a wave hit different banks
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sctx = reinterpret_cast<float*>(smem_raw);   // [H][HS]
  const int HD = H * D;
  const int rows_per_block = 256 / H;
  const int b = blockIdx.y;
  const float* wsb = ws + (size_t)b * ws_stride;
  for (int idx = threadIdx.x; idx < H * D * D; idx += 256) {
    const int h = idx / (D * D), rem = idx % (D * D), d = rem / D;
    sctx[h * HS + rem] = wsb[2 * HD + idx] / wsb[HD + h * D + d];
  }
  __syncthreads();
  const int h = threadIdx.x % H;
  const int r = blockIdx.x * rows_per_block + threadIdx.x / H;
  if (r >= L) return;
  const size_t base = ((size_t)(b * L + r)) * (3 * HD) + h * D;
  float q[D];
  float m = -INFINITY;
#pragma unroll
  for (int d = 0; d < D; ++d) { q[d] = ld1<T>(qkv, base + d); m = fmaxf(m, q[d]); }
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) { q[d] = __expf(q[d] - m); s += q[d]; }
  const float inv = scale / s;
#pragma unroll
  for (int d = 0; d < D; ++d) q[d] *= inv;
  const float* c = sctx + h * HS;
  const size_t obase = ((size_t)(b * L + r)) * HD + h * D;
#pragma unroll
  for (int e0 = 0; e0 < D; e0 += 4) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float4 cv = *reinterpret_cast<const float4*>(&c[d * D + e0]);
      o.x += cv.x * q[d]; o.y += cv.y * q[d]; o.z += cv.z * q[d]; o.w += cv.w * q[d];
    }
    st1<T>(out, obase + e0 + 0, o.x); st1<T>(out, obase + e0 + 1, o.y);
    st1<T>(out, obase + e0 + 2, o.z); st1<T>(out, obase + e0 + 3, o.w);
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 engine: the same two passes on the matrix cores (H = 4 heads x D = 32, one wave per head).
//   ctx : context^T-free form  ctx[d][e] = sum_n p[n][d] v[n][e]  with the contraction over positions: the
//         tile's k/v rows arrive as 16-byte row pieces and are written to LDS transposed ([col][row]) so that
//         both MFMA operands are one ds_read_b128 each; p = exp(k - colmax) is rounded to bf16 once and the
//         same rounded values feed the column sums.
//   out : out^T[e][n] = sum_d ctx'[d][e] q'[n][d]; ctx' (= ctx / ksum * scale) is the stationary A operand, the
//         q row pieces are the B operand straight from global memory, the softmax over d is done in registers
//         (a row's 32 values live in lanes l and l^32), the result lands as 4 consecutive channels per lane.
// ---------------------------------------------------------------------------------------------
typedef float lf32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 lbf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ lf32x16 mfma_bf16(const uint4& a, const uint4& b, lf32x16 acc) {
  lbf16x8 va, vb;
  __builtin_memcpy(&va, &a, 16);
  __builtin_memcpy(&vb, &b, 16);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc, 0, 0, 0);
}

template <int R>   // rows per workgroup
__global__ __launch_bounds__(256) void linattn_ctx_mfma_kernel(const unsigned short* qkv, float* ws, int L, size_t ws_stride) {
  constexpr int H = 4, D = 32, HD = 128, P = R + 8;
  __shared__ __attribute__((aligned(16))) unsigned short spT[HD][P];
  __shared__ __attribute__((aligned(16))) unsigned short svT[HD][P];
  __shared__ float skmax[HD];
  const int b = blockIdx.y, tid = threadIdx.x;
  float* wsb = ws + (size_t)b * ws_stride;
  const int r0 = blockIdx.x * R;
  constexpr int NP = R / 16;   // 16-byte pieces per thread and operand
  uint4 kk[NP], vv[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int id = tid + 256 * i, row = id >> 4, c16 = id & 15;
    if (r0 + row < L) {
      const unsigned short* base = qkv + ((size_t)b * L + r0 + row) * (3 * HD) + 8 * c16;
      kk[i] = *reinterpret_cast<const uint4*>(base + HD);
      vv[i] = *reinterpret_cast<const uint4*>(base + 2 * HD);
    } else {
      kk[i] = make_uint4(0, 0, 0, 0); vv[i] = make_uint4(0, 0, 0, 0);
    }
  }
  if (tid < HD) skmax[tid] = fkey_inv(reinterpret_cast<const unsigned*>(wsb)[tid]);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int id = tid + 256 * i, row = id >> 4, c16 = id & 15;
    const bool live = r0 + row < L;
    const unsigned kw[4] = {kk[i].x, kk[i].y, kk[i].z, kk[i].w};
    const unsigned vw[4] = {vv[i].x, vv[i].y, vv[i].z, vv[i].w};
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int col = 8 * c16 + m;
      const unsigned short kb = (unsigned short)((m & 1) ? (kw[m >> 1] >> 16) : (kw[m >> 1] & 0xffffu));
      const unsigned short vb = (unsigned short)((m & 1) ? (vw[m >> 1] >> 16) : (vw[m >> 1] & 0xffffu));
      spT[col][row] = live ? af2bf(__expf(abf2f(kb) - skmax[col])) : (unsigned short)0;
      svT[col][row] = vb;
    }
  }
  __syncthreads();
  const int h = tid >> 6, lane = tid & 63, i32 = lane & 31, g = lane >> 5;
  lf32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int sx = 0; sx < R / 16; ++sx) {
    const uint4 a = *reinterpret_cast<const uint4*>(&spT[h * D + i32][16 * sx + 8 * g]);
    const uint4 bq = *reinterpret_cast<const uint4*>(&svT[h * D + i32][16 * sx + 8 * g]);
    acc = mfma_bf16(a, bq, acc);
  }
  float* ctx = wsb + 2 * HD + (size_t)h * D * D;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = (r & 3) + 8 * (r >> 2) + 4 * g;
    atomicAdd(&ctx[d * D + i32], acc[r]);
  }
  if (tid < HD) {
    float sum = 0.f;
#pragma unroll
    for (int q8 = 0; q8 < R / 8; ++q8) {
      const uint4 u = *reinterpret_cast<const uint4*>(&spT[tid][8 * q8]);
      const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int m = 0; m < 4; ++m) sum += abf2f((unsigned short)(w[m] & 0xffffu)) + abf2f((unsigned short)(w[m] >> 16));
    }
    atomicAdd(&wsb[HD + tid], sum);
  }
}

template <int TILES>   // 32-row tiles per workgroup
__global__ __launch_bounds__(256) void linattn_out_mfma_kernel(const unsigned short* qkv, unsigned short* out, const float* ws,
                                                               int L, size_t ws_stride, float scale) {
  constexpr int D = 32, HD = 128;
  const int b = blockIdx.y, tid = threadIdx.x;
  const int h = tid >> 6, lane = tid & 63, i32 = lane & 31, g = lane >> 5;
  const int r0 = blockIdx.x * (32 * TILES);
  // q row pieces first (independent of the context)
  uint4 qq[TILES][2];
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    const int n = r0 + 32 * t + i32;
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      if (n < L) qq[t][sx] = *reinterpret_cast<const uint4*>(qkv + ((size_t)b * L + n) * (3 * HD) + h * D + 16 * sx + 8 * g);
      else qq[t][sx] = make_uint4(0, 0, 0, 0);
    }
  }
  // stationary operand: A[i = e][k = d] = ctx[d][e] / ksum[d] * scale
  const float* wsb = ws + (size_t)b * ws_stride;
  uint4 af[2];
#pragma unroll
  for (int sx = 0; sx < 2; ++sx) {
    unsigned w[4];
#pragma unroll
    for (int m = 0; m < 8; m += 2) {
      const int d0 = 16 * sx + 8 * g + m;
      const float c0 = wsb[2 * HD + (size_t)h * D * D + d0 * D + i32] * (scale / wsb[HD + h * D + d0]);
      const float c1 = wsb[2 * HD + (size_t)h * D * D + (d0 + 1) * D + i32] * (scale / wsb[HD + h * D + d0 + 1]);
      w[m >> 1] = (unsigned)af2bf(c0) | ((unsigned)af2bf(c1) << 16);
    }
    af[sx] = make_uint4(w[0], w[1], w[2], w[3]);
  }
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    const int n = r0 + 32 * t + i32;
    float x[16];
    float mx = -INFINITY;
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      const unsigned w[4] = {qq[t][sx].x, qq[t][sx].y, qq[t][sx].z, qq[t][sx].w};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        x[8 * sx + 2 * m] = abf2f((unsigned short)(w[m] & 0xffffu));
        x[8 * sx + 2 * m + 1] = abf2f((unsigned short)(w[m] >> 16));
      }
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) mx = fmaxf(mx, x[m]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m) { x[m] = __expf(x[m] - mx); sum += x[m]; }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    uint4 bq[2];
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      unsigned w[4];
#pragma unroll
      for (int m = 0; m < 4; ++m)
        w[m] = (unsigned)af2bf(x[8 * sx + 2 * m] * inv) | ((unsigned)af2bf(x[8 * sx + 2 * m + 1] * inv) << 16);
      bq[sx] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    lf32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = mfma_bf16(af[0], bq[0], acc);
    acc = mfma_bf16(af[1], bq[1], acc);
    if (n < L) {
      unsigned short* orow = out + ((size_t)b * L + n) * HD + h * D;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const unsigned lo = (unsigned)af2bf(acc[4 * rr]) | ((unsigned)af2bf(acc[4 * rr + 1]) << 16);
        const unsigned hi = (unsigned)af2bf(acc[4 * rr + 2]) | ((unsigned)af2bf(acc[4 * rr + 3]) << 16);
        *reinterpret_cast<uint2*>(orow + 8 * rr + 4 * g) = make_uint2(lo, hi);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// LinearAttention tail in one launch (bf16 engine): out = einsum(context, softmax_d(q) * scale), to_out 1x1 conv (128 -> C) +
// bias, channel LayerNorm, + x  (unet.py:216-222 and the Residual around it).  It replaces three launches per attention
// (linattn_out, the to_out conv, ln_rows): at these sizes every one of them is a single round of workgroups that lives on
// its launch boundary and its load/store latencies.
//   phase 1  wave h = head h: the 32 x 32 context (scaled by scale / ksum) is the stationary MFMA operand, the q rows are
//            soft-maxed in registers; the [R][128] bf16 result goes to LDS, where it is the A operand of
//   phase 2  the to_out GEMM: wave w owns C/4 output channels of all R positions, K = 128 in eight MFMA steps, weight
//            fragments straight from the packed conv image ([chunk][n_pad][32 bf16]) in L2, one step ahead;
//   phase 3  bias, LayerNorm over the C channels of a position (two passes as ln_rows: mean, then centred squares; a position's
//            channels sit in 32 lanes x NT tiles x 4 waves -> a 16-value butterfly across lanes, then LDS across waves),
//            gain in fp32 through LDS into position-major pieces, + residual (read coalesced, issued before the reductions),
//            one rounding to bf16, 8-byte row stores.
// The weight fragments of a wave's channels are all requested at kernel entry (they do not depend on phase 1).
// ---------------------------------------------------------------------------------------------
// sum of each of the 16 values over the 32 lanes that share lane>>5; on return every lane holds the total of value
// index (lane & 31) >> 1 (a transposing butterfly: 16 shuffles instead of 80)
__device__ __forceinline__ float reduce16_over32(const float (&v)[16], int lane) {
  float a8[8], a4[4], a2[2];
  const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
  for (int k = 0; k < 8; ++k) a8[k] = (b4 ? v[8 + k] : v[k]) + __shfl_xor(b4 ? v[k] : v[8 + k], 16);
#pragma unroll
  for (int k = 0; k < 4; ++k) a4[k] = (b3 ? a8[4 + k] : a8[k]) + __shfl_xor(b3 ? a8[k] : a8[4 + k], 8);
#pragma unroll
  for (int k = 0; k < 2; ++k) a2[k] = (b2 ? a4[2 + k] : a4[k]) + __shfl_xor(b2 ? a4[k] : a4[2 + k], 4);
  float a1 = (b1 ? a2[1] : a2[0]) + __shfl_xor(b1 ? a2[0] : a2[1], 2);
  a1 += __shfl_xor(a1, 1);
  return a1;
}

template <int C>   // output channels; 32 positions per workgroup
__global__ __launch_bounds__(256) void linattn_tail_mfma_kernel(const unsigned short* qkv, const float* ws, size_t ws_stride, float scale,
                                                                const unsigned short* wo, int n_pad, const float* bias, const float* gain,
                                                                const unsigned short* resid, unsigned short* out, int L) {
  constexpr int D = 32, HD = 128, R = 32, CPW = C / 4, NT = CPW / 32;
  constexpr int AP = HD + 8;                  // sA pitch (bf16 elements): 272 B rows
  constexpr int TP = CPW * 4 + 16;            // transpose pitch (bytes, fp32 values)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* sA = reinterpret_cast<unsigned short*>(smem);                   // [R][AP]
  float* sred = reinterpret_cast<float*>(smem + (size_t)R * AP * 2);              // [4][R] partial sums of a pass
  float* srow = sred + 4 * R;                                                      // [R] mean, then rstd
  char* strans = smem + (size_t)R * AP * 2 + (size_t)5 * R * 4;                   // [4 waves][R][TP]
  const int b = blockIdx.y, tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63, i32 = lane & 31, g = lane >> 5;
  const int r0 = blockIdx.x * R;
  const float* wsb = ws + (size_t)b * ws_stride;
  const int colw = w * CPW + i32;
  // every weight fragment of this wave's channels (independent of phase 1: they land while the softmax runs)
  uint4 bfr[8][NT];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int j = 0; j < NT; ++j)
      bfr[ks][j] = *reinterpret_cast<const uint4*>(wo + ((size_t)(ks >> 1) * n_pad + colw + 32 * j) * 32 + (ks & 1) * 16 + 8 * g);
  float bv[NT], gv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { bv[j] = bias[colw + 32 * j]; gv[j] = gain[colw + 32 * j]; }
  // ---- phase 1: head w ----
  {
    const int h = w;
    const int n = r0 + i32;
    uint4 qq[2];
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      if (n < L) qq[sx] = *reinterpret_cast<const uint4*>(qkv + ((size_t)b * L + n) * (3 * HD) + h * D + 16 * sx + 8 * g);
      else qq[sx] = make_uint4(0, 0, 0, 0);
    }
    uint4 af[2];
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      unsigned wv[4];
#pragma unroll
      for (int m = 0; m < 8; m += 2) {
        const int d0 = 16 * sx + 8 * g + m;
        const float c0 = wsb[2 * HD + (size_t)h * D * D + d0 * D + i32] * (scale / wsb[HD + h * D + d0]);
        const float c1 = wsb[2 * HD + (size_t)h * D * D + (d0 + 1) * D + i32] * (scale / wsb[HD + h * D + d0 + 1]);
        wv[m >> 1] = (unsigned)af2bf(c0) | ((unsigned)af2bf(c1) << 16);
      }
      af[sx] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    }
    float x[16];
    float mx = -INFINITY;
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      const unsigned wv[4] = {qq[sx].x, qq[sx].y, qq[sx].z, qq[sx].w};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        x[8 * sx + 2 * m] = abf2f((unsigned short)(wv[m] & 0xffffu));
        x[8 * sx + 2 * m + 1] = abf2f((unsigned short)(wv[m] >> 16));
      }
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) mx = fmaxf(mx, x[m]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m) { x[m] = __expf(x[m] - mx); sum += x[m]; }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    uint4 bq[2];
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      unsigned wv[4];
#pragma unroll
      for (int m = 0; m < 4; ++m)
        wv[m] = (unsigned)af2bf(x[8 * sx + 2 * m] * inv) | ((unsigned)af2bf(x[8 * sx + 2 * m + 1] * inv) << 16);
      bq[sx] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    }
    lf32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
    acc1 = mfma_bf16(af[0], bq[0], acc1);
    acc1 = mfma_bf16(af[1], bq[1], acc1);
    unsigned short* arow = sA + (size_t)i32 * AP + h * D;   // position i32, channels of head h
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const unsigned lo = (unsigned)af2bf(acc1[4 * rr]) | ((unsigned)af2bf(acc1[4 * rr + 1]) << 16);
      const unsigned hi = (unsigned)af2bf(acc1[4 * rr + 2]) | ((unsigned)af2bf(acc1[4 * rr + 3]) << 16);
      *reinterpret_cast<uint2*>(arow + 8 * rr + 4 * g) = make_uint2(lo, hi);
    }
  }
  __syncthreads();
  // ---- phase 2: to_out GEMM, wave w -> channels [w * CPW, (w + 1) * CPW) ----
  lf32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const uint4 afr = *reinterpret_cast<const uint4*>(sA + (size_t)i32 * AP + 16 * ks + 8 * g);
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = mfma_bf16(afr, bfr[ks][j], acc[j]);
  }
  // the residual rows in the layout of the final sweep (16-byte fp32 pieces of a position's channels -> 8 bytes of bf16):
  // issued now, consumed after the LayerNorm reductions
  constexpr int LPR = CPW / 4;        // lanes per position in the final sweep
  constexpr int RPS = 64 / LPR;       // positions per sweep
  constexpr int NSW = R / RPS;
  const int rsub = lane / LPR, piece = lane % LPR;
  uint2 rres[NSW];
#pragma unroll
  for (int sw = 0; sw < NSW; ++sw) {
    const int n = r0 + sw * RPS + rsub;
    rres[sw] = n < L ? *reinterpret_cast<const uint2*>(resid + ((size_t)b * L + n) * C + w * CPW + piece * 4) : make_uint2(0, 0);
  }
  // ---- phase 3: bias, LayerNorm over channels, gain, residual ----
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] += bv[j];
  const int rr = i32 >> 1;
  const int my_row = (rr & 3) + 8 * (rr >> 2) + 4 * g;   // the position whose total reduce16_over32 leaves in this lane
  {   // pass 1: mean
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float sacc = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) sacc += acc[j][r];
      v[r] = sacc;
    }
    const float tot = reduce16_over32(v, lane);
    if ((lane & 1) == 0) sred[w * R + my_row] = tot;
  }
  __syncthreads();
  if (tid < R) srow[tid] = (sred[tid] + sred[R + tid] + sred[2 * R + tid] + sred[3 * R + tid]) * (1.0f / (float)C);
  __syncthreads();
  float mean[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) mean[r] = srow[(r & 3) + 8 * (r >> 2) + 4 * g];
  {   // pass 2: centred squares
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float sacc = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) { const float d = acc[j][r] - mean[r]; sacc = fmaf(d, d, sacc); }
      v[r] = sacc;
    }
    const float tot = reduce16_over32(v, lane);
    if ((lane & 1) == 0) sred[w * R + my_row] = tot;
  }
  __syncthreads();
  if (tid < R) srow[tid] = rsqrtf((sred[tid] + sred[R + tid] + sred[2 * R + tid] + sred[3 * R + tid]) * (1.0f / (float)C) + 1e-5f);
  __syncthreads();
  char* tw = strans + (size_t)w * R * TP;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
    const float rstd = srow[row];
#pragma unroll
    for (int j = 0; j < NT; ++j)
      reinterpret_cast<float*>(tw + (size_t)row * TP)[32 * j + i32] = (acc[j][r] - mean[r]) * rstd * gv[j];
  }
  // the same wave wrote and reads its transpose region: LDS executes a wave's operations in order
#pragma unroll
  for (int sw = 0; sw < NSW; ++sw) {
    const int row = sw * RPS + rsub;
    const int n = r0 + row;
    const float4 y = *reinterpret_cast<const float4*>(tw + (size_t)row * TP + piece * 16);
    const float o0 = y.x + abf2f((unsigned short)(rres[sw].x & 0xffffu)), o1 = y.y + abf2f((unsigned short)(rres[sw].x >> 16));
    const float o2 = y.z + abf2f((unsigned short)(rres[sw].y & 0xffffu)), o3 = y.w + abf2f((unsigned short)(rres[sw].y >> 16));
    const uint2 o = make_uint2((unsigned)af2bf(o0) | ((unsigned)af2bf(o1) << 16), (unsigned)af2bf(o2) | ((unsigned)af2bf(o3) << 16));
    if (n < L) *reinterpret_cast<uint2*>(out + ((size_t)b * L + n) * C + w * CPW + piece * 4) = o;
  }
}

template <int C>
static hipError_t launch_tail_cfg(const void* qkv, const float* ws, size_t wss, float scale, const void* wo, int n_pad, const float* bias,
                                  const float* gain, const void* resid, void* out, int B, int L, hipStream_t s) {
  constexpr int R = 32, CPW = C / 4;
  const size_t lds = (size_t)R * (128 + 8) * 2 + (size_t)5 * R * 4 + (size_t)4 * R * (CPW * 4 + 16);
  auto kern = linattn_tail_mfma_kernel<C>;
  static bool lds_opt_in = false;   // one-off, outside any stream capture (the first step of a plan runs eagerly)
  if (!lds_opt_in) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    lds_opt_in = true;
  }
  hipLaunchKernelGGL(kern, dim3((L + R - 1) / R, B), dim3(256), lds, s, reinterpret_cast<const unsigned short*>(qkv), ws, wss, scale,
                     reinterpret_cast<const unsigned short*>(wo), n_pad, bias, gain, reinterpret_cast<const unsigned short*>(resid),
                     reinterpret_cast<unsigned short*>(out), L);
  return hipGetLastError();
}

bool linattn_tail_supported(int dt, int heads, int dim_head, int C) {
  return dt == DT_BF16 && heads == 4 && dim_head == 32 && (C == 256 || C == 512 || C == 1024);
}

// context pass only (the k column maxima are already in `ws`: fused into the qkv conv's epilogue)
hipError_t launch_linattn_ctx(int dt, const void* qkv, float* ws, int B, int L, int heads, int dim_head, hipStream_t s) {
  if (dt != DT_BF16 || heads != 4 || dim_head != 32) return hipErrorInvalidValue;
  const size_t wss = linattn_ws_per_item(heads, dim_head);
  hipLaunchKernelGGL(linattn_ctx_mfma_kernel<64>, dim3((L + 63) / 64, B), dim3(256), 0, s, reinterpret_cast<const unsigned short*>(qkv), ws, L, wss);
  return hipGetLastError();
}

// out = LayerNorm_C(to_out(linear-attention output)) * gain + resid, from the qkv rows and the finished context in `ws`
hipError_t launch_linattn_tail(int dt, const void* qkv, const float* ws, const void* wo_packed, int n_pad, const float* bias, const float* gain,
                               const void* resid, void* out, int B, int L, int heads, int dim_head, int C, hipStream_t s) {
  if (!linattn_tail_supported(dt, heads, dim_head, C) || !bias || !gain || !resid) return hipErrorInvalidValue;
  const size_t wss = linattn_ws_per_item(heads, dim_head);
  const float scale = 1.0f / sqrtf((float)dim_head);
  if (C == 256) return launch_tail_cfg<256>(qkv, ws, wss, scale, wo_packed, n_pad, bias, gain, resid, out, B, L, s);
  if (C == 512) return launch_tail_cfg<512>(qkv, ws, wss, scale, wo_packed, n_pad, bias, gain, resid, out, B, L, s);
  return launch_tail_cfg<1024>(qkv, ws, wss, scale, wo_packed, n_pad, bias, gain, resid, out, B, L, s);
}

size_t linattn_ws_floats_per_item(int heads, int dim_head) { return linattn_ws_per_item(heads, dim_head); }

// kmax_fused: the caller zeroed `ws` before the qkv conv and that conv's epilogue already produced the column
// maxima (conv_device.h epilogue_colmax)
hipError_t launch_linattn(int dt, const void* qkv, void* out, float* ws, int B, int L, int heads, int dim_head,
                          bool kmax_fused, hipStream_t s) {
  if (dim_head != 32 || heads * dim_head > 256 || 256 % (heads * dim_head) || 256 % heads) return hipErrorInvalidValue;
  const int HD = heads * dim_head;
  const size_t wss = linattn_ws_per_item(heads, dim_head);
  if (!kmax_fused) {
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)B * wss * sizeof(float), s);
    if (e != hipSuccess) return e;
  }
  const int rpb = 128;
  const int chunks = (L + rpb - 1) / rpb;
  const float scale = 1.0f / sqrtf((float)dim_head);
  const size_t lds_out = (size_t)heads * (dim_head * dim_head + 8) * sizeof(float);
  const int rows_out = 256 / heads;
  if (dt == DT_BF16 && heads == 4) {
    if (!kmax_fused) hipLaunchKernelGGL(linattn_kmax_kernel<__bf16>, dim3(chunks, B), dim3(256), 0, s, qkv, ws, L, HD, rpb, wss);
    hipLaunchKernelGGL(linattn_ctx_mfma_kernel<64>, dim3((L + 63) / 64, B), dim3(256), 0, s,
                       reinterpret_cast<const unsigned short*>(qkv), ws, L, wss);
    hipLaunchKernelGGL(linattn_out_mfma_kernel<2>, dim3((L + 63) / 64, B), dim3(256), 0, s,
                       reinterpret_cast<const unsigned short*>(qkv), reinterpret_cast<unsigned short*>(out), ws, L, wss, scale);
  } else if (dt == DT_F32) {
    if (!kmax_fused) hipLaunchKernelGGL(linattn_kmax_kernel<float>, dim3(chunks, B), dim3(256), 0, s, qkv, ws, L, HD, rpb, wss);
    hipLaunchKernelGGL((linattn_ctx_kernel<float, 32>), dim3(chunks, B * heads), dim3(256), 0, s, qkv, ws, L, heads, rpb, wss);
    hipLaunchKernelGGL((linattn_out_kernel<float, 32>), dim3((L + rows_out - 1) / rows_out, B), dim3(256), lds_out, s, qkv,
                       out, ws, L, heads, wss, scale);
  } else {
    if (!kmax_fused) hipLaunchKernelGGL(linattn_kmax_kernel<__bf16>, dim3(chunks, B), dim3(256), 0, s, qkv, ws, L, HD, rpb, wss);
    hipLaunchKernelGGL((linattn_ctx_kernel<__bf16, 32>), dim3(chunks, B * heads), dim3(256), 0, s, qkv, ws, L, heads, rpb, wss);
    hipLaunchKernelGGL((linattn_out_kernel<__bf16, 32>), dim3((L + rows_out - 1) / rows_out, B), dim3(256), lds_out, s,
                       qkv, out, ws, L, heads, wss, scale);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Full softmax attention at the bottleneck: one workgroup per (item, head); K and V of the head live in
// LDS as fp32 (n <= 512), each thread owns query rows and runs an online softmax over the keys.
// ---------------------------------------------------------------------------------------------
// 64 queries x 4 key-partitions per workgroup: lane 4*i + p walks keys p, p+4, ... of query i with an online
// softmax, the four partial states are merged with two butterfly exchanges, every lane then writes 8 of the 32
// output channels.  (One thread per query was a 75-key serial loop on 75 of 256 threads: 36 us for L = 75.)
// Keys/values pass through LDS in chunks of CH positions, so the sequence length is unbounded (a 35 s utterance
// has n = 1094 at the bottleneck; the reference's softmax attention has no limit either).
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_full_kernel(const void* qkv, void* out, int L, int H, float scale, int CH) {
  constexpr int P = D + 4;   // padded row: the 4 lanes of a query read 4 different keys without bank conflicts
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sk = reinterpret_cast<float*>(smem_raw);   // [CH][P]
  float* sv = sk + (size_t)CH * P;                  // [CH][P]
  const int HD = H * D;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int part = threadIdx.x & 3;
  const int i = blockIdx.y * 64 + (threadIdx.x >> 2);
  const int iq = min(i, L - 1);
  float q[D], acc[D];
  const size_t qb = ((size_t)(b * L + iq)) * (3 * HD) + h * D;
#pragma unroll
  for (int d = 0; d < D; ++d) { q[d] = ld1<T>(qkv, qb + d) * scale; acc[d] = 0.f; }
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < L; k0 += CH) {
    const int nk = min(CH, L - k0);
    if (k0) __syncthreads();   // everyone is done with the previous chunk
    for (int idx = threadIdx.x; idx < nk * (D / 8); idx += 256) {
      const int r = idx / (D / 8), c8 = idx % (D / 8);
      const size_t base = ((size_t)(b * L + k0 + r)) * (3 * HD) + h * D + c8 * 8;
#pragma unroll
      for (int mm = 0; mm < 8; ++mm) {
        sk[r * P + c8 * 8 + mm] = ld1<T>(qkv, base + HD + mm);
        sv[r * P + c8 * 8 + mm] = ld1<T>(qkv, base + 2 * HD + mm);
      }
    }
    __syncthreads();
    for (int j = part; j < nk; j += 4) {
      const float4* kj = reinterpret_cast<const float4*>(sk + (size_t)j * P);
      float sdot = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < D / 4; ++d4) {
        const float4 kv = kj[d4];
        sdot += q[4 * d4] * kv.x + q[4 * d4 + 1] * kv.y + q[4 * d4 + 2] * kv.z + q[4 * d4 + 3] * kv.w;
      }
      const float mn = fmaxf(m, sdot);
      const float corr = __expf(m - mn);
      const float p = __expf(sdot - mn);
      l = l * corr + p;
      const float4* vj = reinterpret_cast<const float4*>(sv + (size_t)j * P);
#pragma unroll
      for (int d4 = 0; d4 < D / 4; ++d4) {
        const float4 vv = vj[d4];
        acc[4 * d4] = acc[4 * d4] * corr + p * vv.x;
        acc[4 * d4 + 1] = acc[4 * d4 + 1] * corr + p * vv.y;
        acc[4 * d4 + 2] = acc[4 * d4 + 2] * corr + p * vv.z;
        acc[4 * d4 + 3] = acc[4 * d4 + 3] * corr + p * vv.w;
      }
      m = mn;
    }
  }
#pragma unroll
  for (int step = 1; step <= 2; step <<= 1) {
    const float mo = __shfl_xor(m, step), lo = __shfl_xor(l, step);
    const float mn = fmaxf(m, mo);
    const float c1 = mn == -INFINITY ? 0.f : __expf(m - mn), c2 = mn == -INFINITY ? 0.f : __expf(mo - mn);
    l = l * c1 + lo * c2;
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = acc[d] * c1 + __shfl_xor(acc[d], step) * c2;
    m = mn;
  }
  if (i < L) {
    const float inv = 1.0f / l;
    const size_t ob = ((size_t)(b * L + i)) * HD + h * D;
#pragma unroll
    for (int d = 0; d < D; ++d)
      if ((d >> 3) == part) st1<T>(out, ob + d, acc[d] * inv);
  }
}

hipError_t launch_attn_full(int dt, const void* qkv, void* out, int B, int L, int heads, int dim_head, hipStream_t s) {
  if (dim_head != 32) return hipErrorInvalidValue;
  const int CH = std::min(L, 256);
  const size_t lds = (size_t)2 * CH * (dim_head + 4) * sizeof(float);
  const float scale = 1.0f / sqrtf((float)dim_head);
  static bool opt_in = false;
  if (!opt_in) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_full_kernel<float, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_full_kernel<__bf16, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    opt_in = true;
  }
  dim3 grid(B * heads, (L + 63) / 64);
  if (dt == DT_F32)
    hipLaunchKernelGGL((attn_full_kernel<float, 32>), grid, dim3(256), lds, s, qkv, out, L, heads, scale, CH);
  else
    hipLaunchKernelGGL((attn_full_kernel<__bf16, 32>), grid, dim3(256), lds, s, qkv, out, L, heads, scale, CH);
  return hipGetLastError();
}

}  // namespace ldc
