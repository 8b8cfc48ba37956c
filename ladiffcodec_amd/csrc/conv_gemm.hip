// conv_gemm.hip -- implicit-GEMM Conv1d / ConvTranspose1d for gfx950 (CDNA4) on MFMA.
//
// Replaces every F.conv1d / F.conv_transpose1d the reference issues on the decode path
// (WeightStandardizedConv2d unet.py:72-80, nn.Conv1d unet.py:61,65,201-204,231-232,307,369,
//  SConv1d conv.py:217-232, SConvTranspose1d conv.py:252-274) with one kernel family.
//
// Roofline: dense contraction, MFMA-bound (bf16: v_mfma_f32_32x32x16_bf16, ~2.5 PFLOP/s dense;
// f32: v_mfma_f32_32x32x2_f32, exact fp32 at 157 TFLOP/s).  Algorithmic flops per launch =
// 2 * rows * N * taps * Cin.
//
// Structure per workgroup (4 wavefronts of 64 lanes), tile BM x BN of the [rows x N] output:
//   for each 64-byte channel chunk (32 bf16 / 16 f32 channels) of the (concatenated) input:
//     - stage the WINDOW of input rows the tile's positions touch for all taps into LDS once
//       (rows are channels-last, so a tap is just a row offset; padding / reflect / x2 nearest
//       upsampling / striding are row-index arithmetic, a dedicated all-zero LDS row serves
//       out-of-range taps),
//     - stage the packed weight slab [tap][BN][64 B] (contiguous in HBM) into LDS,
//     - every wave issues ds_read_b128 fragments + MFMAs for its (TM x TN) 32x32 sub-tiles.
//   LDS rows are padded 64 -> 80 bytes: the 16 lanes of a ds_read_b128 group then start on 16
//   distinct 4-bank slots (conflict-free, see MI355X LDS banking).
//   Both MFMA operands use the same (lane>>5, element) -> k mapping, so the K order inside a chunk
//   is irrelevant; only the C/D layout (col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)) matters.
#include "ldc_kernels.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace ldc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static constexpr int kRowBytes = 64;   // K bytes per LDS row per chunk
static constexpr int kPitch = 80;      // padded LDS row pitch

struct ConvKArgs {
  const char* x1;
  const char* x2;
  const char* w;
  const float* bias;
  char* y;
  const char* residual;
  int C1, C2;          // channels
  int n, n_pad;
  int B, L_in, L_rows, L_final, y_ld;
  int taps, stride, dil, pad_left, ups, pad_mode, pre_act, post_act;
  int tr_stride, tr_cout, tr_trim_left;
  int win_rows;        // LDS window capacity (rows), zero row lives at index win_rows
  int tg;              // taps staged per weight slab
  int reflect_back, reflect_fwd;
};

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                                  // round to nearest even
  return (unsigned short)(u >> 16);
}

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case ACT_SILU: return v / (1.0f + __expf(-v));
    case ACT_ELU: return v > 0.0f ? v : (expm1f(v));
    case ACT_TANH: return tanhf(v);
    case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    default: return v;
  }
}

template <typename T>
__device__ __forceinline__ uint4 elu16(uint4 v);
template <>
__device__ __forceinline__ uint4 elu16<float>(uint4 v) {
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = f[i] > 0.0f ? f[i] : expm1f(f[i]);
  return v;
}
template <>
__device__ __forceinline__ uint4 elu16<__bf16>(uint4 v) {
  unsigned* u = reinterpret_cast<unsigned*>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float lo = bf16_to_f32((unsigned short)(u[i] & 0xffffu));
    float hi = bf16_to_f32((unsigned short)(u[i] >> 16));
    lo = lo > 0.0f ? lo : expm1f(lo);
    hi = hi > 0.0f ? hi : expm1f(hi);
    u[i] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
  }
  return v;
}

template <typename T>
__device__ __forceinline__ void mfma_step(f32x16& acc, const uint4& a, const uint4& b);
template <>
__device__ __forceinline__ void mfma_step<float>(f32x16& acc, const uint4& a, const uint4& b) {
  const float* fa = reinterpret_cast<const float*>(&a);
  const float* fb = reinterpret_cast<const float*>(&b);
#pragma unroll
  for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mfma_step<__bf16>(f32x16& acc, const uint4& a, const uint4& b) {
  bf16x8 va, vb;
  __builtin_memcpy(&va, &a, 16);
  __builtin_memcpy(&vb, &b, 16);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ void store_out(char* y, size_t idx, float v);
template <>
__device__ __forceinline__ void store_out<float>(char* y, size_t idx, float v) { reinterpret_cast<float*>(y)[idx] = v; }
template <>
__device__ __forceinline__ void store_out<__bf16>(char* y, size_t idx, float v) {
  reinterpret_cast<unsigned short*>(y)[idx] = f32_to_bf16(v);
}
template <typename T>
__device__ __forceinline__ float load_in(const char* p, size_t idx);
template <>
__device__ __forceinline__ float load_in<float>(const char* p, size_t idx) { return reinterpret_cast<const float*>(p)[idx]; }
template <>
__device__ __forceinline__ float load_in<__bf16>(const char* p, size_t idx) {
  return bf16_to_f32(reinterpret_cast<const unsigned short*>(p)[idx]);
}

// flat input row reached from GEMM row m with tap offset `toff` (= tap*dil); -1 when it is padding (zero mode)
__device__ __forceinline__ int gather_row(const ConvKArgs& a, int b, int l, int toff) {
  int u = l * a.stride + toff - a.pad_left;
  const int leff = a.L_in << a.ups;
  if (a.pad_mode == PAD_REFLECT) {
    if (u < 0) u = -u;
    if (u >= leff) u = 2 * (leff - 1) - u;
    if (u < 0 || u >= leff) return -1;
  } else {
    if (u < 0 || u >= leff) return -1;
  }
  return b * a.L_in + (u >> a.ups);
}

template <typename T, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM* WN * 64) void conv_gemm_kernel(const ConvKArgs a) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  constexpr int EPV = 16 / (int)sizeof(T);        // elements per 16-byte vector
  constexpr int BKE = kRowBytes / (int)sizeof(T);  // channels per chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + (size_t)(a.win_rows + 1) * kPitch;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int M = a.B * a.L_rows;
  const int total_in_rows = a.B * a.L_in;
  const int leff = a.L_in << a.ups;

  // ---- window of flat input rows touched by this tile (uniform) ----
  int R_lo, R_hi;
  {
    const int m_last = min(m0 + BM, M) - 1;
    int b = m0 / a.L_rows, l = m0 - b * a.L_rows;
    int u = l * a.stride - a.pad_left;
    u = max(0, min(u, leff - 1));
    R_lo = b * a.L_in + (u >> a.ups) - a.reflect_back;
    b = m_last / a.L_rows;
    l = m_last - b * a.L_rows;
    u = l * a.stride + (a.taps - 1) * a.dil - a.pad_left;
    u = max(0, min(u, leff - 1));
    R_hi = b * a.L_in + (u >> a.ups) + a.reflect_fwd;
    R_lo = max(R_lo, 0);
    R_hi = min(R_hi, total_in_rows - 1);
  }
  const int nrows = min(R_hi - R_lo + 1, a.win_rows);
  const int zero_row = a.win_rows;
  if (tid < kPitch / 4) reinterpret_cast<unsigned*>(sA + (size_t)zero_row * kPitch)[tid] = 0u;

  // ---- per-lane GEMM rows of the A fragments ----
  int row_b[TM], row_l[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + (wm * TM + i) * 32 + (lane & 31);
    if (m < M) {
      row_b[i] = m / a.L_rows;
      row_l[i] = m - row_b[i] * a.L_rows;
    } else {
      row_b[i] = -1;
      row_l[i] = 0;
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nchunks = (a.C1 + a.C2) / BKE;
  const int khalf = (lane >> 5) * 16;
  for (int c = 0; c < nchunks; ++c) {
    const char* src;
    int ld, coff;
    if (c * BKE < a.C1) {
      src = a.x1; ld = a.C1; coff = c * BKE;
    } else {
      src = a.x2; ld = a.C2; coff = c * BKE - a.C1;
    }
    __syncthreads();   // previous chunk's fragments are consumed
    for (int idx = tid; idx < nrows * 4; idx += NT) {
      const int r = idx >> 2, p = idx & 3;
      const size_t g = ((size_t)(R_lo + r) * ld + coff + p * EPV) * sizeof(T);
      uint4 v = *reinterpret_cast<const uint4*>(src + g);
      if (a.pre_act == ACT_ELU) v = elu16<T>(v);
      *reinterpret_cast<uint4*>(sA + (size_t)r * kPitch + p * 16) = v;
    }
    for (int tg0 = 0; tg0 < a.taps; tg0 += a.tg) {
      const int ntg = min(a.tg, a.taps - tg0);
      if (tg0 > 0) __syncthreads();
      {
        const char* wsrc = a.w + ((size_t)(c * a.taps + tg0) * a.n_pad) * kRowBytes;
        for (int idx = tid; idx < ntg * BN * 4; idx += NT) {
          const int t = idx / (BN * 4);
          const int rem = idx - t * (BN * 4);
          const int r = rem >> 2, p = rem & 3;
          const uint4 v = *reinterpret_cast<const uint4*>(wsrc + ((size_t)t * a.n_pad + n0 + r) * kRowBytes + p * 16);
          *reinterpret_cast<uint4*>(sB + (size_t)(t * BN + r) * kPitch + p * 16) = v;
        }
      }
      __syncthreads();
      for (int t = 0; t < ntg; ++t) {
        const int toff = (tg0 + t) * a.dil;
        int aoff[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          int row = zero_row;
          if (row_b[i] >= 0) {
            const int g = gather_row(a, row_b[i], row_l[i], toff);
            if (g >= 0) row = g - R_lo;
          }
          aoff[i] = row * kPitch + khalf;
        }
        const int boff = (t * BN + wn * TN * 32 + (lane & 31)) * kPitch + khalf;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          uint4 af[TM], bfr[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const uint4*>(sA + aoff[i] + ks * 32);
#pragma unroll
          for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const uint4*>(sB + boff + j * 32 * kPitch + ks * 32);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) mfma_step<T>(acc[i][j], af[i], bfr[j]);
        }
      }
    }
  }

  // ---- epilogue: bias (+ residual), activation, store ----
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
    if (col >= a.n) continue;
    const float bv = a.bias ? a.bias[col] : 0.0f;
    int tr_p = 0, tr_co = 0;
    if (a.tr_stride) {
      tr_p = col / a.tr_cout;
      tr_co = col - tr_p * a.tr_cout;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= M) continue;
        float v = acc[i][j][r] + bv;
        if (a.tr_stride) {
          const int b = m / a.L_rows, q = m - b * a.L_rows;
          const int pos = q * a.tr_stride + tr_p - a.tr_trim_left;
          if (pos < 0 || pos >= a.L_final) continue;
          v = act_apply(v, a.post_act);
          store_out<T>(a.y, ((size_t)b * a.L_final + pos) * a.y_ld + tr_co, v);
        } else {
          if (a.residual) v += load_in<T>(a.residual, (size_t)m * a.n + col);
          v = act_apply(v, a.post_act);
          store_out<T>(a.y, (size_t)m * a.y_ld + col, v);
        }
      }
    }
  }
}

// bias (+ residual) (+ activation) and store of a wave's TM x TN accumulators (plain conv, row-major [M][n])
template <typename T, int TM, int TN, bool RES, bool ACT>
__device__ __forceinline__ void epilogue_plain(const ConvKArgs& a, f32x16 (&acc)[TM][TN], int mrow0, int col0, int M) {
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + j * 32;
    const bool col_ok = col < a.n;
    const float bv = (a.bias && col_ok) ? a.bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = mrow0 + i * 32;
      float rv[16];
      if (RES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          rv[r] = (col_ok && m < M) ? load_in<T>(a.residual, (size_t)m * a.n + col) : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        float v = acc[i][j][r] + bv;
        if (RES) v += rv[r];
        if (ACT) v = act_apply(v, a.post_act);
        if (col_ok && m < M) store_out<T>(a.y, (size_t)m * a.y_ld + col, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// v2 fast path: same tiling, but the HBM/L2 -> LDS copies are LDS-DMA (`global_load_lds_dwordx4`,
// no VGPR round trip) into a 2-stage LDS ring, issued one pipeline unit ahead of the MFMAs that
// consume them, so the copy of unit u+1 overlaps the matrix work of unit u.  LDS-DMA writes
// lane-linear (wave base + lane*16 B), so rows cannot be padded; bank conflicts are avoided with an
// XOR swizzle of the 16-byte slot index, slot' = slot ^ ((row >> 2) & 3), applied on the per-lane
// SOURCE address and again on the ds_read address (rows distinct mod 16 -> 16 distinct bank slots).
// Eligible: zero padding, no prologue activation, plain (non-transposed) conv -- every UNet conv.
// ------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// one 16-byte-per-lane LDS-DMA: lane l's 16 bytes at `gsrc` land at LDS byte address lds_addr + 16*l.
// M0 carries the (wave-uniform) LDS address; it is saved/restored around the instruction because hipcc
// does not model M0 inside an asm statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

struct ConvV2Geom {
  int a_rows;       // LDS rows reserved for the input window (multiple of 16 * waves); zero row follows
  int b_rows;       // LDS rows of the weight slab per unit (multiple of 16 * waves)
  int stage_bytes;
  int ngroups;      // tap groups per channel chunk
};

template <typename T, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM* WN * 64) void conv_gemm_v2_kernel(const ConvKArgs a, const ConvV2Geom gm) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NW = WM * WN;
  constexpr int BKE = kRowBytes / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int M = a.B * a.L_rows;
  const int total_in_rows = a.B * a.L_in;
  const int leff = a.L_in << a.ups;

  int R_lo, R_hi;
  {
    const int m_last = min(m0 + BM, M) - 1;
    int b = m0 / a.L_rows, l = m0 - b * a.L_rows;
    int u = l * a.stride - a.pad_left;
    u = max(0, min(u, leff - 1));
    R_lo = b * a.L_in + (u >> a.ups);
    b = m_last / a.L_rows;
    l = m_last - b * a.L_rows;
    u = l * a.stride + (a.taps - 1) * a.dil - a.pad_left;
    u = max(0, min(u, leff - 1));
    R_hi = min(b * a.L_in + (u >> a.ups), total_in_rows - 1);
  }
  const int nrows = min(R_hi - R_lo + 1, gm.a_rows);
  const int zero_row = gm.a_rows;
  if (tid < 32) {
    reinterpret_cast<unsigned*>(smem + (size_t)zero_row * kRowBytes)[tid & 15] = 0u;
    reinterpret_cast<unsigned*>(smem + gm.stage_bytes + (size_t)zero_row * kRowBytes)[tid & 15] = 0u;
  }

  int row_b[TM], row_l[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + (wm * TM + i) * 32 + (lane & 31);
    if (m < M) {
      row_b[i] = m / a.L_rows;
      row_l[i] = m - row_b[i] * a.L_rows;
    } else {
      row_b[i] = -1;
      row_l[i] = 0;
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nchunks = (a.C1 + a.C2) / BKE;
  const int nunits = nchunks * gm.ngroups;
  const int nA = gm.a_rows / (16 * NW), nB = gm.b_rows / (16 * NW);
  const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;

  auto load_unit = [&](int u, int st) {
    const int c = u / gm.ngroups, g = u - c * gm.ngroups;
    const char* src;
    int ld, coff;
    if (c * BKE < a.C1) {
      src = a.x1; ld = a.C1; coff = c * BKE;
    } else {
      src = a.x2; ld = a.C2; coff = c * BKE - a.C1;
    }
    // input window: LDS slot p = row*4 + slot'
    for (int i = 0; i < nA; ++i) {
      const int pbase = (i * NW + wave) * 64;
      const int p = pbase + lane;
      int r = p >> 2;
      const int s = (p & 3) ^ ((r >> 2) & 3);
      if (r >= nrows) r = 0;
      const char* g_addr = src + ((size_t)(R_lo + r) * ld + coff) * sizeof(T) + s * 16;
      lds_dma16(g_addr, lds_base + (unsigned)(st * gm.stage_bytes + pbase * 16));
    }
    // weight slab of tap group g
    const int tg0 = g * a.tg;
    const int ntg = min(a.tg, a.taps - tg0);
    const char* wsrc = a.w + ((size_t)(c * a.taps + tg0) * a.n_pad + n0) * kRowBytes;
    for (int i = 0; i < nB; ++i) {
      const int pbase = (i * NW + wave) * 64;
      const int p = pbase + lane;
      const int rb = p >> 2;
      const int s = (p & 3) ^ ((rb >> 2) & 3);
      int t = rb / BN;
      const int j = rb - t * BN;
      if (t >= ntg) t = 0;
      const char* g_addr = wsrc + ((size_t)t * a.n_pad + j) * kRowBytes + s * 16;
      lds_dma16(g_addr, lds_base + (unsigned)(st * gm.stage_bytes + (gm.a_rows + 2) * kRowBytes + pbase * 16));
    }
  };

  load_unit(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int kh = lane >> 5;
  for (int u = 0; u < nunits; ++u) {
    const int st = u & 1;
    if (u + 1 < nunits) load_unit(u + 1, st ^ 1);
    const int g = u % gm.ngroups;
    const int tg0 = g * a.tg;
    const int ntg = min(a.tg, a.taps - tg0);
    const char* sA = smem + (size_t)st * gm.stage_bytes;
    const char* sB = sA + (size_t)(gm.a_rows + 2) * kRowBytes;
    for (int t = 0; t < ntg; ++t) {
      const int toff = (tg0 + t) * a.dil;
      int arow[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int row = zero_row;
        if (row_b[i] >= 0) {
          const int gr = gather_row(a, row_b[i], row_l[i], toff);
          if (gr >= 0) row = gr - R_lo;
        }
        arow[i] = row;
      }
      const int brow0 = t * BN + wn * TN * 32 + (lane & 31);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint4 af[TM], bfr[TN];
        const int slot = ks * 2 + kh;
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[i] = *reinterpret_cast<const uint4*>(sA + arow[i] * kRowBytes + ((slot ^ ((arow[i] >> 2) & 3)) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int rb = brow0 + j * 32;
          bfr[j] = *reinterpret_cast<const uint4*>(sB + rb * kRowBytes + ((slot ^ ((rb >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mfma_step<T>(acc[i][j], af[i], bfr[j]);
      }
    }
    // the LDS-DMA of unit u+1 was issued through inline asm (invisible to hipcc's waitcnt pass, which would
    // otherwise drain it before the first ds_read above): wait for it here, then release stage st
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  const int mrow0 = m0 + wm * TM * 32 + 4 * (lane >> 5);
  const int col0 = n0 + wn * TN * 32 + (lane & 31);
  if (a.residual) {
    if (a.post_act == ACT_NONE) epilogue_plain<T, TM, TN, true, false>(a, acc, mrow0, col0, M);
    else epilogue_plain<T, TM, TN, true, true>(a, acc, mrow0, col0, M);
  } else {
    if (a.post_act == ACT_NONE) epilogue_plain<T, TM, TN, false, false>(a, acc, mrow0, col0, M);
    else epilogue_plain<T, TM, TN, false, true>(a, acc, mrow0, col0, M);
  }
}

int conv_pick_bn(int n) {
  if (n % 128 == 0) return 128;
  if (n % 64 == 0) return 64;
  return 32;
}

size_t conv_packed_weight_bytes(const ConvLayer& ly) {
  const int bke = kRowBytes / (int)dt_size(ly.dt);
  const int nchunks = (ly.cin1 + ly.cin2) / bke;
  return (size_t)nchunks * ly.taps * ly.n_pad * kRowBytes;
}

static inline uint16_t host_f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// generic packer: value(n, ci, tap) supplied by a callback
template <typename F>
static void pack_generic(const ConvLayer& ly, void* dst, F value) {
  const int es = (int)dt_size(ly.dt);
  const int bke = kRowBytes / es;
  const int cin = ly.cin1 + ly.cin2;
  const int nchunks = cin / bke;
  memset(dst, 0, conv_packed_weight_bytes(ly));
  for (int c = 0; c < nchunks; ++c)
    for (int t = 0; t < ly.taps; ++t)
      for (int n = 0; n < ly.n; ++n) {
        const size_t base = (((size_t)c * ly.taps + t) * ly.n_pad + n) * bke;
        for (int k = 0; k < bke; ++k) {
          const float v = value(n, c * bke + k, t);
          if (ly.dt == DT_F32)
            reinterpret_cast<float*>(dst)[base + k] = v;
          else
            reinterpret_cast<uint16_t*>(dst)[base + k] = host_f32_to_bf16(v);
        }
      }
}

void pack_conv_weights(const ConvLayer& ly, const float* w, void* dst) {
  const int cin = ly.cin1 + ly.cin2, k = ly.taps;
  pack_generic(ly, dst, [&](int n, int ci, int t) { return w[((size_t)n * cin + ci) * k + t]; });
}

void pack_convtr_weights(const ConvLayer& ly, const float* w, int cin, int cout, int stride, void* dst) {
  // w [Cin][Cout][2*stride]; GEMM column n = phase*Cout + co; tap 0 reads row q-1 (kernel index phase+stride),
  // tap 1 reads row q (kernel index phase).
  const int k = 2 * stride;
  pack_generic(ly, dst, [&](int n, int ci, int t) {
    const int p = n / cout, co = n - p * cout;
    const int kk = (t == 0) ? p + stride : p;
    return w[((size_t)ci * cout + co) * k + kk];
  });
}

template <typename T, int WM, int WN, int TM, int TN>
static hipError_t launch_cfg(const ConvKArgs& a, int M, size_t lds, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  dim3 grid((M + BM - 1) / BM, a.n_pad / BN);
  auto kern = conv_gemm_kernel<T, WM, WN, TM, TN>;
  static bool lds_opt_in = false;   // one-off, outside any stream capture (launch_conv is first called eagerly)
  if (!lds_opt_in) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    lds_opt_in = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, s, a);
  return hipGetLastError();
}

template <typename T, int WM, int WN, int TM, int TN>
static hipError_t launch_cfg_v2(const ConvKArgs& a, const ConvV2Geom& gm, int M, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  dim3 grid((M + BM - 1) / BM, a.n_pad / BN);
  auto kern = conv_gemm_v2_kernel<T, WM, WN, TM, TN>;
  static bool lds_opt_in = false;
  if (!lds_opt_in) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    lds_opt_in = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), 2 * (size_t)gm.stage_bytes, s, a, gm);
  return hipGetLastError();
}

static int g_conv_force_v1 = -1;

hipError_t launch_conv(const ConvLayer& ly, const ConvCall& c, hipStream_t s) {
  ConvKArgs a;
  a.x1 = (const char*)c.x1; a.x2 = (const char*)c.x2; a.w = (const char*)ly.w; a.bias = ly.bias;
  a.y = (char*)c.y; a.residual = (const char*)c.residual;
  a.C1 = ly.cin1; a.C2 = ly.cin2; a.n = ly.n; a.n_pad = ly.n_pad;
  a.B = c.B; a.L_in = c.L_in; a.L_rows = c.L_rows; a.L_final = c.L_final; a.y_ld = c.y_ld;
  a.taps = ly.taps; a.stride = ly.stride; a.dil = ly.dil; a.pad_left = ly.pad_left; a.ups = ly.ups;
  a.pad_mode = ly.pad_mode; a.pre_act = ly.pre_act; a.post_act = ly.post_act;
  a.tr_stride = ly.tr_stride; a.tr_cout = ly.tr_cout; a.tr_trim_left = ly.tr_trim_left;
  const int BM = 128;
  const int M = c.B * c.L_rows;
  if (M <= 0) return hipSuccess;
  // reflect reach beyond the clamped window ends
  a.reflect_back = 0; a.reflect_fwd = 0;
  if (ly.pad_mode == PAD_REFLECT) {
    a.reflect_fwd = ly.pad_left;
    const int over = (c.L_rows - 1) * ly.stride + (ly.taps - 1) * ly.dil - ly.pad_left - (c.L_in - 1);
    a.reflect_back = std::max(0, over);
  }
  int span = ((BM - 1) * ly.stride + (ly.taps - 1) * ly.dil);
  span = (span >> ly.ups) + 2 + a.reflect_back + a.reflect_fwd;
  {  // input rows at the end of an item that no output position reaches widen the flat window at item seams
    const int u_last = (c.L_rows - 1) * ly.stride + (ly.taps - 1) * ly.dil - ly.pad_left;
    const int leftover = std::max(0, ((c.L_in << ly.ups) - 1 - u_last)) >> ly.ups;
    span += ((BM + c.L_rows - 1) / c.L_rows + 1) * leftover;
  }
  a.win_rows = std::min(span, c.B * c.L_in) + 1;
  const int bn = ly.bn;
  if (g_conv_force_v1 < 0) g_conv_force_v1 = getenv("LDC_CONV_V1") ? 1 : 0;
  if (!g_conv_force_v1 && ly.pad_mode == PAD_ZERO && ly.pre_act == ACT_NONE && ly.tr_stride == 0) {
    ConvV2Geom gm;
    const int quantum = 16 * 4;   // rows per LDS-DMA sweep of the 4 waves
    gm.a_rows = (span + quantum - 1) / quantum * quantum;
    a.tg = std::max(1, std::min(ly.taps, (32 * 1024) / (bn * kRowBytes)));
    gm.ngroups = (ly.taps + a.tg - 1) / a.tg;
    gm.b_rows = (a.tg * bn + quantum - 1) / quantum * quantum;
    gm.stage_bytes = (gm.a_rows + 2 + gm.b_rows) * kRowBytes;
    if (2 * (size_t)gm.stage_bytes <= 160 * 1024) {
      if (ly.dt == DT_F32) {
        if (bn == 128) return launch_cfg_v2<float, 2, 2, 2, 2>(a, gm, M, s);
        if (bn == 64) return launch_cfg_v2<float, 2, 2, 2, 1>(a, gm, M, s);
        return launch_cfg_v2<float, 4, 1, 1, 1>(a, gm, M, s);
      } else {
        if (bn == 128) return launch_cfg_v2<__bf16, 2, 2, 2, 2>(a, gm, M, s);
        if (bn == 64) return launch_cfg_v2<__bf16, 2, 2, 2, 1>(a, gm, M, s);
        return launch_cfg_v2<__bf16, 4, 1, 1, 1>(a, gm, M, s);
      }
    }
  }
  a.tg = std::max(1, std::min(ly.taps, (40 * 1024) / (bn * kPitch)));
  const size_t lds = (size_t)(a.win_rows + 1) * kPitch + (size_t)a.tg * bn * kPitch;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (ly.dt == DT_F32) {
    if (bn == 128) return launch_cfg<float, 2, 2, 2, 2>(a, M, lds, s);
    if (bn == 64) return launch_cfg<float, 2, 2, 2, 1>(a, M, lds, s);
    return launch_cfg<float, 4, 1, 1, 1>(a, M, lds, s);
  } else {
    if (bn == 128) return launch_cfg<__bf16, 2, 2, 2, 2>(a, M, lds, s);
    if (bn == 64) return launch_cfg<__bf16, 2, 2, 2, 1>(a, M, lds, s);
    return launch_cfg<__bf16, 4, 1, 1, 1>(a, M, lds, s);
  }
}

}  // namespace ldc
