// conv_gemm.hip -- implicit-GEMM Conv1d / ConvTranspose1d for gfx950 (CDNA4) on MFMA: weight packing,
// dispatch, and the GENERIC kernel (reflect padding, prologue ELU, transposed-conv scatter, any window).
// The UNet's zero-padded convs take the pipelined fast path in conv_fast.hip.
//
// Replaces every F.conv1d / F.conv_transpose1d the reference issues on the decode path
// (WeightStandardizedConv2d unet.py:72-80, nn.Conv1d unet.py:61,65,201-204,231-232,307,369,
//  SConv1d conv.py:217-232, SConvTranspose1d conv.py:252-274) with one kernel family.
//
// Roofline: dense contraction, MFMA-bound (bf16: v_mfma_f32_32x32x16_bf16, ~2.5 PFLOP/s dense;
// f32: v_mfma_f32_32x32x2_f32, exact fp32 at 157 TFLOP/s).  Algorithmic flops per launch =
// 2 * rows * N * taps * Cin.
//
// Generic kernel, per workgroup (4 wavefronts of 64 lanes), tile 128 x BN of the [rows x N] output:
//   for each 64-byte channel chunk (32 bf16 / 16 f32 channels) of the (concatenated) input:
//     - stage the WINDOW of input rows the tile's positions touch for all taps into LDS once
//       (rows are channels-last, so a tap is just a row offset; padding / reflect / x2 nearest
//       upsampling / striding are row-index arithmetic, a dedicated all-zero LDS row serves
//       out-of-range taps), applying the prologue activation on the way,
//     - stage the packed weight slab [tap][BN][64 B] (contiguous in HBM) into LDS,
//     - every wave issues ds_read_b128 fragments + MFMAs for its (TM x TN) 32x32 sub-tiles.
//   LDS rows are padded 64 -> 80 bytes: the 16 lanes of a ds_read_b128 group then start on 16
//   distinct 4-bank slots (conflict-free, see MI355X LDS banking).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include <math.h>

#include "conv_device.h"

namespace ldc {

static constexpr int kPitch = 80;      // padded LDS row pitch of the generic kernel

template <typename T>
__device__ __forceinline__ uint4 elu16(uint4 v);
template <>
__device__ __forceinline__ uint4 elu16<float>(uint4 v) {
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = fast_elu(f[i]);
  return v;
}
template <>
__device__ __forceinline__ uint4 elu16<__bf16>(uint4 v) {
  unsigned* u = reinterpret_cast<unsigned*>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float lo = bf16_to_f32((unsigned short)(u[i] & 0xffffu));
    float hi = bf16_to_f32((unsigned short)(u[i] >> 16));
    lo = fast_elu(lo);
    hi = fast_elu(hi);
    u[i] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
  }
  return v;
}

__device__ __forceinline__ uint4 expand_fp8x8_generic(const uint2& v) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, false);
  const f32x2 b = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, true);
  const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, false);
  const f32x2 d = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, true);
  return make_uint4(hw_bf16x2(a[0], a[1]), hw_bf16x2(b[0], b[1]), hw_bf16x2(c[0], c[1]), hw_bf16x2(d[0], d[1]));
}

template <typename T, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM* WN * 64) void conv_gemm_kernel(const ConvKArgs a) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  constexpr int EPV = 16 / (int)sizeof(T);        // elements per 16-byte vector
  constexpr int BKE = kRowBytes / (int)sizeof(T);  // channels per chunk
  constexpr int A_MAX = 6;                         // batched 16-byte loads per thread (window)
  constexpr int B_MAX = 8;                         //                                  (weight slab)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + (size_t)(a.win_rows + 1) * kPitch;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int M = a.B * a.L_rows;

  int R_lo, R_hi;
  tile_window(a, m0, BM, M, R_lo, R_hi);
  const int nrows = min(R_hi - R_lo + 1, a.win_rows);
  const int zero_row = a.win_rows;
  if (tid < kPitch / 4) reinterpret_cast<unsigned*>(sA + (size_t)zero_row * kPitch)[tid] = 0u;

  int row_b[TM], row_l[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + (wm * TM + i) * 32 + (lane & 31);
    if (m < M) {
      row_b[i] = (int)fdiv((unsigned)m, a.lrows_div);
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
  int c_lo = 0, c_hi = nchunks;
  if (a.ksplit > 1) {   // split-K slice blockIdx.z of the channel chunks (generic_splitk below)
    const int per = (nchunks + a.ksplit - 1) / a.ksplit;
    c_lo = min(nchunks, (int)blockIdx.z * per);
    c_hi = min(nchunks, c_lo + per);
  }
  for (int c = c_lo; c < c_hi; ++c) {
    const char* src;
    int ld, coff;
    if (c * BKE < a.C1) {
      src = a.x1; ld = a.C1; coff = c * BKE;
    } else {
      src = a.x2; ld = a.C2; coff = c * BKE - a.C1;
    }
    __syncthreads();   // previous chunk's fragments are consumed
    // window: all loads of a batch are issued before the first LDS write (one latency per batch, not per load)
    for (int base = 0; base < nrows * 4; base += NT * A_MAX) {
      uint4 v[A_MAX];
#pragma unroll
      for (int q = 0; q < A_MAX; ++q) {
        const int idx = min(base + q * NT + tid, nrows * 4 - 1);   // clamped: unconditional load keeps v[] in registers
        const int r = idx >> 2, p = idx & 3;
        v[q] = *reinterpret_cast<const uint4*>(src + ((size_t)(R_lo + r) * ld + coff + p * EPV) * sizeof(T));
      }
#pragma unroll
      for (int q = 0; q < A_MAX; ++q) {
        const int idx = base + q * NT + tid;
        if (idx < nrows * 4) {
          const int r = idx >> 2, p = idx & 3;
          uint4 w = v[q];
          if (a.pre_act == ACT_ELU) w = elu16<T>(w);
          *reinterpret_cast<uint4*>(sA + (size_t)r * kPitch + p * 16) = w;
        }
      }
    }
    for (int tg0 = 0; tg0 < a.taps; tg0 += a.tg) {
      const int ntg = min(a.tg, a.taps - tg0);
      if (tg0 > 0) __syncthreads();
      {
        const char* wsrc = a.w + ((size_t)(c * a.taps + tg0) * a.n_pad) * kRowBytes;
        const int total = ntg * BN * 4;
        for (int base = 0; base < total; base += NT * B_MAX) {
          uint4 v[B_MAX];
#pragma unroll
          for (int q = 0; q < B_MAX; ++q) {
            const int idx = min(base + q * NT + tid, total - 1);
            const int t = idx / (BN * 4);
            const int rem = idx - t * (BN * 4);
            if (a.w8) {   // fp8 row of 32 B: logical 16-byte slot `rem & 3` = 8 values = the (permuted) 8-byte slot, expanded to bf16
              const int nn = n0 + (rem >> 2);
              const uint2 raw = *reinterpret_cast<const uint2*>(a.w + (((size_t)(c * a.taps + tg0 + t) * a.n_pad + nn) * 32) + (((rem & 3) ^ ((nn >> 3) & 3)) << 3));
              v[q] = expand_fp8x8_generic(raw);
            } else {
              v[q] = *reinterpret_cast<const uint4*>(wsrc + ((size_t)t * a.n_pad + n0 + (rem >> 2)) * kRowBytes + (rem & 3) * 16);
            }
          }
#pragma unroll
          for (int q = 0; q < B_MAX; ++q) {
            const int idx = base + q * NT + tid;
            if (idx < total) {
              const int t = idx / (BN * 4);
              const int rem = idx - t * (BN * 4);
              *reinterpret_cast<uint4*>(sB + (size_t)(t * BN + (rem >> 2)) * kPitch + (rem & 3) * 16) = v[q];
            }
          }
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

  const int mrow0 = m0 + wm * TM * 32 + 4 * (lane >> 5);
  const int col0 = n0 + wn * TN * 32 + (lane & 31);
  if (a.ksplit > 1) {   // raw fp32 partial sums [slice][M][n_pad]; conv_splitk_reduce_kernel adds them in slice order and runs the epilogue
    float* part = a.sk_part + (size_t)blockIdx.z * M * a.n_pad;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
          if (m < M) part[(size_t)m * a.n_pad + col0 + j * 32] = acc[i][j][r];
        }
    return;
  }
  if (!a.tr_stride) {
    epilogue_dispatch<T, TM, TN>(a, acc, mrow0, col0, M, m0, BM);
    return;
  }
  // transposed conv: GEMM column = phase*Cout + co ; row q -> output position q*stride + phase - trim_left
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + j * 32;
    if (col >= a.n) continue;
    const float bv = a.bias ? a.bias[col] : 0.0f;
    const int tr_p = col / a.tr_cout;
    const int tr_co = col - tr_p * a.tr_cout;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
        if (m >= M) continue;
        const int b = (int)fdiv((unsigned)m, a.lrows_div), q = m - b * a.L_rows;
        const int pos = q * a.tr_stride + tr_p - a.tr_trim_left;
        if (pos < 0 || pos >= a.L_final) continue;
        const float v = act_apply(acc[i][j][r] + bv, a.post_act);
        store_out<T>(a.y, ((size_t)b * a.L_final + pos) * a.y_ld + tr_co, v);
      }
    }
  }
}

// Split-K of the generic kernel (few-tile, long-K layers: the SEANet stacks' last strided convs, k=7 bottleneck convs and first
// transposed convs put 4..64 workgroups on the chip and walk 32..256 channel chunks each, one exposed load latency per chunk):
// the slices' partial sums are added in slice order (deterministic), then bias (+ residual) (+ activation) / the transposed
// conv's scatter exactly as the unsplit epilogues above do
template <typename T>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvKArgs a, int M) {
  const int n4 = a.n_pad >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)M * n4) return;
  const int m = (int)(idx / n4), c4 = (int)(idx - (long)m * n4) * 4;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < a.ksplit; ++z) {
    const float4 v = *reinterpret_cast<const float4*>(a.sk_part + ((size_t)z * M + m) * a.n_pad + c4);
    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
  }
  const float sv[4] = {sum.x, sum.y, sum.z, sum.w};
  int b = 0, q = 0;
  if (a.tr_stride) {
    b = (int)fdiv((unsigned)m, a.lrows_div);
    q = m - b * a.L_rows;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int col = c4 + e;
    if (col >= a.n) continue;
    float v = sv[e] + (a.bias ? a.bias[col] : 0.0f);
    if (!a.tr_stride) {
      if (a.residual) v += load_in<T>(a.residual, (size_t)m * a.n + col);
      store_out<T>(a.y, (size_t)m * a.y_ld + col, act_apply(v, a.post_act));
    } else {
      const int tr_p = col / a.tr_cout, tr_co = col - tr_p * a.tr_cout;
      const int pos = q * a.tr_stride + tr_p - a.tr_trim_left;
      if (pos >= 0 && pos < a.L_final) store_out<T>(a.y, ((size_t)b * a.L_final + pos) * a.y_ld + tr_co, act_apply(v, a.post_act));
    }
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
  return (size_t)nchunks * (ly.wtaps ? ly.wtaps : ly.taps) * ly.n_pad * (ly.w8 ? 32 : kRowBytes);
}

// OCP fp8 e4m3fn (1-4-3, bias 7, no infinities, max 448): round to nearest even, saturate
uint8_t host_f32_to_e4m3(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
  float a = fabsf(f);
  if (!(a == a)) return (uint8_t)(sign | 0x7f);          // NaN
  if (a >= 448.0f) return (uint8_t)(sign | 0x7e);        // saturate (448 = 0x7e)
  if (a <= 0.0009765625f) return sign;                   // <= 2^-10 = half of the smallest subnormal 2^-9: rounds to zero (tie -> even)
  int e;
  const float m = frexpf(a, &e);                          // a = m * 2^e, m in [0.5, 1)
  int exp = e - 1;                                        // a = (2m) * 2^exp, 2m in [1, 2)
  if (exp < -6) {                                         // subnormal: multiples of 2^-9
    const float q = a * 512.0f;                           // in units of 2^-9
    float r = nearbyintf(q);                              // default rounding mode: nearest even
    if (r >= 8.0f) return (uint8_t)(sign | 0x08);         // rounds up into the smallest normal 2^-6
    return (uint8_t)(sign | (uint8_t)r);
  }
  const float frac = (2.0f * m - 1.0f) * 8.0f;            // mantissa in eighths, [0, 8)
  float r = nearbyintf(frac);
  if (r >= 8.0f) { r = 0.0f; ++exp; }
  if (exp > 8 || (exp == 8 && r > 6.0f)) return (uint8_t)(sign | 0x7e);
  return (uint8_t)(sign | (uint8_t)((exp + 7) << 3) | (uint8_t)r);
}
float host_e4m3_to_f32(uint8_t v) {
  const int sign = v & 0x80, e = (v >> 3) & 0xf, m = v & 7;
  float a;
  if (e == 0) a = (float)m * 0.001953125f;                // m * 2^-9
  else if (e == 15 && m == 7) a = NAN;
  else a = ldexpf(1.0f + (float)m / 8.0f, e - 7);
  return sign ? -a : a;
}

static inline uint16_t host_f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// generic packer: value(n, ci, tap) supplied by a callback; image layout [chunk][tap][n_pad][64 B of K]
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

// fp8 image [chunk][tap][n_pad][32 B]: the four 8-byte slots of a row are stored permuted, slot ^ ((n >> 3) & 3), so that the
// 32 lanes of a ds_read_b64 fragment read (rows n..n+31, one slot each) touch all 64 LDS banks once
void pack_conv_weights_fp8(const ConvLayer& ly, const float* w, void* dst, float* scales) {
  const int cin = ly.cin1 + ly.cin2, k = ly.taps, bke = 32;
  const int nchunks = cin / bke;
  uint8_t* out = reinterpret_cast<uint8_t*>(dst);
  memset(out, 0, conv_packed_weight_bytes(ly));
  for (int n = 0; n < ly.n; ++n) {
    float amax = 0.f;
    for (int i = 0; i < cin * k; ++i) amax = std::max(amax, fabsf(w[(size_t)n * cin * k + i]));
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    scales[n] = sc;
    for (int c = 0; c < nchunks; ++c)
      for (int t = 0; t < k; ++t) {
        uint8_t* row = out + (((size_t)c * k + t) * ly.n_pad + n) * 32;
        for (int kk = 0; kk < bke; ++kk) {
          const int slot = kk >> 3, phys = slot ^ ((n >> 3) & 3);
          row[phys * 8 + (kk & 7)] = host_f32_to_e4m3(w[((size_t)n * cin + c * bke + kk) * k + t] / sc);
        }
      }
  }
}

// fp8 x fp8 path (ly.dt == DT_FP8): [chunk of 64 channels][tap][n_pad][64 B] e4m3 with one scale per output channel; the 16-byte
// slots of a row are stored in order (the swizzle is applied by the LDS-DMA source address, as for bf16 rows)
void pack_conv_weights_fp8act(const ConvLayer& ly, const float* w, void* dst, float* scales) {
  const int cin = ly.cin1 + ly.cin2, k = ly.taps, bke = 64;
  const int nchunks = cin / bke;
  uint8_t* out = reinterpret_cast<uint8_t*>(dst);
  memset(out, 0, conv_packed_weight_bytes(ly));
  for (int n = 0; n < ly.n; ++n) {
    float amax = 0.f;
    for (int i = 0; i < cin * k; ++i) amax = std::max(amax, fabsf(w[(size_t)n * cin * k + i]));
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    scales[n] = sc;
    for (int c = 0; c < nchunks; ++c)
      for (int t = 0; t < k; ++t) {
        uint8_t* row = out + (((size_t)c * k + t) * ly.n_pad + n) * 64;
        for (int kk = 0; kk < bke; ++kk) row[kk] = host_f32_to_e4m3(w[((size_t)n * cin + c * bke + kk) * k + t] / sc);
      }
  }
}

void pack_convtr_weights(const ConvLayer& ly, const float* w, int cin, int cout, int stride, void* dst) {
  // w [Cin][Cout][2*stride]; GEMM column n = phase*Cout + co; tap 0 reads row q-1 (kernel index phase+stride),
  // tap 1 reads row q (kernel index phase).
  (void)cin;
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
  dim3 grid((M + BM - 1) / BM, a.n_pad / BN, std::max(1, a.ksplit));
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

// kernel arguments of one conv call; *M_out = GEMM rows (0: nothing to do), *span_out = input rows a 128-row tile can touch
static hipError_t conv_kargs(const ConvLayer& ly, const ConvCall& c, ConvKArgs& a, int* M_out, int* span_out) {
  a.x1 = (const char*)c.x1; a.x2 = (const char*)c.x2; a.w = (const char*)ly.w; a.bias = ly.bias;
  a.y = (char*)c.y; a.residual = (const char*)c.residual;
  a.C1 = ly.cin1; a.C2 = ly.cin2; a.n = ly.n; a.n_pad = ly.n_pad;
  a.B = c.B; a.L_in = c.L_in; a.L_rows = c.L_rows; a.L_final = c.L_final; a.y_ld = c.y_ld;
  a.lrows_div = make_fastdiv((unsigned)std::max(1, c.L_rows));
  a.taps = ly.taps; a.stride = ly.stride; a.dil = ly.dil; a.pad_left = ly.pad_left; a.ups = ly.ups;
  a.pad_mode = ly.pad_mode; a.pre_act = ly.pre_act; a.post_act = ly.post_act;
  a.tr_stride = ly.tr_stride; a.tr_cout = ly.tr_cout; a.tr_trim_left = ly.tr_trim_left;
  a.tg = 1; a.win_rows = 0;
  a.gn_sum = nullptr; a.gn_groups = 0; a.gn_cpg = 1;
  a.colmax = c.colmax; a.colmax_lo = c.colmax_lo; a.colmax_hi = c.colmax_hi; a.colmax_stride = c.colmax_stride;
  if (ly.tr_stride) a.colmax = nullptr;
  a.qkv_ctx_ws = c.qkv_ctx_ws; a.qkv_ctx_stride = c.qkv_ctx_stride;
  a.ksplit = 1; a.sk_part = c.sk_part; a.sk_count = c.sk_count; a.sk_part_cap = c.sk_part_cap; a.sk_count_cap = c.sk_count_cap;
  a.tune = c.tune; a.sk_need = c.sk_need; a.bm_out = c.bm_out;
  if (c.bm_out) { c.bm_out[0] = 0; c.bm_out[1] = 0; c.bm_out[2] = 1; c.bm_out[3] = 0; }
  a.gn_part = (char*)c.gn_part; a.gn_mslots = c.gn_mslots; a.gn_gamma = c.gn_gamma; a.gn_beta = c.gn_beta; a.gn_ss = c.gn_ss; a.gn_out = c.gn_out;
  a.fail_flag = c.fail_flag;
  a.kst = c.kst; a.kst_step = c.kst_step; a.kst_stride = c.kst_stride;
  a.ln_s = ly.ln_s; a.ln_rowstat = c.ln_rowstat; a.rowstat_out = c.rowstat_out;
  a.gn_nap = c.tune ? c.tune->gn_nap : 1; a.gn_nap0 = c.tune ? c.tune->gn_nap0 : 0;
  a.y2 = (char*)c.y2; a.bias2 = ly.bias2; a.wtaps = ly.wtaps ? ly.wtaps : ly.taps;
  if ((c.y2 != nullptr) != (ly.wtaps != 0)) return hipErrorInvalidValue;   // a folded layer always writes its second output
  a.wscale = (ly.w8 || ly.dt == DT_FP8) ? ly.wscale : nullptr; a.w8 = ly.w8;
  if (c.sk_need) *c.sk_need = 0;
  if (c.gn_sum && c.gn_groups > 0 && !ly.tr_stride) {
    const int cpg = ly.n / c.gn_groups;
    const bool pow2 = cpg >= 4 && (cpg & (cpg - 1)) == 0;
    if (ly.n % c.gn_groups || !pow2) return hipErrorInvalidValue;   // the planner only asks for supported widths
    a.gn_sum = c.gn_sum; a.gn_groups = c.gn_groups; a.gn_cpg = cpg;
  }
  if (c.gn_part && c.gn_groups > 0) { a.gn_groups = c.gn_groups; a.gn_cpg = ly.n / c.gn_groups; }   // fused apply: no gn_sum
  const int BM = 128;
  const int M = c.B * c.L_rows;
  *M_out = M; *span_out = 0;
  if (M <= 0) return hipSuccess;
  // reflect reach beyond the clamped window ends
  a.reflect_back = 0; a.reflect_fwd = 0;
  if (ly.pad_mode == PAD_REFLECT) {
    a.reflect_fwd = ly.pad_left;
    const int over = (c.L_rows - 1) * ly.stride + (ly.taps - 1) * ly.dil - ly.pad_left - (c.L_in - 1);
    a.reflect_back = std::max(0, over);
  }
  // rows of input a 128-row tile can touch (per extra 128 rows of tile: + 128*stride >> ups)
  int span = ((BM - 1) * ly.stride + (ly.taps - 1) * ly.dil);
  span = (span >> ly.ups) + 1 + ly.ups + a.reflect_back + a.reflect_fwd;
  int seam_slack = 0;
  {  // input rows at the end of an item that no output position reaches widen the flat window at item seams
    const int u_last = (c.L_rows - 1) * ly.stride + (ly.taps - 1) * ly.dil - ly.pad_left;
    const int leftover = std::max(0, ((c.L_in << ly.ups) - 1 - u_last)) >> ly.ups;
    seam_slack = ((BM + c.L_rows - 1) / c.L_rows + 1) * leftover;
  }
  span += seam_slack;
  *span_out = span;
  return hipSuccess;
}

// split-K factor of the generic kernel for this call (1: none): only where the caller opted in (ConvCall::generic_split with a
// workspace), for the 64 x 64-tile launches that leave most of the chip idle and have at least 8 channel chunks to share out
static int generic_splitk(const ConvLayer& ly, const ConvCall& c, int M) {
  if (!c.generic_split || !c.sk_part || ly.w8 || ly.dt == DT_FP8 || c.gn_sum || c.colmax || (ly.n_pad & 63)) return 1;
  const int bke = kRowBytes / (int)dt_size(ly.dt);
  const int nchunks = (ly.cin1 + ly.cin2) / bke;
  const long tiles = (long)((M + 63) / 64) * (ly.n_pad / 64);
  if (tiles >= 128 || nchunks < 8) return 1;
  long ks = std::min<long>(nchunks / 2, (256 + tiles - 1) / tiles);
  ks = std::min<long>(ks, c.sk_part_cap / std::max<long>(1, (long)M * ly.n_pad));
  return ks >= 2 ? (int)ks : 1;
}

long long conv_generic_splitk_floats(const ConvLayer& ly, const ConvCall& c) {
  if ((c.tune && c.tune->force_generic ? false : conv_fast_eligible(ly))) return 0;
  ConvCall cc = c;
  cc.generic_split = 1; cc.sk_part = reinterpret_cast<float*>(16); cc.sk_part_cap = 1ll << 50;
  const int M = c.B * c.L_rows;
  const int bn = ly.bn;
  if (M <= 0 || (long)((M + 127) / 128) * (ly.n_pad / bn) >= 128 || ly.n_pad % 64 != 0 || bn < 64) return 0;
  const int ks = generic_splitk(ly, cc, M);
  return ks > 1 ? (long long)ks * M * ly.n_pad : 0;
}

hipError_t launch_conv(const ConvLayer& ly, const ConvCall& c, hipStream_t s) {
  ConvKArgs a;
  int M = 0, span = 0;
  hipError_t e0 = conv_kargs(ly, c, a, &M, &span);
  if (e0 != hipSuccess || M <= 0) return e0;
  if (!(c.tune && c.tune->force_generic) && conv_fast_eligible(ly)) {
    bool launched = false;
    hipError_t e = launch_conv_fast(ly, a, M, span, s, &launched);
    if (e != hipSuccess || launched) return e;
  }
  if (c.sk_need) return hipSuccess;   // dry run: the generic kernel never splits K
  if (c.qkv_ctx_ws) return hipErrorInvalidValue;   // (the context fold exists only on the lean kernel; the planner asks for it only there)
  if (c.gn_part || c.y2 || ly.ln_s) return hipErrorInvalidValue;   // (the planner asks bm_out first)   // the fused GroupNorm apply exists only on the pipelined kernel (the planner checks bm_out)
  if (ly.dt == DT_FP8) return hipErrorInvalidValue;   // fp8 inputs exist only on the pipelined kernel (the planner checks eligibility)
  a.win_rows = std::min(span, c.B * c.L_in) + 1;
  int bn = ly.bn;
  // few-tile GEMMs with a long K (the SEANet encoder's last strided / k=7 convs: 3 840 rows x 3 584..4 096 deep, 30
  // tiles of 128 x 128 took 0.5 ms each on 30 CUs): 64 x 64 tiles put 4x the workgroups on the chip
  const long tiles128 = (long)((M + 127) / 128) * (ly.n_pad / bn);
  const bool small = tiles128 < 128 && ly.n_pad % 64 == 0 && bn >= 64;
  if (small) bn = 64;
  a.tg = std::max(1, std::min(ly.taps, (40 * 1024) / (bn * kPitch)));
  const size_t lds = (size_t)(a.win_rows + 1) * kPitch + (size_t)a.tg * bn * kPitch;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (small) {
    const int ks = generic_splitk(ly, c, M);
    if (ks > 1) {
      a.ksplit = ks;
      hipError_t e = ly.dt == DT_F32 ? launch_cfg<float, 2, 2, 1, 1>(a, M, lds, s) : launch_cfg<__bf16, 2, 2, 1, 1>(a, M, lds, s);
      if (e != hipSuccess) return e;
      const long n4 = (long)M * (a.n_pad / 4);
      if (ly.dt == DT_F32) hipLaunchKernelGGL(conv_splitk_reduce_kernel<float>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a, M);
      else hipLaunchKernelGGL(conv_splitk_reduce_kernel<__bf16>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a, M);
      return hipGetLastError();
    }
    return ly.dt == DT_F32 ? launch_cfg<float, 2, 2, 1, 1>(a, M, lds, s) : launch_cfg<__bf16, 2, 2, 1, 1>(a, M, lds, s);
  }
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
