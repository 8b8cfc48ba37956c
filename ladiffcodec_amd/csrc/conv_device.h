// conv_device.h -- device-side pieces shared by the two conv-GEMM kernels (conv_gemm.hip: generic,
// conv_fast.hip: LDS-DMA pipelined fast path).
#pragma once
#include "ldc_kernels.h"
#include "ldc_math.h"
#include <type_traits>

namespace ldc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int ldc_tid() { return (int)threadIdx.x; }

static constexpr int kRowBytes = 64;   // bytes of K (channels) per LDS row per chunk: 32 bf16 or 16 f32

// Division by a launch-invariant divisor (Granlund-Montgomery, exact for every 32-bit n): gfx950 has no integer divide, so
// every `/` by a run-time value is a ~30-instruction reciprocal sequence -- ten of them sat in the conv prologue in front of
// the first copy.  The host prepares (m, sh1, sh2); the device spends a mul_hi, a sub, an add and two shifts.
struct FastDivU { unsigned m, sh1, sh2; };
inline FastDivU make_fastdiv(unsigned d) {
  if (d < 1) d = 1;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  const unsigned long long m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
  FastDivU r;
  r.m = (unsigned)m; r.sh1 = l < 1 ? l : 1u; r.sh2 = l > 0 ? l - 1 : 0u;
  return r;
}
template <typename D>
__device__ __forceinline__ unsigned fdiv(unsigned n, const D& d) {
  const unsigned t = __umulhi(d.m, n);
  return (t + ((n - t) >> d.sh1)) >> d.sh2;
}

struct ConvKArgs {
  const char* x1;
  const char* x2;
  const char* w;
  const float* bias;
  char* y;
  const char* residual;
  int C1, C2;          // channels of the two inputs
  int n, n_pad;
  int B, L_in, L_rows, L_final, y_ld;
  FastDivU lrows_div;  // division by L_rows
  int taps, stride, dil, pad_left, ups, pad_mode, pre_act, post_act;
  int tr_stride, tr_cout, tr_trim_left;
  int win_rows;        // generic kernel: LDS window capacity (rows); the zero row lives at index win_rows
  int tg;              // taps staged per weight slab
  int reflect_back, reflect_fwd;
  float* gn_sum;       // fused GroupNorm statistics target [B][gn_groups][kGnPad] (pre-zeroed; [0] sum, [1] sum of squares) or null
  int gn_groups, gn_cpg;
  unsigned* colmax;    // fused column max over positions (LinearAttention k softmax): ordered-uint keys, pre-zeroed
  int colmax_lo, colmax_hi, colmax_stride;   // columns [lo, hi) -> colmax[b * stride + col - lo]
  float* qkv_ctx_ws;       // LinearAttention context accumulated by the to_qkv launch itself (ConvCall::qkv_ctx_ws; lean kernel only) or null
  int qkv_ctx_stride;
  // split-K (fast kernel, few-tile long-K layers): ksplit workgroups per tile each take a slice of the units, park
  // their fp32 partial tile in sk_part and the last one to arrive (sk_count) sums them and runs the epilogue
  int ksplit;
  float* sk_part;
  unsigned* sk_count;
  long long sk_part_cap;   // floats
  int sk_count_cap;        // tiles
  int w8;                  // weights are fp8 (generic kernel expands them while staging)
  const float* wscale;     // fp8 weights: per-output-channel scale applied to the accumulator before the bias (else null)
  // Fused GroupNorm APPLY (fast kernel; ResnetBlock "Block", unet.py:137-153): every wave of a finished tile publishes its partial
  // statistics, waits until every tile of the items it touches has published, and then normalises its own accumulators in
  // registers -- GN affine, timestep scale / shift, SiLU (+ residual) (+ tanh) -- before the one store (epilogue_gn_fused).
  // The conv output never goes to HBM un-normalised and the gn_apply launch (one read + one write of the tensor) disappears.
  char* gn_part;           // [B][gn_mslots][WM][n / 32] x 16-byte granule pairs (zeroed by the step's first kernel) or null: not fused
  int gn_mslots;           // M-tile slots per item in gn_part (>= the M tiles one item can touch)
  const float* gn_gamma;   // [n]
  const float* gn_beta;    // [n]
  const float* gn_ss;      // [2 n]: (scale | shift) of the current timestep, or null
  int gn_out;              // bit 2: tanh after the residual add (the final ResnetBlock feeds torch.tanh alone, unet.py:467); bit 0: the output
                           // is written as OCP fp8 e4m3 (its only consumer is an fp8 x fp8 conv; bf16 kernels only)
  unsigned* fail_flag;     // host-mapped word raised when a bounded spin gives up (the output is then wrong, never a hang)
  // Timed-mode stamps (ldc_kstamps_enable): every launch records the earliest workgroup start and the latest workgroup end on
  // the 100 MHz wall clock in its own slot kst[(step index & 2047) * kst_stride + {0, 1}] (atomicMin of t / of ~t), so that
  // per-kernel durations exist for the graph-replayed, two-stream mode the bench times (rocprofv3 serialises the streams)
  unsigned long long* kst;
  const int* kst_step;     // the part's device step state: [1] = iteration index
  int kst_stride;
  int gn_nap, gn_nap0;     // fused GroupNorm exchange: s_sleep(1) repetitions between polls / before the first poll (ConvTune)
  const float* ln_s;       // folded PreNorm LayerNorm (ConvLayer::ln_s) or null
  const float* ln_rowstat; // ... its row statistics as partials [rows][C / 32][2] (sum, centred M2 per 32-column block) written by the
                           // producer of the input (rowstat_out of the fused block2 conv), or null: the conv reads its rows once more
  float* rowstat_out;      // fused GroupNorm apply with residual: also write those partials of the stored rows [rows][n / 32][2]: (sum, M2 about the block mean)
  char* y2;                // folded 1x1 conv (ConvLayer::wtaps): second output [rows][n], or null
  const float* bias2;
  int wtaps;               // weight slabs per channel chunk in the packed image (taps, or taps + 1 with the folded conv)
  const ConvTune* tune;    // host-only (never read on the device)
  long long* sk_need;      // host-only: dry run
  int* bm_out;             // host-only: dry run -- rows per tile the fast kernel would use (0: generic kernel); bm_out[1] = wave rows WM, bm_out[2] = split-K factor, bm_out[3] = columns per tile
};

// ---- geometry of one launch of the pipelined kernel (conv_fast.inc), decided on the host ----
struct FastGeom {
  int a_rows;        // window rows per plane (multiple of 16 * waves); each plane is followed by a zero row
  int b_rows;        // weight rows per unit (NS * BN rounded up to 16 * waves)
  int stage_bytes;
  int ngroups;       // tap groups per chunk group (k=7: 2)
  int m_fastest;     // tile order inside an XCD's run: M-tile fastest (weights dominate) or N-tile fastest
  int item_major;    // fused GroupNorm apply: dispatch order follows the items (see the tile-order comment in the kernel)
  FastDivU d_grp, d_rem, d_per;   // item_major: divisions by 8 * ntn, by ntm % 8 and by ntn / 8
  int grid;          // workgroups of this conv (tiles x split-K slices)
  int ln_off;        // folded LayerNorm: LDS byte offset of the tile's (mean, rstd) table, behind the ring and the epilogue staging
  int debug;         // tuning aid (LDC_CONV_DEBUG) bits: 1 no copies after the prologue, 4 no output stores, 8 return at once, 16 one unit only
  int ntiles, ntn, ntm;                      // output tiles, tiles along N / along M (set by the launcher)
  FastDivU d_ntiles, d_ntn, d_ntm, d_ksplit; // divisions by them, prepared on the host
  int sk_xcd;                                // split-K by 2 / 4: slice and N-tile parity follow the XCD (blockIdx & 7), see conv_fast_body
  FastDivU d_skper;                          // ... division by ntn / (8 / ksplit)
  unsigned long long* stamps;   // tuning aid: per workgroup {start, prologue done, loop done, end} s_memtime
};

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) { return hw_bf16(f); }

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case ACT_SILU: return fast_silu(v);
    case ACT_ELU: return fast_elu(v);
    case ACT_TANH: return fast_tanh(v);
    case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case ACT_RELU: return fmaxf(v, 0.0f);
    default: return v;
  }
}

// One 64-byte-per-row K step of a 32x32 output tile.  Both operands use the same (lane>>5, element) -> k
// mapping, so the K order inside a chunk is irrelevant; only the C/D layout matters
// (col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
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

// flat input row reached from GEMM row (b, l) with tap offset `toff` (= tap*dil); -1 when it is padding
template <typename KA>
__device__ __forceinline__ int gather_row(const KA& a, int b, int l, int toff) {
  int u = l * a.stride + toff - a.pad_left;
  const int leff = a.L_in << a.ups;
  if (a.pad_mode == PAD_REFLECT) {
    if (u < 0) u = -u;
    if (u >= leff) u = 2 * (leff - 1) - u;
  }
  if (u < 0 || u >= leff) return -1;
  return b * a.L_in + (u >> a.ups);
}

// [R_lo, R_hi]: flat input rows a BM-row tile starting at m0 touches (zero-pad convs; monotone in m and tap)
template <typename KA>
__device__ __forceinline__ void tile_window(const KA& a, int m0, int BM, int M, int& R_lo, int& R_hi) {
  const int leff = a.L_in << a.ups;
  const int m_last = min(m0 + BM, M) - 1;
  int b = (int)fdiv((unsigned)m0, a.lrows_div), l = m0 - b * a.L_rows;
  int u = l * a.stride - a.pad_left;
  u = max(0, min(u, leff - 1));
  R_lo = b * a.L_in + (u >> a.ups) - a.reflect_back;
  b = (int)fdiv((unsigned)m_last, a.lrows_div);
  l = m_last - b * a.L_rows;
  u = l * a.stride + (a.taps - 1) * a.dil - a.pad_left;
  u = max(0, min(u, leff - 1));
  R_hi = b * a.L_in + (u >> a.ups) + a.reflect_fwd;
  R_lo = max(R_lo, 0);
  R_hi = min(R_hi, a.B * a.L_in - 1);
}

// bias (+ residual) (+ activation) and store of a wave's TM x TN accumulators (plain conv, row-major [M][n])
template <typename T, int TM, int TN, bool RES, bool ACT, typename KA>
__device__ __forceinline__ void epilogue_plain(const KA& a, f32x16 (&acc)[TM][TN], int mrow0, int col0, int M) {
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + j * 32;
    const bool col_ok = col < a.n;
    const float bv = (a.bias && col_ok) ? a.bias[col] : 0.0f;
    const float sc = (a.wscale && col_ok) ? a.wscale[col] : 1.0f;
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
        float v = fmaf(acc[i][j][r], sc, bv);
        if (RES) v += rv[r];
        if (ACT) v = act_apply(v, a.post_act);
        if (col_ok && m < M) store_out<T>(a.y, (size_t)m * a.y_ld + col, v);
      }
    }
  }
}

// order-preserving float -> uint key: atomicMax(unsigned) from a zeroed buffer implements a float max
__device__ __forceinline__ unsigned float_order_key(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// Fused column max per item of the stored (dtype-rounded) conv output, columns [colmax_lo, colmax_hi): the
// max over positions that LinearAttention's k.softmax(dim=-1) needs (unet.py:214), so that no extra pass over
// the qkv tensor is required for it.
template <typename T, int TM, int TN, typename KA>
__device__ __forceinline__ void epilogue_colmax(const KA& a, f32x16 (&acc)[TM][TN], int mrow0, int col0, int m0, int BM,
                                                int M) {
  const int lane = ldc_tid() & 63;
  const int b_first = (int)fdiv((unsigned)m0, a.lrows_div);
  const int b_last = (int)fdiv((unsigned)(min(m0 + BM, M) - 1), a.lrows_div);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + j * 32;
    const bool in = col >= a.colmax_lo && col < a.colmax_hi;     // uniform over the 32 lanes of a sub-tile
    const float bv = (a.bias && col < a.n) ? a.bias[col] : 0.0f;
    const float sc = (a.wscale && col < a.n) ? a.wscale[col] : 1.0f;
    for (int bb = b_first; bb <= b_last; ++bb) {
      const int lo = bb * a.L_rows, hi = min(lo + a.L_rows, M);
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
          if (m >= lo && m < hi) {
            float v = fmaf(acc[i][j][r], sc, bv);
            if (sizeof(T) == 2) v = bf16_to_f32(f32_to_bf16(v));
            mx = fmaxf(mx, v);
          }
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (in && lane < 32 && mx > -INFINITY) atomicMax(a.colmax + (size_t)bb * a.colmax_stride + (size_t)(col - a.colmax_lo), float_order_key(mx));
    }
  }
}

// Fused GroupNorm statistics (unet.py:142-147 normalises the conv output): every wave adds the sum and the sum
// of squares of (acc + bias) per (item, group) to gn_sum[b][g][2].  A lane owns one column per 32-wide sub-tile,
// i.e. one group; lanes of a group are reduced with xor-shuffles (channels per group: a power of two >= 4 that
// divides or is a multiple of 32), then one lane per group issues the two atomics.  Rows of several items can
// meet in one tile, hence the (wave-uniform) loop over the items the tile touches.
template <int TM, int TN, typename KA>
__device__ __forceinline__ void epilogue_gn_stats(const KA& a, f32x16 (&acc)[TM][TN], int mrow0, int col0, int m0,
                                                  int BM, int M) {
  const int lane = ldc_tid() & 63;
  const int b_first = (int)fdiv((unsigned)m0, a.lrows_div);
  const int b_last = (int)fdiv((unsigned)(min(m0 + BM, M) - 1), a.lrows_div);
  const int cpg = a.gn_cpg;
  const int seg = min(cpg, 32);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + j * 32;
    const bool col_ok = col < a.n;
    const float bv = (a.bias && col_ok) ? a.bias[col] : 0.0f;
    const float sc = (a.wscale && col_ok) ? a.wscale[col] : 1.0f;
    const int g = col_ok ? col / cpg : 0;
    for (int bb = b_first; bb <= b_last; ++bb) {
      const int lo = bb * a.L_rows, hi = min(lo + a.L_rows, M);
      float s = 0.f, ss = 0.f;
      if (b_first == b_last && m0 + BM <= M) {
        // the usual case: the whole tile lies inside one item -- no per-element row predicates
        if (col_ok) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = fmaf(acc[i][j][r], sc, bv);
              s += v;
              ss = fmaf(v, v, ss);
            }
        }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
            if (col_ok && m >= lo && m < hi) {
              const float v = fmaf(acc[i][j][r], sc, bv);
              s += v;
              ss += v * v;
            }
          }
      }
      s += __shfl_xor(s, 32);
      ss += __shfl_xor(ss, 32);
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        if (o < seg) {
          s += __shfl_xor(s, o);
          ss += __shfl_xor(ss, o);
        }
      }
      if (col_ok && lane < 32 && (lane & (seg - 1)) == 0) {
        atomicAdd(&a.gn_sum[((size_t)bb * a.gn_groups + g) * kGnPad], s);
        atomicAdd(&a.gn_sum[((size_t)bb * a.gn_groups + g) * kGnPad + 1], ss);
      }
    }
  }
}

template <typename T, int TM, int TN, typename KA>
__device__ __forceinline__ void epilogue_dispatch(const KA& a, f32x16 (&acc)[TM][TN], int mrow0, int col0, int M,
                                                  int m0 = 0, int BM = 0) {
  if (a.gn_sum) epilogue_gn_stats<TM, TN>(a, acc, mrow0, col0, m0, BM, M);
  if (a.colmax) epilogue_colmax<T, TM, TN>(a, acc, mrow0, col0, m0, BM, M);
  if (a.residual) {
    if (a.post_act == ACT_NONE) epilogue_plain<T, TM, TN, true, false>(a, acc, mrow0, col0, M);
    else epilogue_plain<T, TM, TN, true, true>(a, acc, mrow0, col0, M);
  } else {
    if (a.post_act == ACT_NONE) epilogue_plain<T, TM, TN, false, false>(a, acc, mrow0, col0, M);
    else epilogue_plain<T, TM, TN, false, true>(a, acc, mrow0, col0, M);
  }
}

// Store path of the fast kernel: the direct form above issues 16*TM*TN two-byte stores per lane (64 per lane for
// a 64x64 wave tile; measured ~375 cycles per store instruction, ~10 us per workgroup -- longer than the whole
// MFMA loop of a 256-channel k=3 conv).  Here each wave transposes its tile through LDS (ring stages are free
// once the K loop has drained) and writes whole rows: 16 bytes per lane, 8 rows x 128 B per store instruction.
template <typename T, int TM, int TN, bool RES, bool ACT, typename KA>
__device__ __forceinline__ void epilogue_rows(const KA& a, f32x16 (&acc)[TM][TN], char* wave_lds, int m_wave0,
                                              int col_wave0, int M) {
  constexpr int RB = TN * 32 * (int)sizeof(T);      // bytes per tile row
  constexpr int PITCH = RB + 16;
  constexpr int LPR = RB / 16;                      // lanes per row in the read-back
  constexpr int RPS = 64 / LPR;                     // rows per sweep
  const int lane = ldc_tid() & 63;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col_wave0 + j * 32 + (lane & 31);
    const bool col_ok = col < a.n;
    const float bv = (a.bias && col_ok) ? a.bias[col] : 0.0f;
    const float sc = (a.wscale && col_ok) ? a.wscale[col] : 1.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float rv[16];
      if (RES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m_wave0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          rv[r] = (col_ok && m < M) ? load_in<T>(a.residual, (size_t)m * a.n + col) : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = fmaf(acc[i][j][r], sc, bv);
        if (RES) v += rv[r];
        if (ACT) v = act_apply(v, a.post_act);
        store_out<T>(wave_lds, (size_t)(row * PITCH) / sizeof(T) + j * 32 + (lane & 31), v);
      }
    }
  }
  // same wave wrote and reads: LDS executes a wave's operations in order
  const int rsub = lane / LPR, chunk = lane % LPR;
  const int col = col_wave0 + chunk * (16 / (int)sizeof(T));
#pragma unroll
  for (int sw = 0; sw < TM * 32 / RPS; ++sw) {
    const int row = sw * RPS + rsub;
    const int m = m_wave0 + row;
    const uint4 v = *reinterpret_cast<const uint4*>(wave_lds + row * PITCH + chunk * 16);
    if (m < M && col < a.n) *reinterpret_cast<uint4*>(a.y + ((size_t)m * a.y_ld + col) * sizeof(T)) = v;
  }
}

// The same with a residual: the rows are staged in fp32 and the residual is added in the row phase as 16-byte pieces (one rounding,
// as before) -- the accumulator-layout form above issues sixteen 2-byte residual loads per lane and 32 x 32 sub-tile.
template <typename T, int TM, int TN, bool ACT, typename KA>
__device__ __forceinline__ void epilogue_rows_res(const KA& a, f32x16 (&acc)[TM][TN], char* wave_lds, int m_wave0, int col_wave0, int M) {
  constexpr int PITCH = TN * 32 * 4 + 16;
  constexpr int EPL = 16 / (int)sizeof(T);          // output elements per lane in the row phase (16 bytes)
  constexpr int LPR = TN * 32 / EPL;                // lanes per row
  constexpr int RPS = 64 / LPR;                     // rows per sweep
  constexpr int NSW = TM * 32 / RPS;
  const int lane = ldc_tid() & 63;
  const int rsub = lane / LPR, chunk = lane % LPR;
  const int col = col_wave0 + chunk * EPL;
  uint4 rres[NSW];
#pragma unroll
  for (int sw = 0; sw < NSW; ++sw) {
    const int m = m_wave0 + sw * RPS + rsub;
    const bool ok = m < M && col < a.n;
    rres[sw] = *reinterpret_cast<const uint4*>(a.residual + ((size_t)(ok ? m : 0) * a.n + (ok ? col : 0)) * sizeof(T));
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int c = col_wave0 + j * 32 + (lane & 31);
    const bool col_ok = c < a.n;
    const float bv = (a.bias && col_ok) ? a.bias[c] : 0.0f;
    const float sc = (a.wscale && col_ok) ? a.wscale[c] : 1.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        reinterpret_cast<float*>(wave_lds + (size_t)row * PITCH)[j * 32 + (lane & 31)] = fmaf(acc[i][j][r], sc, bv);
      }
  }
  // same wave wrote and reads: LDS executes a wave's operations in order
#pragma unroll
  for (int sw = 0; sw < NSW; ++sw) {
    const int row = sw * RPS + rsub;
    const int m = m_wave0 + row;
    float f[EPL];
    const float* sp = reinterpret_cast<const float*>(wave_lds + (size_t)row * PITCH + chunk * EPL * 4);
#pragma unroll
    for (int e = 0; e < EPL; e += 4) {
      const float4 t = *reinterpret_cast<const float4*>(sp + e);
      f[e] = t.x; f[e + 1] = t.y; f[e + 2] = t.z; f[e + 3] = t.w;
    }
    const unsigned rw[4] = {rres[sw].x, rres[sw].y, rres[sw].z, rres[sw].w};
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) f[e] += __uint_as_float(rw[e]);
    } else {
#pragma unroll
      for (int e = 0; e < EPL; e += 2) {
        f[e] += __uint_as_float(rw[e >> 1] << 16);
        f[e + 1] += __uint_as_float(rw[e >> 1] & 0xffff0000u);
      }
    }
    if (ACT) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) f[e] = act_apply(f[e], a.post_act);
    }
    uint4 v;
    if constexpr (sizeof(T) == 4) v = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    else v = make_uint4(hw_bf16x2(f[0], f[1]), hw_bf16x2(f[2], f[3]), hw_bf16x2(f[4], f[5]), hw_bf16x2(f[6], f[7]));
    if (m < M && col < a.n) *reinterpret_cast<uint4*>(a.y + ((size_t)m * a.y_ld + col) * sizeof(T)) = v;
  }
}

// 16-byte global accesses that other workgroups of the SAME launch may rely on (the eight XCD L2s are not coherent and a
// CU's L1 is never refreshed by another CU's stores): write-through (sc1) stores, L1-bypassing (sc1) loads.  Inline asm: the
// compiler neither counts nor waits for them -- the caller drains with wait_vm0().
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_wt(char* p, const uint4& v) {
  const u32x4_t x = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void load16_sc1_issue(u32x4_t& dst, const char* p) {
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(dst) : "v"(p) : "memory");
}
// a PLAIN dword store the compiler can neither drop nor promote (a volatile store compiles to `sc0 sc1`): stays in this XCD's L2
__device__ __forceinline__ void store_plain_u32(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct GnStamp { unsigned long long t_pub, t_first, t_done; unsigned polls; };   // tuning aid: s_memtime after the publish / the first poll / the last; polls issued

// Fused GroupNorm apply of a wave's TM x TN accumulators (ConvKArgs::gn_part != null).
//
// Statistics exchange WITHOUT atomics, counters or barriers: every wave publishes the (sum, sum of squares) of its rows x 32
// columns per item as ONE 16-byte write-through store holding two data-tagged granules {1, sum}{1, sumsq} into its own slot
//   gn_part[item][M-tile slot of the item][wave row wm][32-column block]          (zeroed by the step's first kernel),
// then reads back every slot of its (item, group) -- lanes in parallel, L1-bypassing loads -- until all tags are set, and sums
// them with a fixed shuffle tree: the statistics are bit-reproducible from run to run (the atomics were not), and the chain
// a tile waits on is one store latency + one load round trip after the item's last tile has finished its K loop (the
// atomic form -- drain the statistics atomics, arrive at a counter, poll it, load the sums -- measured +5.4 us per launch).
// Then: per-column affine (same arithmetic as gn_apply_cols_kernel: a1 = rstd * gamma, b1 = beta - mean * a1, the timestep
// (scale + 1) and shift folded in) -> SiLU -> LDS transpose -> whole-row stores.  RES: the rows are staged in fp32 and the
// residual is added to the un-rounded value as 16-byte pieces in the row phase (one rounding, like gn_apply), then the
// optional tanh.  A tile may straddle items (flat M tiling); the launcher guarantees 2 * L_rows >= BM, i.e. at most three.
// The spin is bounded by the 100 MHz wall clock: tiles that are not all resident in time raise the host-mapped flag, never a hang.
template <typename T, int TM, int TN, bool RES, typename KA>
__device__ __forceinline__ void epilogue_gn_fused(const KA& a, f32x16 (&acc)[TM][TN], char* wave_lds, int m_wave0,
                                                  int col_wave0, int M, int m0, int BM, int WM, int wm, int m_base = 0,
                                                  GnStamp* gst = nullptr) {   // (tuning aid)
  // m_base: first row of the M tiling this tile belongs to (0: the flat tiling of a launch of its own; an XCD team of a chain tiles
  // its own rows from its first item's first row, and M is then the team's last row + 1)
  const int lane = ldc_tid() & 63;
  const int b_first = (int)fdiv((unsigned)m0, a.lrows_div);
  const int b_last = (int)fdiv((unsigned)(min(m0 + BM, M) - 1), a.lrows_div);
  const int m_split = (b_first + 1) * a.L_rows;   // rows >= m_split belong to item b_first + 1, rows >= m_split2 to b_first + 2
  const int m_split2 = m_split + a.L_rows;
  const int cpg = a.gn_cpg;                       // a multiple of 32: a 32-column block lies inside one group
  const int sub_n = cpg >> 5;                     // 32-column blocks per group (a power of two)
  const int nc32 = a.n >> 5;
  const int mt = (m0 - m_base) / BM;
  float bv[TN], wsc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col_wave0 + j * 32 + (lane & 31);
    bv[j] = (a.bias && col < a.n) ? a.bias[col] : 0.0f;
    wsc[j] = (a.wscale && col < a.n) ? a.wscale[col] : 1.0f;
  }
  // 1. publish this wave's partial statistics
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int c32 = (col_wave0 + j * 32) >> 5;
    for (int bb = b_first; bb <= b_last; ++bb) {
      const int lo = bb * a.L_rows, hi = min(lo + a.L_rows, M);
      float s = 0.f, ss = 0.f;
      if (b_first == b_last && m0 + BM <= M) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = fmaf(acc[i][j][r], wsc[j], bv[j]);
            s += v;
            ss = fmaf(v, v, ss);
          }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m_wave0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m >= lo && m < hi) {
              const float v = fmaf(acc[i][j][r], wsc[j], bv[j]);
              s += v;
              ss = fmaf(v, v, ss);
            }
          }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        ss += __shfl_xor(ss, o);
      }
      if (lane == 0 && c32 < nc32) {
        const int mslot = mt - (lo - m_base) / BM;
        char* dst = a.gn_part + ((((size_t)bb * a.gn_mslots + mslot) * WM + wm) * nc32 + c32) * 16;
        store16_wt(dst, make_uint4(1u, __float_as_uint(s), 1u, __float_as_uint(ss)));
      }
    }
  }
  if (gst) { gst->t_pub = __builtin_amdgcn_s_memtime(); gst->polls = 0; gst->t_first = 0; }
  // 2. gather the statistics of this wave's (item, group) pairs and build the per-column affine
  const float inv_n = 1.0f / ((float)a.L_rows * (float)cpg);
  float ca0[TN], cb0[TN], ca1[TN], cb1[TN], ca2[TN], cb2[TN];   // per-column affine for the (at most three) items of the tile
  const unsigned long long t0 = wall_clock64();
  // statistics of (item bb, the group of 32-column block c32_0) -> mean, rstd (wave-uniform values, computed by every lane)
  auto gather = [&](int bb, int g, float& mean, float& rstd) {
    const int lo = bb * a.L_rows, hi = min(lo + a.L_rows, M);
    const int total = ((hi - 1 - m_base) / BM - (lo - m_base) / BM + 1) * WM * sub_n;   // granule pairs of (item bb, group g)
    const char* base = a.gn_part + (size_t)bb * a.gn_mslots * WM * nc32 * 16 + (size_t)g * sub_n * 16;
    float s = 0.f, ss = 0.f;
    for (int l0 = 0; l0 < total; l0 += 64) {
      const int l = l0 + lane;
      const bool mine = l < total;
      const int ms_wm = mine ? l / sub_n : 0, sub = mine ? l - ms_wm * sub_n : 0;   // (mslot * WM + wm'), block inside the group
      const char* src = base + ((size_t)ms_wm * nc32 + sub) * 16;
      u32x4_t v;
      for (int z = 0; z < a.gn_nap0; ++z) __builtin_amdgcn_s_sleep(1);
      for (unsigned spins = 0;; ++spins) {
        load16_sc1_issue(v, src);
        wait_vm0();
        if (gst) { if (!gst->polls) gst->t_first = __builtin_amdgcn_s_memtime(); ++gst->polls; }
        if (__all(!mine || (v[0] == 1u && v[2] == 1u))) break;
        for (int z = 0; z < a.gn_nap; ++z) __builtin_amdgcn_s_sleep(1);
        // give up after 0.5 s -- or at once when an earlier launch of this call already has (the host-mapped flag is still up: the call's
        // result is void anyway, and every further fused launch waiting its own 0.5 s turned one stall into minutes).  The flag
        // lives in host memory (one PCIe round trip): it is looked at every 256th poll of the slow path only.
        const bool flagged = (spins & 255u) == 255u && a.fail_flag && __hip_atomic_load(a.fail_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
        if (flagged || wall_clock64() - t0 > 50000000ull) {   // raise the flag, poison the output
          if (lane == 0 && a.fail_flag && !flagged) __hip_atomic_store(a.fail_flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          s = __uint_as_float(0x7fc00000u);
          break;
        }
      }
      if (mine) {
        s += __uint_as_float(v[1]);
        ss += __uint_as_float(v[3]);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o);
      ss += __shfl_xor(ss, o);
    }
    mean = s * inv_n;
    rstd = rsqrtf(fmaxf(ss * inv_n - mean * mean, 0.0f) + 1e-5f);
  };
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col_wave0 + j * 32 + (lane & 31);
    const bool col_ok = col < a.n;
    const int cc = col_ok ? col : 0;
    const int g = (col_wave0 + j * 32 < a.n ? col_wave0 + j * 32 : 0) / cpg;   // wave-uniform
    const float gam = a.gn_gamma[cc], bet = a.gn_beta[cc];
    const float sc = a.gn_ss ? a.gn_ss[cc] + 1.0f : 1.0f, sh = a.gn_ss ? a.gn_ss[a.n + cc] : 0.0f;
    auto affine = [&](float mean, float rstd, float& a1, float& b1) {
      a1 = rstd * gam;
      b1 = bet - mean * a1;
      if (a.gn_ss) {
        a1 *= sc;
        b1 = b1 * sc + sh;
      }
    };
    float mean, rstd;
    gather(b_first, g, mean, rstd);
    affine(mean, rstd, ca0[j], cb0[j]);
    ca1[j] = ca0[j]; cb1[j] = cb0[j];
    if (b_first + 1 <= b_last) {
      gather(b_first + 1, g, mean, rstd);
      affine(mean, rstd, ca1[j], cb1[j]);
    }
    ca2[j] = ca1[j]; cb2[j] = cb1[j];
    if (b_first + 2 <= b_last) {
      gather(b_first + 2, g, mean, rstd);
      affine(mean, rstd, ca2[j], cb2[j]);
    }
  }
  if (gst) gst->t_done = __builtin_amdgcn_s_memtime();
  typedef typename std::conditional<RES, float, T>::type TS;   // staging element
  constexpr int RBS = TN * 32 * (int)sizeof(TS);    // staged bytes per tile row
  constexpr int PITCH = RBS + 16;
  constexpr int EPL = 16 / (int)sizeof(T);          // output elements per lane in the row phase (16 bytes)
  constexpr int LPR = TN * 32 / EPL;                // lanes per row
  constexpr int RPS = 64 / LPR;                     // rows per sweep
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const bool second = m_wave0 + row >= m_split, third = m_wave0 + row >= m_split2;
        const float v = fmaf(acc[i][j][r], wsc[j], bv[j]);
        const float o = fast_silu(fmaf(v, third ? ca2[j] : (second ? ca1[j] : ca0[j]), third ? cb2[j] : (second ? cb1[j] : cb0[j])));
        store_out<TS>(wave_lds, (size_t)(row * PITCH) / sizeof(TS) + j * 32 + (lane & 31), o);
      }
  // same wave wrote and reads: LDS executes a wave's operations in order
  const int rsub = lane / LPR, chunk = lane % LPR;
  const int col = col_wave0 + chunk * EPL;
  if (!RES) {
#pragma unroll
    for (int sw = 0; sw < TM * 32 / RPS; ++sw) {
      const int row = sw * RPS + rsub;
      const int m = m_wave0 + row;
      const uint4 v = *reinterpret_cast<const uint4*>(wave_lds + row * PITCH + chunk * 16);
      if (m < M && col < a.n) *reinterpret_cast<uint4*>(a.y + ((size_t)m * a.y_ld + col) * sizeof(T)) = v;
    }
  } else {
    constexpr int NSW = TM * 32 / RPS;
    u32x4_t rres[NSW];
    const bool has_res = a.residual != nullptr;   // (this staged-fp32 form also serves the fp8 output of a conv without residual)
    const bool out8 = (a.gn_out & 1) != 0;
#pragma unroll
    for (int sw = 0; sw < NSW; ++sw) {
      const int m = m_wave0 + sw * RPS + rsub;
      const bool ok = m < M && col < a.n;
      rres[sw] = u32x4_t{0u, 0u, 0u, 0u};
      if (has_res) {
        const uint4 t = *reinterpret_cast<const uint4*>(a.residual + ((size_t)(ok ? m : 0) * a.n + (ok ? col : 0)) * sizeof(T));
        rres[sw] = u32x4_t{t.x, t.y, t.z, t.w};
      }
    }
#pragma unroll
    for (int sw = 0; sw < NSW; ++sw) {
      const int row = sw * RPS + rsub;
      const int m = m_wave0 + row;
      float f[EPL], rr[EPL];
      const float* sp = reinterpret_cast<const float*>(wave_lds + row * PITCH + chunk * EPL * 4);
#pragma unroll
      for (int e = 0; e < EPL; e += 4) {
        const float4 t = *reinterpret_cast<const float4*>(sp + e);
        f[e] = t.x; f[e + 1] = t.y; f[e + 2] = t.z; f[e + 3] = t.w;
      }
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) rr[e] = __uint_as_float(rres[sw][e]);
      } else {
#pragma unroll
        for (int e = 0; e < EPL; e += 2) {
          rr[e] = __uint_as_float(rres[sw][e >> 1] << 16);
          rr[e + 1] = __uint_as_float(rres[sw][e >> 1] & 0xffff0000u);
        }
      }
#pragma unroll
      for (int e = 0; e < EPL; ++e) f[e] += rr[e];
      if (a.rowstat_out) {   // per-row partial of this lane group's 32-column block for the PreNorm of the attention block behind:
        // (sum, M2 about the block's own mean) of the values AS STORED (bf16-rounded: what the consumer's matmul reads), merged
        // Chan-style in the consumer's prologue -- a single-pass sum of squares cancels where |mean| >> std
        constexpr int G = 32 / EPL;   // lanes that share a row's 32-column block (consecutive lanes)
        float fr[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) fr[e] = (sizeof(T) == 2 && !(a.gn_out & 4)) ? bf16_to_f32(f32_to_bf16(f[e])) : f[e];
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) ps += fr[e];
#pragma unroll
        for (int o = 1; o < G; o <<= 1) ps += __shfl_xor(ps, o);
        const float bm = ps * (1.0f / 32.0f);
#pragma unroll
        for (int e = 0; e < EPL; ++e) pq = fmaf(fr[e] - bm, fr[e] - bm, pq);
#pragma unroll
        for (int o = 1; o < G; o <<= 1) pq += __shfl_xor(pq, o);
        if ((chunk & (G - 1)) == 0 && m < M && col < a.n)
          *reinterpret_cast<float2*>(a.rowstat_out + ((size_t)m * (a.n >> 5) + (col >> 5)) * 2) = make_float2(ps, pq);
      }
      if (a.gn_out & 4) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) f[e] = fast_tanh(f[e]);
      }
      uint4 v;
      if constexpr (sizeof(T) == 4) {
        v = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
      } else {
        v = make_uint4(hw_bf16x2(f[0], f[1]), hw_bf16x2(f[2], f[3]), hw_bf16x2(f[4], f[5]), hw_bf16x2(f[6], f[7]));
      }
      if constexpr (sizeof(T) == 2) {
        if (out8) {   // OCP e4m3, saturating, from the un-rounded value (as gn_apply's fp8 output: the consumer is an fp8 x fp8 conv)
          float c8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) c8[e] = fminf(fmaxf(f[e], -448.0f), 448.0f);
          int lo = 0, hi = 0;
          lo = __builtin_amdgcn_cvt_pk_fp8_f32(c8[0], c8[1], lo, false);
          lo = __builtin_amdgcn_cvt_pk_fp8_f32(c8[2], c8[3], lo, true);
          hi = __builtin_amdgcn_cvt_pk_fp8_f32(c8[4], c8[5], hi, false);
          hi = __builtin_amdgcn_cvt_pk_fp8_f32(c8[6], c8[7], hi, true);
          if (m < M && col < a.n) *reinterpret_cast<int2*>(a.y + (size_t)m * a.y_ld + col) = make_int2(lo, hi);
          continue;
        }
      }
      if (m < M && col < a.n) *reinterpret_cast<uint4*>(a.y + ((size_t)m * a.y_ld + col) * sizeof(T)) = v;
    }
  }
}

// Second output of a launch with a folded 1x1 conv (ConvKArgs::y2): acc + bias2, whole-row stores through the same LDS staging.
template <typename T, int TM, int TN, typename KA>
__device__ __forceinline__ void epilogue_rows_second(const KA& a, f32x16 (&acc)[TM][TN], char* wave_lds, int m_wave0,
                                                     int col_wave0, int M) {
  constexpr int RB = TN * 32 * (int)sizeof(T);
  constexpr int PITCH = RB + 16;
  constexpr int LPR = RB / 16;
  constexpr int RPS = 64 / LPR;
  const int lane = ldc_tid() & 63;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col_wave0 + j * 32 + (lane & 31);
    const float bv = (a.bias2 && col < a.n) ? a.bias2[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        store_out<T>(wave_lds, (size_t)(row * PITCH) / sizeof(T) + j * 32 + (lane & 31), acc[i][j][r] + bv);
      }
  }
  const int rsub = lane / LPR, chunk = lane % LPR;
  const int col = col_wave0 + chunk * (16 / (int)sizeof(T));
#pragma unroll
  for (int sw = 0; sw < TM * 32 / RPS; ++sw) {
    const int row = sw * RPS + rsub;
    const int m = m_wave0 + row;
    const uint4 v = *reinterpret_cast<const uint4*>(wave_lds + row * PITCH + chunk * 16);
    if (m < M && col < a.n) *reinterpret_cast<uint4*>(a.y2 + ((size_t)m * a.n + col) * sizeof(T)) = v;
  }
}

template <typename T, int TM, int TN, typename KA>
__device__ __forceinline__ void epilogue_rows_dispatch(const KA& a, f32x16 (&acc)[TM][TN], char* wave_lds, int m_wave0,
                                                       int col_wave0, int M, int m0, int BM, int WM = 1, int wm = 0, int m_base = 0) {
  const int lane = ldc_tid() & 63;
  if (a.gn_part) {   // fused GroupNorm apply (uniform over the launch)
    if (a.residual || (a.gn_out & 1)) epilogue_gn_fused<T, TM, TN, true>(a, acc, wave_lds, m_wave0, col_wave0, M, m0, BM, WM, wm, m_base);
    else epilogue_gn_fused<T, TM, TN, false>(a, acc, wave_lds, m_wave0, col_wave0, M, m0, BM, WM, wm, m_base);
    return;
  }
  if (a.gn_sum) epilogue_gn_stats<TM, TN>(a, acc, m_wave0 + 4 * (lane >> 5), col_wave0 + (lane & 31), m0, BM, M);
  if (a.colmax) epilogue_colmax<T, TM, TN>(a, acc, m_wave0 + 4 * (lane >> 5), col_wave0 + (lane & 31), m0, BM, M);
  if (a.residual) {   // (fp32 staging: the launcher sized the LDS for it)
    if (a.post_act == ACT_NONE) epilogue_rows_res<T, TM, TN, false>(a, acc, wave_lds, m_wave0, col_wave0, M);
    else epilogue_rows_res<T, TM, TN, true>(a, acc, wave_lds, m_wave0, col_wave0, M);
  } else {
    if (a.post_act == ACT_NONE) epilogue_rows<T, TM, TN, false, false>(a, acc, wave_lds, m_wave0, col_wave0, M);
    else epilogue_rows<T, TM, TN, false, true>(a, acc, wave_lds, m_wave0, col_wave0, M);
  }
}

// conv_fast_{bf16,f32}.hip
extern unsigned long long* g_conv_stamps;   // tuning aid, set by ldc_conv_microbench when LDC_CONV_STAMPS is on
bool conv_fast_eligible(const ConvLayer& ly);
hipError_t launch_conv_fast(const ConvLayer& ly, const ConvKArgs& a, int M, int span_rows, hipStream_t s, bool* launched);
hipError_t launch_conv_fast_fp8(const ConvLayer& ly, const ConvKArgs& a_in, int M, int span_rows, hipStream_t s, bool* launched);
hipError_t launch_conv_fast_bf16w8(const ConvLayer& ly, const ConvKArgs& a_in, int M, int span_rows, hipStream_t s, bool* launched);

}  // namespace ldc
