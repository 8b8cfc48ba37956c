// LaDiffCodec on MI355X -- training GEMMs of the diffusion UNet on the bf16 MFMA with fp32-class accuracy (round 3).
//
// What this replaces: the three GEMM shapes of every Conv1d of Unet1D under `loss.backward()` (reference: srcs/modules/unet.py:422-469
// through torch autograd; srcs/train.py:170-205 is the step).  Round 2 ran them on v_mfma_f32_32x32x2_f32 (convmm_kernel in train.hip,
// kept as LDC_TRAIN_FP32_MFMA=1): exact fp32, but that instruction is 1/16 of the bf16 rate and the kernel reached a third of it.
//
// Here every fp32 operand a is split into two bf16 numbers, a = hi + lo + O(2^-17 |a|) with hi = bf16(a), lo = bf16(a - hi), and a
// product is three bf16 MFMAs accumulated in fp32:   a b ~= hi_a hi_b + hi_a lo_b + lo_a hi_b   (the dropped lo_a lo_b and the two
// residuals are 2^-16 |a b| each, random in sign: 4-5e-6 of a tensor's maximum against float64 conv1d where the exact-fp32 MFMA gives 1e-6,
// inside the 1e-4 the training parity tests carry against the reference's own autograd).  Three v_mfma_f32_32x32x16_bf16 cost
// 96 cycles for 16 reduction steps where the fp32 MFMA takes 512: the bound moves from the matrix pipe to staging, so the kernel is
// built around that:
//   * 128 x 128 output tile per workgroup (eight waves of 32 x 64 = 1 x 2 MFMA blocks: 127 registers, four waves per SIMD; LDC_MM3_NW=4
//     selects 2 x 2 waves of 64 x 64), reduction chunks of 32, LDS double-buffered
//     (4 planes -- A hi / A lo / B hi / B lo -- of 128 rows x 64 B, 16-byte slots XOR-swizzled by (row >> 2) & 3: fragment reads and
//     staging writes are conflict-free ds_read/write_b128), one barrier per chunk, two register sets of global loads in flight (chunks
//     i + 1 and i + 2 under the MFMAs of chunk i) with branch-free fetches -- see the notes at the loop;
//   * the batch is folded into the GEMM's N (columns = (item, position)), so the L = 75 / 150 levels fill their tiles;
//   * weights are split once per use by a pack kernel into [tap][row][k/8][hi x 8 | lo x 8] (rows and k zero-padded to the tile), so
//     the A operand of forward / dX is four 16-byte loads per thread and chunk and needs no arithmetic;
//   * activations ([B, C, L] fp32, the training layout) are gathered with the positions along the lanes (coalesced), 8 or 16 channels per
//     thread, split in registers and written as whole 16-byte k-vectors;
//   * no atomics: split reductions (dW over items, forward / dX over taps x channels when the tiles do not fill the chip, the bias
//     gradient) write per-part partials that a second kernel sums in order -- device-scope fp32 atomics cost 43 % of the dW kernel.
//   MODE 0 forward  y  [Cout x (B Lout)] = sum_t  W_t   [Cout x Cin]  . x  shifted by t    (+ bias)
//   MODE 1 dX       dx [Cin  x (B Lin)]  = sum_t  W_t^T [Cin x Cout]  . dy shifted by -t
//   MODE 2 dW       dW_t [Cout x Cin]    = sum_b  dy[b] [Cout x Lout] . x[b]^T shifted by t     (grid z = tap * nsplit + part)
// Roofline: MFMA-bound in the limit (3 x the bf16 flops: 2.5 PFLOP/s / 3 = 833 TFLOP/s of fp32-equivalent work); measured numbers in
// DESIGN.md (training row) and profiles/.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "ldc_kernels.h"

extern long long g_device_syncs;   // ldc_api.cpp: every device-wide synchronisation of the library is counted (ldc_debug_sync_count)

namespace ldc {
namespace mm3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // (not HIP's u32x4: a struct of unions, which kept register arrays of it in scratch / LDS)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned: the hardware takes unaligned dwordx4 loads
union V8 {
  u32x4 u;
  bf16x8 v;
  __bf16 e[8];
};
union V4 {
  u32x2 u;
  __bf16 e[4];
};

// values are passed one by one: an array parameter makes the compiler promote the caller's staging registers to LDS
#define LDC_SPLIT1(f, H, L, j) { const __bf16 a_ = (__bf16)(f); (H).e[j] = a_; (L).e[j] = (__bf16)((f) - (float)a_); }
__device__ __forceinline__ void split8(float f0, float f1, float f2, float f3, float f4, float f5, float f6, float f7, u32x4& hi, u32x4& lo) {
  V8 h, l;
  LDC_SPLIT1(f0, h, l, 0) LDC_SPLIT1(f1, h, l, 1) LDC_SPLIT1(f2, h, l, 2) LDC_SPLIT1(f3, h, l, 3)
  LDC_SPLIT1(f4, h, l, 4) LDC_SPLIT1(f5, h, l, 5) LDC_SPLIT1(f6, h, l, 6) LDC_SPLIT1(f7, h, l, 7)
  hi = h.u;
  lo = l.u;
}
__device__ __forceinline__ void split4(float f0, float f1, float f2, float f3, u32x2& hi, u32x2& lo) {
  V4 h, l;
  LDC_SPLIT1(f0, h, l, 0) LDC_SPLIT1(f1, h, l, 1) LDC_SPLIT1(f2, h, l, 2) LDC_SPLIT1(f3, h, l, 3)
  hi = h.u;
  lo = l.u;
}

// w [Cout][Cin][K] fp32 -> pw [K][MP][RP / 8][hi x 8 | lo x 8] bf16.  TR = 0: rows = Cout, k = Cin (forward); TR = 1: rows = Cin, k = Cout (dX).
template <int TR>
__global__ __launch_bounds__(256) void mm3_pack_kernel(const float* w, int Cin, int Cout, int K, int MP, int RP, u32x4* pw) {
  const int M = TR ? Cin : Cout, R = TR ? Cout : Cin, oct = RP / 8;
  const long total = (long)K * MP * oct;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    // TR = 0: the octet index runs fastest (consecutive threads read consecutive channels of one output row);
    // TR = 1: the row (= input channel) runs fastest within a tap, for the same reason
    int t, row, q;
    if (TR == 0) { q = (int)(idx % oct); row = (int)((idx / oct) % MP); t = (int)(idx / ((long)oct * MP)); }
    else { row = (int)(idx % MP); q = (int)((idx / MP) % oct); t = (int)(idx / ((long)oct * MP)); }
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = q * 8 + j;
      const bool ok = row < M && k < R;
      const int o = TR ? k : row, i = TR ? row : k;
      f[j] = ok ? w[((size_t)o * Cin + i) * K + t] : 0.f;
    }
    u32x4 hi, lo;
    split8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], hi, lo);
    u32x4* dst = pw + (((size_t)t * MP + row) * oct + q) * 2;
    dst[0] = hi;
    dst[1] = lo;
  }
}

// TAIL (MODE 0 / 1): the reduction extent is not a multiple of 32 (never in the UNet itself): the last chunk's loads are clamped and masked;
// TAIL (MODE 2): Lout is not a multiple of 4 (the L = 150 / 75 levels): dy is read as dwords instead of dwordx4
// VECB (MODE 2): stride 1 -- x is read as one (unaligned) dwordx4 per pass from a start clamped into the row; the threads whose four
// positions straddle the padding (first / last of a row) get them shifted, which store() undoes for exactly those lanes
// NW: waves per workgroup (4: 2 x 2 waves of 64 x 64; 8: 4 x 2 waves of 32 x 64 -- half the registers per wave, four waves per SIMD)
// ONE: plain bf16 training GEMMs (hi terms only: one MFMA per product, two LDS planes; 2^-9-class products, the numerics of a bf16
// autocast run -- opt-in, LDC_TRAIN_BF16=1 / option train_bf16, not what the parity tests pin)
template <int MODE, bool TAIL, bool VECB = false, int NW = 4, bool ONE = false>
__global__ __launch_bounds__(64 * NW, NW / 2) void mm3_kernel(const void* __restrict__ Asrc, const float* __restrict__ Bsrc, const float* __restrict__ bias,
                                                     float* __restrict__ out, int B, int Cin, int Cout, int Lin, int Lout, int K, int S, int P,
                                                     int nsplit, int MP, int RP, int xmap) {
  __shared__ u32x4 lds[2][4][512];   // [stage][A hi, A lo, B hi, B lo][row * 4 + (slot ^ swizzle)]
  constexpr int NT = 64 * NW;          // threads
  constexpr int TMW = 8 / NW;          // 32-row blocks per wave (the wave tile is 32 TMW x 64)
  constexpr int KPT = 64 / NW;         // MODE 0 / 1: reduction indices per thread and chunk (two octets or one)
  constexpr int PASSES = 16 / NW;      // MODE 2: row passes per chunk (8 NW rows each)
  constexpr int RPP = 8 * NW;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1, i32 = lane & 31, g = lane >> 5;
  const int M = MODE == 1 ? Cin : Cout;
  const int Lcol = MODE == 0 ? Lout : Lin;                       // MODE 0 / 1: positions per item along N
  const int N = MODE == 2 ? Cin : B * Lcol;
  // Logical block coordinates.  1-D grids (gridDim.y == 1 with more than one row tile, set by the launchers) are XCD-aware: workgroup h
  // runs on XCD h % 8 (observed dispatch rule, conv_fast.inc uses it too; used for speed only), and the eight L2s do not share.
  //   MODE 0 / 1: XCD x takes a contiguous run of column tiles for ALL row tiles, columns fastest -- its activation columns are
  //     fetched into ONE L2 and the weight tile of a row stays there across its columns;
  //   MODE 2 (experiment only, slower): all tiles and taps of a part (= the same dy / x items) on one XCD, taps fastest.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const int Mt = (M + 127) / 128, Nt = (N + 127) / 128;
  if (gridDim.y == 1 && (Mt > 1 || MODE == 2) && xmap) {
    const int hh = blockIdx.x, xcd = hh & 7, slot = hh >> 3;
    if (MODE == 2) {
      const int tpp = Nt * Mt * K, pg = slot / tpp, tl = slot - pg * tpp, prt = pg * 8 + xcd;
      if (prt >= nsplit) return;
      const int tap = tl % K, rest = tl / K;
      bx = rest % Nt; by = rest / Nt; bz = tap * nsplit + prt;
    } else {
      const int W = (Nt + 7) / 8, mrow = slot / W, ncol = xcd * W + (slot - mrow * W);
      if (ncol >= Nt) return;
      bx = ncol; by = mrow;
    }
  }
  const int m0 = by * 128, n0 = bx * 128;
  const int z = MODE == 2 ? bz / nsplit : 0, part = MODE == 2 ? bz % nsplit : 0;
  const int red = MODE == 0 ? Cin : (MODE == 1 ? Cout : Lout);    // inner reduction extent
  const int Lsrc = MODE == 0 ? Lin : Lout;                        // MODE 0 / 1: row length of the gathered activation
  // reduction steps of this workgroup, flattened as (outer, chunk): MODE 2 takes the items of its part (grid z = tap * nsplit + part),
  // MODE 0 / 1 an even share of the (tap, channel chunk) steps when the output tiles alone do not fill the chip (grid z = part;
  // parts write their partial outputs to a workspace, mm3_sum_parts_kernel adds them in order: no atomics anywhere, the step is
  // bit-reproducible)
  const int nchunk = (red + 31) / 32;
  const int outer_lo = MODE == 2 ? (int)((long long)B * part / nsplit) : 0;
  const int outer_hi = MODE == 2 ? (int)((long long)B * (part + 1) / nsplit) : K;
  const int kpart = MODE == 2 ? 0 : bz;
  const int it_all = (outer_hi - outer_lo) * nchunk;
  const int it_begin = MODE == 2 ? 0 : (int)((long long)it_all * kpart / nsplit);
  const int n_it = (MODE == 2 ? it_all : (int)((long long)it_all * (kpart + 1) / nsplit)) - it_begin;

  f32x16 acc[TMW][2];
#pragma unroll
  for (int a = 0; a < TMW; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b2][r] = 0.f;

  // ---- staging ----
  // MODE 0 / 1: thread -> (row / column r = tid & 127, k half h = tid >> 7: 16 reduction indices = LDS slots 2h, 2h + 1)
  // MODE 2:     thread -> (row 32 p + (tid >> 3), four consecutive positions 4 (tid & 7) .. + 3 = half of slot (tid & 7) >> 1), p = 0..3
  // Two register sets: a chunk is loaded two iterations before it is written to LDS (one chunk of MFMAs, 0.3 us, does not cover an
  // HBM / far-L2 round trip with two workgroups per CU).
  struct Stage {
    u32x4 a0, a1, a2, a3;   // MODE 0 / 1: packed weights (hi, lo of two k-octets)
    float a[4 * PASSES];    // MODE 2: dy
    float b[KPT > 4 * PASSES ? KPT : 4 * PASSES];   // activations
    int nv;                 // TAIL: valid reduction indices of this thread's 16
    bool ok;                // MODE 0 / 1: this thread's column is inside the item for the chunk's tap (applied when the set is
                            // written to LDS: a select right after the load would wait for it inside fetch)
  };
  const int r128 = tid & 127, h = tid >> 7;
  int cb = 0, cl = 0;       // MODE 0 / 1: (item, position) of this thread's column
  bool cvalid = false;
  if (MODE != 2) {
    const int n = n0 + r128;
    cvalid = n < N;
    cb = cvalid ? n / Lcol : 0;
    cl = cvalid ? n - cb * Lcol : 0;
  }
  const u32x4* pA = nullptr;
  const float* pB = nullptr;
  bool bok = false;
  // MODE 2: per-pass row pointers (advanced by a chunk of positions), row predicates
  // (32-bit element offsets from the item's wave-uniform base: eight 64-bit pointers spilled)
  int qa[PASSES], qb[PASSES];
  int a_last = 0;
  const float* baseA = nullptr;
  const float* baseB = nullptr;
  bool oka[PASSES], okb[PASSES];
  int f_outer = outer_lo + it_begin / nchunk, f_chunk = it_begin % nchunk;

  auto setup_outer = [&](int outer) {
    if (MODE != 2) {
      pA = (const u32x4*)Asrc + (((size_t)outer * MP + m0 + r128) * (RP / 8) + (KPT / 8) * h) * 2;
      int pos;
      if (MODE == 0) {
        pos = cl * S + outer - P;
        bok = cvalid && pos >= 0 && pos < Lin;
      } else {
        const int u = cl + P - outer;
        bok = cvalid && u >= 0 && (S == 1 || u % S == 0);
        pos = bok ? (S == 1 ? u : u / S) : 0;
        bok = bok && pos < Lout;
      }
      pB = Bsrc + ((size_t)cb * red + KPT * h) * Lsrc + (bok ? pos : 0);
      pA += 8 * f_chunk;                       // (non-zero only for the first step of a split reduction)
      pB += (size_t)32 * f_chunk * Lsrc;
    } else {
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const int row = RPP * p + (tid >> 3);
        const int o = m0 + row, i = n0 + row;
        oka[p] = o < Cout;
        okb[p] = i < Cin;
        qa[p] = (oka[p] ? o : 0) * Lout + 4 * (tid & 7);
        qb[p] = (okb[p] ? i : 0) * Lin;
      }
      baseA = (const float*)Asrc + (size_t)outer * Cout * Lout;
      a_last = Cout * Lout - 4;
      baseB = Bsrc + (size_t)outer * Cin * Lin;
    }
  };
  // Loads are unconditional wherever the address is known to be inside the tensor (the value is then selected against the padding /
  // tail predicate): a predicated load costs an exec-mask branch each, and sixteen of them per chunk showed in the issue slots.
  auto fetch = [&](Stage& r) {
    const int k0 = f_chunk * 32;
    if (MODE != 2) {
      r.a0 = pA[0];
      if (!ONE) r.a1 = pA[1];
      if (KPT == 16) { r.a2 = pA[2]; if (!ONE) r.a3 = pA[3]; }
      pA += 8;
      // straight-line on purpose: with a branch in here the compiler's s_waitcnt bookkeeping falls back to "everything older", which
      // drains the set that was loaded one iteration ago together with the one just issued
      r.ok = bok;
      r.nv = red - (k0 + KPT * h);
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const float* q = pB + (size_t)j * Lsrc;
        if (TAIL) q = j < r.nv ? q : Bsrc;
        r.b[j] = *q;
      }
      pB += (size_t)32 * Lsrc;
    } else {
      // dW, branch-free as well: dy as one dwordx4 per pass when the rows keep it dword-x4-safe to clamp (TAIL == false: Lout % 4 == 0,
      // so a thread's four positions are all inside or all outside the row, and the clamp to the end of the tensor only ever moves
      // fully masked ones), else four dwords clamped into the row; x always as dwords clamped into the row (the tap shift makes the
      // first / last thread of a row straddle the padding).  Masks are applied in store() from the chunk origin kept with the set.
      const int lq = k0 + 4 * (tid & 7);
      r.nv = lq;
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        if (!TAIL) {
          const f32x4u v = *(const f32x4u*)(baseA + min(qa[p] + k0, a_last));
#pragma unroll
          for (int e = 0; e < 4; ++e) r.a[4 * p + e] = v[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) r.a[4 * p + e] = baseA[qa[p] - 4 * (tid & 7) + min(lq + e, Lout - 1)];
        }
        if (VECB) {
          const f32x4u v = *(const f32x4u*)(baseB + qb[p] + min(max(lq + z - P, 0), Lin - 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) r.b[4 * p + e] = v[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) r.b[4 * p + e] = baseB[qb[p] + min(max((lq + e) * S + z - P, 0), Lin - 1)];
        }
      }
    }
    if (++f_chunk == nchunk) {
      f_chunk = 0;
      if (++f_outer < outer_hi) setup_outer(f_outer);
    }
  };
  float rs[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) rs[p] = 0.f;   // MODE 2, bias gradient: sums of this thread's dy values per row pass
  const bool do_db = MODE == 2 && bias != nullptr && bx == 0 && z == 0;
  auto store = [&](const Stage& r, int st) {
    if (MODE != 2) {
      const int sw = (r128 >> 2) & 3;
      constexpr int NOCT = KPT / 8;                       // octets (16-byte LDS slots) per thread: 2 or 1
      lds[st][0][r128 * 4 + ((NOCT * h) ^ sw)] = r.a0;
      if (!ONE) lds[st][1][r128 * 4 + ((NOCT * h) ^ sw)] = r.a1;
      if (NOCT == 2) {
        lds[st][0][r128 * 4 + ((2 * h + 1) ^ sw)] = r.a2;
        if (!ONE) lds[st][1][r128 * 4 + ((2 * h + 1) ^ sw)] = r.a3;
      }
      u32x4 hi, lo;
      float b[KPT];
#pragma unroll
      for (int j = 0; j < KPT; ++j) b[j] = (r.ok && (!TAIL || j < r.nv)) ? r.b[j] : 0.f;
      split8(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], hi, lo);
      lds[st][2][r128 * 4 + ((NOCT * h) ^ sw)] = hi;
      if (!ONE) lds[st][3][r128 * 4 + ((NOCT * h) ^ sw)] = lo;
      if (NOCT == 2) {
        split8(b[KPT - 8], b[KPT - 7], b[KPT - 6], b[KPT - 5], b[KPT - 4], b[KPT - 3], b[KPT - 2], b[KPT - 1], hi, lo);
        lds[st][2][r128 * 4 + ((2 * h + 1) ^ sw)] = hi;
        if (!ONE) lds[st][3][r128 * 4 + ((2 * h + 1) ^ sw)] = lo;
      }
    } else {
      const int slot = (tid & 7) >> 1, half = tid & 1;
      const int lq = r.nv, pos0 = lq * S + z - P;
      // (uniform per chunk up to the halo) interior: every position of the chunk is inside both rows
      const bool edge = lq + 4 > Lout || pos0 < 0 || pos0 + 3 * S >= Lin;
      float ma[4], mb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ma[e] = lq + e < Lout ? 1.f : 0.f;
        const int pos = pos0 + e * S;
        mb[e] = (pos >= 0 && pos < Lin) ? 1.f : 0.f;
      }
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const int row = RPP * p + (tid >> 3), sw = (row >> 2) & 3;
        const int at = (row * 4 + (slot ^ sw)) * 2 + half;     // in 8-byte units
        u32x2 hi, lo;
        // (rows past M / N are clamped reads of row 0: they only reach output rows / columns that are never stored; the clamped
        // loads are of real tensor data, so the 0 / 1 multiply is a safe mask)
        if (edge) {
          const float a0 = r.a[4 * p] * ma[0], a1 = r.a[4 * p + 1] * ma[1], a2 = r.a[4 * p + 2] * ma[2], a3 = r.a[4 * p + 3] * ma[3];
          if (do_db) rs[p] += (a0 + a1) + (a2 + a3);
          split4(a0, a1, a2, a3, hi, lo);
        } else {
          if (do_db) rs[p] += (r.a[4 * p] + r.a[4 * p + 1]) + (r.a[4 * p + 2] + r.a[4 * p + 3]);
          split4(r.a[4 * p], r.a[4 * p + 1], r.a[4 * p + 2], r.a[4 * p + 3], hi, lo);
        }
        ((u32x2*)lds[st][0])[at] = hi;
        if (!ONE) ((u32x2*)lds[st][1])[at] = lo;
        if (edge) {
          float b0 = r.b[4 * p], b1 = r.b[4 * p + 1], b2 = r.b[4 * p + 2], b3 = r.b[4 * p + 3];
          if (VECB) {   // loaded from pos0 + d (d = clamp shift): element e is loaded[e - d]
            const int d = min(max(pos0, 0), Lin - 4) - pos0;
            const float l0 = b0, l1 = b1, l2 = b2, l3 = b3;
            auto pick = [&](int k) { return k == 0 ? l0 : (k == 1 ? l1 : (k == 2 ? l2 : (k == 3 ? l3 : 0.f))); };
            b0 = pick(0 - d); b1 = pick(1 - d); b2 = pick(2 - d); b3 = pick(3 - d);
          }
          split4(b0 * mb[0], b1 * mb[1], b2 * mb[2], b3 * mb[3], hi, lo);
        } else {
          split4(r.b[4 * p], r.b[4 * p + 1], r.b[4 * p + 2], r.b[4 * p + 3], hi, lo);
        }
        ((u32x2*)lds[st][2])[at] = hi;
        if (!ONE) ((u32x2*)lds[st][3])[at] = lo;
      }
    }
  };
  const int swl = (i32 >> 2) & 3;
  auto compute = [&](int st) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int slot = ((2 * ks + g) ^ swl);
      V8 ah[TMW], al[TMW], bh[2], bl[2];
#pragma unroll
      for (int a = 0; a < TMW; ++a) {
        const int row = 32 * TMW * wm + 32 * a + i32;
        ah[a].u = lds[st][0][row * 4 + slot];
        if (!ONE) al[a].u = lds[st][1][row * 4 + slot];
      }
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const int row = 64 * wn + 32 * b2 + i32;
        bh[b2].u = lds[st][2][row * 4 + slot];
        if (!ONE) bl[b2].u = lds[st][3][row * 4 + slot];
      }
      // term-major: independent accumulators between two MFMAs into the same one
#pragma unroll
      for (int a = 0; a < TMW; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) if (!ONE) acc[a][b2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a].v, bh[b2].v, acc[a][b2], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < TMW; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) if (!ONE) acc[a][b2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a].v, bl[b2].v, acc[a][b2], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < TMW; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) acc[a][b2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a].v, bh[b2].v, acc[a][b2], 0, 0, 0);
    }
  };

  if (true) {
    Stage X, Y;
    // iteration `it`: LDS[it & 1] holds chunk it, X / Y hold chunks it + 1 and it + 2 (X the odd ones)
    int it = 0;
    if (n_it >= 5) {
      // steady state, its prologue AND its loop inside one branch, without conditions: every path on which a fetch may not have been
      // issued makes the compiler's s_waitcnt bookkeeping assume it was not, i.e. drain the set that was issued an iteration ago
      // TOGETHER with the one just issued (seen in the ISA as vmcnt(15..0) where vmcnt(35..20) is what the data flow needs)
      setup_outer(f_outer);
      fetch(X);
      store(X, 0);
      fetch(X);
      fetch(Y);
      __syncthreads();
      for (; it + 4 < n_it; it += 2) {
        compute(0);
        store(X, 1);
        fetch(X);
        __syncthreads();
        compute(1);
        store(Y, 0);
        fetch(Y);
        __syncthreads();
      }
    } else {
      if (n_it > 0) {
        setup_outer(f_outer);
        fetch(X);
        store(X, 0);
        if (n_it > 1) fetch(X);
        if (n_it > 2) fetch(Y);
      }
      __syncthreads();
    }
    for (; it < n_it; it += 2) {
      compute(0);
      if (it + 1 < n_it) store(X, 1);
      if (it + 3 < n_it) fetch(X);
      __syncthreads();
      if (it + 1 >= n_it) break;
      compute(1);
      if (it + 2 < n_it) store(Y, 0);
      if (it + 4 < n_it) fetch(Y);
      __syncthreads();
    }
  } else {
    // dW: one register set (two do not fit 256 registers next to the row offsets): the loads of chunk it + 1 fly under chunk it
    Stage X;
    if (n_it > 0) {
      setup_outer(f_outer);
      fetch(X);
      store(X, 0);
    }
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
      if (it + 1 < n_it) fetch(X);
      compute(it & 1);
      if (it + 1 < n_it) store(X, (it & 1) ^ 1);
      __syncthreads();
    }
  }

  if (do_db) {   // bias gradient of this part: db[part][o] = sum over its items and positions of dy (the eight lanes of a row meet by shuffles)
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      float v = rs[p];
      v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
      const int o = m0 + RPP * p + (tid >> 3);
      if ((tid & 7) == 0 && o < Cout) const_cast<float*>(bias)[(size_t)part * Cout + o] = v;   // (`bias` = db, or the parts' rows of the workspace)
    }
  }
  // C/D layout: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int b2 = 0; b2 < 2; ++b2) {
    const int n = n0 + 64 * wn + 32 * b2 + i32;
    if (n >= N) continue;
    float* colp;
    size_t rstride;
    if (MODE == 2) {
      if (nsplit > 1) {   // partial tile of this part: workspace [part][tap][Cout][Cin], summed in fixed order by mm3_dw_reduce_kernel
        colp = out + (((size_t)part * K + z) * Cout) * Cin + n;
        rstride = (size_t)Cin;
      } else {
        colp = out + (size_t)n * K + z;
        rstride = (size_t)Cin * K;
      }
    } else {
      const int bb = n / Lcol, l = n - bb * Lcol;
      colp = out + (size_t)kpart * B * M * Lcol + (size_t)bb * M * Lcol + l;   // (split reduction: `out` is the parts' workspace)
      rstride = (size_t)Lcol;
    }
#pragma unroll
    for (int a = 0; a < TMW; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * TMW * wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (m >= M) continue;
        const float v = acc[a][b2][r];
        const float add = (MODE != 2 && bias && kpart == 0) ? bias[m] : 0.f;
        colp[(size_t)m * rstride] = v + add;
      }
  }
}

// dW of a split reduction: dw[o][i][t] = sum over parts of ws[part][t][o][i], parts in order (the fp32 atomics this replaces cost 43 % of
// the dW kernel at 32 parts -- device-scope atomics go to memory, the eight L2s are not coherent -- and made dW differ from run to run)
__global__ __launch_bounds__(256) void mm3_dw_reduce_kernel(const float* ws, int nsplit, int K, int Cout, int Cin, float* dw, const float* db_ws,
                                                            float* db) {
  const size_t plane = (size_t)Cout * Cin, total = plane * K;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int t = (int)(idx / plane);
    const size_t mn = idx - (size_t)t * plane;
    float sacc = 0.f;
    for (int p = 0; p < nsplit; ++p) sacc += ws[((size_t)p * K + t) * plane + mn];
    dw[mn * K + t] = sacc;
  }
  if (db)
    for (int o = blockIdx.x * 256 + threadIdx.x; o < Cout; o += gridDim.x * 256) {
      float sacc = 0.f;
      for (int p = 0; p < nsplit; ++p) sacc += db_ws[(size_t)p * Cout + o];
      db[o] = sacc;
    }
}
// forward / dX of a split reduction: out = sum over parts (in order) of the partial outputs
__global__ __launch_bounds__(256) void mm3_sum_parts_kernel(const float* ws, int np, size_t n, float* out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float sacc = 0.f;
    for (int p = 0; p < np; ++p) sacc += ws[(size_t)p * n + i];
    out[i] = sacc;
  }
}

}  // namespace mm3
using namespace mm3;
namespace {
// packed-weight workspace (one training context per process; grows on demand, never inside a capture).  Process-wide and bound to
// the device it was first allocated on: a call from another device is refused (nullptr -> hipErrorOutOfMemory at the caller), and
// growth waits for the WHOLE device, not only the calling stream, before the old buffer is freed (another trainer's stream may
// still be reading it).
int g_ws_device = -1;
bool ws_device_ok() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (g_ws_device < 0) g_ws_device = dev;
  return dev == g_ws_device;
}
// Two lanes of every workspace: lane 1 serves the launches on the training side stream (train.hip: g_train_side_stream -- parameter
// gradients and the time-embedding branch's backward run there concurrently with the main stream's dX chain), so that the two streams
// never share a buffer.
inline int lane_of(hipStream_t s) { return (g_train_side_stream && s == g_train_side_stream) ? 1 : 0; }
void* g_pw[2] = {nullptr, nullptr};
size_t g_pw_bytes[2] = {0, 0};
u32x4* pack_workspace(size_t bytes, hipStream_t s) {
  if (!ws_device_ok()) return nullptr;
  const int lane = lane_of(s);
  if (bytes > g_pw_bytes[lane]) {
    // (counted like every device-wide sync of the library: ldc_debug_sync_count; a failed sync means work may still read the buffer)
    if (hipStreamSynchronize(s) != hipSuccess) return nullptr;
    ++g_device_syncs;
    if (hipDeviceSynchronize() != hipSuccess) return nullptr;
    if (g_pw[lane]) (void)hipFree(g_pw[lane]);
    g_pw[lane] = nullptr;
    g_pw_bytes[lane] = 0;
    const size_t want = std::max(bytes + bytes / 2, (size_t)32 << 20);
    if (hipMalloc(&g_pw[lane], want) != hipSuccess) return nullptr;
    g_pw_bytes[lane] = want;
  }
  return (u32x4*)g_pw[lane];
}
// partial dW tiles / split forward-dX reductions (same ownership rules as g_pw; a lane per stream, see above)
void* g_dw_ws[2] = {nullptr, nullptr};
size_t g_dw_ws_bytes[2] = {0, 0};
float* dw_workspace(size_t bytes, hipStream_t s) {
  if (!ws_device_ok()) return nullptr;
  const int lane = lane_of(s);
  if (bytes > g_dw_ws_bytes[lane]) {
    if (hipStreamSynchronize(s) != hipSuccess) return nullptr;
    ++g_device_syncs;
    if (hipDeviceSynchronize() != hipSuccess) return nullptr;
    if (g_dw_ws[lane]) (void)hipFree(g_dw_ws[lane]);
    g_dw_ws[lane] = nullptr;
    g_dw_ws_bytes[lane] = 0;
    const size_t want = std::max(bytes + bytes / 4, (size_t)64 << 20);
    if (hipMalloc(&g_dw_ws[lane], want) != hipSuccess) return nullptr;
    g_dw_ws_bytes[lane] = want;
  }
  return (float*)g_dw_ws[lane];
}
inline int up(int v, int q) { return (v + q - 1) / q * q; }
// waves per workgroup of the GEMM kernels (LDC_MM3_NW = 4 | 8; see mm3_kernel)
// XCD-aware 1-D grids (LDC_MM3_XCD=0 restores the 3-D ones)
inline int mm3_xcd() {
  static const int on = getenv("LDC_MM3_XCD") ? atoi(getenv("LDC_MM3_XCD")) : 1;
  return on;
}
inline int mm3_nw() {
  static const int nw = getenv("LDC_MM3_NW") ? atoi(getenv("LDC_MM3_NW")) : 8;   // 8 measured 3 % faster per step (dW 10 %)
  return nw == 4 ? 4 : 8;
}
// forward / dX: parts of the (tap, channel chunk) reduction per output tile.  The chip holds 512 workgroups (two per CU); below
// ~0.75 of that the tiles are split, as long as a part keeps >= 16 steps (the two-deep prefetch needs a few to reach its stride).
inline int reduction_split(long tiles, int steps) {
  static const int forced = getenv("LDC_MM3_KSPLIT") ? atoi(getenv("LDC_MM3_KSPLIT")) : 0;
  if (forced > 0) return std::max(1, std::min(forced, steps));
  if (tiles >= 384) return 1;
  const int want = (int)((512 + tiles - 1) / tiles);
  return std::max(1, std::min({want, steps / 16, 8}));
}

}  // namespace

hipError_t launch_mm3_forward(const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int Lin, int Lout, int K, int S, int P,
                              float* y, hipStream_t s) {
  const int MP = up(Cout, 128), RP = up(Cin, 32);
  u32x4* pw = pack_workspace((size_t)K * MP * RP * 4, s);
  if (!pw) return hipErrorOutOfMemory;
  const long total = (long)K * MP * (RP / 8);
  hipLaunchKernelGGL((mm3_pack_kernel<0>), dim3((unsigned)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, s, w, Cin, Cout, K, MP, RP, pw);
  const long N = (long)B * Lout;
  const int ks = reduction_split((long)((N + 127) / 128) * (MP / 128), K * (RP / 32));
  const size_t n_out = (size_t)B * Cout * Lout;
  float* target = y;
  if (ks > 1) {
    target = dw_workspace((size_t)ks * n_out * sizeof(float), s);
    if (!target) return hipErrorOutOfMemory;
  }
  const int Ntl = (int)((N + 127) / 128), Mtl = MP / 128;
  const int xm = mm3_xcd() && Mtl > 1;
  const dim3 grid = xm ? dim3((unsigned)(8 * ((Ntl + 7) / 8) * Mtl), 1, ks) : dim3((unsigned)Ntl, Mtl, ks);
  if (Cin % 32 == 0) {
    if (g_train_bf16) hipLaunchKernelGGL((mm3_kernel<0, false, false, 8, true>), grid, dim3(512), 0, s, (const void*)pw, x, bias, target, B, Cin, Cout, Lin, Lout, K, S, P, ks, MP, RP, xm);
    else if (mm3_nw() == 8) hipLaunchKernelGGL((mm3_kernel<0, false, false, 8>), grid, dim3(512), 0, s, (const void*)pw, x, bias, target, B, Cin, Cout, Lin, Lout, K, S, P, ks, MP, RP, xm);
    else hipLaunchKernelGGL((mm3_kernel<0, false, false, 4>), grid, dim3(256), 0, s, (const void*)pw, x, bias, target, B, Cin, Cout, Lin, Lout, K, S, P, ks, MP, RP, xm);
  } else
    hipLaunchKernelGGL((mm3_kernel<0, true>), grid, dim3(256), 0, s, (const void*)pw, x, bias, target, B, Cin, Cout, Lin, Lout, K, S, P, ks, MP, RP, xm);
  if (ks > 1)
    hipLaunchKernelGGL(mm3_sum_parts_kernel, dim3((unsigned)std::min<size_t>((n_out + 255) / 256, 8192)), dim3(256), 0, s, (const float*)target, ks, n_out, y);
  return hipGetLastError();
}
hipError_t launch_mm3_dx(const float* dy, const float* w, int B, int Cin, int Cout, int Lin, int Lout, int K, int S, int P, float* dx, hipStream_t s,
                         const float* bias) {
  const int MP = up(Cin, 128), RP = up(Cout, 32);
  u32x4* pw = pack_workspace((size_t)K * MP * RP * 4, s);
  if (!pw) return hipErrorOutOfMemory;
  const long total = (long)K * MP * (RP / 8);
  hipLaunchKernelGGL((mm3_pack_kernel<1>), dim3((unsigned)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, s, w, Cin, Cout, K, MP, RP, pw);
  const long N = (long)B * Lin;
  const int ks = reduction_split((long)((N + 127) / 128) * (MP / 128), K * (RP / 32));
  const size_t n_out = (size_t)B * Cin * Lin;
  float* target = dx;
  if (ks > 1) {
    target = dw_workspace((size_t)ks * n_out * sizeof(float), s);
    if (!target) return hipErrorOutOfMemory;
  }
  const int Ntl = (int)((N + 127) / 128), Mtl = MP / 128;
  const int xm = mm3_xcd() && Mtl > 1;
  const dim3 grid = xm ? dim3((unsigned)(8 * ((Ntl + 7) / 8) * Mtl), 1, ks) : dim3((unsigned)Ntl, Mtl, ks);
  if (Cout % 32 == 0) {
    if (g_train_bf16) hipLaunchKernelGGL((mm3_kernel<1, false, false, 8, true>), grid, dim3(512), 0, s, (const void*)pw, dy, bias, target, B, Cin, Cout, Lin, Lout, K, S, P, ks, MP, RP, xm);
    else if (mm3_nw() == 8) hipLaunchKernelGGL((mm3_kernel<1, false, false, 8>), grid, dim3(512), 0, s, (const void*)pw, dy, bias, target, B, Cin, Cout, Lin, Lout, K, S, P, ks, MP, RP, xm);
    else hipLaunchKernelGGL((mm3_kernel<1, false, false, 4>), grid, dim3(256), 0, s, (const void*)pw, dy, bias, target, B, Cin, Cout, Lin, Lout, K, S, P, ks, MP, RP, xm);
  } else
    hipLaunchKernelGGL((mm3_kernel<1, true>), grid, dim3(256), 0, s, (const void*)pw, dy, bias, target, B, Cin, Cout, Lin, Lout, K, S, P, ks, MP, RP, xm);
  if (ks > 1)
    hipLaunchKernelGGL(mm3_sum_parts_kernel, dim3((unsigned)std::min<size_t>((n_out + 255) / 256, 8192)), dim3(256), 0, s, (const float*)target, ks, n_out, dx);
  return hipGetLastError();
}
hipError_t launch_mm3_dw(const float* dy, const float* x, int B, int Cin, int Cout, int Lin, int Lout, int K, int S, int P, float* dw, hipStream_t s,
                         float* db) {
  // few output tiles, a long reduction over the items: split the items over workgroups until the grid fills the chip (two workgroups
  // per CU); the parts write their tiles to a workspace and a second kernel sums them in order (deterministic)
  const int tiles = ((Cin + 127) / 128) * ((Cout + 127) / 128) * K;
  static const int want_wgs = getenv("LDC_MM3_DW_WGS") ? atoi(getenv("LDC_MM3_DW_WGS")) : 512;
  const int nsplit = std::max(1, std::min(B, (want_wgs + tiles - 1) / tiles));
  float* target = dw;
  float* db_target = db;          // the first column tile of tap 0 writes the row sums of its dy tiles (per part)
  if (nsplit > 1) {
    const size_t n_dw = (size_t)nsplit * K * Cout * Cin;
    target = dw_workspace((n_dw + (size_t)nsplit * Cout) * sizeof(float), s);
    if (!target) return hipErrorOutOfMemory;
    if (db) db_target = target + n_dw;
  }
  // (measured: with every tile and tap of a part on one XCD the dW kernels are 1.6x SLOWER -- 3.22 vs 1.98 ms over the probe shapes --
  // so dW keeps the 3-D grid; LDC_MM3_XCD=2 selects the mapping for experiments)
  const int xm = mm3_xcd() == 2;
  const int tpp = ((Cin + 127) / 128) * ((Cout + 127) / 128) * K;
  const dim3 grid = xm ? dim3((unsigned)(8 * ((nsplit + 7) / 8) * tpp), 1, 1) : dim3((Cin + 127) / 128, (Cout + 127) / 128, K * nsplit);
  const bool vecb = S == 1 && Lin >= 4;
#define LDC_MM3_DW(TAIL_, VECB_) \
  do { \
    if (g_train_bf16) hipLaunchKernelGGL((mm3_kernel<2, TAIL_, VECB_, 8, true>), grid, dim3(512), 0, s, (const void*)dy, x, db_target, target, B, Cin, Cout, Lin, Lout, K, S, P, nsplit, 0, 0, xm); \
    else if (mm3_nw() == 8) hipLaunchKernelGGL((mm3_kernel<2, TAIL_, VECB_, 8>), grid, dim3(512), 0, s, (const void*)dy, x, db_target, target, B, Cin, Cout, Lin, Lout, K, S, P, nsplit, 0, 0, xm); \
    else hipLaunchKernelGGL((mm3_kernel<2, TAIL_, VECB_, 4>), grid, dim3(256), 0, s, (const void*)dy, x, db_target, target, B, Cin, Cout, Lin, Lout, K, S, P, nsplit, 0, 0, xm); \
  } while (0)
  if (Lout % 4 == 0) { if (vecb) LDC_MM3_DW(false, true); else LDC_MM3_DW(false, false); }
  else { if (vecb) LDC_MM3_DW(true, true); else LDC_MM3_DW(true, false); }
#undef LDC_MM3_DW
  if (nsplit > 1) {
    const size_t total = (size_t)K * Cout * Cin;
    hipLaunchKernelGGL(mm3_dw_reduce_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, (const float*)target, nsplit, K, Cout, Cin, dw, (const float*)db_target, db);
  }
  return hipGetLastError();
}

}  // namespace ldc
