// conv_fast.hip -- the pipelined conv-GEMM used by every UNet conv (zero padding, no prologue
// activation, plain Conv1d incl. stride 2, nearest-x2-upsample folding and two concatenated inputs).
//
// Same implicit-GEMM tiling as conv_gemm.hip (BM output positions x BN output channels per
// workgroup, 32x32 MFMA sub-tiles), restructured around the CDNA4 memory path:
//   * HBM/L2 -> LDS copies are LDS-DMA (`global_load_lds_dwordx4`: 64 lanes x 16 B land lane-linear in
//     LDS, no VGPR round trip), issued through inline asm so that hipcc's waitcnt pass neither waits
//     for them nor drains them before the ds_reads of the stage being consumed (with the builtin it
//     inserts `s_waitcnt vmcnt(0)` in front of the first ds_read and serialises copy and math);
//   * an S-stage LDS ring (S compile-time, the unit loop is unrolled by S so every LDS address is a
//     precomputed register + immediate): the copies of units u+1..u+S-1 are in flight while unit u is
//     multiplied, released with a counted `s_waitcnt vmcnt(N)` + one barrier per unit;
//   * a pipeline UNIT = NS sub-steps of (one tap, one 64-byte channel chunk): 3 taps x 1 chunk for k=3,
//     4 x 1 for k=4 / k=7 (two groups), 1 tap x 2 chunks for 1x1 convs;
//   * LDS-DMA cannot pad rows, so bank conflicts are avoided with an XOR swizzle of the 16-byte slot
//     index, slot' = slot ^ ((row >> 2) & 3), applied to the per-lane SOURCE address and again to the
//     ds_read address (16 rows distinct mod 16 -> 16 distinct bank slots; SQ_LDS_BANK_CONFLICT < 1 %);
//   * everything that does not depend on the channel chunk -- the per-tap input-row gather incl. padding
//     / stride / upsample, the swizzled fragment addresses for every ring stage, the per-lane DMA source
//     rows -- is computed once per workgroup: the first version spent 11 VALU + 9 SALU instructions per
//     MFMA on address arithmetic (rocprofv3 SQ_INSTS_*), which, not LDS or L2, was what bounded it.
#include <stdlib.h>

#include <algorithm>

#include "conv_device.h"

namespace ldc {

// counted wait on the VMEM counter
__device__ __forceinline__ void wait_vmcnt(int n) {
  switch (n) {
#define LDC_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    LDC_W(0) LDC_W(1) LDC_W(2) LDC_W(3) LDC_W(4) LDC_W(5) LDC_W(6) LDC_W(7) LDC_W(8) LDC_W(9) LDC_W(10) LDC_W(11)
    LDC_W(12) LDC_W(13) LDC_W(14) LDC_W(15) LDC_W(16) LDC_W(17) LDC_W(18) LDC_W(19) LDC_W(20) LDC_W(21) LDC_W(22)
    LDC_W(23) LDC_W(24) LDC_W(25) LDC_W(26) LDC_W(27) LDC_W(28) LDC_W(29) LDC_W(30) LDC_W(31) LDC_W(32) LDC_W(33)
    LDC_W(34) LDC_W(35) LDC_W(36) LDC_W(37) LDC_W(38) LDC_W(39) LDC_W(40) LDC_W(41) LDC_W(42) LDC_W(43) LDC_W(44)
    LDC_W(45) LDC_W(46) LDC_W(47) LDC_W(48)
#undef LDC_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// one LDS-DMA: lane l's 16 bytes at (base + voff) land at LDS byte address lds_addr + 16*l.  M0 carries the
// wave-uniform LDS address and is saved/restored (hipcc does not model M0 inside an asm statement).
__device__ __forceinline__ void lds_dma16(const char* base, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(base), "s"(lds_addr)
               : "memory");
}

struct FastGeom {
  int a_rows;        // window rows per plane (multiple of 16 * waves); each plane is followed by a zero row
  int b_rows;        // weight rows per unit (NS * BN rounded up to 16 * waves)
  int stage_bytes;
  int ngroups;       // tap groups per chunk group (k=7: 2)
  int debug;         // tuning aid (LDC_CONV_DEBUG): 1 = skip the copies after the prologue, 2 = skip the math
  unsigned long long* stamps;   // tuning aid: per workgroup {start, prologue done, loop done, end} s_memtime
};


template <typename T, int WM, int WN, int TM, int TN, int TG, int KC, int S, int NA>
__global__ __launch_bounds__(WM* WN * 64) void conv_fast_kernel(const ConvKArgs a, const FastGeom gm) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NW = WM * WN, NS = TG * KC;
  constexpr int BKE = kRowBytes / (int)sizeof(T);
  constexpr int NB = (NS * BN + 16 * NW - 1) / (16 * NW);   // LDS-DMA sweeps per wave: weight slabs of a unit
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int M = a.B * a.L_rows;
  unsigned long long t_start = 0, t_pro = 0, t_loop = 0;
  if (gm.stamps) t_start = __builtin_amdgcn_s_memtime();
  // XCD-aware tile order: workgroup h runs on XCD h % 8 (observed dispatch rule, used for speed only), so give
  // every XCD a contiguous run of tiles, N-tile fastest: workgroups sharing an input window hit the same L2.
  int m0, n0;
  {
    const int ntn = a.n_pad / BN;
    const int ntiles = gridDim.x;
    const int h = blockIdx.x;
    const int xcd = h & 7, slot = h >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int mt = t / ntn;
    m0 = mt * BM;
    n0 = (t - mt * ntn) * BN;
  }

  int R_lo, R_hi;
  tile_window(a, m0, BM, M, R_lo, R_hi);
  const int nrows = min(R_hi - R_lo + 1, gm.a_rows);
  const int plane_bytes = (gm.a_rows + 1) * kRowBytes;
  const int zero_off = gm.a_rows * kRowBytes;
  const int b_off = KC * plane_bytes;
  // zero rows (one behind every plane of every stage)
  for (int i = tid; i < S * KC * 16; i += NW * 64) {
    const int st = i / (KC * 16), rem = i - st * (KC * 16);
    reinterpret_cast<unsigned*>(smem + (size_t)st * gm.stage_bytes + (rem >> 4) * plane_bytes + zero_off)[rem & 15] = 0u;
  }

  const int kh = lane >> 5;
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
  // LDS byte addresses of this lane's fragments for every ring stage / plane (wave-uniform offsets folded in)
  unsigned a_ad[S][KC][TM][TG][2];
  auto calc_arows = [&](int g) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int t = 0; t < TG; ++t) {
        int off0 = zero_off, off1 = zero_off;
        const int tap = g * TG + t;
        if (row_b[i] >= 0 && tap < a.taps) {
          const int gr = gather_row(a, row_b[i], row_l[i], tap * a.dil);
          if (gr >= 0) {
            const int row = gr - R_lo;
            const int sw = (row >> 2) & 3;
            off0 = row * kRowBytes + ((kh ^ sw) << 4);
            off1 = row * kRowBytes + (((2 + kh) ^ sw) << 4);
          }
        }
#pragma unroll
        for (int ss = 0; ss < S; ++ss)
#pragma unroll
          for (int kc = 0; kc < KC; ++kc) {
            a_ad[ss][kc][i][t][0] = (unsigned)(ss * gm.stage_bytes + kc * plane_bytes + off0);
            a_ad[ss][kc][i][t][1] = (unsigned)(ss * gm.stage_bytes + kc * plane_bytes + off1);
          }
      }
  };
  calc_arows(0);
  unsigned b_ad[S][TN][2];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int rb = wn * TN * 32 + j * 32 + (lane & 31);
    const int sw = (rb >> 2) & 3;
#pragma unroll
    for (int ss = 0; ss < S; ++ss) {
      b_ad[ss][j][0] = (unsigned)(ss * gm.stage_bytes + b_off + rb * kRowBytes + ((kh ^ sw) << 4));
      b_ad[ss][j][1] = (unsigned)(ss * gm.stage_bytes + b_off + rb * kRowBytes + (((2 + kh) ^ sw) << 4));
    }
  }

  // per-lane DMA sources, chunk-independent part
  unsigned dma_row[NA], dma_s16[NA];   // window: flat input row and swizzled 16-byte slot offset
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int p = (i * NW + wave) * 64 + lane;
    int r = p >> 2;
    dma_s16[i] = (unsigned)(((p & 3) ^ ((r >> 2) & 3)) << 4);
    if (r >= nrows) r = 0;
    dma_row[i] = (unsigned)(R_lo + r);
  }
  unsigned dma_b[NB];                      // weights: byte offset inside the (chunk group 0, tap group 0) slab set
  const unsigned slab_bytes = (unsigned)a.n_pad * kRowBytes;       // one tap of one chunk
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int p = (i * NW + wave) * 64 + lane;
    const int rb = p >> 2;
    const int s = (p & 3) ^ ((rb >> 2) & 3);
    int q = rb / BN;
    const int j = rb - q * BN;
    if (q >= NS) q = 0;
    const int kc = q / TG, t = q - kc * TG;
    dma_b[i] = (unsigned)(kc * a.taps + t) * slab_bytes + (unsigned)(n0 + j) * kRowBytes + (unsigned)(s << 4);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nchunks = (a.C1 + a.C2) / BKE;
  const int nunits = (nchunks / KC) * gm.ngroups;
  constexpr int per_unit = KC * NA + NB;     // LDS-DMA instructions per wave per unit (wave-uniform)

  // producer cursor: (chunk group, tap group) of the next unit to copy
  int lu_c0 = 0, lu_g = 0;
  auto load_unit = [&](int st) {
    const unsigned stage_lds = lds_base + (unsigned)(st * gm.stage_bytes) + (unsigned)(wave * 1024);
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int c = lu_c0 + kc;
      const char* src;
      unsigned ldb, cofb;
      if (c * BKE < a.C1) {
        src = a.x1; ldb = a.C1 * (unsigned)sizeof(T); cofb = (unsigned)c * kRowBytes;
      } else {
        src = a.x2; ldb = a.C2 * (unsigned)sizeof(T); cofb = (unsigned)(c * BKE - a.C1) * (unsigned)sizeof(T);
      }
#pragma unroll
      for (int i = 0; i < NA; ++i)
        lds_dma16(src, dma_row[i] * ldb + dma_s16[i] + cofb, stage_lds + (unsigned)(kc * plane_bytes + i * NW * 1024));
    }
    const int tg0 = lu_g * TG;
    const unsigned unit_b = (unsigned)(lu_c0 * a.taps + tg0) * slab_bytes;
    if (gm.ngroups == 1) {
#pragma unroll
      for (int i = 0; i < NB; ++i) lds_dma16(a.w, dma_b[i] + unit_b, stage_lds + (unsigned)(b_off + i * NW * 1024));
    } else {
      // k=7: the last tap group is short; its unused slab is pointed at a valid one
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int p = (i * NW + wave) * 64 + lane;
        int q = (p >> 2) / BN;
        if (q >= NS) q = 0;
        const int t = q % TG;
        const unsigned off = (tg0 + t >= a.taps) ? dma_b[i] - (unsigned)t * slab_bytes : dma_b[i];
        lds_dma16(a.w, off + unit_b, stage_lds + (unsigned)(b_off + i * NW * 1024));
      }
    }
    if (++lu_g == gm.ngroups) { lu_g = 0; lu_c0 += KC; }
  };

#pragma unroll
  for (int p = 0; p < S - 1; ++p)
    if (p < nunits) load_unit(p);
  int cu_g = 0;   // consumer cursor: tap group of the unit being multiplied
  if (gm.stamps) t_pro = __builtin_amdgcn_s_memtime();
  for (int u0 = 0; u0 < nunits; u0 += S) {
#pragma unroll
    for (int ss = 0; ss < S; ++ss) {
      const int u = u0 + ss;
      if (u >= nunits) break;
      // my copies of unit u have landed once at most min(S-2, units after u) later units are outstanding
      if (S == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else wait_vmcnt(min(S - 2, nunits - 1 - u) * per_unit);
      __syncthreads();   // every wave's part of unit u is visible; every wave is done with unit u-1 -> its stage is free
      if (u + S - 1 < nunits && gm.debug != 1) load_unit((ss + S - 1) % S);
      if (gm.ngroups > 1) calc_arows(cu_g);
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const int kc = q / TG, t = q % TG;
        if (gm.debug == 2) continue;
        if (gm.ngroups > 1 && cu_g * TG + t >= a.taps) continue;   // k=7: second group has 3 taps
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          uint4 af[TM], bfr[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const uint4*>(smem + a_ad[ss][kc][i][t][ks]);
#pragma unroll
          for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const uint4*>(smem + b_ad[ss][j][ks] + q * (BN * kRowBytes));
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) mfma_step<T>(acc[i][j], af[i], bfr[j]);
        }
      }
      if (++cu_g == gm.ngroups) cu_g = 0;
    }
  }

  if (gm.stamps) t_loop = __builtin_amdgcn_s_memtime();
  __syncthreads();   // every wave has consumed the last ring stage: LDS is free for the output transpose
  {
    constexpr int WREG = TM * 32 * (TN * 32 * (int)sizeof(T) + 16);
    epilogue_rows_dispatch<T, TM, TN>(a, acc, smem + (size_t)wave * WREG, m0 + wm * TM * 32, n0 + wn * TN * 32, M, m0, BM);
  }
  if (gm.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      unsigned long long* o = gm.stamps + (size_t)blockIdx.x * 4;
      o[0] = t_start; o[1] = t_pro; o[2] = t_loop; o[3] = t_end;
    }
  }
}

unsigned long long* g_conv_stamps = nullptr;   // set by ldc_conv_microbench when LDC_CONV_STAMPS is on

bool conv_fast_eligible(const ConvLayer& ly) {
  return ly.pad_mode == PAD_ZERO && ly.pre_act == ACT_NONE && ly.tr_stride == 0 && ly.taps <= 8;
}

template <typename T, int WM, int WN, int TM, int TN, int TG, int KC, int S, int NA>
static hipError_t launch_fast_cfg(const ConvKArgs& a, const FastGeom& gm, int M, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  dim3 grid(((M + BM - 1) / BM) * (a.n_pad / BN));
  auto kern = conv_fast_kernel<T, WM, WN, TM, TN, TG, KC, S, NA>;
  static bool lds_opt_in = false;
  if (!lds_opt_in) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    lds_opt_in = true;
  }
  const size_t epi_bytes = (size_t)WM * WN * TM * 32 * (TN * 32 * sizeof(T) + 16);
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), std::max((size_t)S * gm.stage_bytes, epi_bytes), s, a, gm);
  return hipGetLastError();
}

template <typename T, int TG, int KC, int NA>
static hipError_t launch_fast_bn(int bn, const ConvKArgs& a, const FastGeom& gm, int M, hipStream_t s) {
  if (bn == 128) return launch_fast_cfg<T, 2, 2, 2, 2, TG, KC, 2, NA>(a, gm, M, s);
  if (bn == 64) return launch_fast_cfg<T, 2, 2, 2, 1, TG, KC, 2, NA>(a, gm, M, s);
  return launch_fast_cfg<T, 4, 1, 1, 1, TG, KC, 2, NA>(a, gm, M, s);
}

hipError_t launch_conv_fast(const ConvLayer& ly, const ConvKArgs& a_in, int M, int span_rows, hipStream_t s, bool* launched) {
  *launched = false;
  FastGeom gm;
  ConvKArgs a = a_in;
  const int bn = ly.bn;
  const int bke = kRowBytes / (int)dt_size(ly.dt);
  const int nchunks = (ly.cin1 + ly.cin2) / bke;
  // 32-bit byte offsets inside the kernel
  if ((double)a.B * a.L_in * std::max(a.C1, a.C2) * dt_size(ly.dt) >= 2.0e9) return hipSuccess;
  if ((double)nchunks * ly.taps * ly.n_pad * kRowBytes >= 2.0e9) return hipSuccess;
  int TG, KC = 1;
  if (ly.taps == 1) {
    TG = 1;
    const bool kc2 = nchunks % 2 == 0 && (ly.cin1 / bke) % 2 == 0;
    KC = kc2 ? 2 : 1;
  } else if (ly.taps == 3) {
    TG = 3;
  } else if (ly.taps == 4 || ly.taps == 7 || ly.taps == 8) {
    TG = 4;
  } else {
    return hipSuccess;   // other kernel sizes: generic path
  }
  const int quantum = 16 * 4;
  gm.ngroups = (ly.taps + TG - 1) / TG;
  gm.a_rows = (span_rows + quantum - 1) / quantum * quantum;
  gm.b_rows = (TG * KC * bn + quantum - 1) / quantum * quantum;
  gm.stage_bytes = (KC * (gm.a_rows + 1) + gm.b_rows) * kRowBytes;
  if ((size_t)gm.stage_bytes * 2 > 160 * 1024) return hipSuccess;
  // window sweeps per wave are a template parameter: 2 or 3 (stride 1 / upsample), 5 (k=4 stride 2)
  int na = gm.a_rows / quantum;
  if (TG == 4 && na <= 3) na = 3;
  else if (TG == 4 && na <= 5) na = 5;
  else if (TG != 4 && na <= 2) na = 2;
  else if (TG != 4 && na <= 3) na = 3;
  else return hipSuccess;
  gm.a_rows = na * quantum;
  gm.stage_bytes = (KC * (gm.a_rows + 1) + gm.b_rows) * kRowBytes;
  if ((size_t)gm.stage_bytes * 2 > 160 * 1024) return hipSuccess;
  {
    static int dbg = -1;
    if (dbg < 0) dbg = getenv("LDC_CONV_DEBUG") ? atoi(getenv("LDC_CONV_DEBUG")) : 0;
    gm.debug = dbg;
    gm.stamps = g_conv_stamps;
  }
  a.tg = TG;
  *launched = true;
#define LDC_FAST(TT)                                                                                   \
  if (TG == 3) return na == 2 ? launch_fast_bn<TT, 3, 1, 2>(bn, a, gm, M, s) : launch_fast_bn<TT, 3, 1, 3>(bn, a, gm, M, s); \
  if (TG == 4) return na == 3 ? launch_fast_bn<TT, 4, 1, 3>(bn, a, gm, M, s) : launch_fast_bn<TT, 4, 1, 5>(bn, a, gm, M, s); \
  if (KC == 2) return na == 2 ? launch_fast_bn<TT, 1, 2, 2>(bn, a, gm, M, s) : launch_fast_bn<TT, 1, 2, 3>(bn, a, gm, M, s); \
  return na == 2 ? launch_fast_bn<TT, 1, 1, 2>(bn, a, gm, M, s) : launch_fast_bn<TT, 1, 1, 3>(bn, a, gm, M, s);
  if (ly.dt == DT_F32) { LDC_FAST(float) } else { LDC_FAST(__bf16) }
#undef LDC_FAST
}

}  // namespace ldc
