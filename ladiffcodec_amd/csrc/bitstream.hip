// bitstream.hip -- the on-wire format between RVQ encode and decode: index packing and the
// range (arithmetic) coder of the reference, one independent stream per utterance.
//
//   pack / unpack   : BitPacker / BitUnpacker, srcs/encodec/binary.py:55-118 -- values of `bits` bits, LSB first, in
//                     the push order of srcs/encodec/compress.py:74-84 (for t: for k: codes[k][t]); byte-parallel:
//                     one thread per output byte / per symbol.  HBM-bound byte work (8 B in, 1.25 B out per code).
//   quantised cdf   : build_stable_quantized_cdf, srcs/quantization/ac.py:18-53 -- float32 floor arithmetic exactly as
//                     torch evaluates it (IEEE division and multiplication, no contraction), one wave per pdf row
//                     with a wave prefix sum.
//   range coder     : ArithmeticCoder.push/flush and ArithmeticDecoder.pull, ac.py:131-174, 218-260 -- sequential by
//                     construction: one lane per stream; the interval arithmetic is the reference's (Python ints and
//                     doubles -> 64-bit integers and IEEE doubles here), the output is bit-identical.
// All integer / byte results are bit-exact against the reference (tests/golden/bitstream.npz).
#include <algorithm>

#include "ldc_kernels.h"

namespace ldc {

// ---------------------------------------------------------------------------------------------
// index packing
// ---------------------------------------------------------------------------------------------
// symbol s of item b is codes[k = s % n_q][b][t = s / n_q]
__global__ __launch_bounds__(256) void pack_codes_kernel(const int64_t* codes, int n_q, int B, int F, int bits, uint8_t* out,
                                                         int64_t out_stride, int64_t nbytes) {
  const int b = blockIdx.y;
  const int64_t nsym = (int64_t)n_q * F;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < nbytes; j += (int64_t)gridDim.x * 256) {
    const int64_t bit0 = 8 * j;
    unsigned v = 0;
    for (int64_t s = bit0 / bits; s < nsym && s * bits < bit0 + 8; ++s) {
      const int k = (int)(s % n_q);
      const int64_t t = s / n_q;
      const uint64_t val = (uint64_t)codes[((size_t)k * B + b) * F + t] & ((1ull << bits) - 1);
      const int64_t sh = s * bits - bit0;          // bit position of the symbol's LSB relative to this byte
      v |= (unsigned)((sh >= 0 ? (val << sh) : (val >> (-sh))) & 0xffu);
    }
    out[(size_t)b * out_stride + j] = (uint8_t)v;
  }
}

__global__ __launch_bounds__(256) void unpack_codes_kernel(const uint8_t* in, int64_t in_stride, int64_t nbytes, int n_q, int B,
                                                           int F, int bits, int64_t* codes) {
  const int b = blockIdx.y;
  const int64_t nsym = (int64_t)n_q * F;
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < nsym; s += (int64_t)gridDim.x * 256) {
    const int64_t bit0 = s * bits, j0 = bit0 >> 3;
    uint64_t acc = 0;
    for (int q = 0; q < 4; ++q)
      if (j0 + q < nbytes) acc |= (uint64_t)in[(size_t)b * in_stride + j0 + q] << (8 * q);
    const int64_t val = (int64_t)((acc >> (bit0 & 7)) & ((1ull << bits) - 1));
    codes[((size_t)(s % n_q) * B + b) * F + s / n_q] = val;
  }
}

hipError_t launch_pack_codes(const int64_t* codes, int n_q, int B, int F, int bits, uint8_t* out, int64_t out_stride, hipStream_t s) {
  const int64_t nbytes = ((int64_t)n_q * F * bits + 7) / 8;
  if (nbytes == 0) return hipSuccess;
  hipLaunchKernelGGL(pack_codes_kernel, dim3((unsigned)std::min<int64_t>((nbytes + 255) / 256, 1024), B), dim3(256), 0, s, codes, n_q, B,
                     F, bits, out, out_stride, nbytes);
  return hipGetLastError();
}
hipError_t launch_unpack_codes(const uint8_t* in, int64_t in_stride, int n_q, int B, int F, int bits, int64_t* codes, hipStream_t s) {
  const int64_t nsym = (int64_t)n_q * F, nbytes = (nsym * bits + 7) / 8;
  if (nsym == 0) return hipSuccess;
  hipLaunchKernelGGL(unpack_codes_kernel, dim3((unsigned)std::min<int64_t>((nsym + 255) / 256, 1024), B), dim3(256), 0, s, in, in_stride,
                     nbytes, n_q, B, F, bits, codes);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// quantised cdf: one wave per row
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void build_cdf_kernel(const float* pdf, int rows, int card, float scale, float roundoff,
                                                       int min_range, int* cdf) {
  const int row = blockIdx.x, lane = threadIdx.x;
  if (row >= rows) return;
  const float* p = pdf + (size_t)row * card;
  int* o = cdf + (size_t)row * card;
  int carry = 0;
  for (int c0 = 0; c0 < card; c0 += 64) {
    const int c = c0 + lane;
    int r = 0;
    if (c < card) {
      float v = p[c];
      if (roundoff != 0.0f) v = __fmul_rn(floorf(__fdiv_rn(v, roundoff)), roundoff);     // ac.py:37-38
      r = (int)floorf(__fmul_rn(scale, v)) + min_range;                                    // ac.py:44-45
    }
    // inclusive wave prefix sum
#pragma unroll
    for (int o2 = 1; o2 < 64; o2 <<= 1) {
      const int up = __shfl_up(r, o2);
      if (lane >= o2) r += up;
    }
    if (c < card) o[c] = r + carry;
    carry += __shfl(r, 63);
  }
}

hipError_t launch_build_cdf(const float* pdf, int rows, int card, int total_range_bits, float roundoff, int min_range, int* cdf,
                            hipStream_t s) {
  if (rows <= 0 || card <= 0) return hipSuccess;
  const double total_range = (double)(1ll << total_range_bits);
  const double alpha = (double)min_range * card / total_range;
  if (alpha > 1.0 || min_range < 2) return hipErrorInvalidValue;
  const float scale = (float)((1.0 - alpha) * total_range);    // the python float the float32 tensor is multiplied by
  hipLaunchKernelGGL(build_cdf_kernel, dim3(rows), dim3(64), 0, s, pdf, rows, card, scale, roundoff, min_range, cdf);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// range coder: one lane per stream
// ---------------------------------------------------------------------------------------------
struct BitSink {            // BitPacker(bits = 1): LSB first
  uint8_t* p; int64_t cap, n; unsigned cur; int nb; bool overflow;
  __device__ void push(unsigned bit) {
    cur |= bit << nb;
    if (++nb == 8) { if (n < cap) p[n] = (uint8_t)cur; else overflow = true; ++n; cur = 0; nb = 0; }
  }
  __device__ void flush() {
    if (nb) { if (n < cap) p[n] = (uint8_t)cur; else overflow = true; ++n; cur = 0; nb = 0; }
  }
};

__device__ __forceinline__ const int* cdf_row(const int* cdf, int card, int mode, int period, int b, int S, int s) {
  // mode 0: a table per (stream, step): row b*S + s;  mode 1: `period` static tables used round-robin (codebook k = s % n_q)
  return cdf + (size_t)(mode == 0 ? (size_t)b * S + s : (size_t)(s % period)) * card;
}

__global__ __launch_bounds__(64) void ac_encode_kernel(const int* symbols, const int* cdf, int B, int S, int card, int mode,
                                                       int period, int trb, uint8_t* out, int64_t out_stride, int64_t cap,
                                                       int64_t* nbytes) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  BitSink sink{out + (size_t)b * out_stride, cap, 0, 0u, 0, false};
  unsigned long long low = 0, high = 0;
  int max_bit = -1;
  const unsigned long long full = 1ull << trb;
  const double inv = 1.0 / (double)full;
  bool bad = false;
  for (int s = 0; s < S && !bad; ++s) {
    const int sym = symbols[(size_t)b * S + s];
    const int* q = cdf_row(cdf, card, mode, period, b, S, s);
    if (sym < 0 || sym >= card) { bad = true; break; }
    while (high - low + 1 < full) { low *= 2; high = high * 2 + 1; ++max_bit; }                 // ac.py:140-143
    if (max_bit > 61) { bad = true; break; }
    const unsigned long long delta = high - low + 1;
    const double f = (double)delta * inv;                                                       // delta / 2**total_range_bits
    const long long range_low = sym == 0 ? 0 : q[sym - 1];
    const long long range_high = (long long)q[sym] - 1;
    const unsigned long long eff_low = (unsigned long long)ceil((double)range_low * f);
    const unsigned long long eff_high = (unsigned long long)floor((double)range_high * f);
    high = low + eff_high;
    low = low + eff_low;
    if (low > high) { bad = true; break; }
    while (max_bit >= 0) {                                                                       // _flush_common_prefix
      const unsigned long long b1 = low >> max_bit, b2 = high >> max_bit;
      if (b1 != b2) break;
      low -= b1 << max_bit; high -= b1 << max_bit;
      --max_bit;
      sink.push((unsigned)b1);
    }
  }
  while (max_bit >= 0) { sink.push((unsigned)((low >> max_bit) & 1ull)); --max_bit; }           // flush, ac.py:167-174
  sink.flush();
  nbytes[b] = (bad || sink.overflow) ? -1 : sink.n;
}

__global__ __launch_bounds__(64) void ac_decode_kernel(const uint8_t* in, int64_t in_stride, const int64_t* nbytes, const int* cdf,
                                                       int B, int S, int card, int mode, int period, int trb, int* symbols,
                                                       int* status) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const uint8_t* p = in + (size_t)b * in_stride;
  const int64_t nbits = 8 * nbytes[b];
  int64_t bitpos = 0;
  unsigned long long low = 0, high = 0, current = 0;
  int max_bit = -1;
  const unsigned long long full = 1ull << trb;
  const double inv = 1.0 / (double)full;
  int st = 0;
  for (int s = 0; s < S && st == 0; ++s) {
    const int* q = cdf_row(cdf, card, mode, period, b, S, s);
    while (high - low + 1 < full) {                                                              // ac.py:228-236
      if (bitpos >= nbits) { st = 1; break; }                                                    // stream exhausted
      const unsigned bit = (p[bitpos >> 3] >> (bitpos & 7)) & 1u;
      ++bitpos;
      low *= 2; high = high * 2 + 1; current = current * 2 + bit; ++max_bit;
    }
    if (st) break;
    const double f = (double)(high - low + 1) * inv;
    int lo_i = 0, hi_i = card - 1, mid = 0;
    unsigned long long lo = 0, hi = 0;
    for (;;) {                                                                                    // bin_search, ac.py:238-254
      if (hi_i < lo_i) { st = 2; break; }
      mid = (lo_i + hi_i) / 2;
      const long long range_low = mid > 0 ? q[mid - 1] : 0;
      const long long range_high = (long long)q[mid] - 1;
      lo = (unsigned long long)ceil((double)range_low * f) + low;
      hi = (unsigned long long)floor((double)range_high * f) + low;
      if (current >= lo) {
        if (current <= hi) break;
        lo_i = mid + 1;
      } else {
        hi_i = mid - 1;
      }
    }
    if (st) break;
    low = lo; high = hi;
    while (max_bit >= 0) {
      const unsigned long long b1 = low >> max_bit, b2 = high >> max_bit;
      if (b1 != b2) break;
      low -= b1 << max_bit; high -= b1 << max_bit; current -= b1 << max_bit;
      --max_bit;
    }
    symbols[(size_t)b * S + s] = mid;
  }
  status[b] = st;
}

hipError_t launch_ac_encode(const int* symbols, const int* cdf, int B, int S, int card, int mode, int period, int trb, uint8_t* out,
                            int64_t out_stride, int64_t cap, int64_t* nbytes, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(ac_encode_kernel, dim3((B + 63) / 64), dim3(64), 0, s, symbols, cdf, B, S, card, mode, period, trb, out, out_stride,
                     cap, nbytes);
  return hipGetLastError();
}
hipError_t launch_ac_decode(const uint8_t* in, int64_t in_stride, const int64_t* nbytes, const int* cdf, int B, int S, int card, int mode,
                            int period, int trb, int* symbols, int* status, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(ac_decode_kernel, dim3((B + 63) / 64), dim3(64), 0, s, in, in_stride, nbytes, cdf, B, S, card, mode, period, trb,
                     symbols, status);
  return hipGetLastError();
}

}  // namespace ldc
