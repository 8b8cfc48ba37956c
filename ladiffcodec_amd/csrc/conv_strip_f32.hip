// f32 (exact-fp32 MFMA) instantiations of the strip conv (see conv_strip.inc) + the dtype dispatch, eligibility and
// weight packing
#define LDC_STRIP_T float
#define LDC_STRIP_NS strip_f32
#define LDC_STRIP_GEOM strip_geom_f32
#define LDC_STRIP_ENTRY launch_conv_strip_f32
#include "conv_strip.inc"
#include "conv_strip_entry.inc"

#include <string.h>

namespace ldc {

hipError_t launch_conv_strip_bf16(const StripCall& sc, hipStream_t s);

static int gcd_int(int a, int b) { return b ? gcd_int(b, a % b) : a; }

bool conv_strip_eligible(int dt, int N, int groups, int L, int cin, int cres) {
  if (groups <= 0 || N % groups || L < 1) return false;
  const int che = kRowBytes / (int)dt_size(dt);
  if (cin % che || cres % che) return false;
  int bn, wr, rtw, ctw, nch, plane, nbuf;
  size_t lds;
  const int g = cres ? gcd_int(cin / che, cres / che) : cin / che;
  return strip_geom_f32(dt, N, N / groups, L, g, &bn, &wr, &rtw, &ctw, &nch, &plane, &nbuf, &lds);
}

hipError_t launch_conv_strip(const StripCall& sc, hipStream_t s) {
  return sc.conv->dt == DT_F32 ? launch_conv_strip_f32(sc, s) : launch_conv_strip_bf16(sc, s);
}

size_t strip_packed_weight_bytes(int dt, int cin, int n, int taps) {
  const int che = kRowBytes / (int)dt_size(dt);
  return (size_t)(n / 32) * (cin / che) * taps * 2 * 1024;
}

static inline uint16_t strip_host_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// w_oik fp32 [n][cin][taps] -> [column tile][chunk][tap][k half][lane][16 B]: the MFMA B fragment image
void pack_strip_weights(int dt, int cin, int n, int taps, const float* w_oik, void* dst) {
  const int es = (int)dt_size(dt), che = kRowBytes / es, per = 16 / es;
  const int nchunks = cin / che;
  char* out = reinterpret_cast<char*>(dst);
  for (int ct = 0; ct < n / 32; ++ct)
    for (int c = 0; c < nchunks; ++c)
      for (int t = 0; t < taps; ++t)
        for (int ks = 0; ks < 2; ++ks) {
          char* frag = out + ((((size_t)ct * nchunks + c) * taps + t) * 2 + ks) * 1024;
          for (int lane = 0; lane < 64; ++lane) {
            const int col = ct * 32 + (lane & 31), kh = lane >> 5;
            for (int e = 0; e < per; ++e) {
              const int ch = c * che + (ks * 2 + kh) * per + e;
              const float v = w_oik[((size_t)col * cin + ch) * taps + t];
              if (dt == DT_F32) reinterpret_cast<float*>(frag + lane * 16)[e] = v;
              else reinterpret_cast<uint16_t*>(frag + lane * 16)[e] = strip_host_bf16(v);
            }
          }
        }
}

}  // namespace ldc
