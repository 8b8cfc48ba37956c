// train.hip -- the diffusion TRAINING step (BASELINE config 4), fp32 correctness path:
//
//   q_sample                      srcs/losses/ddpm_loss.py:386-392   x_t = sqrt(abar_t) x0 + sqrt(1 - abar_t) eps
//   loss of p_losses              ddpm_loss.py:434-438               mean_b( p2w[t_b] * mean_{c,l} |out - target| ), and d/d out
//   Block forward / backward      srcs/modules/unet.py:137-154 with WeightStandardizedConv2d (:67-80), GroupNorm(8), SiLU:
//                                 y = SiLU( GN(conv_k3(x; WS(W), b)) * (scale + 1) + shift )
//                                 backward: dx, dW (THROUGH the weight standardisation), db, dgamma, dbeta, dscale, dshift
//   channel LayerNorm fwd / bwd   unet.py:82-101 (PreNorm and to_out of the attention blocks)
//   pointwise linear fwd / bwd    unet.py:163-166,171 (time-embedding MLP SiLU -> Linear, 1x1 res_conv): with the two Blocks the
//                                 whole ResnetBlock (unet.py:157-192) runs forward and backward (ladiffcodec_amd/train.py)
//   LinearAttention core fwd/bwd  unet.py:208-221 (both softmaxes and both einsums); with the pointwise maps and the LayerNorm the whole
//                                 Residual(PreNorm(LinearAttention)) block runs forward and backward (ladiffcodec_amd/train.py)
//   plain Conv1d, nearest x2, tanh / GELU / SiLU, softmax Attention core, condition upsampler (transposed conv) + max-abs scaling:
//                                 the remaining layers of Unet1D (unet.py:248-470), each forward and backward
//   Adam step                     srcs/train.py:365-371 (optim.Adam(params, lr)), flat parameter / gradient / moment buffers
// ladiffcodec_amd/train.py assembles them into Unet1D.forward / backward and DiffusionTrainer.step.
//
// fp32 throughout, reference layouts [B, C, L].  Every layer, the assembled UNet and a short optimisation run are pinned to the
// reference's autograd / torch.optim.Adam (tests/golden/train_block.npz, train_unet.npz).  The GEMM-shaped pieces (conv / pointwise
// forward, dX, dW) run on the exact-fp32 MFMA (convmm_kernel below); their first VALU forms stay as the reference (LDC_TRAIN_VALU=1)
// and for sequences shorter than 16.  bench.py --config c4 times the step; a bf16 path on conv_fast is the next slice.
#include <stdlib.h>

#include <algorithm>

#include "ldc_kernels.h"
#include "ldc_math.h"

namespace ldc {

// ---------------------------------------------------------------------------------------------
// q_sample and the L1 objective
// ---------------------------------------------------------------------------------------------
// t comes from device memory: an index outside [0, T) would read past the schedule tables, so it is clamped (the host entry
// points refuse such t when they can see it; device-drawn t cannot be checked there)
__device__ __forceinline__ int64_t clamp_t(int64_t t, int T) { return t < 0 ? 0 : (t >= T ? (int64_t)T - 1 : t); }

__global__ __launch_bounds__(256) void q_sample_kernel(const float* x0, const float* noise, const int64_t* t, const float* sa,
                                                       const float* s1ma, int64_t n_per_item, float* out, int T) {
  const int b = blockIdx.y;
  const int64_t tb = clamp_t(t[b], T);
  const float a = sa[tb], c = s1ma[tb];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per_item; i += (int64_t)gridDim.x * 256) {
    const size_t k = (size_t)b * n_per_item + i;
    out[k] = a * x0[k] + c * noise[k];
  }
}
hipError_t launch_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac, const float* sqrt_1mac, int B,
                           int64_t n_per_item, float* out, int T, hipStream_t s) {
  hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)std::min<int64_t>((n_per_item + 255) / 256, 256), B), dim3(256), 0, s, x0, noise, t,
                     sqrt_ac, sqrt_1mac, n_per_item, out, T);
  return hipGetLastError();
}

// predicted_x_start of p_losses (ddpm_loss.py:416-420 -> model_predictions -> predict_start_from_noise, :175-179; no clamp on this
// path: clip_x_start defaults to False): x0 = sqrt(1 / abar_t) x_t - sqrt(1 / abar_t - 1) eps, per item t
__global__ __launch_bounds__(256) void predict_x_start_kernel(const float* x_t, const float* eps, const int64_t* t, const float* r,
                                                              const float* rm1, int64_t n_per_item, float* out, int T) {
  const int b = blockIdx.y;
  const int64_t tb = clamp_t(t[b], T);
  const float a = r[tb], c = rm1[tb];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per_item; i += (int64_t)gridDim.x * 256) {
    const size_t k = (size_t)b * n_per_item + i;
    out[k] = a * x_t[k] - c * eps[k];
  }
}
hipError_t launch_predict_x_start(const float* x_t, const float* eps, const int64_t* t, const float* sqrt_recip_ac, const float* sqrt_recipm1_ac,
                                  int B, int64_t n_per_item, float* out, int T, hipStream_t s) {
  hipLaunchKernelGGL(predict_x_start_kernel, dim3((unsigned)std::min<int64_t>((n_per_item + 255) / 256, 256), B), dim3(256), 0, s, x_t, eps,
                     t, sqrt_recip_ac, sqrt_recipm1_ac, n_per_item, out, T);
  return hipGetLastError();
}

// The monitoring loss of DiffAudioRep.forward (model.py:194): ClippedSDR (losses_fn.py:56-66) = clamp(MultiSrcNegSDR("sdsdr"), min
// -30) with est_targets = the clean input x and targets = x_hat (the reference passes them in this order).  asteroid 0.6.0's
// published algorithm, one source per item: zero-mean both, s = <e, g> g / (|g|^2 + EPS), noise = e - g,
// loss = -10 log10(|s|^2 / (|noise|^2 + EPS) + EPS), EPS = 1e-8.  One workgroup per item, double accumulators, two passes.
__global__ __launch_bounds__(256) void neg_sdsdr_kernel(const float* est, const float* tgt, int64_t n, float clip_min, float* out) {
  __shared__ double red[4][4];
  const int b = blockIdx.x;
  const float* e = est + (size_t)b * n;
  const float* g = tgt + (size_t)b * n;
  auto reduce4 = [&](double& v0, double& v1, double& v2, double& v3) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { v0 += __shfl_xor(v0, o); v1 += __shfl_xor(v1, o); v2 += __shfl_xor(v2, o); v3 += __shfl_xor(v3, o); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = v0; red[threadIdx.x >> 6][1] = v1; red[threadIdx.x >> 6][2] = v2; red[threadIdx.x >> 6][3] = v3; }
    __syncthreads();
    v0 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    v1 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    v2 = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
    v3 = (red[0][3] + red[1][3]) + (red[2][3] + red[3][3]);
  };
  double se = 0, sg = 0, z0 = 0, z1 = 0;
  for (int64_t i = threadIdx.x; i < n; i += 256) { se += e[i]; sg += g[i]; }
  reduce4(se, sg, z0, z1);
  const double me = se / (double)n, mg = sg / (double)n;
  double dot = 0, eg = 0, en = 0, z = 0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const double a = (double)e[i] - me, c = (double)g[i] - mg;
    dot += a * c; eg += c * c; en += (a - c) * (a - c);
  }
  reduce4(dot, eg, en, z);
  if (threadIdx.x == 0) {
    const double eps = 1e-8;
    const double k = dot / (eg + eps);                 // scaled target = k * g: |s|^2 = k^2 |g|^2
    const double ratio = k * k * eg / (en + eps);
    const double loss = -10.0 * log10(ratio + eps);
    out[b] = fmaxf((float)loss, clip_min);
  }
}
hipError_t launch_neg_sdsdr(const float* est, const float* tgt, int B, int64_t n_per_item, float clip_min, float* per_item, hipStream_t s) {
  hipLaunchKernelGGL(neg_sdsdr_kernel, dim3(B), dim3(256), 0, s, est, tgt, n_per_item, clip_min, per_item);
  return hipGetLastError();
}

// per item: sum |d| in double (deterministic two-stage: block partials, then one block)
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* pred, const float* target, int64_t n_per_item, double* part) {
  const int b = blockIdx.y;
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per_item; i += (int64_t)gridDim.x * 256)
    s += (double)fabsf(pred[(size_t)b * n_per_item + i] - target[(size_t)b * n_per_item + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[(size_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void l1_final_kernel(const double* part, int nblk, int B, int64_t n_per_item, const int64_t* t, const float* p2w, float* loss, int T) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double tot = 0.0;
  for (int b = 0; b < B; ++b) {
    double s = 0.0;
    for (int k = 0; k < nblk; ++k) s += part[(size_t)b * nblk + k];
    tot += (double)(float)(s / (double)n_per_item) * (double)p2w[clamp_t(t[b], T)];
  }
  loss[0] = (float)(tot / (double)B);
}
__global__ __launch_bounds__(256) void l1_grad_kernel(const float* pred, const float* target, const int64_t* t, const float* p2w,
                                                      int B, int64_t n_per_item, float* grad, int T) {
  const int b = blockIdx.y;
  const float w = p2w[clamp_t(t[b], T)] / ((float)n_per_item * (float)B);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per_item; i += (int64_t)gridDim.x * 256) {
    const size_t k = (size_t)b * n_per_item + i;
    const float d = pred[k] - target[k];
    grad[k] = d > 0.f ? w : (d < 0.f ? -w : 0.f);        // torch: sign(0) = 0
  }
}
size_t l1_loss_ws_bytes(int B) { return (size_t)B * 64 * sizeof(double); }
hipError_t launch_l1_loss(const float* pred, const float* target, const int64_t* t, const float* p2w, int B, int64_t n_per_item,
                          float* loss, float* grad, void* ws, int T, hipStream_t s) {
  const int nblk = (int)std::min<int64_t>((n_per_item + 255) / 256, 64);
  hipLaunchKernelGGL(l1_partial_kernel, dim3(nblk, B), dim3(256), 0, s, pred, target, n_per_item, (double*)ws);
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, nblk, B, n_per_item, t, p2w, loss, T);
  if (grad) hipLaunchKernelGGL(l1_grad_kernel, dim3(nblk, B), dim3(256), 0, s, pred, target, t, p2w, B, n_per_item, grad, T);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// weight standardisation: forward (with saved per-channel 1/std) and backward
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red) {   // 256 threads
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
// 1024 threads (16 waves): red[16]
__device__ __forceinline__ float block_sum16(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) t += red[k];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {   // 256 threads
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void ws_forward_kernel(const float* w, int inner, float* wn, float* rstd) {
  __shared__ float red[4];
  const int o = blockIdx.x;
  const float* p = w + (size_t)o * inner;
  float s = 0.f;
  for (int i = threadIdx.x; i < inner; i += 256) s += p[i];
  const float mean = block_sum(s, red) / (float)inner;
  float q = 0.f;
  for (int i = threadIdx.x; i < inner; i += 256) { const float d = p[i] - mean; q += d * d; }
  const float r = rsqrtf(block_sum(q, red) / (float)inner + 1e-5f);
  for (int i = threadIdx.x; i < inner; i += 256) wn[(size_t)o * inner + i] = (p[i] - mean) * r;
  if (threadIdx.x == 0) rstd[o] = r;
}
// dW = r * (dWn - mean(dWn) - Wn * mean(dWn * Wn))
__global__ __launch_bounds__(256) void ws_backward_kernel(const float* dwn, const float* wn, const float* rstd, int inner, float* dw) {
  __shared__ float red[4];
  const int o = blockIdx.x;
  float s = 0.f, sx = 0.f;
  for (int i = threadIdx.x; i < inner; i += 256) {
    const float g = dwn[(size_t)o * inner + i];
    s += g; sx += g * wn[(size_t)o * inner + i];
  }
  const float ms = block_sum(s, red) / (float)inner;
  const float msx = block_sum(sx, red) / (float)inner;
  const float r = rstd[o];
  for (int i = threadIdx.x; i < inner; i += 256) {
    const size_t k = (size_t)o * inner + i;
    dw[k] = r * (dwn[k] - ms - wn[k] * msx);
  }
}

// ---------------------------------------------------------------------------------------------
// The three GEMM shapes of a Conv1d (any kernel size K, stride S, zero padding P) on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32:
// bitwise an fmaf chain), [B, C, L] layouts, weights [Cout, Cin, K] -- no packing, the weights change every step:
//   MODE 0 forward  y[b]  [Cout x Lout] = sum_t  W_t [Cout x Cin]   . x[b] shifted by t   [Cin x Lout]     (+ bias)
//   MODE 1 dX       dx[b] [Cin x Lin]   = sum_t  W_t^T [Cin x Cout] . dy[b] shifted by -t [Cout x Lin]
//   MODE 2 dW       dW_t  [Cout x Cin]  = sum_b  dy[b] [Cout x Lout] . x[b]^T shifted by t [Lout x Cin]     (grid z = tap)
// 64 x 64 output tile per workgroup (2 x 2 waves of 32 x 32), reduction in chunks of 16 through LDS (k-major, so the 32 lanes
// of a fragment read consecutive words).  The VALU forms above stay as the reference (LDC_TRAIN_VALU=1) and for L < 16.
// ---------------------------------------------------------------------------------------------
int g_train_bf16 = 0;        // set by ldc_create from LDC_TRAIN_BF16 / ldc_set_option("train_bf16")
int g_train_fp32_mfma = 0;   // set by ldc_create from LDC_TRAIN_FP32_MFMA
int g_train_valu = 0;   // set by ldc_create from LDC_TRAIN_VALU (process-wide tuning aid, like g_conv_stamps)

typedef float tf32x16 __attribute__((ext_vector_type(16)));

// TI x TJ 32 x 32 blocks per wave (2 x 2 waves): 64 x 64 output tiles (TI = TJ = 1, the round-2 form: four loads and eight MFMAs per
// thread and reduction chunk) or 128 x 128 (TI = TJ = 2: eight loads per operand and 32 MFMAs per chunk -- the load / LDS / barrier
// work per MFMA halves; round 3, opt-in: see convmm_big).
template <int MODE, int TI, int TJ>
__global__ __launch_bounds__(256) void convmm_kernel(const float* w, const float* src, const float* bias, float* out, int B, int Cin, int Cout,
                                                     int Lin, int Lout, int K, int S, int P, int nsplit) {
  constexpr int BM = 64 * TI, BN = 64 * TJ;
  __shared__ float As[16][BM + 4];
  __shared__ float Bs[16][BN + 4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1, i32 = lane & 31, g = lane >> 5;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  // MODE 2: grid z = tap * nsplit + part; every part reduces its own items and adds its tile to the (pre-zeroed) gradient
  const int z = MODE == 2 ? (int)blockIdx.z / nsplit : (int)blockIdx.z, part = MODE == 2 ? (int)blockIdx.z % nsplit : 0;
  // M, N of this mode
  const int M = MODE == 1 ? Cin : Cout;
  const int N = MODE == 0 ? Lout : (MODE == 1 ? Lin : Cin);
  tf32x16 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b2 = 0; b2 < TJ; ++b2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b2][r] = 0.f;
  // outer / inner reduction ranges: (tap, channel chunks) for MODE 0 / 1, (item, position chunks) for MODE 2
  const int outer_n = MODE == 2 ? B : K;
  const int inner_n = MODE == 0 ? Cin : (MODE == 1 ? Cout : Lout);
  const int outer_lo = MODE == 2 ? (int)((long long)B * part / nsplit) : 0;
  const int outer_hi = MODE == 2 ? (int)((long long)B * (part + 1) / nsplit) : outer_n;
  const int nchunk = (inner_n + 15) / 16;
  const int n_it = (outer_hi - outer_lo) * nchunk;
  // The tiles of reduction step (outer, chunk) -> registers; the global loads of the next step are in flight under the MFMAs of
  // the current one.  Addressing is incremental (round 3): everything that depends on the thread and on `outer` (tap / item) is
  // computed once per `outer` -- the base pointers and the padding / stride predicates -- and a chunk step adds a wave-uniform
  // stride.  The first form recomputed every 64-bit index and bounds check per load: ~240 VALU instructions per 8 MFMAs, which,
  // not the loads, is what held these kernels at a third of the fp32 MFMA rate (eight resident waves per SIMD share its VALU).
  const int kq = tid & 15;                          // reduction index inside a chunk for the k-fastest operands
  const float* pa[4 * TI];
  const float* pb[4 * TJ];
  bool va[4 * TI], vb[4 * TJ];
  int pos2 = 0;                                     // MODE 2: input position of reduction index kq at chunk 0
  const long sA = MODE == 0 ? 16L * K : (MODE == 1 ? 16L * Cin * K : 16L);
  const long sB = MODE == 0 ? 16L * Lin : (MODE == 1 ? 16L * Lout : 16L * S);
  auto setup_outer = [&](int outer) {
#pragma unroll
    for (int p = 0; p < 4 * TI; ++p) {
      const int m = (tid >> 4) + 16 * p;
      if (MODE == 0) { const int o = m0 + m; va[p] = o < Cout; pa[p] = w + ((size_t)(va[p] ? o : 0) * Cin + kq) * K + outer; }
      else if (MODE == 1) { const int i = m0 + m; va[p] = i < Cin; pa[p] = w + ((size_t)kq * Cin + (va[p] ? i : 0)) * K + outer; }
      else { const int o = m0 + m; va[p] = o < Cout; pa[p] = src + ((size_t)outer * Cout + (va[p] ? o : 0)) * Lout + kq; }
    }
    if (MODE == 2) {
      pos2 = kq * S + z - P;
#pragma unroll
      for (int p = 0; p < 4 * TJ; ++p) {
        const int i = n0 + (tid >> 4) + 16 * p;
        vb[p] = i < Cin;
        pb[p] = w + ((size_t)outer * Cin + (vb[p] ? i : 0)) * Lin + pos2;
      }
    } else {
#pragma unroll
      for (int q = 0; q < TJ; ++q) {
        const int n = (tid & 63) + 64 * q;
        bool ok;
        long off;
        if (MODE == 0) {          // B(i, l) = x[b][i][l*S + t - P]
          const int l = n0 + n, pos = l * S + outer - P;
          ok = l < Lout && pos >= 0 && pos < Lin;
          off = ok ? pos : 0;
        } else {                  // B(o, m) = dy[b][o][(m + P - t) / S]
          const int mpos = n0 + n, u = mpos + P - outer;
          ok = mpos < Lin && u >= 0 && (S == 1 || u % S == 0);
          const int l = ok ? (S == 1 ? u : u / S) : 0;
          ok = ok && l < Lout;
          off = ok ? l : 0;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int kk = (tid >> 6) + 4 * p;
          vb[q * 4 + p] = ok;
          pb[q * 4 + p] = src + ((size_t)z * (MODE == 0 ? Cin : Cout) + kk) * (MODE == 0 ? Lin : Lout) + off;
        }
      }
    }
  };
  int f_outer = outer_lo, f_chunk = 0;              // fetch cursor
  auto fetch = [&](float (&ra)[4 * TI], float (&rb)[4 * TJ]) {
    const int k0 = f_chunk * 16;
    const bool ka = k0 + kq < inner_n;
#pragma unroll
    for (int p = 0; p < 4 * TI; ++p) {
      ra[p] = (va[p] && ka) ? *pa[p] : 0.f;
      pa[p] += sA;
    }
    if (MODE == 2) {
      const int pos = pos2 + k0 * S;
      const bool kb = ka && pos >= 0 && pos < Lin;
#pragma unroll
      for (int p = 0; p < 4 * TJ; ++p) {
        rb[p] = (vb[p] && kb) ? *pb[p] : 0.f;
        pb[p] += sB;
      }
    } else {
#pragma unroll
      for (int q = 0; q < TJ; ++q)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const bool kb = k0 + (tid >> 6) + 4 * p < inner_n;
          rb[q * 4 + p] = (vb[q * 4 + p] && kb) ? *pb[q * 4 + p] : 0.f;
          pb[q * 4 + p] += sB;
        }
    }
    if (++f_chunk == nchunk) {
      f_chunk = 0;
      if (++f_outer < outer_hi) setup_outer(f_outer);
    }
  };
  float ra[4 * TI], rb[4 * TJ];
  if (n_it > 0) { setup_outer(f_outer); fetch(ra, rb); }
  for (int it = 0; it < n_it; ++it) {
#pragma unroll
    for (int p = 0; p < 4 * TI; ++p) As[tid & 15][(tid >> 4) + 16 * p] = ra[p];
    if (MODE == 2) {
#pragma unroll
      for (int p = 0; p < 4 * TJ; ++p) Bs[tid & 15][(tid >> 4) + 16 * p] = rb[p];
    } else {
#pragma unroll
      for (int q = 0; q < TJ; ++q)
#pragma unroll
        for (int p = 0; p < 4; ++p) Bs[(tid >> 6) + 4 * p][(tid & 63) + 64 * q] = rb[q * 4 + p];
    }
    __syncthreads();
    if (it + 1 < n_it) fetch(ra, rb);
#pragma unroll
    for (int sx = 0; sx < 8; ++sx) {
      float fa[TI], fb[TJ];
#pragma unroll
      for (int a = 0; a < TI; ++a) fa[a] = As[2 * sx + g][32 * (wm * TI + a) + i32];
#pragma unroll
      for (int b2 = 0; b2 < TJ; ++b2) fb[b2] = Bs[2 * sx + g][32 * (wn * TJ + b2) + i32];
#pragma unroll
      for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b2 = 0; b2 < TJ; ++b2) acc[a][b2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b2], acc[a][b2], 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D layout: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b2 = 0; b2 < TJ; ++b2) {
      const int n = n0 + 32 * (wn * TJ + b2) + i32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * (wm * TI + a) + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (m >= M || n >= N) continue;
        const float v = acc[a][b2][r];
        if (MODE == 0) out[((size_t)z * Cout + m) * Lout + n] = v + (bias ? bias[m] : 0.f);
        else if (MODE == 1) out[((size_t)z * Cin + m) * Lin + n] = v;
        else if (nsplit == 1) out[((size_t)m * Cin + n) * K + z] = v;
        else atomicAdd(&out[((size_t)m * Cin + n) * K + z], v);
      }
    }
}
static bool convmm_ok(int Lin, int Lout) { return !g_train_valu && Lin >= 16 && Lout >= 16; }

// The weight-gradient GEMMs of a backward pass on a side stream (round 6).  dW of a layer is needed by nothing before the optimiser,
// while dX is the critical path and the kernels between two dX GEMMs (GroupNorm / LayerNorm / attention backward: streaming kernels)
// leave the matrix pipes idle.  A Block's backward records an event behind its GroupNorm backward, the side stream waits for it and
// runs dW (+ its ordered reduction, the bias gradient, the weight-standardisation backward) from its own workspace lane, the main
// stream goes on with dX.  launch_train_join makes the main stream wait for the side stream before the gradients are used.
// Everything a side launch reads lives in the Block's saved workspace / saved input (kept by the Python layer until the next forward)
// and everything it writes in that workspace or the flat gradient buffer.  Off by default: DiffusionTrainer switches it on around
// net.backward only (a per-layer caller reads dw right after the call).  Not inside a stream capture.
int g_train_dw_side = 0;
hipStream_t g_train_side_stream = nullptr;
static hipEvent_t g_side_fork_ev = nullptr, g_side_join_ev = nullptr;
hipStream_t train_side_stream();
static hipStream_t dw_side_fork(hipStream_t s) {
  if (!g_train_dw_side || g_train_fp32_mfma || g_train_valu) return nullptr;
  if (g_train_side_stream && s == g_train_side_stream) return nullptr;   // (the caller runs this whole branch on the side stream already)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
  if (!train_side_stream()) return nullptr;
  if (hipEventRecord(g_side_fork_ev, s) != hipSuccess || hipStreamWaitEvent(g_train_side_stream, g_side_fork_ev, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return g_train_side_stream;
}
// the side stream itself (created on first use), for callers that must tell their allocator about it
hipStream_t train_side_stream() {
  if (!g_train_side_stream) {
    if (hipStreamCreateWithFlags(&g_train_side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&g_side_fork_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g_side_join_ev, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      g_train_side_stream = nullptr;
    }
  }
  return g_train_side_stream;
}
hipError_t launch_train_join(hipStream_t s) {
  if (!g_train_side_stream || s == g_train_side_stream) return hipSuccess;
  hipError_t e = hipEventRecord(g_side_join_ev, g_train_side_stream);   // (unconditional: callers may have launched on the side stream themselves)
  if (e == hipSuccess) e = hipStreamWaitEvent(s, g_side_join_ev, 0);
  return e;
}
// 128 x 128 tiles (opt-in, LDC_TRAIN_BIG_TILES=1) when both output dimensions fill them reasonably and the grid still covers the
// chip.  Measured on the full-width step (32 x 2.4 s): 125.1 ms against 122.5 ms with 64 x 64 tiles everywhere -- the kernel is not
// bound by its loads per MFMA (the fp32 MFMA is 64 cycles; eight resident workgroups per CU cover the rest), so the default stays.
static bool convmm_big(int M, int N, long tiles_other) {
  const bool on = false;   // (measured slower, see above; the variant stays compiled for the exact-fp32 path's probes)
  if (!on) return false;
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * tiles_other;
  const double fill = (double)M * N / ((double)((M + 127) / 128 * 128) * ((N + 127) / 128 * 128));
  return M >= 128 && N >= 96 && fill >= 0.7 && t128 >= 192;
}
static void convmm_forward(const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int Lin, int Lout, int K, int S, int P,
                           float* y, hipStream_t s) {
  if (!g_train_fp32_mfma) { (void)launch_mm3_forward(x, w, bias, B, Cin, Cout, Lin, Lout, K, S, P, y, s); return; }
  if (convmm_big(Cout, Lout, B))
    hipLaunchKernelGGL((convmm_kernel<0, 2, 2>), dim3((Lout + 127) / 128, (Cout + 127) / 128, B), dim3(256), 0, s, w, x, bias, y, B, Cin, Cout, Lin, Lout, K, S, P, 1);
  else
    hipLaunchKernelGGL((convmm_kernel<0, 1, 1>), dim3((Lout + 63) / 64, (Cout + 63) / 64, B), dim3(256), 0, s, w, x, bias, y, B, Cin, Cout, Lin, Lout, K, S, P, 1);
}
static void convmm_dx(const float* dy, const float* w, int B, int Cin, int Cout, int Lin, int Lout, int K, int S, int P, float* dx, hipStream_t s) {
  if (!g_train_fp32_mfma) { (void)launch_mm3_dx(dy, w, B, Cin, Cout, Lin, Lout, K, S, P, dx, s); return; }
  if (convmm_big(Cin, Lin, B))
    hipLaunchKernelGGL((convmm_kernel<1, 2, 2>), dim3((Lin + 127) / 128, (Cin + 127) / 128, B), dim3(256), 0, s, w, dy, nullptr, dx, B, Cin, Cout, Lin, Lout, K, S, P, 1);
  else
    hipLaunchKernelGGL((convmm_kernel<1, 1, 1>), dim3((Lin + 63) / 64, (Cin + 63) / 64, B), dim3(256), 0, s, w, dy, nullptr, dx, B, Cin, Cout, Lin, Lout, K, S, P, 1);
}
// -> true when the bias gradient was produced too (split-bf16 path: fused into the dW kernel)
static bool convmm_dw(const float* dy, const float* x, int B, int Cin, int Cout, int Lin, int Lout, int K, int S, int P, float* dw, hipStream_t s,
                      float* db = nullptr) {
  if (!g_train_fp32_mfma) { (void)launch_mm3_dw(dy, x, B, Cin, Cout, Lin, Lout, K, S, P, dw, s, db); return db != nullptr; }
  // few output tiles, a long reduction over the items: split the items over workgroups (fp32 atomics into the zeroed gradient:
  // the sum order varies from run to run at the 1e-7 level) until the grid fills the chip
  const bool big = convmm_big(Cout, Cin, (long)K * B);
  const int T = big ? 128 : 64;
  const int tiles = ((Cin + T - 1) / T) * ((Cout + T - 1) / T) * K;
  const int nsplit = std::max(1, std::min(B, (768 + tiles - 1) / tiles));
  if (nsplit > 1) (void)hipMemsetAsync(dw, 0, (size_t)Cout * Cin * K * sizeof(float), s);
  if (big)
    hipLaunchKernelGGL((convmm_kernel<2, 2, 2>), dim3((Cin + 127) / 128, (Cout + 127) / 128, K * nsplit), dim3(256), 0, s, x, dy, nullptr, dw, B, Cin, Cout, Lin, Lout,
                       K, S, P, nsplit);
  else
    hipLaunchKernelGGL((convmm_kernel<2, 1, 1>), dim3((Cin + 63) / 64, (Cout + 63) / 64, K * nsplit), dim3(256), 0, s, x, dy, nullptr, dw, B, Cin, Cout, Lin, Lout, K,
                       S, P, nsplit);
  return false;
}

// ---------------------------------------------------------------------------------------------
// Conv1d k = 3, padding 1: forward, dX, dW, db  ([B, C, L] fp32)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv3_forward_kernel(const float* x, const float* w, const float* bias, int Cin, int Cout, int L,
                                                            float* y) {
  const int b = blockIdx.z, o = blockIdx.y;
  const float* wr = w + (size_t)o * Cin * 3;
  for (int l = blockIdx.x * 256 + threadIdx.x; l < L; l += gridDim.x * 256) {
    float acc = bias ? bias[o] : 0.f;
    for (int i = 0; i < Cin; ++i) {
      const float* xr = x + ((size_t)b * Cin + i) * L;
      const float xm = l > 0 ? xr[l - 1] : 0.f, xc = xr[l], xp = l + 1 < L ? xr[l + 1] : 0.f;
      acc = fmaf(wr[3 * i], xm, acc); acc = fmaf(wr[3 * i + 1], xc, acc); acc = fmaf(wr[3 * i + 2], xp, acc);
    }
    y[((size_t)b * Cout + o) * L + l] = acc;
  }
}
// dx[b,i,l] = sum_{o,t} w[o,i,t] * dh[b,o,l - t + 1]
__global__ __launch_bounds__(256) void conv3_dx_kernel(const float* dh, const float* w, int Cin, int Cout, int L, float* dx) {
  const int b = blockIdx.z, i = blockIdx.y;
  for (int l = blockIdx.x * 256 + threadIdx.x; l < L; l += gridDim.x * 256) {
    float acc = 0.f;
    for (int o = 0; o < Cout; ++o) {
      const float* dr = dh + ((size_t)b * Cout + o) * L;
      const float* wr = w + ((size_t)o * Cin + i) * 3;
      const float dp = l + 1 < L ? dr[l + 1] : 0.f, dc = dr[l], dm = l > 0 ? dr[l - 1] : 0.f;
      acc = fmaf(wr[0], dp, acc); acc = fmaf(wr[1], dc, acc); acc = fmaf(wr[2], dm, acc);
    }
    dx[((size_t)b * Cin + i) * L + l] = acc;
  }
}
// dw[o,i,t] = sum_{b,l} dh[b,o,l] * x[b,i,l + t - 1];  one block per (o, i), fixed-order reduction
__global__ __launch_bounds__(256) void conv3_dw_kernel(const float* dh, const float* x, int B, int Cin, int Cout, int L, float* dw) {
  __shared__ float red[4];
  const int o = blockIdx.y, i = blockIdx.x;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* dr = dh + ((size_t)b * Cout + o) * L;
    const float* xr = x + ((size_t)b * Cin + i) * L;
    for (int l = threadIdx.x; l < L; l += 256) {
      const float d = dr[l];
      if (l > 0) a0 = fmaf(d, xr[l - 1], a0);
      a1 = fmaf(d, xr[l], a1);
      if (l + 1 < L) a2 = fmaf(d, xr[l + 1], a2);
    }
  }
  const float s0 = block_sum(a0, red), s1 = block_sum(a1, red), s2 = block_sum(a2, red);
  if (threadIdx.x == 0) {
    float* p = dw + ((size_t)o * Cin + i) * 3;
    p[0] = s0; p[1] = s1; p[2] = s2;
  }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm + (scale + 1, shift) + SiLU: forward and backward.  One block per (item, group).
// ---------------------------------------------------------------------------------------------
// One workgroup of 1024 threads per (item, group) (round 3; the first form had 256 threads -- one workgroup of four waves per CU
// does not keep enough loads in flight to stream its 150 KB three times -- and an integer division per element).
__global__ __launch_bounds__(1024) void gn_silu_forward_kernel(const float* h, const float* gamma, const float* beta, const float* ss,
                                                               int C, int L, int groups, float* y, float* stats) {
  __shared__ float red[16];
  const int b = blockIdx.y, g = blockIdx.x, cpg = C / groups, n = cpg * L;
  const float* p = h + ((size_t)b * C + (size_t)g * cpg) * L;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) s += p[i];
  const float mean = block_sum16(s, red) / (float)n;
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) { const float d = p[i] - mean; q += d * d; }
  const float rstd = rsqrtf(block_sum16(q, red) / (float)n + 1e-5f);
  for (int cc = 0; cc < cpg; ++cc) {
    const int c = g * cpg + cc;
    const float sc = ss ? ss[(size_t)b * 2 * C + c] + 1.0f : 1.0f, sh = ss ? ss[(size_t)b * 2 * C + C + c] : 0.0f;
    const float* pr = p + (size_t)cc * L;
    float* yr = y + ((size_t)b * C + c) * L;
    for (int l = threadIdx.x; l < L; l += 1024) {
      float v = (pr[l] - mean) * rstd * gamma[c] + beta[c];
      if (ss) v = v * sc + sh;
      yr[l] = v / (1.0f + expf(-v));
    }
  }
  if (threadIdx.x == 0) { stats[((size_t)b * groups + g) * 2] = mean; stats[((size_t)b * groups + g) * 2 + 1] = rstd; }
}

// The same op with the (item, group) slab kept in LDS (150 KB at every level of the full-width UNet: 32 x 1200, 64 x 600 ... floats; gfx950
// gives one workgroup up to 160 KB): ONE read of h from memory instead of three, identical arithmetic in identical order (the
// partial sums are per thread over i = tid, tid + 1024, ... exactly as above), so the results are bit-identical.
__global__ __launch_bounds__(1024) void gn_silu_forward_lds_kernel(const float* h, const float* gamma, const float* beta, const float* ss,
                                                                   int C, int L, int groups, float* y, float* stats) {
  extern __shared__ float slab[];
  __shared__ float red[16];
  const int b = blockIdx.y, g = blockIdx.x, cpg = C / groups, n = cpg * L;
  const float* p = h + ((size_t)b * C + (size_t)g * cpg) * L;
  float s = 0.f;
  int i0 = threadIdx.x;
  // (round 6: eight loads in flight per thread -- one workgroup per CU streams its 150 KB with 16 waves; same elements per thread in the
  // same order: bit-identical sums)
  for (; i0 + 7 * 1024 < n; i0 += 8 * 1024) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[i0 + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) { slab[i0 + u * 1024] = v[u]; s += v[u]; }
  }
  for (; i0 < n; i0 += 1024) { const float v = p[i0]; slab[i0] = v; s += v; }
  const float mean = block_sum16(s, red) / (float)n;       // (the barriers inside make the slab visible to every thread)
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) { const float d = slab[i] - mean; q += d * d; }
  const float rstd = rsqrtf(block_sum16(q, red) / (float)n + 1e-5f);
  for (int cc = 0; cc < cpg; ++cc) {
    const int c = g * cpg + cc;
    const float sc = ss ? ss[(size_t)b * 2 * C + c] + 1.0f : 1.0f, sh = ss ? ss[(size_t)b * 2 * C + C + c] : 0.0f;
    const float* pr = slab + (size_t)cc * L;
    float* yr = y + ((size_t)b * C + c) * L;
    for (int l = threadIdx.x; l < L; l += 1024) {
      float v = (pr[l] - mean) * rstd * gamma[c] + beta[c];
      if (ss) v = v * sc + sh;
      yr[l] = v / (1.0f + expf(-v));
    }
  }
  if (threadIdx.x == 0) { stats[((size_t)b * groups + g) * 2] = mean; stats[((size_t)b * groups + g) * 2 + 1] = rstd; }
}
static void launch_gn_silu_forward(const float* h, const float* gamma, const float* beta, const float* ss, int B, int C, int L, int groups,
                                   float* y, float* stats, hipStream_t s) {
  const size_t slab = (size_t)(C / groups) * L * sizeof(float);
  static bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(gn_silu_forward_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            160 * 1024 - 256) == hipSuccess;
  if (attr_ok && slab <= (size_t)160 * 1024 - 256 && !g_train_valu)
    hipLaunchKernelGGL(gn_silu_forward_lds_kernel, dim3(groups, B), dim3(1024), slab, s, h, gamma, beta, ss, C, L, groups, y, stats);
  else
    hipLaunchKernelGGL(gn_silu_forward_kernel, dim3(groups, B), dim3(1024), 0, s, h, gamma, beta, ss, C, L, groups, y, stats);
}

// pass 1: per (item, channel): dz -> dn = dz * (scale + 1); writes dxhat = dn * gamma into `tmp`, the per-(b,c) sums
// dscale = sum dz * nval, dshift = sum dz, and pgam[b][c] = sum dn * xhat, pbet[b][c] = sum dn
__global__ __launch_bounds__(256) void gn_silu_backward1_kernel(const float* dy, const float* h, const float* gamma, const float* beta,
                                                                const float* ss, const float* stats, int C, int L, int groups,
                                                                float* tmp, float* dss, float* pgam, float* pbet) {
  __shared__ float red[4];
  const int b = blockIdx.y, c = blockIdx.x, g = c / (C / groups);
  const float mean = stats[((size_t)b * groups + g) * 2], rstd = stats[((size_t)b * groups + g) * 2 + 1];
  const float sc = ss ? ss[(size_t)b * 2 * C + c] + 1.0f : 1.0f, sh = ss ? ss[(size_t)b * 2 * C + C + c] : 0.0f;
  const float ga = gamma[c], be = beta[c];
  const size_t base = ((size_t)b * C + c) * L;
  float s_dz = 0.f, s_dzn = 0.f, s_dnx = 0.f;
  for (int l = threadIdx.x; l < L; l += 256) {
    const float xh = (h[base + l] - mean) * rstd;
    const float nv = xh * ga + be;
    const float z = nv * sc + sh;
    const float sg = 1.0f / (1.0f + expf(-z));
    const float dz = dy[base + l] * (sg * (1.0f + z * (1.0f - sg)));
    const float dn = dz * sc;
    s_dz += dz; s_dzn += dz * nv; s_dnx += dn * xh;
    tmp[base + l] = dn * ga;
  }
  const float t_dz = block_sum(s_dz, red), t_dzn = block_sum(s_dzn, red), t_dnx = block_sum(s_dnx, red);
  if (threadIdx.x == 0) {
    if (dss) { dss[(size_t)b * 2 * C + c] = t_dzn; dss[(size_t)b * 2 * C + C + c] = t_dz; }
    pgam[(size_t)b * C + c] = t_dnx;
    pbet[(size_t)b * C + c] = t_dz * sc;
  }
}
// pass 2: per (item, group): dh = rstd / n * (n * dxhat - sum dxhat - xhat * sum(dxhat * xhat))
__global__ __launch_bounds__(1024) void gn_silu_backward2_kernel(const float* h, const float* stats, int C, int L, int groups, float* tmp_dh) {
  __shared__ float red[16];
  const int b = blockIdx.y, g = blockIdx.x, cpg = C / groups, n = cpg * L;
  const float mean = stats[((size_t)b * groups + g) * 2], rstd = stats[((size_t)b * groups + g) * 2 + 1];
  const size_t base = ((size_t)b * C + (size_t)g * cpg) * L;
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float d = tmp_dh[base + i];
    s1 += d; s2 += d * (h[base + i] - mean) * rstd;
  }
  const float t1 = block_sum16(s1, red), t2 = block_sum16(s2, red);
  const float inv_n = 1.0f / (float)n;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float xh = (h[base + i] - mean) * rstd;
    tmp_dh[base + i] = rstd * (tmp_dh[base + i] - t1 * inv_n - xh * t2 * inv_n);
  }
}
// dgamma[c] = sum_b pgam[b][c], dbeta likewise; db[o] = sum_{b,l} dh[b,o,l]
__global__ __launch_bounds__(256) void reduce_items_kernel(const float* p, int B, int C, float* out) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += p[(size_t)b * C + c];
    out[c] = s;
  }
}
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* dh, int B, int C, int L, float* db) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int b = 0; b < B; ++b)
    for (int l = threadIdx.x; l < L; l += 256) s += dh[((size_t)b * C + c) * L + l];
  const float t = block_sum(s, red);
  if (threadIdx.x == 0) db[c] = t;
}

size_t train_block_ws_floats(int B, int Cin, int Cout, int L, int groups) {
  // wn [Cout*Cin*3] | rstd_w [Cout] | h [B*Cout*L] | gn stats [B*groups*2] | tmp [B*Cout*L] | dwn [Cout*Cin*3] | pgam, pbet [B*Cout] x 2
  return (size_t)2 * Cout * Cin * 3 + Cout + (size_t)2 * B * Cout * L + (size_t)B * groups * 2 + (size_t)2 * B * Cout + 64;
}

struct BlockWs { float *wn, *rstd_w, *h, *stats, *tmp, *dwn, *pgam, *pbet; };
static BlockWs carve(float* ws, int B, int Cin, int Cout, int L, int groups) {
  BlockWs w;
  float* p = ws;
  w.wn = p; p += (size_t)Cout * Cin * 3;
  w.rstd_w = p; p += Cout;
  w.h = p; p += (size_t)B * Cout * L;
  w.stats = p; p += (size_t)B * groups * 2;
  w.tmp = p; p += (size_t)B * Cout * L;
  w.dwn = p; p += (size_t)Cout * Cin * 3;
  w.pgam = p; p += (size_t)B * Cout;
  w.pbet = p;
  return w;
}

hipError_t launch_train_block_forward(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                                      const float* ss, int B, int Cin, int Cout, int L, int groups, float* y, float* ws, hipStream_t s) {
  const BlockWs k = carve(ws, B, Cin, Cout, L, groups);
  hipLaunchKernelGGL(ws_forward_kernel, dim3(Cout), dim3(256), 0, s, w, Cin * 3, k.wn, k.rstd_w);
  if (convmm_ok(L, L)) convmm_forward(x, k.wn, bias, B, Cin, Cout, L, L, 3, 1, 1, k.h, s);
  else hipLaunchKernelGGL(conv3_forward_kernel, dim3((L + 255) / 256, Cout, B), dim3(256), 0, s, x, k.wn, bias, Cin, Cout, L, k.h);
  launch_gn_silu_forward(k.h, gamma, beta, ss, B, Cout, L, groups, y, k.stats, s);
  return hipGetLastError();
}

hipError_t launch_train_block_backward(const float* dy, const float* x, const float* gamma, const float* beta, const float* ss, int B,
                                       int Cin, int Cout, int L, int groups, float* ws, float* dx, float* dw, float* db, float* dgamma,
                                       float* dbeta, float* dss, hipStream_t s) {
  const BlockWs k = carve(ws, B, Cin, Cout, L, groups);
  hipLaunchKernelGGL(gn_silu_backward1_kernel, dim3(Cout, B), dim3(256), 0, s, dy, k.h, gamma, beta, ss, k.stats, Cout, L, groups, k.tmp,
                     dss, k.pgam, k.pbet);
  hipLaunchKernelGGL(gn_silu_backward2_kernel, dim3(groups, B), dim3(1024), 0, s, k.h, k.stats, Cout, L, groups, k.tmp);   // tmp := dh
  // the parameter gradients (the GroupNorm's: sums of pgam / pbet over the items; the weight's: reads tmp and x, writes dwn, dw, db): on
  // the side stream when the trainer has switched that on
  hipStream_t sd = convmm_ok(L, L) ? dw_side_fork(s) : nullptr;
  hipStream_t sw = sd ? sd : s;
  hipLaunchKernelGGL(reduce_items_kernel, dim3((Cout + 255) / 256), dim3(256), 0, sw, k.pgam, B, Cout, dgamma);
  hipLaunchKernelGGL(reduce_items_kernel, dim3((Cout + 255) / 256), dim3(256), 0, sw, k.pbet, B, Cout, dbeta);
  bool db_done = false;
  if (convmm_ok(L, L)) db_done = convmm_dw(k.tmp, x, B, Cin, Cout, L, L, 3, 1, 1, k.dwn, sw, db);
  else hipLaunchKernelGGL(conv3_dw_kernel, dim3(Cin, Cout), dim3(256), 0, sw, k.tmp, x, B, Cin, Cout, L, k.dwn);
  if (!db_done) hipLaunchKernelGGL(bias_grad_kernel, dim3(Cout), dim3(256), 0, sw, k.tmp, B, Cout, L, db);
  hipLaunchKernelGGL(ws_backward_kernel, dim3(Cout), dim3(256), 0, sw, k.dwn, k.wn, k.rstd_w, Cin * 3, dw);
  if (dx) {
    if (convmm_ok(L, L)) convmm_dx(k.tmp, k.wn, B, Cin, Cout, L, L, 3, 1, 1, dx, s);
    else hipLaunchKernelGGL(conv3_dx_kernel, dim3((L + 255) / 256, Cin, B), dim3(256), 0, s, k.tmp, k.wn, Cin, Cout, L, dx);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Adam step (srcs/train.py:365-371: optim.Adam(params, lr), default betas (0.9, 0.999), eps 1e-8, no weight decay), flat buffers.
// torch.optim.Adam's arithmetic: m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2; p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                                                   float eps, float step_size, float inv_sqrt_bc2) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.0f - b1) * gi;          // (torch: m.lerp_(g, 1 - b1) == m + (g - m)(1 - b1); same to rounding)
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] -= step_size * (mi / denom);
  }
}
// The same with the step count on the device (a captured optimisation step: the count advances with every replay).  `step_dev` is
// counted up by a one-thread kernel in front; the bias corrections are evaluated per thread in fp32 (exp2 / log2: ~1e-7 relative, against
// the host's double in launch_adam).
__global__ void adam_tick_kernel(int* step_dev) { step_dev[0] += 1; }
__global__ __launch_bounds__(256) void adam_dev_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                                                       float eps, const int* step_dev) {
  const float st = (float)step_dev[0];
  const float bc1 = 1.0f - exp2f(st * log2f(b1)), bc2 = 1.0f - exp2f(st * log2f(b2));
  const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] -= step_size * (mi / denom);
  }
}
hipError_t launch_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, int* step_dev, float lr, float b1, float b2, float eps, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, s, step_dev);
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, step_dev);
  return hipGetLastError();
}
hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr, float b1, float b2, float eps, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)));
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// channel LayerNorm of the UNet (srcs/modules/unet.py:82-101): per position, over C: y = (x - mean) * rsqrt(var + 1e-5) * g
// ([B, C, L] fp32; var biased).  backward: dx = rstd * (dxhat - mean_c(dxhat) - xhat * mean_c(dxhat * xhat)), dxhat = dy * g;
// dg[c] = sum_{b,l} dy * xhat (two-stage, fixed order).
// ---------------------------------------------------------------------------------------------
// Layout [B, C, L]: the reduction over channels has stride L.  A workgroup takes 64 consecutive positions of one item; its four
// waves split the channels (wave w takes c = w, w + 4, ..), every load instruction of a wave is one coalesced 256-byte row piece,
// four loads are in flight per thread, and the four partial sums meet in LDS.  (The first version gave one thread all C
// channels of its position: 3 x C dependent strided loads per thread on 160 workgroups -- 340-390 us per call at 32 x 256 x
// 1200, 12 % of the optimisation step.)
__device__ __forceinline__ float ln_quad_sum(float v, float (*red)[64], int w, int lx) {
  red[w][lx] = v;
  __syncthreads();
  const float r = (red[0][lx] + red[1][lx]) + (red[2][lx] + red[3][lx]);
  __syncthreads();
  return r;
}
__global__ __launch_bounds__(256) void ln_forward_kernel(const float* x, const float* g, int C, int L, float* y, float* stats) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, lx = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int l = blockIdx.x * 64 + lx;
  const bool ok = l < L;
  const float* xb = x + (size_t)b * C * L + (ok ? l : 0);
  float s = 0.f;
  int c = w;
  for (; c + 12 < C; c += 16) {
    const float v0 = xb[(size_t)c * L], v1 = xb[(size_t)(c + 4) * L], v2 = xb[(size_t)(c + 8) * L], v3 = xb[(size_t)(c + 12) * L];
    s += (v0 + v1) + (v2 + v3);
  }
  for (; c < C; c += 4) s += xb[(size_t)c * L];
  const float mean = ln_quad_sum(s, red, w, lx) / (float)C;
  float ss = 0.f;
  c = w;
  for (; c + 12 < C; c += 16) {
    const float d0 = xb[(size_t)c * L] - mean, d1 = xb[(size_t)(c + 4) * L] - mean, d2 = xb[(size_t)(c + 8) * L] - mean, d3 = xb[(size_t)(c + 12) * L] - mean;
    ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  for (; c < C; c += 4) { const float d = xb[(size_t)c * L] - mean; ss += d * d; }
  const float rstd = rsqrtf(ln_quad_sum(ss, red, w, lx) / (float)C + 1e-5f);
  if (!ok) return;
  float* yb = y + (size_t)b * C * L + l;
  c = w;
  for (; c + 28 < C; c += 32) {   // (eight loads in flight, as ln_backward_dx_kernel)
    float xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = xb[(size_t)(c + 4 * u) * L];
#pragma unroll
    for (int u = 0; u < 8; ++u) yb[(size_t)(c + 4 * u) * L] = (xv[u] - mean) * rstd * g[c + 4 * u];
  }
  for (; c < C; c += 4) yb[(size_t)c * L] = (xb[(size_t)c * L] - mean) * rstd * g[c];
  if (w == 0) {
    stats[((size_t)b * L + l) * 2] = mean;
    stats[((size_t)b * L + l) * 2 + 1] = rstd;
  }
}
__global__ __launch_bounds__(256) void ln_backward_dx_kernel(const float* dy, const float* x, const float* g, const float* stats, int C, int L,
                                                             float* dx) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, lx = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int l = blockIdx.x * 64 + lx;
  const bool ok = l < L;
  const size_t base = (size_t)b * C * L + (ok ? l : 0);
  const size_t sidx = ((size_t)b * L + (ok ? l : 0)) * 2;
  const float mean = stats[sidx], rstd = stats[sidx + 1];
  float s1 = 0.f, s2 = 0.f;
  int c = w;
  // (round 6: eight channels' loads in flight per thread instead of two -- the kernel is a latency chain of 256-byte row segments; the
  // order in which a thread adds its channels is unchanged: bit-identical sums)
  for (; c + 28 < C; c += 32) {
    float xv[8], dv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { xv[u] = x[base + (size_t)(c + 4 * u) * L]; dv[u] = dy[base + (size_t)(c + 4 * u) * L]; }
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      const float da = dv[u] * g[c + 4 * u], db2 = dv[u + 1] * g[c + 4 * u + 4];
      s1 += da + db2;
      s2 += da * ((xv[u] - mean) * rstd) + db2 * ((xv[u + 1] - mean) * rstd);
    }
  }
  for (; c + 4 < C; c += 8) {
    const float xa = x[base + (size_t)c * L], xb2 = x[base + (size_t)(c + 4) * L];
    const float da = dy[base + (size_t)c * L] * g[c], db2 = dy[base + (size_t)(c + 4) * L] * g[c + 4];
    s1 += da + db2;
    s2 += da * ((xa - mean) * rstd) + db2 * ((xb2 - mean) * rstd);
  }
  for (; c < C; c += 4) {
    const float xh = (x[base + (size_t)c * L] - mean) * rstd, dxh = dy[base + (size_t)c * L] * g[c];
    s1 += dxh;
    s2 += dxh * xh;
  }
  s1 = ln_quad_sum(s1, red, w, lx) / (float)C;
  s2 = ln_quad_sum(s2, red, w, lx) / (float)C;
  if (!ok) return;
  c = w;
  for (; c + 28 < C; c += 32) {
    float xv[8], dv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { xv[u] = x[base + (size_t)(c + 4 * u) * L]; dv[u] = dy[base + (size_t)(c + 4 * u) * L]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float xh = (xv[u] - mean) * rstd, dxh = dv[u] * g[c + 4 * u];
      dx[base + (size_t)(c + 4 * u) * L] = rstd * (dxh - s1 - xh * s2);
    }
  }
  for (; c < C; c += 4) {
    const float xh = (x[base + (size_t)c * L] - mean) * rstd, dxh = dy[base + (size_t)c * L] * g[c];
    dx[base + (size_t)c * L] = rstd * (dxh - s1 - xh * s2);
  }
}
// Register-resident forward for C = 4 * NC (NC = 64: the C = 256 level, the largest tensors): a thread's channels are loaded once, all loads
// in flight together, and the three passes run over registers -- same operations in the same order as the kernel above, so the results
// are bit-identical; x is read once instead of three times.  (NC = 128 needs 239 registers, and a register-resident backward 286-444:
// one or two waves per SIMD: not kept.)
template <int NC>
__global__ __launch_bounds__(256) void ln_forward_reg_kernel(const float* x, const float* g, int C, int L, float* y, float* stats) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, lx = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int l = blockIdx.x * 64 + lx;
  const bool ok = l < L;
  const float* xb = x + (size_t)b * C * L + (ok ? l : 0);
  float v[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) v[k] = xb[(size_t)(w + 4 * k) * L];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NC; k += 4) s += (v[k] + v[k + 1]) + (v[k + 2] + v[k + 3]);
  const float mean = ln_quad_sum(s, red, w, lx) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < NC; k += 4) {
    const float d0 = v[k] - mean, d1 = v[k + 1] - mean, d2 = v[k + 2] - mean, d3 = v[k + 3] - mean;
    ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  const float rstd = rsqrtf(ln_quad_sum(ss, red, w, lx) / (float)C + 1e-5f);
  if (!ok) return;
  float* yb = y + (size_t)b * C * L + l;
#pragma unroll
  for (int k = 0; k < NC; ++k) yb[(size_t)(w + 4 * k) * L] = (v[k] - mean) * rstd * g[w + 4 * k];
  if (w == 0) {
    stats[((size_t)b * L + l) * 2] = mean;
    stats[((size_t)b * L + l) * 2 + 1] = rstd;
  }
}
// one block per channel: dg[c] = sum_{b,l} dy * xhat in a fixed order
__global__ __launch_bounds__(256) void ln_backward_dg_kernel(const float* dy, const float* x, const float* stats, int B, int C, int L, float* dg) {
  const int c = blockIdx.x;
  __shared__ float red[256];
  float acc = 0.f;
  for (int idx = threadIdx.x; idx < B * L; idx += 256) {
    const int b = idx / L, l = idx - b * L;
    const float mean = stats[((size_t)b * L + l) * 2], rstd = stats[((size_t)b * L + l) * 2 + 1];
    const size_t off = ((size_t)b * C + c) * L + l;
    acc += dy[off] * (x[off] - mean) * rstd;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) dg[c] = red[0];
}
hipError_t launch_train_ln_forward(const float* x, const float* g, int B, int C, int L, float* y, float* stats, hipStream_t s) {
  const dim3 grid((L + 63) / 64, B);
  if (C == 256) hipLaunchKernelGGL((ln_forward_reg_kernel<64>), grid, dim3(256), 0, s, x, g, C, L, y, stats);
  else hipLaunchKernelGGL(ln_forward_kernel, grid, dim3(256), 0, s, x, g, C, L, y, stats);
  return hipGetLastError();
}
hipError_t launch_train_ln_backward(const float* dy, const float* x, const float* g, const float* stats, int B, int C, int L, float* dx,
                                    float* dg, hipStream_t s) {
  const dim3 grid((L + 63) / 64, B);
  hipLaunchKernelGGL(ln_backward_dx_kernel, grid, dim3(256), 0, s, dy, x, g, stats, C, L, dx);
  hipStream_t sd = dw_side_fork(s);   // the gain's gradient feeds the optimiser only (as launch_train_conv_backward)
  hipLaunchKernelGGL(ln_backward_dg_kernel, dim3(C), dim3(256), 0, sd ? sd : s, dy, x, stats, B, C, L, dg);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// pointwise linear maps on [B, C, L] (L = 1: nn.Linear on [B, K]): the 1x1 res_conv of a ResnetBlock (unet.py:171,192) and its
// time-embedding MLP SiLU -> Linear (unet.py:163-166: `pre_silu`), forward and backward
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float silu_grad_f(float v) { const float sg = 1.0f / (1.0f + expf(-v)); return sg * (1.0f + v * (1.0f - sg)); }

__global__ __launch_bounds__(256) void pw_forward_kernel(const float* x, const float* w, const float* bias, int Cin, int Cout, int L, int pre_silu,
                                                         float* y) {
  const int b = blockIdx.z, o = blockIdx.y;
  const float* wr = w + (size_t)o * Cin;
  for (int l = blockIdx.x * 256 + threadIdx.x; l < L; l += gridDim.x * 256) {
    float acc = bias ? bias[o] : 0.f;
    for (int i = 0; i < Cin; ++i) {
      float v = x[((size_t)b * Cin + i) * L + l];
      if (pre_silu) v = silu_f(v);
      acc = fmaf(wr[i], v, acc);
    }
    y[((size_t)b * Cout + o) * L + l] = acc;
  }
}
__global__ __launch_bounds__(256) void pw_dx_kernel(const float* dy, const float* x, const float* w, int Cin, int Cout, int L, int pre_silu, float* dx) {
  const int b = blockIdx.z, i = blockIdx.y;
  for (int l = blockIdx.x * 256 + threadIdx.x; l < L; l += gridDim.x * 256) {
    float acc = 0.f;
    for (int o = 0; o < Cout; ++o) acc = fmaf(w[(size_t)o * Cin + i], dy[((size_t)b * Cout + o) * L + l], acc);
    if (pre_silu) acc *= silu_grad_f(x[((size_t)b * Cin + i) * L + l]);
    dx[((size_t)b * Cin + i) * L + l] = acc;
  }
}
// dw[o,i] = sum_{b,l} dy[b,o,l] * a[b,i,l], a = x or SiLU(x); one block per (o, i), fixed-order reduction
__global__ __launch_bounds__(256) void pw_dw_kernel(const float* dy, const float* x, int B, int Cin, int Cout, int L, int pre_silu, float* dw) {
  __shared__ float red[4];
  const int o = blockIdx.y, i = blockIdx.x;
  float a = 0.f;
  for (int idx = threadIdx.x; idx < B * L; idx += 256) {
    const int b = idx / L, l = idx - b * L;
    float v = x[((size_t)b * Cin + i) * L + l];
    if (pre_silu) v = silu_f(v);
    a = fmaf(dy[((size_t)b * Cout + o) * L + l], v, a);
  }
  const float t = block_sum(a, red);
  if (threadIdx.x == 0) dw[(size_t)o * Cin + i] = t;
}
// L == 1 (nn.Linear on [B, Cin], the time-embedding MLPs): one block per output, the reduction spread over the threads
__global__ __launch_bounds__(256) void lin_forward_kernel(const float* x, const float* w, const float* bias, int Cin, int Cout, int pre_silu, float* y) {
  __shared__ float red[4];
  const int o = blockIdx.x, b = blockIdx.y;
  float a = 0.f;
  for (int i = threadIdx.x; i < Cin; i += 256) {
    float v = x[(size_t)b * Cin + i];
    if (pre_silu) v = silu_f(v);
    a = fmaf(w[(size_t)o * Cin + i], v, a);
  }
  a = block_sum(a, red);
  if (threadIdx.x == 0) y[(size_t)b * Cout + o] = a + (bias ? bias[o] : 0.f);
}
__global__ __launch_bounds__(256) void lin_dx_kernel(const float* dy, const float* x, const float* w, int Cin, int Cout, int pre_silu, float* dx) {
  __shared__ float red[4];
  const int i = blockIdx.x, b = blockIdx.y;
  float a = 0.f;
  for (int o = threadIdx.x; o < Cout; o += 256) a = fmaf(w[(size_t)o * Cin + i], dy[(size_t)b * Cout + o], a);
  a = block_sum(a, red);
  if (threadIdx.x == 0) dx[(size_t)b * Cin + i] = pre_silu ? a * silu_grad_f(x[(size_t)b * Cin + i]) : a;
}
// dw[o, i] = sum_b dy[b, o] * a(x[b, i]): one block per o, threads over i
__global__ __launch_bounds__(256) void lin_dw_kernel(const float* dy, const float* x, int B, int Cin, int Cout, int pre_silu, float* dw) {
  const int o = blockIdx.x;
  for (int i = threadIdx.x; i < Cin; i += 256) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) {
      float v = x[(size_t)b * Cin + i];
      if (pre_silu) v = silu_f(v);
      a = fmaf(dy[(size_t)b * Cout + o], v, a);
    }
    dw[(size_t)o * Cin + i] = a;
  }
}
// elementwise helpers of the L == 1 (nn.Linear) path on the GEMM kernels: a = SiLU(x) into the workspace, dx *= SiLU'(x)
__global__ __launch_bounds__(256) void silu_map_kernel(const float* x, int64_t n, float* y) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = silu_f(x[i]);
}
__global__ __launch_bounds__(256) void silu_grad_mul_kernel(const float* x, int64_t n, float* dx) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dx[i] *= silu_grad_f(x[i]);
}
static float* g_lin_ws[2] = {nullptr, nullptr};     // SiLU(x) of the Linear being processed (one training context per process; a lane per stream: 1 = the side stream)
static size_t g_lin_ws_floats[2] = {0, 0};
static float* lin_workspace(size_t n, hipStream_t s) {
  const int lane = (g_train_side_stream && s == g_train_side_stream) ? 1 : 0;
  if (n > g_lin_ws_floats[lane]) {
    (void)hipStreamSynchronize(s);
    if (g_lin_ws[lane]) (void)hipFree(g_lin_ws[lane]);
    g_lin_ws[lane] = nullptr;
    g_lin_ws_floats[lane] = 0;
    const size_t want = std::max(n, (size_t)1 << 18);
    if (hipMalloc((void**)&g_lin_ws[lane], want * sizeof(float)) != hipSuccess) return nullptr;
    g_lin_ws_floats[lane] = want;
  }
  return g_lin_ws[lane];
}
// nn.Linear on [B, Cin] (the time-embedding MLPs; 24 of them per step at B x 1024 x <= 2048): the round-2 kernels read the weight once
// per item (forward), column-wise (dX: 179 us a call) or the activations once per output (dW); here they are the same three GEMM
// shapes as a k = 1 conv at L = 1 (N = B columns: a quarter of one tile, which the matrix pipe does not notice).
static bool lin_mm() { return !g_train_valu && !g_train_fp32_mfma; }
hipError_t launch_train_pw_forward(const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int L, int pre_silu, float* y,
                                   hipStream_t s) {
  if (L == 1 && lin_mm()) {
    const float* a = x;
    if (pre_silu) {
      float* ws = lin_workspace((size_t)B * Cin, s);
      if (!ws) return hipErrorOutOfMemory;
      hipLaunchKernelGGL(silu_map_kernel, dim3((B * Cin + 255) / 256), dim3(256), 0, s, x, (int64_t)B * Cin, ws);
      a = ws;
    }
    return launch_mm3_forward(a, w, bias, B, Cin, Cout, 1, 1, 1, 1, 0, y, s);
  }
  if (L == 1 && !g_train_valu) hipLaunchKernelGGL(lin_forward_kernel, dim3(Cout, B), dim3(256), 0, s, x, w, bias, Cin, Cout, pre_silu, y);
  else if (!pre_silu && convmm_ok(L, L)) convmm_forward(x, w, bias, B, Cin, Cout, L, L, 1, 1, 0, y, s);
  else hipLaunchKernelGGL(pw_forward_kernel, dim3((L + 255) / 256, Cout, B), dim3(256), 0, s, x, w, bias, Cin, Cout, L, pre_silu, y);
  return hipGetLastError();
}
hipError_t launch_train_pw_backward(const float* dy, const float* x, const float* w, int B, int Cin, int Cout, int L, int pre_silu, float* dx,
                                    float* dw, float* db, hipStream_t s) {
  const bool mm = !pre_silu && convmm_ok(L, L);
  if (L == 1 && lin_mm()) {
    if (dx) {
      hipError_t e = launch_mm3_dx(dy, w, B, Cin, Cout, 1, 1, 1, 1, 0, dx, s);
      if (e != hipSuccess) return e;
      if (pre_silu) hipLaunchKernelGGL(silu_grad_mul_kernel, dim3((B * Cin + 255) / 256), dim3(256), 0, s, x, (int64_t)B * Cin, dx);
    }
    const float* a = x;
    if (pre_silu) {
      float* ws = lin_workspace((size_t)B * Cin, s);
      if (!ws) return hipErrorOutOfMemory;
      hipLaunchKernelGGL(silu_map_kernel, dim3((B * Cin + 255) / 256), dim3(256), 0, s, x, (int64_t)B * Cin, ws);
      a = ws;
    }
    return launch_mm3_dw(dy, a, B, Cin, Cout, 1, 1, 1, 1, 0, dw, s, db);
  }
  if (L == 1 && !g_train_valu) {
    if (dx) hipLaunchKernelGGL(lin_dx_kernel, dim3(Cin, B), dim3(256), 0, s, dy, x, w, Cin, Cout, pre_silu, dx);
    hipLaunchKernelGGL(lin_dw_kernel, dim3(Cout), dim3(256), 0, s, dy, x, B, Cin, Cout, pre_silu, dw);
    if (db) hipLaunchKernelGGL(bias_grad_kernel, dim3(Cout), dim3(256), 0, s, dy, B, Cout, L, db);
    return hipGetLastError();
  }
  if (dx) {
    if (mm) convmm_dx(dy, w, B, Cin, Cout, L, L, 1, 1, 0, dx, s);
    else hipLaunchKernelGGL(pw_dx_kernel, dim3((L + 255) / 256, Cin, B), dim3(256), 0, s, dy, x, w, Cin, Cout, L, pre_silu, dx);
  }
  hipStream_t sd = dw_side_fork(s);   // (as launch_train_conv_backward)
  hipStream_t sw = sd ? sd : s;
  bool db_done = false;
  if (mm) db_done = convmm_dw(dy, x, B, Cin, Cout, L, L, 1, 1, 0, dw, sw, db);
  else hipLaunchKernelGGL(pw_dw_kernel, dim3(Cin, Cout), dim3(256), 0, sw, dy, x, B, Cin, Cout, L, pre_silu, dw);
  if (db && !db_done) hipLaunchKernelGGL(bias_grad_kernel, dim3(Cout), dim3(256), 0, sw, dy, B, Cout, L, db);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// LinearAttention core (srcs/modules/unet.py:208-221, between to_qkv and to_out), forward and backward, [B, C, L] fp32:
//   qs = softmax_d(q) * scale,  ks = softmax_n(k),  ctx[d, e] = sum_n ks[d, n] v[e, n],  o[e, n] = sum_d ctx[d, e] qs[d, n]
// per (item, head); qkv = [q | k | v] with channel = h * D + d.  Workspace (floats): qs, ks [B, H*D, N] each and ctx [B, H, D, D]
// saved by the forward pass; the backward pass adds dctx [B, H, D, D] and a [B, H*D] row-dot buffer.  D <= 64.
//   dctx[d, e] = sum_n do[e, n] qs[d, n]          dqs[d, n] = sum_e ctx[d, e] do[e, n]
//   dks[d, n]  = sum_e dctx[d, e] v[e, n]         dv[e, n]  = sum_d ks[d, n] dctx[d, e]
//   dq = s (g - sum_d s g), s = qs / scale, g = dqs * scale        dk = ks (dks - sum_n ks dks)
// ---------------------------------------------------------------------------------------------
constexpr int kLaParts = 4;   // position ranges a (item, head)'s D x D product is split into (partials summed in order)
size_t train_linattn_ws_floats(int B, int H, int D, int N) {
  return (size_t)2 * B * H * D * N + (size_t)(2 + kLaParts) * B * H * D * D + (size_t)B * H * D + 64;
}
struct LaWs { float *qs, *ks, *ctx, *dctx, *rowdot, *part; };
static LaWs la_carve(float* ws, int B, int H, int D, int N) {
  LaWs w;
  float* p = ws;
  w.qs = p; p += (size_t)B * H * D * N;
  w.ks = p; p += (size_t)B * H * D * N;
  w.ctx = p; p += (size_t)B * H * D * D;
  w.dctx = p; p += (size_t)B * H * D * D;
  w.rowdot = p; p += (size_t)B * H * D + 64;
  w.part = p;
  return w;
}
// one block per (b, h, d): ks[d, :] = softmax over positions of k[d, :]
__global__ __launch_bounds__(256) void la_ksoftmax_kernel(const float* qkv, int H, int D, int N, float* ks) {
  __shared__ float red[4];
  const int d = blockIdx.x, h = blockIdx.y, b = blockIdx.z, HD = H * D;
  const float* kr = qkv + ((size_t)b * 3 * HD + HD + h * D + d) * N;
  float m = -INFINITY;
  for (int n = threadIdx.x; n < N; n += 256) m = fmaxf(m, kr[n]);
  m = block_max(m, red);
  float sum = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) sum += expf(kr[n] - m);
  sum = block_sum(sum, red);
  float* out = ks + ((size_t)b * HD + h * D + d) * N;
  for (int n = threadIdx.x; n < N; n += 256) out[n] = expf(kr[n] - m) / sum;
}
// one thread per (b, h, n): qs[:, n] = softmax over d of q[:, n], times scale
__global__ __launch_bounds__(256) void la_qsoftmax_kernel(const float* qkv, int H, int D, int N, float scale, float* qs) {
  const int n = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y, b = blockIdx.z, HD = H * D;
  if (n >= N) return;
  const float* qr = qkv + ((size_t)b * 3 * HD + h * D) * N + n;
  float m = -INFINITY;
  for (int d = 0; d < D; ++d) m = fmaxf(m, qr[(size_t)d * N]);
  float sum = 0.f;
  for (int d = 0; d < D; ++d) sum += expf(qr[(size_t)d * N] - m);
  float* out = qs + ((size_t)b * HD + h * D) * N + n;
  for (int d = 0; d < D; ++d) out[(size_t)d * N] = expf(qr[(size_t)d * N] - m) / sum * scale;
}
// out[d, e] = sum_n a[d, n] * bsrc[e, n]   (ctx from (ks, v); dctx from (qs, do)).  One workgroup per (item, head, position range):
// 64-position chunks of both operands staged through LDS with coalesced row reads, thread (e = tid % D, d = tid / D + (256 / D) j) owns
// D * D / 256 outputs; the ranges' partials go to `part` and la_outer_sum_kernel adds them in order.  (The first form gave every lane
// its own row of bsrc -- 64 cache lines per load instruction -- and re-read bsrc once per d: 112 us a call, 318 at N = 1200.)
__global__ __launch_bounds__(256) void la_outer_kernel(const float* a, const float* bsrc, size_t a_item, size_t b_item, int H, int D, int N,
                                                       float* part) {
  __shared__ float As[64][65], Bs[64][65];
  const int h = blockIdx.y, b = blockIdx.z, pr = blockIdx.x, np = gridDim.x;
  const int n_lo = (int)((long long)N * pr / np), n_hi = (int)((long long)N * (pr + 1) / np);
  const float* ab = a + (size_t)b * a_item + (size_t)(h * D) * N;
  const float* bb = bsrc + (size_t)b * b_item + (size_t)(h * D) * N;
  const int e = threadIdx.x % D, d0 = threadIdx.x / D, dstep = 256 / D;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  for (int n0 = n_lo; n0 < n_hi; n0 += 64) {
    const int len = min(64, n_hi - n0);
    for (int idx = threadIdx.x; idx < D * 64; idx += 256) {
      const int d = idx >> 6, nn = idx & 63;
      const bool ok = nn < len;
      As[d][nn] = ok ? ab[(size_t)d * N + n0 + nn] : 0.f;
      Bs[d][nn] = ok ? bb[(size_t)d * N + n0 + nn] : 0.f;
    }
    __syncthreads();
    for (int nn = 0; nn < 64; ++nn) {
      const float bv = Bs[e][nn];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int d = d0 + dstep * j;
        if (d < D) acc[j] = fmaf(As[d][nn], bv, acc[j]);
      }
    }
    __syncthreads();
  }
  float* o = part + (((size_t)pr * gridDim.z + b) * H + h) * D * D;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int d = d0 + dstep * j;
    if (d < D) o[d * D + e] = acc[j];
  }
}
__global__ __launch_bounds__(256) void la_outer_sum_kernel(const float* part, int np, size_t n, float* out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float t = 0.f;
    for (int p = 0; p < np; ++p) t += part[(size_t)p * n + i];
    out[i] = t;
  }
}
static void la_outer(const float* a, const float* bsrc, size_t a_item, size_t b_item, int B, int H, int D, int N, float* part, float* out,
                     hipStream_t s) {
  const int np = N >= 256 ? kLaParts : 1;
  hipLaunchKernelGGL(la_outer_kernel, dim3(np, H, B), dim3(256), 0, s, a, bsrc, a_item, b_item, H, D, N, np == 1 ? out : part);
  if (np > 1) {
    const size_t n = (size_t)B * H * D * D;
    hipLaunchKernelGGL(la_outer_sum_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, s, part, np, n, out);
  }
}
// one thread per (b, h, n): y[e, n] = sum_d m[d, e] * x[d, n]  (TRANS = false: o from (ctx, qs)) or y[d, n] = sum_e m[d, e] * x[e, n] (TRANS = true)
template <bool TRANS>
__global__ __launch_bounds__(256) void la_apply_kernel(const float* m, const float* x, size_t x_item, int H, int D, int N, float* y, size_t y_item) {
  extern __shared__ float sm[];   // [D][D]
  const int h = blockIdx.y, b = blockIdx.z;
  for (int i = threadIdx.x; i < D * D; i += 256) sm[i] = m[((size_t)b * H + h) * D * D + i];
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float* xr = x + (size_t)b * x_item + (size_t)(h * D) * N + n;
  float* yr = y + (size_t)b * y_item + (size_t)(h * D) * N + n;
  for (int o = 0; o < D; ++o) {
    float acc = 0.f;
    for (int i = 0; i < D; ++i) acc = fmaf(TRANS ? sm[o * D + i] : sm[i * D + o], xr[(size_t)i * N], acc);
    yr[(size_t)o * N] = acc;
  }
}
// dq from dqs (in place in `dq`, which holds dqs on entry): softmax over d backward; one thread per (b, h, n)
__global__ __launch_bounds__(256) void la_dq_kernel(const float* qs, int H, int D, int N, float scale, float* dq, size_t dq_item) {
  const int n = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y, b = blockIdx.z, HD = H * D;
  if (n >= N) return;
  const float* sr = qs + ((size_t)b * HD + h * D) * N + n;
  float* gr = dq + (size_t)b * dq_item + (size_t)(h * D) * N + n;
  const float inv = 1.0f / scale;
  float dot = 0.f;
  for (int d = 0; d < D; ++d) dot = fmaf(sr[(size_t)d * N] * inv, gr[(size_t)d * N] * scale, dot);
  for (int d = 0; d < D; ++d) { const float sv = sr[(size_t)d * N] * inv; gr[(size_t)d * N] = sv * (gr[(size_t)d * N] * scale - dot); }
}
// dk from dks (in place): softmax over n backward; one block per (b, h, d)
__global__ __launch_bounds__(256) void la_dk_kernel(const float* ks, int H, int D, int N, float* dk, size_t dk_item) {
  __shared__ float red[4];
  const int d = blockIdx.x, h = blockIdx.y, b = blockIdx.z, HD = H * D;
  const float* kr = ks + ((size_t)b * HD + h * D + d) * N;
  float* gr = dk + (size_t)b * dk_item + (size_t)(h * D + d) * N;
  float dot = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) dot = fmaf(kr[n], gr[n], dot);
  dot = block_sum(dot, red);
  for (int n = threadIdx.x; n < N; n += 256) gr[n] = kr[n] * (gr[n] - dot);
}
hipError_t launch_train_linattn_forward(const float* qkv, int B, int H, int D, int N, float* o, float* ws, hipStream_t s) {
  if (D < 1 || D > 64 || 256 % D) return hipErrorInvalidValue;
  const LaWs w = la_carve(ws, B, H, D, N);
  const size_t HD = (size_t)H * D;
  const float scale = 1.0f / sqrtf((float)D);
  hipLaunchKernelGGL(la_ksoftmax_kernel, dim3(D, H, B), dim3(256), 0, s, qkv, H, D, N, w.ks);
  hipLaunchKernelGGL(la_qsoftmax_kernel, dim3((N + 255) / 256, H, B), dim3(256), 0, s, qkv, H, D, N, scale, w.qs);
  la_outer(w.ks, qkv + 2 * HD * N, HD * N, 3 * HD * N, B, H, D, N, w.part, w.ctx, s);
  hipLaunchKernelGGL(la_apply_kernel<false>, dim3((N + 255) / 256, H, B), dim3(256), (size_t)D * D * 4, s, w.ctx, w.qs, HD * N, H, D, N, o, HD * N);
  return hipGetLastError();
}
// dqkv [B, 3*H*D, N] = [dq | dk | dv]
hipError_t launch_train_linattn_backward(const float* d_o, const float* qkv, int B, int H, int D, int N, float* ws, float* dqkv, hipStream_t s) {
  if (D < 1 || D > 64 || 256 % D) return hipErrorInvalidValue;
  const LaWs w = la_carve(ws, B, H, D, N);
  const size_t HD = (size_t)H * D;
  const float scale = 1.0f / sqrtf((float)D);
  const dim3 gn((N + 255) / 256, H, B), gd(D, H, B);
  la_outer(w.qs, d_o, HD * N, HD * N, B, H, D, N, w.part, w.dctx, s);                                                       // dctx[d, e]
  hipLaunchKernelGGL(la_apply_kernel<true>, gn, dim3(256), (size_t)D * D * 4, s, w.ctx, d_o, HD * N, H, D, N, dqkv, 3 * HD * N);   // dqs -> dq slot
  hipLaunchKernelGGL(la_dq_kernel, gn, dim3(256), 0, s, w.qs, H, D, N, scale, dqkv, 3 * HD * N);
  hipLaunchKernelGGL(la_apply_kernel<true>, gn, dim3(256), (size_t)D * D * 4, s, w.dctx, qkv + 2 * HD * N, 3 * HD * N, H, D, N, dqkv + HD * N,
                     3 * HD * N);                                                                                          // dks -> dk slot
  hipLaunchKernelGGL(la_dk_kernel, gd, dim3(256), 0, s, w.ks, H, D, N, dqkv + HD * N, 3 * HD * N);
  hipLaunchKernelGGL(la_apply_kernel<false>, gn, dim3(256), (size_t)D * D * 4, s, w.dctx, w.ks, HD * N, H, D, N, dqkv + 2 * HD * N, 3 * HD * N);   // dv
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The rest of Unet1D (srcs/modules/unet.py:248-470) for the assembled forward / backward:
//   plain Conv1d with any kernel size / stride / zero padding (init_conv k7 p3, Downsample k4 s2 p1, the k3 p1 convs of
//   Upsample and of the last levels, final_conv k1), nearest x2 upsampling, tanh / GELU, and the bottleneck softmax Attention core.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void convg_forward_kernel(const float* x, const float* w, const float* bias, int Cin, int Cout, int Lin, int Lout,
                                                            int K, int S, int P, float* y) {
  const int b = blockIdx.z, o = blockIdx.y;
  for (int l = blockIdx.x * 256 + threadIdx.x; l < Lout; l += gridDim.x * 256) {
    float acc = bias ? bias[o] : 0.f;
    for (int i = 0; i < Cin; ++i) {
      const float* xr = x + ((size_t)b * Cin + i) * Lin;
      const float* wr = w + ((size_t)o * Cin + i) * K;
      for (int t = 0; t < K; ++t) {
        const int m = l * S + t - P;
        if (m >= 0 && m < Lin) acc = fmaf(wr[t], xr[m], acc);
      }
    }
    y[((size_t)b * Cout + o) * Lout + l] = acc;
  }
}
// dx[b,i,m] = sum_{o,t : (m + P - t) % S == 0} w[o,i,t] * dy[b,o,(m + P - t) / S]
__global__ __launch_bounds__(256) void convg_dx_kernel(const float* dy, const float* w, int Cin, int Cout, int Lin, int Lout, int K, int S, int P,
                                                       float* dx) {
  const int b = blockIdx.z, i = blockIdx.y;
  for (int m = blockIdx.x * 256 + threadIdx.x; m < Lin; m += gridDim.x * 256) {
    float acc = 0.f;
    for (int t = 0; t < K; ++t) {
      const int u = m + P - t;
      if (u < 0 || u % S) continue;
      const int l = u / S;
      if (l >= Lout) continue;
      for (int o = 0; o < Cout; ++o) acc = fmaf(w[((size_t)o * Cin + i) * K + t], dy[((size_t)b * Cout + o) * Lout + l], acc);
    }
    dx[((size_t)b * Cin + i) * Lin + m] = acc;
  }
}
// dw[o,i,t] = sum_{b,l} dy[b,o,l] * x[b,i,l*S + t - P]; one block per (o, i, t), fixed-order reduction
__global__ __launch_bounds__(256) void convg_dw_kernel(const float* dy, const float* x, int B, int Cin, int Cout, int Lin, int Lout, int K, int S,
                                                       int P, float* dw) {
  __shared__ float red[4];
  const int i = blockIdx.x, o = blockIdx.y, t = blockIdx.z;
  float a = 0.f;
  for (int idx = threadIdx.x; idx < B * Lout; idx += 256) {
    const int b = idx / Lout, l = idx - b * Lout, m = l * S + t - P;
    if (m >= 0 && m < Lin) a = fmaf(dy[((size_t)b * Cout + o) * Lout + l], x[((size_t)b * Cin + i) * Lin + m], a);
  }
  const float tot = block_sum(a, red);
  if (threadIdx.x == 0) dw[((size_t)o * Cin + i) * K + t] = tot;
}
hipError_t launch_train_conv_forward(const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int Lin, int K, int S, int P,
                                     float* y, hipStream_t s) {
  const int Lout = (Lin + 2 * P - K) / S + 1;
  if (Lout < 1) return hipErrorInvalidValue;
  if (convmm_ok(Lin, Lout)) convmm_forward(x, w, bias, B, Cin, Cout, Lin, Lout, K, S, P, y, s);
  else hipLaunchKernelGGL(convg_forward_kernel, dim3((Lout + 255) / 256, Cout, B), dim3(256), 0, s, x, w, bias, Cin, Cout, Lin, Lout, K, S, P, y);
  return hipGetLastError();
}
hipError_t launch_train_conv_backward(const float* dy, const float* x, const float* w, int B, int Cin, int Cout, int Lin, int K, int S, int P,
                                      float* dx, float* dw, float* db, hipStream_t s) {
  const int Lout = (Lin + 2 * P - K) / S + 1;
  if (Lout < 1) return hipErrorInvalidValue;
  const bool mm = convmm_ok(Lin, Lout);
  if (dx) {
    if (mm) convmm_dx(dy, w, B, Cin, Cout, Lin, Lout, K, S, P, dx, s);
    else hipLaunchKernelGGL(convg_dx_kernel, dim3((Lin + 255) / 256, Cin, B), dim3(256), 0, s, dy, w, Cin, Cout, Lin, Lout, K, S, P, dx);
  }
  // parameter gradients: on the side stream when the trainer has switched that on (the caller then keeps dy alive -- record_stream -- and x
  // is the layer's saved input)
  hipStream_t sd = dw_side_fork(s);
  hipStream_t sw = sd ? sd : s;
  bool db_done = false;
  if (mm) db_done = convmm_dw(dy, x, B, Cin, Cout, Lin, Lout, K, S, P, dw, sw, db);
  else hipLaunchKernelGGL(convg_dw_kernel, dim3(Cin, Cout, K), dim3(256), 0, sw, dy, x, B, Cin, Cout, Lin, Lout, K, S, P, dw);
  if (db && !db_done) hipLaunchKernelGGL(bias_grad_kernel, dim3(Cout), dim3(256), 0, sw, dy, B, Cout, Lout, db);
  return hipGetLastError();
}

// nearest x2 upsampling along L (nn.Upsample(scale_factor = 2, mode = 'nearest')): y[.., 2l] = y[.., 2l+1] = x[.., l]; backward sums the pair
__global__ __launch_bounds__(256) void up2_kernel(const float* in, size_t rows, int L, int backward, float* out) {
  const size_t n = rows * (size_t)L;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
    const size_t r = idx / L;
    const int l = (int)(idx - r * L);
    if (backward) out[idx] = in[r * 2 * L + 2 * l] + in[r * 2 * L + 2 * l + 1];
    else { out[r * 2 * L + 2 * l] = in[idx]; out[r * 2 * L + 2 * l + 1] = in[idx]; }
  }
}
hipError_t launch_train_upsample2(const float* in, int64_t rows, int L, int backward, float* out, hipStream_t s) {
  const size_t n = (size_t)rows * L;
  hipLaunchKernelGGL(up2_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, in, (size_t)rows, L, backward, out);
  return hipGetLastError();
}

// elementwise activations: kind 0 tanh, 1 GELU (exact, erf), 2 SiLU.  forward: y = f(x); backward: dx = dy * f'(x)
__device__ __forceinline__ float act_fwd(float v, int kind) {
  if (kind == 0) return tanhf(v);
  if (kind == 1) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return silu_f(v);
}
__device__ __forceinline__ float act_bwd(float v, int kind) {
  if (kind == 0) { const float t = tanhf(v); return 1.0f - t * t; }
  if (kind == 1) return 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * 0.3989422804014327f * expf(-0.5f * v * v);
  return silu_grad_f(v);
}
__global__ __launch_bounds__(256) void act_train_kernel(const float* x, const float* dy, int64_t n, int kind, float* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = dy ? dy[i] * act_bwd(x[i], kind) : act_fwd(x[i], kind);
}
hipError_t launch_train_act(const float* x, const float* dy, int64_t n, int kind, float* out, hipStream_t s) {
  if (kind < 0 || kind > 2) return hipErrorInvalidValue;
  hipLaunchKernelGGL(act_train_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, x, dy, n, kind, out);
  return hipGetLastError();
}

// bottleneck Attention core (unet.py:234-245): per (item, head), sim[i, j] = scale * sum_d q[d, i] k[d, j], attn = softmax_j(sim),
// out[d, i] = sum_j attn[i, j] v[d, j].  `attn` [B, H, N, N] is saved for the backward pass (N is the bottleneck length, <= ~1k).
//   dv[d, j] = sum_i attn[i, j] do[d, i];  dattn[i, j] = sum_d do[d, i] v[d, j];  dsim = attn (dattn - sum_j attn dattn)
//   dq[d, i] = scale * sum_j dsim[i, j] k[d, j];  dk[d, j] = scale * sum_i dsim[i, j] q[d, i]
__global__ __launch_bounds__(256) void attn_scores_kernel(const float* qkv, int H, int D, int N, float scale, float* attn) {
  __shared__ float red[4];
  const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, HD = H * D;
  const float* q = qkv + ((size_t)b * 3 * HD + h * D) * N;
  const float* k = q + (size_t)HD * N;
  float* row = attn + (((size_t)b * H + h) * N + i) * N;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < N; j += 256) {
    float a = 0.f;
    for (int d = 0; d < D; ++d) a = fmaf(q[(size_t)d * N + i] * scale, k[(size_t)d * N + j], a);
    row[j] = a;
    m = fmaxf(m, a);
  }
  m = block_max(m, red);
  float sum = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) { const float e = expf(row[j] - m); row[j] = e; sum += e; }
  sum = block_sum(sum, red);
  for (int j = threadIdx.x; j < N; j += 256) row[j] /= sum;
}
// out[d, i] = sum_j attn[i, j] v[d, j]: one block per (i, h, b), thread d-groups reduce over j
__global__ __launch_bounds__(256) void attn_out_kernel(const float* attn, const float* qkv, int H, int D, int N, float* out) {
  __shared__ float part[256];
  const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, HD = H * D;
  const float* v = qkv + ((size_t)b * 3 * HD + 2 * HD + h * D) * N;
  const float* row = attn + (((size_t)b * H + h) * N + i) * N;
  const int d = threadIdx.x % D, grp = threadIdx.x / D, ngrp = 256 / D;
  float a = 0.f;
  for (int j = grp; j < N; j += ngrp) a = fmaf(row[j], v[(size_t)d * N + j], a);
  part[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < D) {
    float t = 0.f;
    for (int g = 0; g < ngrp; ++g) t += part[g * D + threadIdx.x];
    out[((size_t)b * HD + h * D + threadIdx.x) * N + i] = t;
  }
}
// dsim (in place of attn's copy `ds`): one block per (i, h, b): dattn[i, j] = sum_d do[d, i] v[d, j]; dsim = attn (dattn - dot)
__global__ __launch_bounds__(256) void attn_dsim_kernel(const float* attn, const float* d_o, const float* qkv, int H, int D, int N, float* ds) {
  __shared__ float red[4];
  const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, HD = H * D;
  const float* v = qkv + ((size_t)b * 3 * HD + 2 * HD + h * D) * N;
  const float* dor = d_o + ((size_t)b * HD + h * D) * N + i;
  const float* arow = attn + (((size_t)b * H + h) * N + i) * N;
  float* drow = ds + (((size_t)b * H + h) * N + i) * N;
  float dot = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) {
    float a = 0.f;
    for (int d = 0; d < D; ++d) a = fmaf(dor[(size_t)d * N], v[(size_t)d * N + j], a);
    drow[j] = a;
    dot = fmaf(arow[j], a, dot);
  }
  dot = block_sum(dot, red);
  for (int j = threadIdx.x; j < N; j += 256) drow[j] = arow[j] * (drow[j] - dot);
}
// dq[d, i] = scale * sum_j ds[i, j] k[d, j]   (block per (i, h, b), like attn_out with k);  mode 0
// dk[d, j] = scale * sum_i ds[i, j] q[d, i],  dv[d, j] = sum_i attn[i, j] do[d, i]          (block per (j, h, b));  mode 1 / 2
__global__ __launch_bounds__(256) void attn_grad_kernel(const float* mat, const float* src, size_t src_item, int H, int D, int N, float mul, int mode,
                                                        float* dst, size_t dst_item) {
  __shared__ float part[256];
  const int p = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const float* m = mat + ((size_t)b * H + h) * N * N;
  const float* sr = src + (size_t)b * src_item + (size_t)(h * D) * N;
  const int d = threadIdx.x % D, grp = threadIdx.x / D, ngrp = 256 / D;
  float a = 0.f;
  for (int r = grp; r < N; r += ngrp) a = fmaf(mode == 0 ? m[(size_t)p * N + r] : m[(size_t)r * N + p], sr[(size_t)d * N + r], a);
  part[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < D) {
    float t = 0.f;
    for (int g = 0; g < ngrp; ++g) t += part[g * D + threadIdx.x];
    dst[(size_t)b * dst_item + (size_t)(h * D + threadIdx.x) * N + p] = t * mul;
  }
}
size_t train_attn_ws_floats(int B, int H, int N) { return (size_t)2 * B * H * N * N + 64; }
hipError_t launch_train_attn_forward(const float* qkv, int B, int H, int D, int N, float* out, float* ws, hipStream_t s) {
  if (D < 1 || D > 64 || 256 % D) return hipErrorInvalidValue;
  const float scale = 1.0f / sqrtf((float)D);
  hipLaunchKernelGGL(attn_scores_kernel, dim3(N, H, B), dim3(256), 0, s, qkv, H, D, N, scale, ws);
  hipLaunchKernelGGL(attn_out_kernel, dim3(N, H, B), dim3(256), 0, s, ws, qkv, H, D, N, out);
  return hipGetLastError();
}
hipError_t launch_train_attn_backward(const float* d_o, const float* qkv, int B, int H, int D, int N, float* ws, float* dqkv, hipStream_t s) {
  if (D < 1 || D > 64 || 256 % D) return hipErrorInvalidValue;
  const size_t HD = (size_t)H * D;
  const float scale = 1.0f / sqrtf((float)D);
  float* attn = ws;
  float* ds = ws + (size_t)B * H * N * N;
  const dim3 g(N, H, B);
  hipLaunchKernelGGL(attn_dsim_kernel, g, dim3(256), 0, s, attn, d_o, qkv, H, D, N, ds);
  hipLaunchKernelGGL(attn_grad_kernel, g, dim3(256), 0, s, ds, qkv + HD * N, 3 * HD * N, H, D, N, scale, 0, dqkv, 3 * HD * N);            // dq from k
  hipLaunchKernelGGL(attn_grad_kernel, g, dim3(256), 0, s, ds, qkv, 3 * HD * N, H, D, N, scale, 1, dqkv + HD * N, 3 * HD * N);            // dk from q
  hipLaunchKernelGGL(attn_grad_kernel, g, dim3(256), 0, s, attn, d_o, HD * N, H, D, N, 1.0f, 2, dqkv + 2 * HD * N, 3 * HD * N);          // dv from do
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Unet1D.process_cond (unet.py:407-420) for training: the condition upsampler SConvTranspose1d(C, C, kernel 2r, stride r,
// non-causal: conv.py:235-274 trims r - r/2 samples on the left and r/2 on the right) forward / backward, and the per-item
// max-abs scaling x / (max|x| + 1e-20) (unet.py:401-403) forward / backward.  ConvTranspose1d weight layout [Cin, Cout, 2r].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void convtr_forward_kernel(const float* x, const float* w, const float* bias, int Cin, int Cout, int L, int r,
                                                             float* y) {
  const int b = blockIdx.z, o = blockIdx.y, pl = r - r / 2, K = 2 * r, Lo = L * r;
  for (int mp = blockIdx.x * 256 + threadIdx.x; mp < Lo; mp += gridDim.x * 256) {
    const int m = mp + pl;
    float acc = bias ? bias[o] : 0.f;
    for (int t = m % r; t < K; t += r) {
      const int l = (m - t) / r;
      if (l < 0 || l >= L) continue;
      for (int i = 0; i < Cin; ++i) acc = fmaf(x[((size_t)b * Cin + i) * L + l], w[((size_t)i * Cout + o) * K + t], acc);
    }
    y[((size_t)b * Cout + o) * Lo + mp] = acc;
  }
}
__global__ __launch_bounds__(256) void convtr_dx_kernel(const float* dy, const float* w, int Cin, int Cout, int L, int r, float* dx) {
  const int b = blockIdx.z, i = blockIdx.y, pl = r - r / 2, K = 2 * r, Lo = L * r;
  for (int l = blockIdx.x * 256 + threadIdx.x; l < L; l += gridDim.x * 256) {
    float acc = 0.f;
    for (int t = 0; t < K; ++t) {
      const int mp = l * r + t - pl;
      if (mp < 0 || mp >= Lo) continue;
      for (int o = 0; o < Cout; ++o) acc = fmaf(dy[((size_t)b * Cout + o) * Lo + mp], w[((size_t)i * Cout + o) * K + t], acc);
    }
    dx[((size_t)b * Cin + i) * L + l] = acc;
  }
}
// dw[i,o,t] = sum_{b,l} x[b,i,l] * dy[b,o,l*r + t - pl]; one block per (o, i, t)
__global__ __launch_bounds__(256) void convtr_dw_kernel(const float* dy, const float* x, int B, int Cin, int Cout, int L, int r, float* dw) {
  __shared__ float red[4];
  const int o = blockIdx.x, i = blockIdx.y, t = blockIdx.z, pl = r - r / 2, K = 2 * r, Lo = L * r;
  float a = 0.f;
  for (int idx = threadIdx.x; idx < B * L; idx += 256) {
    const int b = idx / L, l = idx - b * L, mp = l * r + t - pl;
    if (mp >= 0 && mp < Lo) a = fmaf(x[((size_t)b * Cin + i) * L + l], dy[((size_t)b * Cout + o) * Lo + mp], a);
  }
  const float tot = block_sum(a, red);
  if (threadIdx.x == 0) dw[((size_t)i * Cout + o) * K + t] = tot;
}
hipError_t launch_train_convtr_forward(const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int L, int r, float* y,
                                       hipStream_t s) {
  // the transposed conv IS the dX GEMM of a stride-r conv with the same weight layout ([Cin, Cout, 2r] = [Cout', Cin', K]) and
  // padding r - r/2; its dX is that conv's forward, its dW that conv's dW with the operands swapped
  if (lin_mm()) return launch_mm3_dx(x, w, B, Cout, Cin, L * r, L, 2 * r, r, r - r / 2, y, s, bias);
  hipLaunchKernelGGL(convtr_forward_kernel, dim3((L * r + 255) / 256, Cout, B), dim3(256), 0, s, x, w, bias, Cin, Cout, L, r, y);
  return hipGetLastError();
}
hipError_t launch_train_convtr_backward(const float* dy, const float* x, const float* w, int B, int Cin, int Cout, int L, int r, float* dx,
                                        float* dw, float* db, hipStream_t s) {
  if (lin_mm()) {
    if (dx) {
      hipError_t e = launch_mm3_forward(dy, w, nullptr, B, Cout, Cin, L * r, L, 2 * r, r, r - r / 2, dx, s);
      if (e != hipSuccess) return e;
    }
    hipError_t e = launch_mm3_dw(x, dy, B, Cout, Cin, L * r, L, 2 * r, r, r - r / 2, dw, s);
    if (e != hipSuccess) return e;
    if (db) hipLaunchKernelGGL(bias_grad_kernel, dim3(Cout), dim3(256), 0, s, dy, B, Cout, L * r, db);
    return hipGetLastError();
  }
  if (dx) hipLaunchKernelGGL(convtr_dx_kernel, dim3((L + 255) / 256, Cin, B), dim3(256), 0, s, dy, w, Cin, Cout, L, r, dx);
  hipLaunchKernelGGL(convtr_dw_kernel, dim3(Cout, Cin, 2 * r), dim3(256), 0, s, dy, x, B, Cin, Cout, L, r, dw);
  if (db) hipLaunchKernelGGL(bias_grad_kernel, dim3(Cout), dim3(256), 0, s, dy, B, Cout, L * r, db);
  return hipGetLastError();
}
// one block per item.  forward (dy == nullptr): out = x / (max|x| + 1e-20).  backward: out = dy / (s + eps) - [j == argmax] * sign(x_j) *
// sum_k dy_k x_k / (s + eps)^2 (torch's max sends the gradient to the first maximal element)
// (1024 threads: one workgroup per item is all the parallelism this op has at B = 32, so each at least keeps 16 waves of loads in flight;
// the block reductions keep their fixed order)
__device__ __forceinline__ float block_max16(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int k = 1; k < 16; ++k) t = fmaxf(t, red[k]);
  return t;
}
__global__ __launch_bounds__(1024) void maxscale_kernel(const float* x, const float* dy, int64_t n, float* out) {
  __shared__ float red[16];
  __shared__ long long s_arg;
  const float* xb = x + (size_t)blockIdx.x * n;
  float m = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(xb[i]));
  m = block_max16(m, red);
  const float inv = 1.0f / (m + 1e-20f);
  float* ob = out + (size_t)blockIdx.x * n;
  if (!dy) {
    for (int64_t i = threadIdx.x; i < n; i += 1024) ob[i] = xb[i] * inv;
    return;
  }
  const float* db = dy + (size_t)blockIdx.x * n;
  float dot = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) dot = fmaf(db[i], xb[i], dot);
  dot = block_sum16(dot, red);
  if (threadIdx.x == 0) s_arg = (long long)n;
  __syncthreads();
  for (int64_t i = threadIdx.x; i < n; i += 1024)
    if (fabsf(xb[i]) == m) atomicMin(&s_arg, (long long)i);
  __syncthreads();
  const long long arg = s_arg;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    float g = db[i] * inv;
    if (i == arg) g -= (xb[i] < 0.f ? -1.0f : 1.0f) * dot * inv * inv;
    ob[i] = g;
  }
}
hipError_t launch_train_maxscale(const float* x, const float* dy, int B, int64_t n_per_item, float* out, hipStream_t s) {
  hipLaunchKernelGGL(maxscale_kernel, dim3(B), dim3(1024), 0, s, x, dy, n_per_item, out);
  return hipGetLastError();
}

}  // namespace ldc
