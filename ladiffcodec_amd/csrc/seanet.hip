// seanet.hip -- the two SEANet pieces that are not a channel GEMM: the Cin=1 input conv and the LSTM
// time recurrence.
//
//   conv_cin1 : SEANetEncoder.model[0] = SConv1d(1, n_filters, 7, causal, reflect)  (reference
//               srcs/modules/seanet.py:108-111, conv.py:217-232).  HBM-bound: reads T floats, writes T*Cout.
//   lstm      : SLSTM.forward (lstm.py:22-28) = nn.LSTM (gate order i,f,g,o) + skip.  The input
//               projection for all T runs as one GEMM on the conv kernel; what is left is the strictly
//               sequential h_{t-1} -> h_t chain, latency-bound: one workgroup per utterance keeps W_hh in
//               registers (H <= 128: one gate row per thread) or streams a k-major copy from L2 (H = 512).
#include <type_traits>
#include "ldc_kernels.h"
#include "ldc_math.h"

namespace ldc {

__device__ __forceinline__ float sbf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short sf2bf(float f) { return hw_bf16(f); }
template <typename T>
__device__ __forceinline__ float sld(const void* p, size_t i);
template <>
__device__ __forceinline__ float sld<float>(const void* p, size_t i) { return reinterpret_cast<const float*>(p)[i]; }
template <>
__device__ __forceinline__ float sld<__bf16>(const void* p, size_t i) {
  return sbf2f(reinterpret_cast<const unsigned short*>(p)[i]);
}
template <typename T>
__device__ __forceinline__ void sst(void* p, size_t i, float v);
template <>
__device__ __forceinline__ void sst<float>(void* p, size_t i, float v) { reinterpret_cast<float*>(p)[i] = v; }
template <>
__device__ __forceinline__ void sst<__bf16>(void* p, size_t i, float v) {
  reinterpret_cast<unsigned short*>(p)[i] = sf2bf(v);
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_cin1_kernel(const float* x, void* y, const float* w, const float* bias,
                                                        int L, int Cout, int k) {
  extern __shared__ float sw[];   // [Cout][k] + [Cout]
  for (int i = threadIdx.x; i < Cout * k; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < Cout; i += 256) sw[Cout * k + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int b = blockIdx.y;
  const int pad = k - 1;
  const size_t total = (size_t)L * Cout;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int l = (int)(idx / Cout), co = (int)(idx % Cout);
    float acc = sw[Cout * k + co];
    for (int t = 0; t < k; ++t) {
      int u = l + t - pad;
      if (u < 0) u = -u;            // causal reflect (left only; L > pad)
      if (u >= L) u = 2 * (L - 1) - u;
      acc += sw[co * k + t] * x[(size_t)b * L + u];
    }
    sst<T>(y, ((size_t)b * L + l) * Cout + co, acc);
  }
}

// Cout/4 divides 256 (n_filters = 32): thread -> 4 fixed output channels (one 16-byte store per row), rows advance
// by 1024/Cout per trip; the taps of the thread's channels live in registers (k <= 8).  The kernel writes
// T*Cout*4 bytes per item and reads T*4: store width is what matters.
template <typename T>
__global__ __launch_bounds__(256) void conv_cin1_rows_kernel(const float* x, void* y, const float* w, const float* bias,
                                                             int L, int Cout, int k, int rows_per_block) {
  const int b = blockIdx.y;
  const int tpr = Cout / 4;   // threads per row
  const int c4 = (threadIdx.x % tpr) * 4, r0 = threadIdx.x / tpr, rstep = 256 / tpr;
  float wk[4][8], bv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    bv[c] = bias ? bias[c4 + c] : 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) wk[c][t] = t < k ? w[(c4 + c) * k + t] : 0.f;
  }
  const int pad = k - 1;
  const int lbeg = blockIdx.x * rows_per_block, lend = min(L, lbeg + rows_per_block);
  const float* xb = x + (size_t)b * L;
  for (int l = lbeg + r0; l < lend; l += rstep) {
    float xv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      int u = l + t - pad;
      if (u < 0) u = -u;            // causal reflect (left only; L > pad)
      xv[t] = t < k ? xb[u] : 0.f;
    }
    float acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[c] = bv[c];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[c] = fmaf(wk[c][t], xv[t], acc[c]);
    }
    const size_t o = ((size_t)b * L + l) * Cout + c4;
    if (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(y) + o) = make_uint2(hw_bf16x2(acc[0], acc[1]), hw_bf16x2(acc[2], acc[3]));
    }
  }
}

hipError_t launch_conv_cin1(int dt, const float* x, void* y, const float* w, const float* bias, int B, int L, int Cout,
                            int k, hipStream_t s) {
  if (L <= k - 1) return hipErrorInvalidValue;
  if (Cout % 4 == 0 && Cout <= 1024 && 256 % (Cout / 4) == 0 && k <= 8) {
    const int rpb = 8 * (1024 / Cout);
    dim3 grid((L + rpb - 1) / rpb, B);
    if (dt == DT_F32) hipLaunchKernelGGL(conv_cin1_rows_kernel<float>, grid, dim3(256), 0, s, x, y, w, bias, L, Cout, k, rpb);
    else hipLaunchKernelGGL(conv_cin1_rows_kernel<__bf16>, grid, dim3(256), 0, s, x, y, w, bias, L, Cout, k, rpb);
    return hipGetLastError();
  }
  const size_t lds = (size_t)(Cout * k + Cout) * sizeof(float);
  int bx = (int)std::min<size_t>(((size_t)L * Cout + 255) / 256, 512);
  if (dt == DT_F32)
    hipLaunchKernelGGL(conv_cin1_kernel<float>, dim3(bx, B), dim3(256), lds, s, x, y, w, bias, L, Cout, k);
  else
    hipLaunchKernelGGL(conv_cin1_kernel<__bf16>, dim3(bx, B), dim3(256), lds, s, x, y, w, bias, L, Cout, k);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_acc(float v) { return 1.0f / (1.0f + expf(-v)); }

// 4H threads, W_hh in registers, h in LDS.  Thread (row group rg, k segment g) holds four consecutive gate rows x H/4 weights: it reads
// its QUARTER of h once for four rows and the four segments meet in a two-step DPP reduction.  (Round 1-4: one thread per gate row, every
// thread reading all of h -- 4H x H x 4 bytes of LDS reads per step: half of the step's 2 050 cycles at H = 128.)
// q (LstmSeq): the launch covers time steps [t0, t1) of a longer sequence -- the recurrent state (h | c per item) comes from / goes to
// q.state when the chunk is not the first / always -- and addresses rows as item * bs + t * ts, so that a layer can write its output
// time-major for the next layer's chunked input GEMM (run_seanet: the two layers of the decoder's LSTM as a two-stage pipeline).
template <typename T, int H>
__global__ __launch_bounds__(4 * H) void lstm_reg_kernel(const void* pre, const float* w_hh, void* out, const void* skip,
                                                         const LstmSeq q) {
  constexpr int KS = H / 4;          // k range of a thread
  constexpr int KP = KS + 4;         // LDS pitch of a segment: the four segments a quad reads at once sit in different banks
  __shared__ __attribute__((aligned(16))) float sh[4 * KP];
  __shared__ float sg[4 * H];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int rg = tid >> 2, g = tid & 3;
  float w[4][KS];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int k = 0; k < KS; k += 4) {
      const float4 v = *reinterpret_cast<const float4*>(w_hh + (size_t)(rg * 4 + q) * H + g * KS + k);
      w[q][k] = v.x; w[q][k + 1] = v.y; w[q][k + 2] = v.z; w[q][k + 3] = v.w;
    }
  const int row = tid;               // gate phase: one thread per hidden unit (tid < H); it fetches the unit's four pre-activations a step ahead
  const bool resume = q.state && q.t0 > 0;
  float c = (resume && row < H) ? q.state[(size_t)b * 2 * H + H + row] : 0.f;
  float h_last = (resume && row < H) ? q.state[(size_t)b * 2 * H + row] : 0.f;
  if (row < H) sh[row + 4 * (row / KS)] = h_last;
  float p_next[4] = {0.f, 0.f, 0.f, 0.f};
  const size_t pre0 = (size_t)b * q.pre_bs, out0 = (size_t)b * q.out_bs, skip0 = (size_t)b * q.skip_bs;
  if (row < H) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) p_next[gq] = sld<T>(pre, (pre0 + (size_t)q.t0 * q.pre_ts) * (4 * H) + gq * H + row);
  }
  // the skip input of the coming step, fetched a step ahead like the gate pre-activation (loaded inside the step its latency sat on the
  // recurrence's critical path: the wave waited for it before the h + skip store, in front of the barrier)
  float sk_next = (skip && row < H) ? sld<T>(skip, (skip0 + (size_t)q.t0 * q.skip_ts) * H + row) : 0.f;
  __syncthreads();
  for (int t = q.t0; t < q.t1; ++t) {
    const float p_cur[4] = {p_next[0], p_next[1], p_next[2], p_next[3]};
    const float sk = sk_next;
    if (t + 1 < q.t1 && row < H) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) p_next[gq] = sld<T>(pre, (pre0 + (size_t)(t + 1) * q.pre_ts) * (4 * H) + gq * H + row);
      if (skip) sk_next = sld<T>(skip, (skip0 + (size_t)(t + 1) * q.skip_ts) * H + row);
    }
    // packed fp32 FMAs (v_pk_fma_f32): the mat-vec is VALU-issue-bound, one instruction per two products
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 a01[4], a23[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a01[q] = f32x2{0.f, 0.f}; a23[q] = f32x2{0.f, 0.f}; }
#pragma unroll
    for (int k = 0; k < KS; k += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(&sh[g * KP + k]);
      const f32x2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2 w01 = {w[q][k], w[q][k + 1]}, w23 = {w[q][k + 2], w[q][k + 3]};
        a01[q] = __builtin_elementwise_fma(w01, h01, a01[q]);
        a23[q] = __builtin_elementwise_fma(w23, h23, a23[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v = (a01[q][0] + a01[q][1]) + (a23[q][0] + a23[q][1]);
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1, 0, 3, 2]
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2, 3, 0, 1]
      if (g == q) sg[rg * 4 + q] = v;       // (every lane of the quad holds the sum: lane q stores row q)
    }
    __syncthreads();
    if (row < H) {
      // v_exp_f32 / v_rcp_f32 forms (absolute error < 2e-7, ldc_math.h): the gate arithmetic of H threads is the serial part of a
      // step, libm's expf / tanhf were ~40 % of it.  (This kernel serves the H <= 128 LSTMs of the main codec, which feed no RVQ:
      // the cond codec's H = 512 kernels keep the libm forms so that no code index can move.)
      const float ig = fast_sigmoid(sg[row] + p_cur[0]), fg = fast_sigmoid(sg[H + row] + p_cur[1]);
      const float gg = fast_tanh(sg[2 * H + row] + p_cur[2]), og = fast_sigmoid(sg[3 * H + row] + p_cur[3]);
      c = fg * c + ig * gg;
      const float h = og * fast_tanh(c);
      h_last = h;
      sh[row + 4 * (row / KS)] = h;
      const size_t o = (out0 + (size_t)t * q.out_ts) * H + row;
      sst<T>(out, o, h + sk);
    }
    __syncthreads();
  }
  if (q.state && row < H) {
    q.state[(size_t)b * 2 * H + row] = h_last;
    q.state[(size_t)b * 2 * H + H + row] = c;
  }
}

// Generic H: 1024 threads, thread owns gate rows {tid, tid+1024, ...}; W_hh pre-transposed k-major
// [H/4][4H][4] so that a wavefront reads 1 KiB contiguous per instruction from L2.
template <typename T>
__global__ __launch_bounds__(1024) void lstm_stream_kernel(const void* pre, const float* w_km, void* out, const void* skip,
                                                           int T_len, int H) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sh = sm;            // [H]
  float* sg = sm + H;        // [4H]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int G = 4 * H;
  for (int i = tid; i < H; i += 1024) sh[i] = 0.f;
  // cell state lives with the threads that own rows < H
  float c_state[4] = {0.f, 0.f, 0.f, 0.f};   // up to H = 4096
  __syncthreads();
  for (int t = 0; t < T_len; ++t) {
    for (int row = tid; row < G; row += 1024) {
      float acc0 = 0.f, acc1 = 0.f;
      const float4* wp = reinterpret_cast<const float4*>(w_km) + row;
      for (int k4 = 0; k4 < H / 4; k4 += 2) {
        const float4 w0 = wp[(size_t)k4 * G];
        const float4 w1 = wp[(size_t)(k4 + 1) * G];
        const float4 h0 = *reinterpret_cast<const float4*>(&sh[4 * k4]);
        const float4 h1 = *reinterpret_cast<const float4*>(&sh[4 * k4 + 4]);
        acc0 += w0.x * h0.x + w0.y * h0.y + w0.z * h0.z + w0.w * h0.w;
        acc1 += w1.x * h1.x + w1.y * h1.y + w1.z * h1.z + w1.w * h1.w;
      }
      sg[row] = acc0 + acc1 + sld<T>(pre, ((size_t)b * T_len + t) * G + row);
    }
    __syncthreads();
    int ci = 0;
    for (int j = tid; j < H; j += 1024, ++ci) {
      const float ig = sigmoid_acc(sg[j]), fg = sigmoid_acc(sg[H + j]);
      const float gg = tanhf(sg[2 * H + j]), og = sigmoid_acc(sg[3 * H + j]);
      const float c = fg * c_state[ci] + ig * gg;
      c_state[ci] = c;
      const float h = og * tanhf(c);
      sh[j] = h;
      const size_t o = ((size_t)b * T_len + t) * H + j;
      sst<T>(out, o, skip ? h + sld<T>(skip, o) : h);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Weight-stationary cooperative recurrence for wide layers (H = 256 / 512, the cond codec's SLSTM).
// H/4 workgroups; workgroup j owns hidden units [4j, 4j+4) = 16 gate rows whose W_hh slices stay in
// registers as the B operand of the exact-fp32 MFMA (16x16x4), K = H split over the 4 waves.  Per step:
// every workgroup reads h_{t-1} of all (<= 32) items (the MFMA M dimension) from a double-buffered exchange
// buffer, reduces the 4 waves' partials through LDS, applies the gates for its 4 units, publishes its slice
// of h_t with write-through (sc1) stores and arrives on one of 8 counters.  Visibility follows the
// placement-independent protocol: drained agent-scope stores -> relaxed arrive; relaxed poll -> agent acquire.
// Every spin is bounded; a timeout poisons the output with NaN instead of hanging.
// ---------------------------------------------------------------------------------------------
constexpr int LSTM_COOP_SHARDS = 8;
constexpr int LSTM_COOP_SHARD_STRIDE = 16;   // unsigneds: one 64-byte line per counter
size_t lstm_coop_ws_bytes(int H) { return (size_t)2 * 32 * H * sizeof(float) + 256 * sizeof(unsigned); }

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <typename T, int H>
__global__ __launch_bounds__(256) void lstm_coop_kernel(const void* pre, const float* w_hh, void* out, const void* skip,
                                                        int B, int T_len, float* hbuf, unsigned* sync, unsigned* host_flag) {
  constexpr int KW = H / 4;      // k range of one wave
  constexpr int NI = KW / 16;    // float4 k-groups per lane
  __shared__ float part[4][2][16][16];
  __shared__ int s_dead;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = blockIdx.x, nb = gridDim.x;
  const int n = lane & 15, q = lane >> 4;
  const int MT = (B + 15) / 16;
  // B operand: lane (n, q) holds W[row(n)][w*KW + 16*jj + 4*q + e]; the k order inside a wave is permuted
  // identically for A and B, which the dot product does not see
  f32x4 breg[NI];
  {
    const size_t row = (size_t)(n >> 2) * H + 4 * j + (n & 3);
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) breg[jj] = *reinterpret_cast<const f32x4*>(w_hh + row * H + w * KW + 16 * jj + 4 * q);
  }
  if (tid == 0) s_dead = 0;
  // gate owner threads: (item b, unit u)
  const int ob = tid >> 2, ou = tid & 3;
  const bool owner = tid < 4 * B;
  float c_state = 0.f;
  float pnext[4] = {0.f, 0.f, 0.f, 0.f};
  float sknext = 0.f;      // the skip input of the coming step, fetched a step ahead like the gate pre-activations: loaded inside the
                           // step (round 4) its HBM latency sat between the h stores and the arrival that releases the other workgroups
  if (owner) {
#pragma unroll
    for (int g = 0; g < 4; ++g) pnext[g] = sld<T>(pre, ((size_t)ob * T_len) * (4 * H) + g * H + 4 * j + ou);
    if (skip) sknext = sld<T>(skip, ((size_t)ob * T_len) * H + 4 * j + ou);
  }
  __syncthreads();
  bool dead = false;
  int t = 0;
  for (; t < T_len; ++t) {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (t > 0) {
      if (w == 0) {
        const unsigned target = (unsigned)(t * (nb / LSTM_COOP_SHARDS));   // arrivals per shard so far
        unsigned spins = 0;
        for (;;) {
          unsigned v = target;
          if (lane < LSTM_COOP_SHARDS) v = __hip_atomic_load(sync + lane * LSTM_COOP_SHARD_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned flag = lane == 0 ? __hip_atomic_load(sync + 200, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
          if (__any(flag != 0u) || ++spins > 400000u) {
            if (lane == 0) { __hip_atomic_store(sync + 200, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_dead = 1; }
            break;
          }
          if (__all(v >= target)) break;
          __builtin_amdgcn_s_sleep(2);
        }
      }
      __syncthreads();
      if (s_dead) { dead = true; break; }
      // agent-scope (sc0 sc1) 16-byte loads of the freshly published h: they go past the non-coherent L2 themselves, so
      // no acquire fence (= L2 invalidate) is needed; issued through asm, released by one vmcnt(0)
      const float* hp = hbuf + (size_t)((t + 1) & 1) * 32 * H;
      f32x4 areg[2][NI];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if (mt < MT) {
          const int item = min(mt * 16 + n, B - 1);
#pragma unroll
          for (int jj = 0; jj < NI; ++jj) {
            const float* src = hp + (size_t)item * H + w * KW + 16 * jj + 4 * q;
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(areg[mt][jj]) : "v"(src) : "memory");
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        if (mt < MT) {
#pragma unroll
          for (int jj = 0; jj < NI; ++jj) asm volatile("" : "+v"(areg[mt][jj]));
        }
#pragma unroll
      for (int jj = 0; jj < NI; ++jj) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[0][jj][e], breg[jj][e], acc[0], 0, 0, 0);
          if (MT > 1) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[1][jj][e], breg[jj][e], acc[1], 0, 0, 0);
        }
      }
    }
    // D[row = 4*q + r][col = n]: row = item within the tile, col = gate row
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[w][mt][4 * q + r][n] = acc[mt][r];
    __syncthreads();
    if (owner) {
      float gates[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = g * 4 + ou;
        gates[g] = pnext[g] + ((part[0][ob >> 4][ob & 15][col] + part[1][ob >> 4][ob & 15][col]) +
                               (part[2][ob >> 4][ob & 15][col] + part[3][ob >> 4][ob & 15][col]));
      }
      const float ig = sigmoid_acc(gates[0]), fg = sigmoid_acc(gates[1]);
      const float gg = tanhf(gates[2]), og = sigmoid_acc(gates[3]);
      c_state = fg * c_state + ig * gg;
      const float h = og * tanhf(c_state);
      __hip_atomic_store(hbuf + (size_t)(t & 1) * 32 * H + (size_t)ob * H + 4 * j + ou, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the h store has reached memory; nothing else of this thread is in flight here)
      const size_t o = ((size_t)ob * T_len + t) * H + 4 * j + ou;
      sst<T>(out, o, h + sknext);
      if (t + 1 < T_len) {
#pragma unroll
        for (int g = 0; g < 4; ++g) pnext[g] = sld<T>(pre, ((size_t)ob * T_len + t + 1) * (4 * H) + g * H + 4 * j + ou);
        if (skip) sknext = sld<T>(skip, o + H);
      }
    }
    __syncthreads();
    if (tid == 0 && t + 1 < T_len)
      __hip_atomic_fetch_add(sync + (j % LSTM_COOP_SHARDS) * LSTM_COOP_SHARD_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (dead && owner) {
    for (int tt = t; tt < T_len; ++tt) sst<T>(out, ((size_t)ob * T_len + tt) * H + 4 * j + ou, __builtin_nanf(""));
  }
  // the host learns about the timeout through a mapped word it checks at the next API call / synchronisation
  if (dead && tid == 0 && host_flag) __hip_atomic_store(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------
// One or two items (configs[0]'s single clip, a file decoded alone by the CLI): the weight-stationary recurrence on SIXTEEN workgroups
// of 1024 threads, all on one XCD, one per CU (round 5).
//   * profiles/r05_xcd_team_probe.md: a plain store stays in the writing XCD's L2 and an L1-bypassing (sc1) load from another CU of
//     the SAME XCD reads it there -- flag latency 0.4 us against 0.8-1.0 us + memory-rate reads for the write-through / memory-side-
//     atomic protocol of lstm_coop_kernel, of which a step has three dependent legs.
//   * Workgroup j keeps the four gate rows of units [j H/16, (j + 1) H/16) in registers (4H/16 rows x H weights = 64 per thread at
//     H = 512); a step is H/16-wide dot products on the VALU (a 16-item MFMA tile would be 15/16 padding), a shuffle reduction, the
//     gates of H/16 units, H/16 plain stores of h and ONE flag word; sixteen participants instead of 128, each alone on its CU (the
//     128-workgroup XCD-local forms measured slower than the chip-wide kernel: four workgroups' pollers and gate arithmetic per CU).
//   * The launch has 8 x 16 workgroups; the ones on XCC 0 take a rank (one atomic per launch), everybody else leaves at once.  Team
//     membership comes from the hardware id, so the L2 the data sits in IS the L2 the readers ask; a team that does not fill ends in
//     the bounded spin (NaN output + host flag) and the context goes back to lstm_coop_kernel.
// ---------------------------------------------------------------------------------------------
// sum over groups of G consecutive lanes (G = 8, 16, 32 or 64), left in every lane of the group: DPP lane permutations on the VALU for the
// first steps (the generic __shfl_xor goes through the LDS crossbar: ds_bpermute)
template <int G>
__device__ __forceinline__ float dpp_sum_group(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>());    // quad_perm [1, 0, 3, 2]
  v += dpp(v, std::integral_constant<int, 0x4E>());    // quad_perm [2, 3, 0, 1]
  v += dpp(v, std::integral_constant<int, 0x141>());   // row_half_mirror: the other quad of the 8 lanes
  if (G >= 16) v += dpp(v, std::integral_constant<int, 0x140>());   // row_mirror: the other half of the 16 lanes
  if (G >= 32) v += __shfl_xor(v, 16);
  if (G >= 64) v += __shfl_xor(v, 32);
  return v;
}

template <typename T, int H>
__global__ __launch_bounds__(1024) void lstm_xcd_kernel(const void* pre, const float* w_hh, void* out, const void* skip,
                                                        int B, int T_len, float* hbuf, unsigned* sync, unsigned* host_flag, unsigned long long* dbg = nullptr) {
  constexpr int NB = 16;              // workgroups of the team
  constexpr int U = H / NB;           // hidden units of a workgroup
  constexpr int ROWS = 4 * U;         // gate rows of a workgroup
  constexpr int RPT = 2;              // gate rows per thread: a thread's h segment is read from LDS once for two rows.  (One row per thread:
                                      // every thread reads 256 bytes of h, 1 000 cycles of LDS bandwidth per step; four rows: a 32-lane reduction
                                      // whose last step leaves the DPP for the LDS crossbar.  Products of a step: 1 560 / this / 3 070 cycles.)
  constexpr int KSEG = 1024 / (ROWS / RPT);   // threads per row group (32 at H = 512, 64 at H = 256)
  constexpr int KS = H / KSEG;        // k range of one thread (16 / 4)
  constexpr int NV = KS / 4;
  constexpr int KPAD = KS >= 8 ? 4 : 0;      // LDS: segment g starts at g * (KS + KPAD) floats -- the segments a wave reads at once sit in different banks
  __shared__ float sgate[2][ROWS];
  __shared__ __attribute__((aligned(16))) float sh[2][H + KPAD * KSEG];   // (unpadded 256-byte segments: an 8-way bank conflict on every read, 9 200 cycles per step)
  __shared__ int s_dead, s_rank;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11)) & 7u;   // HW_REG_XCC_ID
  if (xcc != 0u) return;
  if (tid == 0) {
    s_rank = (int)__hip_atomic_fetch_add(sync + 202, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_dead = 0;
  }
  __syncthreads();
  const int j = s_rank;
  if (j >= NB) return;
  unsigned* flags = sync;             // [NB] words on one line: the step a workgroup has published (zeroed before the launch)
  const int rg = tid / KSEG, g = tid % KSEG;        // row group (rows rg * RPT ..), k segment
  f32x4 wreg[RPT][NV];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = rg * RPT + q;
    const size_t row = (size_t)(r / U) * H + (size_t)j * U + (r % U);
#pragma unroll
    for (int v = 0; v < NV; ++v) wreg[q][v] = *reinterpret_cast<const f32x4*>(w_hh + row * H + g * KS + 4 * v);
  }
  const int ob = tid / U, ou = tid % U;             // gate owner: (item, unit)
  const bool owner = tid < U * B;
  float c_state = 0.f;
  float pnext[4] = {0.f, 0.f, 0.f, 0.f};
  float sknext = 0.f;
  if (owner) {
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) pnext[gg] = sld<T>(pre, ((size_t)ob * T_len) * (4 * H) + gg * H + j * U + ou);
    if (skip) sknext = sld<T>(skip, ((size_t)ob * T_len) * H + j * U + ou);
  }
  bool dead = false;
  int t = 0;
  for (; t < T_len; ++t) {
    float dot[2][RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) dot[0][q] = dot[1][q] = 0.f;
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0;
    const bool rec = dbg && j == 3 && tid == 0 && t >= 16 && t < 48;
    if (rec) st0 = __builtin_amdgcn_s_memtime();
    if (t > 0) {
      if (w == 0) {
        const unsigned long long t0 = wall_clock64();
        for (unsigned spins = 0;; ++spins) {
          const bool ok = lane >= NB || __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)t;
          if (__all(ok)) break;
          if ((spins & 255u) == 255u) {
            const unsigned flag = lane == 0 ? __hip_atomic_load(sync + 200, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (__any(flag != 0u) || wall_clock64() - t0 > 20000000ull) {   // 0.2 s
              if (lane == 0) { __hip_atomic_store(sync + 200, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_dead = 1; }
              break;
            }
          }
        }
      }
      __syncthreads();
      if (rec) st1 = __builtin_amdgcn_s_memtime();
      if (s_dead) { dead = true; break; }
      // h_{t-1} of the (<= 2) items: ONE L1-bypassing load per 16 bytes and workgroup into LDS (every thread loading its own k segment
      // past the L1 was 128-fold redundant L2 traffic on sixteen hot lines: 8.4 us per step), then broadcast reads
      const float* hp = hbuf + (size_t)((t + 1) & 1) * 32 * H;
      if (tid < B * (H / 4)) {
        const int b = tid / (H / 4), v = tid % (H / 4);
        f32x4 hv;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(hv) : "v"(hp + (size_t)b * H + 4 * v) : "memory");
        *reinterpret_cast<f32x4*>(&sh[b][4 * v + KPAD * ((4 * v) / KS)]) = hv;
      }
      __syncthreads();
      if (rec) st2 = __builtin_amdgcn_s_memtime();
#pragma unroll
      for (int b = 0; b < 2; ++b)
        if (b < B) {
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            const f32x4 hv = *reinterpret_cast<const f32x4*>(&sh[b][g * (KS + KPAD) + 4 * v]);
#pragma unroll
            for (int q = 0; q < RPT; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) dot[b][q] = fmaf(hv[e], wreg[q][v][e], dot[b][q]);
          }
        }
#pragma unroll
      for (int b = 0; b < 2; ++b)
        if (b < B) {
#pragma unroll
          for (int q = 0; q < RPT; ++q) dot[b][q] = dpp_sum_group<KSEG>(dot[b][q]);
        }
    }
    if (g == 0) {
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        sgate[0][rg * RPT + q] = dot[0][q];
        sgate[1][rg * RPT + q] = dot[1][q];
      }
    }
    __syncthreads();
    if (rec) st3 = __builtin_amdgcn_s_memtime();
    if (owner) {
      float gates[4];
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) gates[gg] = pnext[gg] + sgate[ob][gg * U + ou];
      const float ig = sigmoid_acc(gates[0]), fg = sigmoid_acc(gates[1]);
      const float gv = tanhf(gates[2]), og = sigmoid_acc(gates[3]);
      c_state = fg * c_state + ig * gv;
      const float h = og * tanhf(c_state);
      // plain store: the line stays in this XCD's L2, where the team's sc1 loads find it
      asm volatile("global_store_dword %0, %1, off" ::"v"(hbuf + (size_t)(t & 1) * 32 * H + (size_t)ob * H + j * U + ou), "v"(h) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (only the h store is outstanding here: the flag may follow it at once)
      if (rec) st4 = __builtin_amdgcn_s_memtime();
      const size_t o = ((size_t)ob * T_len + t) * H + j * U + ou;
      sst<T>(out, o, h + sknext);
      if (t + 1 < T_len) {
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) pnext[gg] = sld<T>(pre, ((size_t)ob * T_len + t + 1) * (4 * H) + gg * H + j * U + ou);
        if (skip) sknext = sld<T>(skip, o + H);
      }
    }
    __syncthreads();
    if (tid == 0 && t + 1 < T_len) {
      asm volatile("global_store_dword %0, %1, off" ::"v"(flags + j), "v"((unsigned)(t + 1)) : "memory");
    }
    if (rec) { st5 = __builtin_amdgcn_s_memtime(); unsigned long long* o = dbg + (size_t)(t - 16) * 6; o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = st4; o[5] = st5; }
  }
  if (dead && owner) {
    for (int tt = t; tt < T_len; ++tt) sst<T>(out, ((size_t)ob * T_len + tt) * H + j * U + ou, __builtin_nanf(""));
  }
  if (dead && tid == 0 && host_flag) __hip_atomic_store(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename T>
static hipError_t lstm_coop_launch(const void* pre, const float* w_rm, void* out, const void* skip, int B, int T_len, int H,
                                   void* ws, unsigned* host_flag, int coop_launch, hipStream_t s) {
  float* hbuf = reinterpret_cast<float*>(ws);
  unsigned* sync = reinterpret_cast<unsigned*>(hbuf + (size_t)2 * 32 * H);
  const size_t esz = sizeof(T);
  for (int b0 = 0; b0 < B; b0 += 32) {
    const int nb = std::min(32, B - b0);
    hipError_t e = hipMemsetAsync(sync, 0, 256 * sizeof(unsigned), s);
    if (e != hipSuccess) return e;
    const char* p = reinterpret_cast<const char*>(pre) + (size_t)b0 * T_len * 4 * H * esz;
    char* o = reinterpret_cast<char*>(out) + (size_t)b0 * T_len * H * esz;
    const char* k = skip ? reinterpret_cast<const char*>(skip) + (size_t)b0 * T_len * H * esz : nullptr;
    // All H/4 workgroups must be resident together (the kernel's hand-rolled h exchange spins on them).  That is checked
    // ONCE per kernel against the occupancy query (below); the launch itself is a plain one.  hipLaunchCooperativeKernel
    // makes the same check per launch, but on ROCm 7.2 it goes through a device-wide cooperative queue: enqueued while an
    // earlier decode is still running it cost ~90 ms per decode (247 vs 157 ms with two decodes queued on a caller's
    // stream).  `coop_launch` (LDC_COOP_LAUNCH=1) restores it.  A grid that does not become resident at run time (another process holding
    // the CUs) still ends in the bounded spin's timeout: NaN output + the host-mapped failure flag.
    const void* pp = p; void* oo = o; const void* kk = k; int nbv = nb, tl = T_len;
    void* args[] = {&pp, &w_rm, &oo, &kk, &nbv, &tl, &hbuf, &sync, &host_flag};
    const void* fn = H == 512 ? reinterpret_cast<const void*>(lstm_coop_kernel<T, 512>) : reinterpret_cast<const void*>(lstm_coop_kernel<T, 256>);
    if (coop_launch == 2 && nb <= 2) {   // one or two items: XCD-local exchange, 8 x 16 workgroups of 1024 threads, the ones on XCC 0 form the team
      const void* fx = H == 512 ? reinterpret_cast<const void*>(lstm_xcd_kernel<T, 512>) : reinterpret_cast<const void*>(lstm_xcd_kernel<T, 256>);
      static unsigned long long* dbg = nullptr;
      static int dbg_n = 0;
      if (getenv("LDC_LSTM_STAMPS") && !dbg) (void)hipMalloc((void**)&dbg, 32 * 6 * 8);
      void* args2[] = {&pp, &w_rm, &oo, &kk, &nbv, &tl, &hbuf, &sync, &host_flag, &dbg};
      e = hipLaunchKernel(fx, dim3(8 * 16), dim3(1024), args2, 0, s);
      if (dbg && ++dbg_n == 20) {   // tuning aid: one launch's per-step phase stamps (shader cycles) of workgroup 3
        (void)hipStreamSynchronize(s);
        unsigned long long hb[32 * 6];
        (void)hipMemcpy(hb, dbg, sizeof(hb), hipMemcpyDeviceToHost);
        double acc[6] = {0};
        for (int i = 1; i < 32; ++i) {
          acc[0] += (double)(hb[i * 6 + 1] - hb[i * 6 + 0]); acc[1] += (double)(hb[i * 6 + 2] - hb[i * 6 + 1]); acc[2] += (double)(hb[i * 6 + 3] - hb[i * 6 + 2]);
          acc[3] += (double)(hb[i * 6 + 4] - hb[i * 6 + 3]); acc[4] += (double)(hb[i * 6 + 5] - hb[i * 6 + 4]); acc[5] += (double)(hb[i * 6 + 0] - hb[(i - 1) * 6 + 0]);
        }
        fprintf(stderr, "[lstm_xcd] cycles per step: poll %.0f | h load %.0f | dot+reduce %.0f | gates+h store %.0f | rest+flag %.0f | step %.0f\n", acc[0] / 31, acc[1] / 31,
                acc[2] / 31, acc[3] / 31, acc[4] / 31, acc[5] / 31);
      }
    } else if (coop_launch == 1) e = hipLaunchCooperativeKernel(fn, dim3(H / 4), dim3(256), args, 0, s);
    else e = hipLaunchKernel(fn, dim3(H / 4), dim3(256), args, 0, s);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

bool lstm_coop_eligible(int H) { return H == 256 || H == 512; }
// the XCD-local form needs its sixteen 1024-thread workgroups resident on one XCD (32 CUs): how many such teams fit on one XCD (0 = none).
// The caller asks for room for twice the teams it may have in flight (batch parts run their codec ends concurrently).
int lstm_xcd_resident(int H) {
  if (!lstm_coop_eligible(H)) return 0;
  const void* fn = H == 512 ? reinterpret_cast<const void*>(lstm_xcd_kernel<float, 512>) : reinterpret_cast<const void*>(lstm_xcd_kernel<float, 256>);
  int per_cu = 0, cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 1024, 0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  if (cus % 8 != 0 || per_cu < 1) return 0;
  return per_cu * (cus / 8) / 16;
}

// Can the H/4 workgroups of the cooperative kernel be resident together on the current device?  Asked once per context
// (ldc_create); the occupancy API can over-report by one block per CU for SGPR-heavy kernels (MI355X_MICROARCH.md), hence the margin.
bool lstm_coop_resident(int H) {
  if (!lstm_coop_eligible(H)) return false;
  const void* fn = H == 512 ? reinterpret_cast<const void*>(lstm_coop_kernel<float, 512>) : reinterpret_cast<const void*>(lstm_coop_kernel<float, 256>);
  int dev = 0, per_cu = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return (long long)std::max(0, per_cu - 1) * cus >= H / 4 || (long long)per_cu * cus >= 2 * (H / 4);
}

// w_rm: row-major [4H][H] fp32; ws: lstm_coop_ws_bytes(H) bytes of device scratch owned by the caller
hipError_t launch_lstm_coop(int dt, const void* pre, const float* w_rm, void* out, const void* skip, int B, int T, int H,
                            void* ws, unsigned* host_flag, int coop_launch, hipStream_t s) {
  if (!lstm_coop_eligible(H)) return hipErrorInvalidValue;
  return dt == DT_F32 ? lstm_coop_launch<float>(pre, w_rm, out, skip, B, T, H, ws, host_flag, coop_launch, s)
                      : lstm_coop_launch<__bf16>(pre, w_rm, out, skip, B, T, H, ws, host_flag, coop_launch, s);
}

// w_hh points at: [4H][H] row-major for the register variants (H = 64, 128), k-major [H/4][4H][4] otherwise.
bool lstm_seq_supported(int H) { return H == 64 || H == 128; }

// time steps [q.t0, q.t1) of one layer on the register kernel (H = 64 / 128), rows addressed through q (LstmSeq)
hipError_t launch_lstm_seq(int dt, const void* pre, const float* w_hh, void* out, const void* skip, int B, int H, const LstmSeq& q,
                           hipStream_t s) {
  if (!lstm_seq_supported(H) || q.t1 <= q.t0) return hipErrorInvalidValue;
  if (H == 64) {
    if (dt == DT_F32) hipLaunchKernelGGL((lstm_reg_kernel<float, 64>), dim3(B), dim3(256), 0, s, pre, w_hh, out, skip, q);
    else hipLaunchKernelGGL((lstm_reg_kernel<__bf16, 64>), dim3(B), dim3(256), 0, s, pre, w_hh, out, skip, q);
  } else {
    if (dt == DT_F32) hipLaunchKernelGGL((lstm_reg_kernel<float, 128>), dim3(B), dim3(512), 0, s, pre, w_hh, out, skip, q);
    else hipLaunchKernelGGL((lstm_reg_kernel<__bf16, 128>), dim3(B), dim3(512), 0, s, pre, w_hh, out, skip, q);
  }
  return hipGetLastError();
}

hipError_t launch_lstm_layer(int dt, const void* pre, const float* w_hh, void* out, const void* skip, int B, int T,
                             int H, hipStream_t s) {
  if (H % 8 || H > 4096) return hipErrorInvalidValue;
  if (H == 64 || H == 128) {
    LstmSeq q;
    q.t0 = 0; q.t1 = T; q.pre_bs = q.out_bs = q.skip_bs = T; q.pre_ts = q.out_ts = q.skip_ts = 1; q.state = nullptr;
    return launch_lstm_seq(dt, pre, w_hh, out, skip, B, H, q, s);
  } else {
    const size_t lds = (size_t)5 * H * sizeof(float);
    if (dt == DT_F32) hipLaunchKernelGGL(lstm_stream_kernel<float>, dim3(B), dim3(1024), lds, s, pre, w_hh, out, skip, T, H);
    else hipLaunchKernelGGL(lstm_stream_kernel<__bf16>, dim3(B), dim3(1024), lds, s, pre, w_hh, out, skip, T, H);
  }
  return hipGetLastError();
}

}  // namespace ldc
