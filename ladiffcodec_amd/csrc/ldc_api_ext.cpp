// ldc_api_ext.cpp -- the C ABI entry points of SURVEY.md section 8(f)'s "next" rows: the bit-stream layer (index packing,
// arithmetic coder), the resampling front end and the training step's kernels (q_sample, objective, Block / conv / attention /
// LayerNorm forward + backward, Adam).  Thin argument checks around the launchers of bitstream.hip, diffusion.hip, train.hip and
// train_mm3.hip; the context and its helpers live in ldc_api.cpp (ldc_internal.h).
#include "ldc_internal.h"

// ------------------------------------------------------------------------------------------------
// bit-stream layer.  These calls need no weights: any context of the device will do.
// ------------------------------------------------------------------------------------------------

extern "C" int64_t ldc_packed_bytes(int n_q, int F, int bits) { return ((int64_t)n_q * F * bits + 7) / 8; }

extern "C" int ldc_pack_codes(ldc_ctx* c, const int64_t* codes, int n_q, int B, int F, int bits, uint8_t* out, int64_t out_stride,
                              void* stream) {
  LDCCHK(check_dev(c));
  if (!codes || !out || n_q < 1 || B < 1 || F < 0 || bits < 1 || bits > 24) return fail(LDC_E_INVALID, "bad arguments (bits must be in [1,24])");
  if (out_stride < ldc_packed_bytes(n_q, F, bits)) return fail(LDC_E_INVALID, "out_stride %lld < %lld packed bytes per item", (long long)out_stride, (long long)ldc_packed_bytes(n_q, F, bits));
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_pack_codes(codes, n_q, B, F, bits, out, out_stride, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_unpack_codes(ldc_ctx* c, const uint8_t* in, int64_t in_stride, int n_q, int B, int F, int bits, int64_t* codes_out,
                                void* stream) {
  LDCCHK(check_dev(c));
  if (!in || !codes_out || n_q < 1 || B < 1 || F < 0 || bits < 1 || bits > 24) return fail(LDC_E_INVALID, "bad arguments (bits must be in [1,24])");
  if (in_stride < ldc_packed_bytes(n_q, F, bits)) return fail(LDC_E_INVALID, "in_stride smaller than the packed size: the stream ended sooner than expected");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_unpack_codes(in, in_stride, n_q, B, F, bits, codes_out, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_ac_build_cdf(ldc_ctx* c, const float* pdf, int rows, int card, int total_range_bits, float roundoff, int min_range,
                                int32_t* cdf_out, void* stream) {
  LDCCHK(check_dev(c));
  if (!pdf || !cdf_out || rows < 1 || card < 1) return fail(LDC_E_INVALID, "bad arguments");
  if (total_range_bits < 2 || total_range_bits > 30) return fail(LDC_E_INVALID, "total_range_bits must be <= 30 (ac.py:98)");
  if (min_range < 2) return fail(LDC_E_INVALID, "min_range must be at least 2. (ac.py:47-48)");
  if ((double)min_range * card > (double)(1ll << total_range_bits)) return fail(LDC_E_INVALID, "you must reduce min_range (ac.py:43)");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_build_cdf(pdf, rows, card, total_range_bits, roundoff, min_range, cdf_out, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_ac_encode(ldc_ctx* c, const int32_t* symbols, const int32_t* cdf, int B, int S, int card, int n_static,
                             int total_range_bits, uint8_t* out, int64_t out_stride, int64_t* nbytes_out, void* stream) {
  LDCCHK(check_dev(c));
  if (!symbols || !cdf || !out || !nbytes_out || B < 1 || S < 0 || card < 1 || n_static < 0 || out_stride < 1) return fail(LDC_E_INVALID, "bad arguments");
  if (total_range_bits < 2 || total_range_bits > 30) return fail(LDC_E_INVALID, "total_range_bits must be <= 30 (ac.py:98)");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_ac_encode(symbols, cdf, B, S, card, n_static ? 1 : 0, std::max(1, n_static), total_range_bits, out, out_stride, out_stride,
                          nbytes_out, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_ac_decode(ldc_ctx* c, const uint8_t* in, int64_t in_stride, const int64_t* nbytes, const int32_t* cdf, int B, int S,
                             int card, int n_static, int total_range_bits, int32_t* symbols_out, int32_t* status_out, void* stream) {
  LDCCHK(check_dev(c));
  if (!in || !nbytes || !cdf || !symbols_out || !status_out || B < 1 || S < 0 || card < 1 || n_static < 0) return fail(LDC_E_INVALID, "bad arguments");
  if (total_range_bits < 2 || total_range_bits > 30) return fail(LDC_E_INVALID, "total_range_bits must be <= 30 (ac.py:98)");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_ac_decode(in, in_stride, nbytes, cdf, B, S, card, n_static ? 1 : 0, std::max(1, n_static), total_range_bits, symbols_out,
                          status_out, s));
  return finish_stream(c, stream);
}

// ------------------------------------------------------------------------------------------------
// audio front end: torchaudio.functional.resample as srcs/sample.py:84 calls it
// ------------------------------------------------------------------------------------------------
static int gcd_i(int a, int b) { return b ? gcd_i(b, a % b) : a; }

extern "C" int64_t ldc_resample_out_len(int64_t T, int orig_freq, int new_freq) {
  if (orig_freq <= 0 || new_freq <= 0 || T < 0) return -1;
  const int g = gcd_i(orig_freq, new_freq);
  const int64_t orig = orig_freq / g, nnew = new_freq / g;
  return (nnew * T + orig - 1) / orig;       // ceil(new * T / orig), torchaudio's target_length
}

extern "C" int ldc_resample(ldc_ctx* c, const float* wav, int C, int64_t T, int orig_freq, int new_freq, float* out, void* stream) {
  LDCCHK(check_dev(c));
  if (!wav || !out || C < 1 || T < 1 || orig_freq <= 0 || new_freq <= 0) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  if (orig_freq == new_freq) {
    HIPCHK(hipMemcpyAsync(out, wav, (size_t)C * T * 4, hipMemcpyDeviceToDevice, s));
    return finish_stream(c, stream);
  }
  const int g = gcd_i(orig_freq, new_freq);
  const int orig = orig_freq / g, nnew = new_freq / g;
  // torchaudio 0.13 _get_sinc_resample_kernel: lowpass_filter_width 6, rolloff 0.99, Hann window.  functional.resample (the entry
  // srcs/sample.py:84 calls) hands the WAVEFORM's dtype to the kernel builder, so for the fp32 tensors torchaudio.load returns the bank
  // is built in fp32 tensor arithmetic (float64 is the transforms.Resample path, dtype=None): the same operations in the same order
  // here, every Python-float scalar rounded to fp32 where torch applies it to an fp32 tensor.
  const double lpw = 6.0, rolloff = 0.99, pi = 3.14159265358979323846;
  const double base_freq = std::min(orig, nnew) * rolloff;
  const int width = (int)ceil(lpw * orig / base_freq);
  const int K = 2 * width + orig;
  if ((double)nnew * K > 64e6) return fail(LDC_E_INVALID, "resampling %d -> %d needs a %d x %d filter bank: rates too incommensurate", orig_freq, new_freq, nnew, K);
  std::vector<float> bank((size_t)nnew * K);
  const float scale_f = (float)(base_freq / orig), base_f = (float)base_freq, pi_f = (float)pi, lpw_f = (float)lpw;
  for (int p = 0; p < nnew; ++p)
    for (int k = 0; k < K; ++k) {
      const float idx = (float)(k - width) / (float)orig;        // arange(-width, width + orig) / orig
      float t = (float)(-p) / (float)nnew + idx;                 // arange(0, -new, -1) / new + idx
      t *= base_f;
      t = std::max(-lpw_f, std::min(lpw_f, t));
      const float cw = cosf(t * pi_f / lpw_f / 2.0f);
      const float win = cw * cw;                                 // ** 2
      t *= pi_f;
      const float sinc = t == 0.0f ? 1.0f : sinf(t) / t;
      bank[(size_t)p * K + k] = sinc * (win * scale_f);          // kernels *= window * scale
    }
  const int64_t target = ldc_resample_out_len(T, orig_freq, new_freq);
  LDCCHK(with_scratch(c, s, [&](Arena& ar, bool dry) -> int {
    float* dbank = (float*)ar.alloc(bank.size() * 4);
    if (!dry) {
      // (pageable host memory: the copy is staged before the call returns, `bank` may go out of scope)
      HIPCHK(hipMemcpyAsync(dbank, bank.data(), bank.size() * 4, hipMemcpyHostToDevice, s));
      HIPCHK(hipStreamSynchronize(s));
      HIPCHK(launch_resample(wav, C, T, dbank, orig, nnew, width, target, out, s));
    }
    return LDC_OK;
  }));
  return finish_stream(c, stream);
}

// ------------------------------------------------------------------------------------------------
// training step, first slice
// ------------------------------------------------------------------------------------------------
extern "C" int ldc_train_q_sample(ldc_ctx* c, const float* x0, const int64_t* t, const float* noise, int B, int C, int L, float* x_t,
                                  void* stream) {
  LDCCHK(check_ready(c, LDC_MODEL_MAIN));
  if (!x0 || !t || !noise || !x_t || B < 1 || C < 1 || L < 1) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_q_sample(x0, noise, t, c->sqrt_alphas_cumprod, c->sqrt_one_minus_alphas_cumprod, B, (int64_t)C * L, x_t, c->unet.timesteps, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_num_timesteps(ldc_ctx* c) { return c ? c->unet.timesteps : 0; }

extern "C" int ldc_train_predict_x_start(ldc_ctx* c, const float* x_t, const float* eps, const int64_t* t, int B, int C, int L, float* x0_out,
                                         void* stream) {
  LDCCHK(check_ready(c, LDC_MODEL_MAIN));
  if (!x_t || !eps || !t || !x0_out || B < 1 || C < 1 || L < 1) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_predict_x_start(x_t, eps, t, c->sched.sqrt_recip_alphas_cumprod, c->sched.sqrt_recipm1_alphas_cumprod, B, (int64_t)C * L, x0_out,
                                c->unet.timesteps, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_neg_sdsdr(ldc_ctx* c, const float* est, const float* tgt, int B, int64_t n_per_item, float clip_min, float* per_item_out,
                                   void* stream) {
  LDCCHK(check_dev(c));
  if (!est || !tgt || !per_item_out || B < 1 || n_per_item < 1) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_neg_sdsdr(est, tgt, B, n_per_item, clip_min, per_item_out, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_l1_loss(ldc_ctx* c, const float* model_out, const float* target, const int64_t* t, int B, int C, int L,
                                 float* loss_out, float* grad_out, void* stream) {
  LDCCHK(check_ready(c, LDC_MODEL_MAIN));
  if (!model_out || !target || !t || !loss_out || B < 1 || C < 1 || L < 1) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  LDCCHK(with_scratch(c, s, [&](Arena& ar, bool dry) -> int {
    void* ws = ar.alloc(l1_loss_ws_bytes(B));
    if (!dry) HIPCHK(launch_l1_loss(model_out, target, t, c->p2_loss_weight, B, (int64_t)C * L, loss_out, grad_out, ws, c->unet.timesteps, s));
    return LDC_OK;
  }));
  return finish_stream(c, stream);
}

extern "C" int64_t ldc_train_block_ws_floats(int B, int Cin, int Cout, int L, int groups) {
  return (int64_t)train_block_ws_floats(B, Cin, Cout, L, groups);
}

extern "C" int ldc_train_block_forward(ldc_ctx* c, const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                                       const float* scale_shift, int B, int Cin, int Cout, int L, int groups, float* y, float* ws,
                                       void* stream) {
  LDCCHK(check_dev(c));
  if (!x || !w || !gamma || !beta || !y || !ws || B < 1 || Cin < 1 || Cout < 1 || L < 1 || groups < 1 || Cout % groups)
    return fail(LDC_E_INVALID, "bad arguments (Cout must be a multiple of groups)");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_block_forward(x, w, bias, gamma, beta, scale_shift, B, Cin, Cout, L, groups, y, ws, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_block_backward(ldc_ctx* c, const float* dy, const float* x, const float* gamma, const float* beta,
                                        const float* scale_shift, int B, int Cin, int Cout, int L, int groups, float* ws, float* dx,
                                        float* dw, float* db, float* dgamma, float* dbeta, float* dscale_shift, void* stream) {
  LDCCHK(check_dev(c));
  if (!dy || !x || !gamma || !beta || !ws || !dw || !db || !dgamma || !dbeta || B < 1 || Cout % groups) return fail(LDC_E_INVALID, "bad arguments");
  if (scale_shift && !dscale_shift) return fail(LDC_E_INVALID, "dscale_shift is required when scale_shift was given");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_block_backward(dy, x, gamma, beta, scale_shift, B, Cin, Cout, L, groups, ws, dx, dw, db, dgamma, dbeta,
                                     scale_shift ? dscale_shift : nullptr, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_pointwise_forward(ldc_ctx* c, const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int L,
                                           int pre_silu, float* y, void* stream) {
  LDCCHK(check_dev(c));
  if (!x || !w || !y || B < 1 || Cin < 1 || Cout < 1 || L < 1) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_pw_forward(x, w, bias, B, Cin, Cout, L, pre_silu ? 1 : 0, y, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_pointwise_backward(ldc_ctx* c, const float* dy, const float* x, const float* w, int B, int Cin, int Cout, int L,
                                            int pre_silu, float* dx, float* dw, float* db, void* stream) {
  LDCCHK(check_dev(c));
  if (!dy || !x || !w || !dw || B < 1 || Cin < 1 || Cout < 1 || L < 1) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_pw_backward(dy, x, w, B, Cin, Cout, L, pre_silu ? 1 : 0, dx, dw, db, s));
  return finish_stream(c, stream);
}

extern "C" int64_t ldc_train_linattn_ws_floats(int B, int heads, int dim_head, int N) {
  return (int64_t)train_linattn_ws_floats(B, heads, dim_head, N);
}

extern "C" int ldc_train_linattn_forward(ldc_ctx* c, const float* qkv, int B, int heads, int dim_head, int N, float* out, float* ws, void* stream) {
  LDCCHK(check_dev(c));
  if (!qkv || !out || !ws || B < 1 || heads < 1 || N < 1 || dim_head < 1 || dim_head > 64 || 256 % dim_head)
    return fail(LDC_E_INVALID, "bad arguments (dim_head must divide 256 and be <= 64)");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_linattn_forward(qkv, B, heads, dim_head, N, out, ws, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_linattn_backward(ldc_ctx* c, const float* dout, const float* qkv, int B, int heads, int dim_head, int N, float* ws,
                                          float* dqkv, void* stream) {
  LDCCHK(check_dev(c));
  if (!dout || !qkv || !ws || !dqkv || B < 1 || heads < 1 || N < 1 || dim_head < 1 || dim_head > 64 || 256 % dim_head)
    return fail(LDC_E_INVALID, "bad arguments (dim_head must divide 256 and be <= 64)");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_linattn_backward(dout, qkv, B, heads, dim_head, N, ws, dqkv, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_conv_forward(ldc_ctx* c, const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int Lin, int K,
                                      int stride, int pad, float* y, void* stream) {
  LDCCHK(check_dev(c));
  if (!x || !w || !y || B < 1 || Cin < 1 || Cout < 1 || Lin < 1 || K < 1 || stride < 1 || pad < 0 || (Lin + 2 * pad - K) / stride + 1 < 1)
    return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_conv_forward(x, w, bias, B, Cin, Cout, Lin, K, stride, pad, y, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_conv_backward(ldc_ctx* c, const float* dy, const float* x, const float* w, int B, int Cin, int Cout, int Lin, int K,
                                       int stride, int pad, float* dx, float* dw, float* db, void* stream) {
  LDCCHK(check_dev(c));
  if (!dy || !x || !w || !dw || B < 1 || Cin < 1 || Cout < 1 || Lin < 1 || K < 1 || stride < 1 || pad < 0 || (Lin + 2 * pad - K) / stride + 1 < 1)
    return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_conv_backward(dy, x, w, B, Cin, Cout, Lin, K, stride, pad, dx, dw, db, s));
  return finish_stream(c, stream);
}

/* The caller's stream waits for the weight-gradient launches the backward pass put on the training side stream (option
 * "train_dw_side"); a no-op when there are none.  Call it between the backward pass and the first use of a parameter gradient. */
/* The side stream's handle (hipStream_t), for callers whose allocator must know that a tensor handed to a backward call is still read
 * there (torch: tensor.record_stream(torch.cuda.ExternalStream(handle))). */
extern "C" int ldc_train_side_stream(ldc_ctx* c, void** out) {
  LDCCHK(check_dev(c));
  if (!out) return fail(LDC_E_INVALID, "bad arguments");
  *out = (void*)train_side_stream();
  if (!*out) return fail(LDC_E_HIP, "could not create the training side stream");
  return LDC_OK;
}
extern "C" int ldc_train_join(ldc_ctx* c, void* stream) {
  LDCCHK(check_dev(c));
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_join(s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_upsample2(ldc_ctx* c, const float* in, int64_t rows, int L, int backward, float* out, void* stream) {
  LDCCHK(check_dev(c));
  if (!in || !out || rows < 1 || L < 1) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_upsample2(in, rows, L, backward ? 1 : 0, out, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_activation(ldc_ctx* c, const float* x, const float* dy, int64_t n, int kind, float* out, void* stream) {
  LDCCHK(check_dev(c));
  if (!x || !out || n < 1 || kind < 0 || kind > 2) return fail(LDC_E_INVALID, "bad arguments (kind: 0 tanh, 1 GELU, 2 SiLU)");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_act(x, dy, n, kind, out, s));
  return finish_stream(c, stream);
}

extern "C" int64_t ldc_train_attn_ws_floats(int B, int heads, int N) { return (int64_t)train_attn_ws_floats(B, heads, N); }

extern "C" int ldc_train_attn_forward(ldc_ctx* c, const float* qkv, int B, int heads, int dim_head, int N, float* out, float* ws, void* stream) {
  LDCCHK(check_dev(c));
  if (!qkv || !out || !ws || B < 1 || heads < 1 || N < 1 || dim_head < 1 || dim_head > 64 || 256 % dim_head)
    return fail(LDC_E_INVALID, "bad arguments (dim_head must divide 256 and be <= 64)");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_attn_forward(qkv, B, heads, dim_head, N, out, ws, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_attn_backward(ldc_ctx* c, const float* dout, const float* qkv, int B, int heads, int dim_head, int N, float* ws,
                                       float* dqkv, void* stream) {
  LDCCHK(check_dev(c));
  if (!dout || !qkv || !ws || !dqkv || B < 1 || heads < 1 || N < 1 || dim_head < 1 || dim_head > 64 || 256 % dim_head)
    return fail(LDC_E_INVALID, "bad arguments (dim_head must divide 256 and be <= 64)");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_attn_backward(dout, qkv, B, heads, dim_head, N, ws, dqkv, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_convtr_forward(ldc_ctx* c, const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int L, int ratio,
                                        float* y, void* stream) {
  LDCCHK(check_dev(c));
  if (!x || !w || !y || B < 1 || Cin < 1 || Cout < 1 || L < 1 || ratio < 1) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_convtr_forward(x, w, bias, B, Cin, Cout, L, ratio, y, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_convtr_backward(ldc_ctx* c, const float* dy, const float* x, const float* w, int B, int Cin, int Cout, int L, int ratio,
                                         float* dx, float* dw, float* db, void* stream) {
  LDCCHK(check_dev(c));
  if (!dy || !x || !w || !dw || B < 1 || Cin < 1 || Cout < 1 || L < 1 || ratio < 1) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_convtr_backward(dy, x, w, B, Cin, Cout, L, ratio, dx, dw, db, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_maxscale(ldc_ctx* c, const float* x, const float* dy, int B, int64_t n_per_item, float* out, void* stream) {
  LDCCHK(check_dev(c));
  if (!x || !out || B < 1 || n_per_item < 1) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_maxscale(x, dy, B, n_per_item, out, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_adam_step(ldc_ctx* c, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int step,
                                   float lr, float beta1, float beta2, float eps, void* stream) {
  LDCCHK(check_dev(c));
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f))
    return fail(LDC_E_INVALID, "bad arguments (step counts from 1)");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_adam(param, grad, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_adam_step_dev(ldc_ctx* c, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t* step_dev,
                                       float lr, float beta1, float beta2, float eps, void* stream) {
  LDCCHK(check_dev(c));
  if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev || n < 0 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f))
    return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_adam_dev(param, grad, exp_avg, exp_avg_sq, n, step_dev, lr, beta1, beta2, eps, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_layernorm_forward(ldc_ctx* c, const float* x, const float* g, int B, int C, int L, float* y, float* stats, void* stream) {
  LDCCHK(check_dev(c));
  if (!x || !g || !y || !stats || B < 1 || C < 1 || L < 1) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_ln_forward(x, g, B, C, L, y, stats, s));
  return finish_stream(c, stream);
}

extern "C" int ldc_train_layernorm_backward(ldc_ctx* c, const float* dy, const float* x, const float* g, const float* stats, int B, int C, int L,
                                            float* dx, float* dg, void* stream) {
  LDCCHK(check_dev(c));
  if (!dy || !x || !g || !stats || !dx || !dg || B < 1 || C < 1 || L < 1) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_train_ln_backward(dy, x, g, stats, B, C, L, dx, dg, s));
  return finish_stream(c, stream);
}

