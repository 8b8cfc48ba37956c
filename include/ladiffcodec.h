/*
 * ladiffcodec.h  --  C ABI of the MI355X-native LaDiffCodec decode path (libladiffcodec.so).
 *
 * The reference (haiciyang/LaDiffCodec) is pure Python/PyTorch and exposes no plugin / FFI layer;
 * its hot path is `nn.Module` calls made by `synthesis()` (srcs/sample.py:50-136).  Each entry point
 * below replaces one of those calls; the reference call site it stands in for is cited next to it.
 * The Python host in `ladiffcodec_amd/` binds these with ctypes and mirrors the reference's module
 * attributes (`get_cond`, `diff_model.upsampling_layers`, `diffusion.halfway_sampling`, `decoder`).
 *
 * Conventions
 *   - Every function returns 0 on success or a negative LDC_E_* code; `ldc_last_error()` returns a
 *     thread-local human-readable message for the last failure.  No C++ exception crosses the ABI.
 *   - Tensor arguments are DEVICE pointers owned by the caller, contiguous, in the reference's own
 *     layouts: activations [B, C, L] float32, RVQ codes [n_q, B, F] int64.  The caller keeps them
 *     alive until `stream` has been synchronised.  `stream` is a hipStream_t passed as void*.
 *     stream != NULL: all work is queued asynchronously on that stream.  stream == NULL: the call runs
 *     on the context's own (non-blocking) stream and returns after synchronising it.
 *   - Device-wide synchronisation happens only in three documented places, never in the steady
 *     state: (1) when a context-owned workspace has to grow (first call at a larger shape),
 *     (2) when a captured step graph is replaced or evicted (a new (B, L, F), a new noise pointer,
 *     the LRU plan cache bound, see LDC_PLAN_CACHE_GB / LDC_PLAN_CACHE_N), (3) in ldc_destroy.
 *     Besides those, a sampler call may wait ON THE HOST for an earlier step-graph replay of the same
 *     context (bounded look-ahead, LDC_FLOW_DEPTH = 8 replays; an event wait, not a device sync).
 *   - Several contexts may be used side by side on one device (one per stream of work): two batches
 *     in flight, each decoded by its own context on its own stream, is how `python -m srcs.sample`
 *     and `bench.py --in-flight 2` raise throughput.
 *   - Reproducibility: GroupNorm statistics and the fused column maxima are accumulated with fp32
 *     atomics, so two identical UNet calls agree to rounding (~1e-6 relative), not bit for bit; the
 *     split-K reduction order is fixed.  The codec stages (SEANet, LSTM, RVQ) use no atomics: RVQ code
 *     indices are bit-reproducible.
 *   - An asynchronous device-side failure (the cooperative LSTM's bounded spin timing out) is
 *     reported as LDC_E_HIP by the synchronous call that hit it or by the next call on the context.
 *   - Weights are handed over on the HOST (float32, the `.amlt` state-dict tensors,
 *     srcs/utils.py:98-108), one call per state-dict key, then folded / packed / uploaded by
 *     `ldc_finalize_weights`.
 *   - A context is bound to one device and is not thread-safe (one context per rank / stream).
 *   - Internally activations are channels-last [B, L, C] in the context's compute dtype
 *     (LDC_F32: exact-fp32 MFMA path for parity; LDC_BF16: bf16 storage + bf16 MFMA, fp32
 *     accumulation, fp32 diffusion state).
 */
#ifndef LADIFFCODEC_H_
#define LADIFFCODEC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDC_OK 0
#define LDC_E_INVALID -1   /* bad argument / unsupported configuration          */
#define LDC_E_STATE -2     /* call out of order (e.g. stage before finalize)    */
#define LDC_E_MISSING -3   /* strict load: missing / unexpected / mis-shaped key */
#define LDC_E_HIP -4       /* a HIP runtime call failed                          */
#define LDC_E_NOMEM -5

#define LDC_F32 0
#define LDC_BF16 1
#define LDC_BF16_W8 2   /* bf16 activations, UNet conv weights stored as OCP fp8 e4m3 + one fp32 scale per output channel
                          (BASELINE config 5); codec stages stay fp32 */

/* which of the two DiffAudioRep instances a call addresses (srcs/sample.py:56 and :63) */
#define LDC_MODEL_MAIN 0   /* --model_path      : autoencoder [enc_ratios] + Unet1D + diffusion */
#define LDC_MODEL_COND 1   /* --model_for_cond  : EnCodec-style codec, always ratios [8,5,4,2] (quirk Q1) */

#define LDC_MAX_RATIOS 8

typedef struct ldc_ctx ldc_ctx;

/* Mirrors the argparse flags of srcs/sample.py:141-201 that shape the two models. */
typedef struct ldc_config {
  int32_t compute_dtype;                 /* LDC_F32 | LDC_BF16 | LDC_BF16_W8 */
  /* shared SEANet hyper-parameters (model.py:52-55) */
  int32_t rep_dims;                      /* --rep_dims            (128) */
  int32_t n_filters;                     /* --n_filters           (32)  */
  int32_t n_residual_layers;             /* --n_residual_layers   (1)   */
  int32_t lstm;                          /* --lstm                (2)   */
  /* main model */
  int32_t n_enc_ratios;                  /* --enc_ratios          ([8]) */
  int32_t enc_ratios[LDC_MAX_RATIOS];
  int32_t diff_dims;                     /* --diff_dims           (256) */
  int32_t n_upsampling_ratios;           /* --upsampling_ratios   ([5,4,2]); 0 = None */
  int32_t upsampling_ratios[LDC_MAX_RATIOS];
  int32_t unet_scale_cond;               /* --unet_scale_cond */
  int32_t unet_scale_x;                  /* --unet_scale_x    */
  /* cond model */
  int32_t has_cond_model;                /* --model_for_cond given */
  float cond_bandwidth;                  /* --cond_bandwidth      (3.0): builds floor(1000*bw / (50*10)) codebooks */
  /* capacity hints (workspaces are sized lazily; these only pre-size) */
  int32_t max_batch;
  int32_t max_latent_len;
  uint64_t noise_seed;                   /* device Philox stream used when noise == NULL */
  int32_t final_activation;              /* --final_activation: LDC_ACT_* applied after both encoders' last conv */
  int32_t reserved_;
} ldc_config;

/* --final_activation names (getattr(nn, name)() with default arguments, seanet.py:144-149) */
#define LDC_ACT_NONE 0
#define LDC_ACT_TANH 1
#define LDC_ACT_SIGMOID 2
#define LDC_ACT_ELU 3
#define LDC_ACT_SILU 4
#define LDC_ACT_GELU 5
#define LDC_ACT_RELU 6

const char* ldc_last_error(void);
const char* ldc_version(void);

/* lifecycle ---------------------------------------------------------------------------------- */
int ldc_create(const ldc_config* cfg, int device, ldc_ctx** out);
int ldc_destroy(ldc_ctx* ctx);

/* Per-context run-time options (instead of environment variables read at ldc_create): "split" = independent chains a batch is
 * decoded as (1..4, default 2; the reference has no counterpart: utterances never interact inside the UNet), "lstm_stream" =
 * 1: never use the cooperative (co-resident workgroups) LSTM kernel, "side_streams" = 1: res_conv on a side stream (split 1
 * only), "fp8_act" (fp8 contexts, before ldc_finalize_weights), "train_fp32_mfma" = 1: the training GEMMs on the exact-fp32
 * MFMA (round-2 kernels) instead of the split-bf16 ones (three bf16 MFMAs per product, 2^-16-class: csrc/train_mm3.hip) -- this
 * one is process-wide, like "train_bf16" = 1: one bf16 MFMA per product in the training GEMMs (autocast-class numerics, opt-in).
 * Launch structure of the UNet step (round 4; each keeps the reference's arithmetic, unet.py:137-246): "fuse_gn_epi" (default 1) = the
 * GroupNorm apply of a ResnetBlock's Block inside the producing conv, behind an in-launch exchange of per-wave statistics -- 0 restores
 * the conv + gn_apply launch pairs and is the fallback when a "[gn_wait]" device-side failure is reported; "fold_res" (1) = res_conv as a
 * fourth weight slab of block1's conv; "fold_ln" (1) = the attention blocks' PreNorm LayerNorm inside to_qkv.
 * Round 6: "conv_lean" (1) = the instruction-diet conv kernel (csrc/conv_lean.inc) where its shapes allow, 0 = conv_fast_kernel
 * everywhere (same tiles, same results: for A/B runs); "part_graphs" (1) = one single-stream step graph per batch part (two parts),
 * 0 = one fork / join graph, 2 = per-part graphs for three / four parts as well.  (Removed in round 6, measured slower in rounds 4-5:
 * "chain_convs" -- block1's and block2's convs as one launch -- and "xcd_teams" -- runs of convs as persistent XCD-team launches;
 * profiles/r04_fusion_experiments.md, profiles/r05_team_chain_experiments.md keep the numbers.)
 * "conv_xcd_order" (1) = the lean kernel's dispatch order as an xm x xn arrangement of the eight XCDs chosen per launch by operand
 * bytes (0 = conv_fast_kernel's order; 2 / 4 / 8 = xn forced); "fold_ctx" (1) = the LinearAttention context inside to_qkv's epilogue.
 * The whole table (name, default, effect) is in tools/README.md; LDC_OPTIONS="name=value,..." in the environment sets the same
 * names at ldc_create.  Cached plans and graphs are dropped when a value changes. */
int ldc_set_option(ldc_ctx* ctx, const char* name, int value);

/* Device-drawn noise (noise == NULL): Philox4x32-10 keyed by (noise_seed, call counter); every sampler call that
 * draws advances the counter, as every torch.randn_like of the reference (ddpm_loss.py:249) advances the global
 * generator.  ldc_reseed sets the seed and rewinds the counter (the counterpart of torch.manual_seed). */
int ldc_reseed(ldc_ctx* ctx, uint64_t seed);
/* The fp8 weight packer's rounding (OCP e4m3fn, nearest even, saturating at 448), host-side: codes and/or decoded values. */
void ldc_quantize_e4m3(const float* in, int64_t n, uint8_t* out_codes, float* out_values);

/* load_model(model, path, strict) -- srcs/utils.py:98-108.  One call per state-dict entry (after
 * the caller stripped any `module.` prefix).  `data` is a host float32 buffer of prod(shape)
 * elements.  Keys under `diffusion.model.*` alias `diff_model.*` and may be passed or skipped. */
int ldc_set_weight(ldc_ctx* ctx, int which, const char* key, const float* data, const int64_t* shape, int ndim);
/* strict != 0 reproduces load_state_dict(strict=True): every expected key present with the expected
 * shape, no unexpected key.  Folds weight-norm (conv.py:27-30) and weight-standardisation
 * (unet.py:73-78, eps 1e-5, fp32) once, packs for MFMA, builds the timestep scale/shift table. */
int ldc_finalize_weights(ldc_ctx* ctx, int strict);

/* stages ---------------------------------------------------------------------------------------- */
/* model.encoder(wav)  (seanet.py:153)            wav [B,1,T] -> z [B,rep_dims,T/hop]            */
int ldc_seanet_encode(ldc_ctx* ctx, int which, const float* wav, int B, int T, float* z_out, void* stream);
/* model.decoder(z)    (seanet.py:246; sample.py:131)   z [B,rep_dims,L] -> wav [B,1,L*hop]      */
int ldc_seanet_decode(ldc_ctx* ctx, int which, const float* z, int B, int L, float* wav_out, void* stream);
/* model.quantizer(z, frame_rate, bandwidth) in eval (vq.py:69-84, core_vq.py:324-342).
 * n_q = number of codebooks used.  codes_out [n_q,B,F] int64 (may be NULL), quantized_out [B,D,F]. */
int ldc_rvq_encode(ldc_ctx* ctx, const float* z, int B, int F, int n_q, int64_t* codes_out, float* quantized_out,
                   void* stream);
/* quantizer.decode(codes) (core_vq.py:356-362) */
int ldc_rvq_decode(ldc_ctx* ctx, const int64_t* codes, int B, int F, int n_q, float* quantized_out, void* stream);
/* model_for_cond.get_cond(wav) (model.py:223-231) = encode + rvq, fused on one stream.
 * codes_out may be NULL.  bandwidth <= 0 means the configured cond_bandwidth. */
int ldc_get_cond(ldc_ctx* ctx, const float* wav, int B, int T, float bandwidth, float* cond_out, int64_t* codes_out,
                 void* stream);
/* for layer in diff_model.upsampling_layers: img = layer(img)   (sample.py:125-128, unet.py:372-377)
 * normalise: 0 = raw; 1 = img /= max|img|+1e-8 over the whole tensor (sample.py:129);
 *            2 = the same per batch item (a batch of independent utterances). cond [B,C,F] -> [B,C,L] */
int ldc_cond_upsample(ldc_ctx* ctx, const float* cond, int B, int F, int normalise, float* img_out, void* stream);
/* diff_model(x, t, cond)  (Unet1D.forward, unet.py:422-469).  cond is the RAW condition [B,C,F]
 * (process_cond runs inside, as in the reference).  t is one timestep for the whole batch: the sampler
 * only ever calls the model with torch.full((b,), t) (ddpm_loss.py:247); per-item t is not supported. */
int ldc_unet_forward(ldc_ctx* ctx, const float* x, int t, const float* cond, int B, int L, int F, float* eps_out,
                     void* stream);
/* diffusion.p_sample(x, t, cond) (ddpm_loss.py:244-251).  noise [B,C,L] or NULL (NULL: Philox draw;
 * ignored when t == 0). x is updated in place. */
int ldc_p_sample(ldc_ctx* ctx, float* x_inout, int t, const float* cond, const float* noise, int B, int L, int F,
                 void* stream);
/* diffusion.halfway_sampling(img, t=n_steps, condition=cond) (ddpm_loss.py:370-385): t = n_steps-1..0.
 * noise [n_steps,B,C,L] (entry j is consumed at iteration j; the last is unused) or NULL.
 * Five steps of every batch part are captured in one hipGraph keyed by (B, L, F) and replayed; the loop runs on a
 * context-owned copy of img, so a new tensor per call does not force a re-capture (a new noise pointer does). */
int ldc_denoise(ldc_ctx* ctx, float* img_inout, const float* cond, const float* noise, int n_steps, int B, int L,
                int F, void* stream);
/* SURVEY.md section 8(f) row 1 -- the two alternative decode modes left commented in sample.py:96-122; same kernels,
 * different loop driver.
 * diffusion.p_sample_loop(shape, condition) (ddpm_loss.py:253-266): ancestral sampling over all timesteps.
 * fill_start != 0: img is drawn ~N(0,1) on the device first (ddpm_loss.py:257); 0: img already holds the start image.
 * noise [timesteps,B,C,L] or NULL as for ldc_denoise. */
int ldc_p_sample_loop(ldc_ctx* ctx, float* img_inout, const float* cond, const float* noise, int fill_start, int B, int L,
                      int F, void* stream);
/* diffusion.infilling(infill_img, condition, midway_t, lam=lam) (ddpm_loss.py:331-367): per t = midway_t-1..0
 *   img <- p_sample(img,t); img <- (1-lam) img + lam infill; infill <- p_sample(infill,t); img <- (1-lam) img + lam infill.
 * Both img and infill_img are updated in place (the reference returns img).  fill_start != 0: img is drawn ~U[0,1) first
 * (ddpm_loss.py:336).  noise [2*midway_t,B,C,L] (draw 2i for img, 2i+1 for infill at iteration i) or NULL. */
int ldc_infilling(ldc_ctx* ctx, float* img_inout, float* infill_inout, const float* cond, int midway_t, const float* noise,
                  float lam, int fill_start, int B, int L, int F, void* stream);
/* sample.py:133-134: x /= std(x)+1e-8 ; x /= max|x|+1e-8.  per_item: 0 whole tensor, 1 per item. */
int ldc_output_normalise(ldc_ctx* ctx, float* wav_inout, int B, int T, int per_item, void* stream);
/* The whole per-batch body of synthesis() (sample.py:94-134) on resident buffers:
 * wav [B,1,T] -> wav_out [B,1,T]; optional stage outputs may be NULL. per_item as above. */
int ldc_decode(ldc_ctx* ctx, const float* wav, int B, int T, int n_steps, const float* noise, int per_item,
               float* wav_out, float* latents_out, float* cond_out, int64_t* codes_out, void* stream);

/* bit-stream layer: the on-wire format between ldc_rvq_encode and ldc_rvq_decode -- SURVEY.md section 8(f) row 3 ------------
 * Every batch item is an independent stream.  All results are bit-exact with the reference classes.  These calls need no
 * weights (any context of the device).
 * BitPacker / BitUnpacker (srcs/encodec/binary.py:55-118) over a frame's codes in compress.py's push order
 * (srcs/encodec/compress.py:74-84: for t: for k: codes[k][t]); `bits` per code (10 for the 1024-entry codebooks).
 * codes [n_q,B,F] int64 (device) <-> out [B][out_stride] bytes (device), ldc_packed_bytes() of them used per item. */
int64_t ldc_packed_bytes(int n_q, int F, int bits);
int ldc_pack_codes(ldc_ctx* ctx, const int64_t* codes, int n_q, int B, int F, int bits, uint8_t* out, int64_t out_stride, void* stream);
int ldc_unpack_codes(ldc_ctx* ctx, const uint8_t* in, int64_t in_stride, int n_q, int B, int F, int bits, int64_t* codes_out,
                     void* stream);
/* build_stable_quantized_cdf (srcs/quantization/ac.py:18-53): pdf [rows][card] float32 -> cdf [rows][card] int32. */
int ldc_ac_build_cdf(ldc_ctx* ctx, const float* pdf, int rows, int card, int total_range_bits, float roundoff, int min_range,
                     int32_t* cdf_out, void* stream);
/* ArithmeticCoder.push ... flush (ac.py:131-174) of S symbols per stream.  symbols [B][S] int32.  n_static == 0: cdf is
 * [B][S][card] (a table per step, as an LM would provide, compress.py:79-82); n_static > 0: cdf is [n_static][card] and
 * symbol s uses table s % n_static (one static table per codebook).  nbytes_out[b] = bytes written, -1 = out_stride too
 * small or a symbol outside its table. */
int ldc_ac_encode(ldc_ctx* ctx, const int32_t* symbols, const int32_t* cdf, int B, int S, int card, int n_static, int total_range_bits,
                  uint8_t* out, int64_t out_stride, int64_t* nbytes_out, void* stream);
/* ArithmeticDecoder.pull x S (ac.py:218-260).  status_out[b]: 0 ok, 1 stream exhausted (pull returned None), 2 search failed. */
int ldc_ac_decode(ldc_ctx* ctx, const uint8_t* in, int64_t in_stride, const int64_t* nbytes, const int32_t* cdf, int B, int S, int card,
                  int n_static, int total_range_bits, int32_t* symbols_out, int32_t* status_out, void* stream);

/* audio front end -- SURVEY.md section 8(f) row 4: torchaudio.functional.resample(wav, orig_freq, new_freq) with its defaults
 * (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99), the call at srcs/sample.py:84.  wav [C][T] -> out [C][ldc_resample_out_len]. */
int64_t ldc_resample_out_len(int64_t T, int orig_freq, int new_freq);
int ldc_resample(ldc_ctx* ctx, const float* wav, int C, int64_t T, int orig_freq, int new_freq, float* out, void* stream);

/* training step of the diffusion UNet -- SURVEY.md section 8(f) row 2 (BASELINE config 4); fp32 correctness path, reference layouts [B,C,L]
 * diffusion.q_sample(x_start, t, noise) (ddpm_loss.py:386-392); t [B] int64 (device). */
int ldc_train_q_sample(ldc_ctx* ctx, const float* x_start, const int64_t* t, const float* noise, int B, int C, int L, float* x_t,
                       void* stream);
/* Number of diffusion timesteps T of the loaded schedule (len(diffusion.betas)); device-side t is clamped to [0, T). */
int ldc_train_num_timesteps(ldc_ctx* ctx);
/* predicted_x_start of p_losses (ddpm_loss.py:416-420 -> :175-179, no clamp): x0 = sqrt(1/abar_t) x_t - sqrt(1/abar_t - 1) eps. */
int ldc_train_predict_x_start(ldc_ctx* ctx, const float* x_t, const float* eps, const int64_t* t, int B, int C, int L, float* x0_out,
                              void* stream);
/* Monitoring loss of DiffAudioRep.forward (model.py:194): per item clamp(-SD-SDR(est, tgt), min clip_min) -- ClippedSDR
 * (losses_fn.py:56-66) over asteroid 0.6.0's MultiSrcNegSDR("sdsdr"), one source; the reference calls it as (x, x_hat). */
int ldc_train_neg_sdsdr(ldc_ctx* ctx, const float* est, const float* tgt, int B, int64_t n_per_item, float clip_min, float* per_item_out,
                        void* stream);

/* The objective of p_losses (ddpm_loss.py:434-438, loss_type l1): loss = mean_b(p2_loss_weight[t_b] * mean_{c,l}|out - target|);
 * loss_out [1]; grad_out (nullable) = d loss / d model_out. */
int ldc_train_l1_loss(ldc_ctx* ctx, const float* model_out, const float* target, const int64_t* t, int B, int C, int L, float* loss_out,
                      float* grad_out, void* stream);
/* Block (unet.py:137-154): y = SiLU(GroupNorm(conv_k3(x; WS(w), bias)) * (scale + 1) + shift), weights as trained (raw `w`
 * [Cout,Cin,3]: the weight standardisation of unet.py:73-78 is part of the graph).  scale_shift [B][2*Cout] (scale | shift)
 * or NULL.  `ws`: ldc_train_block_ws_floats() floats of device scratch that carry the saved tensors from forward to backward. */
int64_t ldc_train_block_ws_floats(int B, int Cin, int Cout, int L, int groups);
int ldc_train_block_forward(ldc_ctx* ctx, const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                            const float* scale_shift, int B, int Cin, int Cout, int L, int groups, float* y, float* ws, void* stream);
/* gradients of everything: dx [B,Cin,L] (nullable), dw [Cout,Cin,3], db / dgamma / dbeta [Cout], dscale_shift [B][2*Cout]. */
int ldc_train_block_backward(ldc_ctx* ctx, const float* dy, const float* x, const float* gamma, const float* beta,
                             const float* scale_shift, int B, int Cin, int Cout, int L, int groups, float* ws, float* dx, float* dw,
                             float* db, float* dgamma, float* dbeta, float* dscale_shift, void* stream);

/* Channel LayerNorm of the UNet's attention blocks (srcs/modules/unet.py:82-101), forward and backward, [B, C, L] float32:
 * y = (x - mean_c) * rsqrt(var_c + 1e-5) * g.  `stats`: [B, L, 2] floats written by the forward pass (mean, 1/std), read by the
 * backward pass, which returns dx and dg [C]. */
int ldc_train_layernorm_forward(ldc_ctx* ctx, const float* x, const float* g, int B, int C, int L, float* y, float* stats, void* stream);
int ldc_train_layernorm_backward(ldc_ctx* ctx, const float* dy, const float* x, const float* g, const float* stats, int B, int C, int L,
                                 float* dx, float* dg, void* stream);

/* y[b, o, l] = bias[o] + sum_i w[o, i] * a(x[b, i, l]) on [B, C, L] float32, a = identity or SiLU (`pre_silu`): the 1x1 res_conv of a
 * ResnetBlock (srcs/modules/unet.py:171,192) and, with L = 1, its time-embedding MLP SiLU -> Linear (unet.py:163-166); backward
 * returns dx (may be NULL), dw [Cout, Cin] and db (may be NULL).  With ldc_train_block_* this is a whole ResnetBlock. */
int ldc_train_pointwise_forward(ldc_ctx* ctx, const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int L, int pre_silu,
                                float* y, void* stream);
int ldc_train_pointwise_backward(ldc_ctx* ctx, const float* dy, const float* x, const float* w, int B, int Cin, int Cout, int L, int pre_silu,
                                 float* dx, float* dw, float* db, void* stream);

/* LinearAttention core (srcs/modules/unet.py:208-221: both softmaxes and both einsums, between to_qkv and to_out), forward and backward.
 * qkv [B, 3*heads*dim_head, N] float32 = [q | k | v] (the to_qkv output), out [B, heads*dim_head, N]; `ws`:
 * ldc_train_linattn_ws_floats() floats carrying the saved softmaxes and context to the backward pass; dqkv in the layout of qkv. */
int64_t ldc_train_linattn_ws_floats(int B, int heads, int dim_head, int N);
int ldc_train_linattn_forward(ldc_ctx* ctx, const float* qkv, int B, int heads, int dim_head, int N, float* out, float* ws, void* stream);
int ldc_train_linattn_backward(ldc_ctx* ctx, const float* dout, const float* qkv, int B, int heads, int dim_head, int N, float* ws, float* dqkv,
                               void* stream);

/* The remaining layers of Unet1D (srcs/modules/unet.py:248-470) for the assembled forward / backward, [B, C, L] float32:
 * plain Conv1d with any kernel size / stride / zero padding (init_conv k7 p3 :307, Downsample k4 s2 p1 :64-65, the k3 p1 convs of
 * Upsample :58-62 and of the last levels, final_conv k1 :372); nearest x2 upsampling and its adjoint; tanh / GELU / SiLU
 * (`kind` 0 / 1 / 2; dy == NULL: out = f(x), else out = dy * f'(x)); the bottleneck softmax Attention core (:234-245) with
 * the [B, heads, N, N] attention matrix kept in `ws` (2 * B * heads * N * N floats) for the backward pass. */
int ldc_train_conv_forward(ldc_ctx* ctx, const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int Lin, int K, int stride,
                           int pad, float* y, void* stream);
int ldc_train_conv_backward(ldc_ctx* ctx, const float* dy, const float* x, const float* w, int B, int Cin, int Cout, int Lin, int K, int stride,
                            int pad, float* dx, float* dw, float* db, void* stream);
/* Option "train_dw_side" (process-wide, 0 / 1, default 0; DiffusionTrainer sets it around its backward pass): ldc_train_block_backward puts
 * the weight-gradient GEMM of the Block (and what hangs on it: the ordered reduction of its parts, the bias gradient, the
 * weight-standardisation backward) on an internal side stream behind an event, the caller's stream goes on with dX.  ldc_train_join makes
 * `stream` wait for everything the side stream holds: call it between the backward pass and the first use of a parameter gradient.  The
 * caller keeps the Block's workspace and saved input alive until then.  (The reference's autograd engine does the same thing with its
 * own streams; srcs/train.py:150-158 is `loss.backward()`.) */
int ldc_train_join(ldc_ctx* ctx, void* stream);
/* Round 6 (second half): ldc_train_conv_backward, ldc_train_pointwise_backward (L > 1) and ldc_train_layernorm_backward put their parameter
 * gradients on the side stream as well; those read `dy`, which the caller therefore keeps alive until ldc_train_join -- a torch caller marks
 * it with dy.record_stream(torch.cuda.ExternalStream(handle)), handle from ldc_train_side_stream. */
int ldc_train_side_stream(ldc_ctx* ctx, void** stream_out);
int ldc_train_upsample2(ldc_ctx* ctx, const float* in, int64_t rows, int L, int backward, float* out, void* stream);
int ldc_train_activation(ldc_ctx* ctx, const float* x, const float* dy, int64_t n, int kind, float* out, void* stream);
int64_t ldc_train_attn_ws_floats(int B, int heads, int N);
int ldc_train_attn_forward(ldc_ctx* ctx, const float* qkv, int B, int heads, int dim_head, int N, float* out, float* ws, void* stream);
int ldc_train_attn_backward(ldc_ctx* ctx, const float* dout, const float* qkv, int B, int heads, int dim_head, int N, float* ws, float* dqkv,
                            void* stream);

/* Unet1D.process_cond for training (srcs/modules/unet.py:372-377,401-420): the condition upsampler SConvTranspose1d(C, C, 2 * ratio,
 * stride ratio, non-causal; srcs/modules/conv.py:235-274) forward / backward on [B, C, L] -> [B, C, L * ratio] (weight [Cin, Cout,
 * 2 * ratio]), and the per-item max-abs scaling x / (max|x| + 1e-20): dy == NULL forward, else the gradient w.r.t. x. */
int ldc_train_convtr_forward(ldc_ctx* ctx, const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int L, int ratio, float* y,
                             void* stream);
int ldc_train_convtr_backward(ldc_ctx* ctx, const float* dy, const float* x, const float* w, int B, int Cin, int Cout, int L, int ratio, float* dx,
                              float* dw, float* db, void* stream);
int ldc_train_maxscale(ldc_ctx* ctx, const float* x, const float* dy, int B, int64_t n_per_item, float* out, void* stream);

/* One Adam step over flat device buffers, in place (srcs/train.py:365-371: optim.Adam(params, lr); torch's defaults are
 * beta1 0.9, beta2 0.999, eps 1e-8, no weight decay, no amsgrad).  `step` counts from 1 (bias correction). */
int ldc_train_adam_step(ldc_ctx* ctx, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr,
                        float beta1, float beta2, float eps, void* stream);
/* The same with the step count in device memory (*step_dev is counted up first, then used): the form a captured (hipGraph) optimisation
 * step needs -- a host-side count would be frozen into the graph. */
int ldc_train_adam_step_dev(ldc_ctx* ctx, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t* step_dev,
                            float lr, float beta1, float beta2, float eps, void* stream);

/* L1 primitives (reference srcs/modules/conv.py, lstm.py), exposed for the parity tests ---------- */
/* SConv1d.forward (conv.py:217-232), reflect padding.  w [Cout,Cin,k] (already weight-norm folded),
 * all HOST float32; x/y DEVICE [B,Cin,L] / [B,Cout,Lout].  pre_elu applies ELU to the input. */
int ldc_sconv1d(ldc_ctx* ctx, const float* x, int B, int Cin, int L, const float* w_host, const float* b_host,
                int Cout, int k, int stride, int dilation, int causal, int pre_elu, float* y, void* stream);
/* SConvTranspose1d.forward (conv.py:252-274), w [Cin,Cout,k] host. */
int ldc_sconvtr1d(ldc_ctx* ctx, const float* x, int B, int Cin, int L, const float* w_host, const float* b_host,
                  int Cout, int k, int stride, int causal, float* y, void* stream);
/* SLSTM.forward (lstm.py:22-28): weights host, PyTorch layout [4H,H] / [4H] per layer, order
 * w_ih, w_hh, b_ih, b_hh for layer 0 then layer 1 ... */
int ldc_slstm(ldc_ctx* ctx, const float* x, int B, int H, int T, const float* const* weights_host, int layers,
              float* y, void* stream);

/* introspection ------------------------------------------------------------------------------- */
/* Copy a named intermediate of the last ldc_unet_forward (channels-last -> [B,C,L] fp32).
 * Names: "cond_proc", "init", "down0".."downN", "mid", "up0".."upN".  For tests. */
int ldc_unet_debug_tap(ldc_ctx* ctx, const char* name, float* out, int64_t capacity_elems, void* stream);
/* Algorithmic work of one UNet step for (B, L): flops and bytes (activations + weights, compute dtype). */
int ldc_unet_step_cost(ldc_ctx* ctx, int B, int L, double* flops, double* bytes);
/* Timing of the dominant kernel class, measured with hipEvents on the launch stream when enabled.
 * ldc_profile_enable(ctx, 1) makes ldc_denoise bracket every conv-GEMM launch (eager, no graph). */
int ldc_profile_enable(ldc_ctx* ctx, int on);
/* Tuning aid: histogram of the XCC (die) ids `wgs` workgroups land on when launched on a stream created with the given CU mask
 * (n_words 32-bit words; NULL: unmasked). */
int ldc_xcc_census(ldc_ctx* ctx, const uint32_t* mask, int n_words, int wgs, int* hist16);

/* Test hook: raises the context's device-side failure flag exactly as a kernel whose bounded spin gave up would (code 1: the
 * cooperative LSTM's hidden-state exchange, 2: the in-launch GroupNorm exchange of a fused conv).  The next call on the context
 * reports LDC_E_HIP with "device-side failure [coop_lstm]" / "[gn_wait]" in ldc_last_error and clears the flag. */
int ldc_debug_raise_failure(ldc_ctx* ctx, int code);
/* Test hook: number of device-wide synchronisations (hipDeviceSynchronize) this library has issued in this process.  Steady-state
 * stage calls issue none: the count does not move across warm ldc_decode calls. */
long long ldc_debug_sync_count(void);

/* Host-side cost of the step-graph replays since the last reset: milliseconds spent inside hipGraphLaunch, milliseconds spent
 * waiting for the bounded look-ahead window (LDC_FLOW_DEPTH), number of replays. */
int ldc_host_stats(ldc_ctx* ctx, int reset, double* graph_launch_ms, double* lookahead_wait_ms, int64_t* graph_launches);
/* Preconditions of the timed mode (round 6): what the part-stream calibration measured -- streams of the context that overlap with the
 * caller's stream and each other / candidates (good = -1: none has run), wall ms of one 150 us spin alone and of all accepted streams
 * spinning together, parts a batch is decoded as. */
int ldc_stream_info(ldc_ctx* ctx, int* good, int* candidates, double* one_spin_ms, double* all_spin_ms, int* parts);
/* One sample of the device clocks behind everything queued on `stream` (synchronises it): out2[0] = 100 MHz wall clock,
 * out2[1] = shader cycles (s_memtime).  Two samples around a region give the shader clock it ran at. */
int ldc_clock_sample(ldc_ctx* ctx, uint64_t* out2, void* stream);

/* Device-side timeline of the timed (graph-replayed, multi-stream) mode: every step of every batch part stamps a 100 MHz
 * clock at its first and last kernel.  ldc_timeline_read: ticks[2j], ticks[2j+1] = begin / end of step j of `part`. */
int ldc_timeline_enable(ldc_ctx* ctx, int on);
int ldc_timeline_read(ldc_ctx* ctx, int part, int n, uint64_t* ticks);
/* Per-kernel durations of the TIMED mode (hipGraph replay, batch parts on their own streams): with stamps enabled every launch of
 * the pipelined conv kernel records its earliest workgroup start and latest workgroup end (100 MHz clock) in a slot of its own per
 * denoise step.  enable / disable rebuilds the plans; reset re-arms the buffers; read returns plan `idx` (creation order = batch
 * part) as ticks[n_steps][n_ops][2] (0 where the op is not a pipelined conv) with the ops' descriptions and LDC_CLASS_* codes.
 * ticks == NULL: only *n_ops is returned. */
int ldc_kstamps_enable(ldc_ctx* ctx, int on);
int ldc_kstamps_reset(ldc_ctx* ctx);
int ldc_kstamps_read(ldc_ctx* ctx, int idx, int n_steps, int* n_ops, uint64_t* ticks, char* infos, int info_cap, int* classes);
/* Tuning aid: times the GroupNorm-apply kernel on [B,L,C] (random data, fixed statistics). */
int ldc_gn_microbench(ldc_ctx* ctx, int dtype, int B, int L, int C, int with_residual, int iters, double* ms_per_launch);
/* Tuning aid: times `iters` launches of one conv-GEMM (random weights/inputs) of the given shape with
 * hipEvents on the context's stream; dtype LDC_F32 | LDC_BF16; ups = 1 folds nearest x2 upsampling. */
int ldc_conv_microbench(ldc_ctx* ctx, int dtype, int B, int L, int cin1, int cin2, int cout, int k, int stride, int ups,
                        int iters, double* ms_per_launch);
/* Self-check: the same layer and pseudo-random operands through the pipelined conv-GEMM with tile shape `tile_cfg` forced
 * (-1: the launcher's choice) and through the generic kernel; largest output difference, largest |output|, largest relative
 * difference of the fused GroupNorm statistics / column maxima.  tile_cfg >= 100 (round 6): both passes on the pipelined path with tile
 * shape tile_cfg - 100 (199: the launcher's choice) -- conv_fast_kernel, then conv_lean_kernel (csrc/conv_lean.inc): the outputs must be bit-identical
 * (max_abs_diff == 0; pass with_gn = 0, the unfused statistics stay on conv_fast_kernel). */
int ldc_conv_compare(ldc_ctx* ctx, int dtype, int B, int L, int cin1, int cin2, int cout, int k, int stride, int ups, int tile_cfg,
                     int with_gn, int with_colmax, int with_residual, double* max_abs_diff, double* max_abs_ref, double* max_rel_stat);
/* Self-check of the LayerNorm folded into a 1x1 conv (the attention blocks' PreNorm in front of to_qkv) on rows x = dc + U(-1, 1):
 * against launch_ln_rows + a plain conv; max_abs_diff2[0]: the folded conv computing the row statistics itself, [1]: from per-row
 * (sum, centred M2) partials per 32-column block as the fused ResnetBlock conv in front leaves them. */
int ldc_ln_fold_compare(ldc_ctx* ctx, int dtype, int rows, int C, int n_out, double dc, double* max_abs_diff2, double* max_abs_ref);
/* Self-check of the fp8 x fp8 conv-GEMM (block-scaled fp8 MFMA): operands drawn on the e4m3 grid through that kernel and through
 * the bf16-activation x fp8-weight kernel, which then computes the same products exactly. */
int ldc_conv_compare_fp8(ldc_ctx* ctx, int B, int L, int cin1, int cin2, int cout, int k, int stride, int ups, int with_gn, int with_colmax,
                         double* max_abs_diff, double* max_abs_ref, double* max_rel_stat);
int ldc_profile_read(ldc_ctx* ctx, double* conv_ms_total, int64_t* conv_launches, double* conv_flops_total);
/* Per kernel class (LDC_CLASS_*) totals of the same profiling pass: event-timed milliseconds, launches, algorithmic
 * flops and algorithmic HBM bytes; arrays of n >= LDC_N_CLASSES entries (any may be NULL). */
enum {
  LDC_CLASS_OTHER = 0,       /* memsets, statistics fallbacks */
  LDC_CLASS_CONV = 1,        /* implicit-GEMM Conv1d (F.conv1d in unet.py) -- MFMA bound */
  LDC_CLASS_GN_APPLY = 2,    /* GroupNorm affine + scale/shift + SiLU (+ residual), unet.py:137-156 -- HBM bound */
  LDC_CLASS_LAYERNORM = 3,   /* channel LayerNorm (+ residual), unet.py:82-101 -- HBM bound */
  LDC_CLASS_LINATTN = 4,     /* LinearAttention core, unet.py:208-222 -- HBM bound */
  LDC_CLASS_ATTN_FULL = 5,   /* bottleneck Attention core, unet.py:234-246 -- latency bound */
  LDC_CLASS_ELEMENTWISE = 6, /* tanh before final_conv, unet.py:467 -- HBM bound */
  LDC_N_CLASSES = 7
};
int ldc_profile_read_classes(ldc_ctx* ctx, int n, double* ms, int64_t* launches, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* LADIFFCODEC_H_ */
