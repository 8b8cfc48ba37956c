// Internal kernel-launcher interface of libladiffcodec (gfx950 / CDNA4 only).
//
// Internal activation layout is channels-last: a tensor of B items, L positions, C channels is a
// row-major matrix [B*L][C] in the context's compute dtype (float or bf16).  This makes every conv
// an "NT" GEMM whose two MFMA operands are both K(channel)-contiguous, turns channel concatenation
// into a second input pointer, channel LayerNorm / RVQ / LSTM inputs into row operations, and
// nearest-upsampling / striding / causal reflect padding into row-index arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ldc {

enum { DT_F32 = 0, DT_BF16 = 1, DT_FP8 = 2 };   // DT_FP8: OCP e4m3 conv INPUTS of the fp8 x fp8 MFMA path (conv_fast_fp8.hip); outputs stay bf16
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_ELU = 2, ACT_TANH = 3, ACT_GELU = 4, ACT_SIGMOID = 5, ACT_RELU = 6 };
enum { PAD_ZERO = 0, PAD_REFLECT = 1 };
// GroupNorm statistics accumulators: every (item, group) pair owns a 64-byte line ([0] = sum, [1] = sum of squares).  Packed
// ([B][groups][2] floats) the 4800 fp32 atomics of one conv launch landed on 8 cache lines and were serialised by the L2
// (~10 ns each: +7.7 us per launch, measured on the same conv with and without fused statistics).
static constexpr int kGnPad = 16;

inline size_t dt_size(int dt) { return dt == DT_F32 ? 4 : (dt == DT_FP8 ? 1 : 2); }

// ------------------------------------------------------------------------------------------------
// conv_gemm.hip : implicit-GEMM Conv1d / ConvTranspose1d on MFMA
// ------------------------------------------------------------------------------------------------
// GEMM view: rows m = (item b, output position l), columns n = output channel, K = taps x Cin.
// Input row gathered for (l, tap):  u = l*stride + tap*dil - pad_left ;  row = u >> ups
//   (ups=1 folds nn.Upsample(scale 2, nearest) into the index), zero or reflect outside [0, L_in<<ups).
// Transposed conv (kernel 2s, stride s) is the 2-tap conv over q in [0, L_in] with N = s*Cout
// (phase-major); the epilogue scatters (q, phase) to position q*s + phase - trim_left.
struct ConvLayer {
  int dt = DT_F32;
  int cin1 = 0, cin2 = 0;     // channels of the two (concatenated) inputs; cin2 = 0 for one input
  int n = 0;                  // GEMM N actually stored (Cout, or s*Cout for transposed)
  int n_pad = 0;              // N padded to the tile
  int bn = 128;               // N tile: 32 | 64 | 128
  int taps = 1, stride = 1, dil = 1, pad_left = 0, ups = 0, pad_mode = PAD_ZERO;
  int pre_act = ACT_NONE, post_act = ACT_NONE;
  int tr_stride = 0, tr_cout = 0, tr_trim_left = 0;   // transposed-conv scatter (tr_stride = 0: plain)
  void* w = nullptr;          // packed [chunk][tap][n_pad][64 B of K]  (w8: [chunk][tap][n_pad][32 B], 8-byte slots permuted)
  float* bias = nullptr;      // [n] fp32 or null
  int w8 = 0;                 // weights are OCP fp8 e4m3 (dt must be DT_BF16): value = fp8 * wscale[n]
  float* wscale = nullptr;    // [n] fp32 per-output-channel scale (w8)
  double flops_per_row = 0;   // 2*K*N, for accounting
  // ResnetBlock.res_conv folded into block1's conv (unet.py:171,189-192: both read the same input): the packed image holds
  // wtaps = taps + 1 slabs per channel chunk, the extra one being the 1x1 res_conv weight; the pipelined kernel multiplies it
  // with the centre-tap window it has in LDS anyway into a second accumulator set and writes ConvCall::y2 (+ bias2).
  // PreNorm LayerNorm folded into a 1x1 conv (Residual(PreNorm(attention)).to_qkv, unet.py:82-101,208-246): the weight is packed
  // as W diag(g) and  conv(LN(x))[r][n] = rstd_r * (acc[r][n] - mean_r * ln_s[n]),  ln_s[n] = sum_c (W diag(g))[n][c] of the ROUNDED
  // packed weight; the pipelined kernel computes (mean_r, rstd_r) of its rows in its prologue and applies the identity to the
  // accumulators (one launch and one write + read of the normalised tensor less)
  float* ln_s = nullptr;
  int wtaps = 0;              // 0: no folded second conv
  float* bias2 = nullptr;     // [n] bias of the folded 1x1 conv
};

// tuning knobs of the conv launchers; owned by the context (read from the environment once at ldc_create)
struct ConvTune {
  int force_generic = 0;        // LDC_CONV_V1: every conv on the generic kernel
  int small_max = 100;          // LDC_CONV_SMALL_TILES: 64x64 tiles up to this many 128x128-equivalents (60 through round 5; with the lean kernel 80-150 measure 0.8-1.2 % faster per decode, 200+ slower: tools/sweep_knobs_r06b.sh)
  int splitk = 1;               // LDC_CONV_SPLITK: 0 off | 1 by layer | 2 | 3
  int sk_tiles = 200, sk_u2 = 24, sk_u3 = 60;   // LDC_SK_TILES / LDC_SK_U2 / LDC_SK_U3
  int m_fastest = 1;            // LDC_CONV_MFAST: 0 N-tile fastest | 1 by operand size | 2 M-tile fastest
  int debug = 0;                // LDC_CONV_DEBUG (bits, see conv_fast.inc)
  int gn_nap = 16, gn_nap0 = 0;  // LDC_GN_NAP / LDC_GN_NAP0: 64-clock naps between the polls of the fused GroupNorm exchange / before the first
  int force_tile = -1;          // self-check / tuning: 0 = 64x64, 1 = 128x64 tiles wherever the layer's N allows
  int xcd_order = 1;            // round 6: lean kernel's dispatch order as an xm x xn arrangement of the XCDs chosen by operand bytes (conv_lean.inc: lean_xcd_order); 0 = conv_fast_body's order; 2 / 4 / 8 = xn forced
  int lean = 1;                 // round 6: the instruction-diet kernel (conv_lean.inc) where its shapes allow; 0 = conv_fast_kernel everywhere (LDC_CONV_LEAN)
};

struct ConvCall {
  const void* x1 = nullptr;
  const void* x2 = nullptr;
  void* y = nullptr;
  void* y2 = nullptr;              // output of the folded 1x1 conv (ConvLayer::wtaps), [rows][n]
  const void* residual = nullptr;  // [rows][n], same dtype, added before post_act (plain conv only)
  int B = 0;
  int L_in = 0;               // positions per item of the input
  int L_rows = 0;             // GEMM rows per item (output positions; L_in+1 for transposed)
  int L_final = 0;            // transposed: final positions per item after trimming
  int y_ld = 0;               // channels per output row
  float* gn_sum = nullptr;    // optional fused GroupNorm statistics: [B][groups][2] (sum, sumsq), pre-zeroed
  int gn_groups = 0;
  unsigned* colmax = nullptr; // optional fused per-item column max, columns [colmax_lo, colmax_hi), pre-zeroed keys
  int colmax_lo = 0, colmax_hi = 0, colmax_stride = 0;
  // LinearAttention context inside to_qkv (round 6, lean kernel, bf16): the layer's output columns are ordered q | (k_h v_h) x heads, the
  // q tiles are stored, a (k_h | v_h) tile accumulates exp(k)^T v and the column sums of exp(k) into the item's workspace
  // (qkv_ctx_ws + item * qkv_ctx_stride floats: kmax keys [HD] (unused) | ksum [HD] | ctx [H][D][D], zeroed every step) and stores nothing
  float* qkv_ctx_ws = nullptr;
  int qkv_ctx_stride = 0;
  float* sk_part = nullptr;   // optional split-K workspace (fp32 partial tiles) and arrival counters (pre-zeroed)
  unsigned* sk_count = nullptr;
  long long sk_part_cap = 0;  // floats
  int sk_count_cap = 0;       // tiles
  int generic_split = 0;      // the generic kernel may split K through sk_part (conv_generic_splitk_floats floats; no counters: a reduce launch follows)
  // Fused GroupNorm apply in the conv epilogue (pipelined kernel only; needs gn_sum): y = SiLU(GN(conv) * (scale + 1) + shift)
  // (+ residual) (tanh).  gn_part: [B][gn_mslots][WM][n / 32] 16-byte granule pairs the waves exchange their partial statistics
  // through, zeroed before every launch (the step's first kernel does it); see ConvKArgs / epilogue_gn_fused in conv_device.h.
  void* gn_part = nullptr;
  int gn_mslots = 0;                // M-tile slots per item (>= (L_rows + BM - 1) / BM + 1 for the tile height the dry run reports)
  const float* gn_gamma = nullptr;
  const float* gn_beta = nullptr;
  const float* gn_ss = nullptr;     // [2 n] scale | shift of the current timestep, or null
  int gn_out = 0;                   // bit 2: tanh after the residual add
  unsigned* fail_flag = nullptr;    // host-mapped word raised when the bounded in-launch wait gives up
  float* rowstat_out = nullptr;        // fused apply with residual: per-row (sum, sumsq) partials of the output per 32-column block, [rows][n / 32][2]
  const float* ln_rowstat = nullptr;   // LayerNorm-folded conv (ConvLayer::ln_s): those partials of its input, or null (it reads its rows once more)
  unsigned long long* kst = nullptr;   // timed-mode stamps of this launch (ConvKArgs::kst), pipelined kernel only
  const int* kst_step = nullptr;
  int kst_stride = 0;
  const ConvTune* tune = nullptr;   // null: defaults
  long long* sk_need = nullptr;     // dry run: no launch, *sk_need = split-K workspace floats this call would use
  int* bm_out = nullptr;            // dry run (with sk_need): int[4] = rows per tile of the pipelined kernel (0 when the generic kernel would run), wave rows WM, split-K factor, columns per tile
};

hipError_t launch_conv(const ConvLayer& ly, const ConvCall& c, hipStream_t s);
long long conv_generic_splitk_floats(const ConvLayer& ly, const ConvCall& c);   // workspace floats a generic_split call of this shape uses (0: it would not split)
size_t conv_packed_weight_bytes(const ConvLayer& ly);
// host-side packers (fp32 [Cout][Cin][k] or, transposed, [Cin][Cout][k]) -> packed image in ly.dt
void pack_conv_weights(const ConvLayer& ly, const float* w_oik, void* dst_host);
// w8: also fills scales[n] (max |w| of the output channel / 448)
void pack_conv_weights_fp8(const ConvLayer& ly, const float* w_oik, void* dst_host, float* scales);
// ly.dt == DT_FP8 (fp8 inputs): [chunk of 64][tap][n_pad][64 B], fills scales[n]
void pack_conv_weights_fp8act(const ConvLayer& ly, const float* w_oik, void* dst_host, float* scales);
// OCP e4m3fn, round to nearest even, saturating at +-448 (host)
uint8_t host_f32_to_e4m3(float f);
float host_e4m3_to_f32(uint8_t v);
void pack_convtr_weights(const ConvLayer& ly, const float* w_iok, int cin, int cout, int stride, void* dst_host);
int conv_pick_bn(int n);

// ------------------------------------------------------------------------------------------------
// norm_act.hip
// ------------------------------------------------------------------------------------------------
// GroupNorm statistics of x [B][L][C]: stats[b][g] = (sum, sumsq) accumulated with fp32 atomics into a
// pre-zeroed buffer.
hipError_t launch_gn_stats(int dt, const void* x, int B, int L, int C, int groups, float* stats, hipStream_t s);
// y = act( GN(x)*(scale+1)+shift ) (+ residual).  scale_shift: fp32 [2*C] (scale then shift) selected
// by *t_ptr from a table with row stride ss_stride, or null.  eps 1e-5.
// out8 bit 0: y is written as OCP fp8 e4m3 ([rows][C] bytes, saturating) instead of dt; bit 1: the same for y_ln; bit 2: y = tanh(y)
// after the residual add (the final ResnetBlock, whose output feeds torch.tanh alone).  (The fp8 x fp8
// conv path: a tensor whose only consumer is a conv is produced in the conv's input type.)
hipError_t launch_gn_apply(int dt, const void* x, void* y, const void* residual, int B, int L, int C, int groups,
                           const float* stats, const float* gamma, const float* beta, const float* ss_table,
                           int ss_stride, const int* t_ptr, int act, hipStream_t s, void* y_ln = nullptr, const float* ln_g = nullptr, int out8 = 0);
// y_ln != null: also write channel-LayerNorm(y) * ln_g (needs gn_apply_ln_fusable(C) and ACT_SILU)
bool gn_apply_ln_fusable(int C);
// channel LayerNorm (gain only, biased var, eps 1e-5) per row; y = LN(x)*g (+ residual)
hipError_t launch_ln_rows(int dt, const void* x, void* y, const void* residual, const float* g, int rows, int C,
                          hipStream_t s, int out8 = 0);
// elementwise tanh in place / out of place
hipError_t launch_act(int dt, const void* x, void* y, int64_t n, int act, hipStream_t s, int out8 = 0);

// ------------------------------------------------------------------------------------------------
// attention.hip   (heads x dim_head = 4 x 32 fixed by the reference, unet.py:195,225)
// ------------------------------------------------------------------------------------------------
// qkv [B*L][3*H*D] (q | k | v, head-major).  ctx_ws: fp32 [B][H][D][D].
hipError_t launch_linattn(int dt, const void* qkv, void* out, float* ctx_ws, int B, int L, int heads, int dim_head,
                          bool kmax_fused, hipStream_t s);
size_t linattn_ws_floats_per_item(int heads, int dim_head);
// LinearAttention in three launches (bf16 engine): qkv conv (+ k column max) -> context -> tail (out + to_out conv + LayerNorm + residual)
bool linattn_tail_supported(int dt, int heads, int dim_head, int C);
hipError_t launch_linattn_ctx(int dt, const void* qkv, float* ctx_ws, int B, int L, int heads, int dim_head, hipStream_t s);
hipError_t launch_linattn_tail(int dt, const void* qkv, const float* ctx_ws, const void* wo_packed, int n_pad, const float* bias, const float* gain,
                               const void* resid, void* out, int B, int L, int heads, int dim_head, int C, hipStream_t s);
hipError_t launch_attn_full(int dt, const void* qkv, void* out, int B, int L, int heads, int dim_head, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// diffusion.hip
// ------------------------------------------------------------------------------------------------
// [B][C][L] fp32 -> [B][L][C] dt  (optionally multiplied by 1/(maxabs[b or 0] + eps))
hipError_t launch_to_cl(int dt, const float* x_bcl, void* y_blc, int B, int C, int L, const float* maxabs,
                        int maxabs_per_item, float eps, hipStream_t s);
// [B][L][C] dt -> [B][C][L] fp32 (same optional scaling)
hipError_t launch_from_cl(int dt, const void* x_blc, float* y_bcl, int B, int C, int L, const float* maxabs,
                          int maxabs_per_item, float eps, hipStream_t s);
// maxabs[b] (or maxabs[0] when !per_item) = max |x| ; buffer must be pre-zeroed.  Works on raw element
// streams: n_per_item elements per item.
hipError_t launch_maxabs(int dt, const void* x, int B, int64_t n_per_item, int per_item, float* maxabs,
                         hipStream_t s);
// Step state on the device: st[0] = current t, st[1] = iteration index j, st[2..3] = the 64-bit Philox key of the
// current sampler call (seed mixed with the context's call counter on the host, written by launch_step_set: a
// captured step graph therefore draws fresh noise on every replayed call).
struct StepTables {            // device pointers, fp32 [T]
  const float* sqrt_recip_alphas_cumprod;
  const float* sqrt_recipm1_alphas_cumprod;
  const float* posterior_mean_coef1;
  const float* posterior_mean_coef2;
  const float* posterior_log_variance_clipped;
};
// x [B][C][L] fp32 in place; eps_cl [B][L][C] dt; noise [.. j ..][B][C][L] fp32 or null (Philox);
// also writes x_cl [B][L][C] dt (the next step's UNet input).  Reads t, j from st.
hipError_t launch_random_fill(float* x, int64_t n, int uniform, uint64_t seed, unsigned step, hipStream_t s);
hipError_t launch_axpby(float* x, const float* y, float a, float b, int64_t n, hipStream_t s);
hipError_t launch_p_sample_update(int dt, float* x, const void* eps_cl, const float* noise, int64_t noise_step_stride,
                                  void* x_cl, int B, int C, int L, StepTables tb, const int* st,
                                  uint64_t elem_base, hipStream_t s);
// x /= (maxabs[b or 0] + eps) in place on a raw element stream (n_per_item elements per item)
hipError_t launch_scale_by_maxabs(int dt, void* x, int B, int64_t n_per_item, const float* maxabs, int per_item,
                                  float eps, hipStream_t s);
// torchaudio-style polyphase sinc resampling; bank [nnew][2*width+orig] fp32 (device)
hipError_t launch_resample(const float* wav, int C, int64_t T, const float* bank, int orig, int nnew, int width, int64_t target, float* out,
                           hipStream_t s);
hipError_t launch_scale_copy(int dt, const void* x, void* y, int B, int64_t n_per_item, const float* maxabs, float eps,
                             hipStream_t s);
hipError_t launch_step_advance(int* st, unsigned long long* tl, hipStream_t s);      // t -= 1, j += 1 (tl: timeline slot or null); behind the LAST step of a loop only (launch_step_begin advances)
// cur[0..stride) = table[st[0]][0..stride): the current timestep's scale/shift row, so that consumers need no
// dependent load through the step counter
// also zeroes [zero, zero + zero_bytes) (rounded up to 16 bytes: the caller pads the region): the step's accumulators
// advance = 1: first move the step state on from the previous step (t - 1, iteration + 1), i.e. a loop starts from (t + 1, -1)
hipError_t launch_step_begin(const float* table, int stride, int* st, float* cur, unsigned long long* tl, hipStream_t s,
                             void* zero = nullptr, size_t zero_bytes = 0, int advance = 0);
hipError_t launch_step_set(int* st, int t, int j, uint64_t noise_key, hipStream_t s);
hipError_t launch_clock_sample(unsigned long long* out2, hipStream_t s);   // out2[0] = 100 MHz wall clock, out2[1] = s_memtime (shader cycles)
hipError_t launch_spin_us(unsigned us, hipStream_t s);   // one workgroup busy for `us` microseconds (stream-overlap calibration)
// output normalisation (sample.py:133-134); ws: double [B][2] + float [B] zeroed by the launcher
hipError_t launch_output_normalise(float* x, int B, int64_t n_per_item, int per_item, void* ws, hipStream_t s);
size_t output_normalise_ws_bytes(int B);

// ------------------------------------------------------------------------------------------------
// seanet.hip
// ------------------------------------------------------------------------------------------------
// First SEANet conv: Cin = 1, causal reflect pad.  x [B][L] fp32 -> y [B][L][Cout] dt.  w [Cout][k] fp32.
hipError_t launch_conv_cin1(int dt, const float* x, void* y, const float* w, const float* bias, int B, int L, int Cout,
                            int k, hipStream_t s);
// LSTM recurrence over T for one layer.  pre [B][T][4H] dt_pre (input GEMM + both biases), w_hh [4H][H]
// fp32, out [B][T][H]; if skip != null: out = h + skip (SLSTM skip, lstm.py:25-26).
// a stretch of time steps of one LSTM layer on the register kernel: rows are item * bs + t * ts (in rows of 4H / H values), the
// recurrent state [B][2H] (h | c) is read when t0 > 0 and always written
struct LstmSeq {
  int t0 = 0, t1 = 0;
  long long pre_bs = 0, pre_ts = 1, out_bs = 0, out_ts = 1, skip_bs = 0, skip_ts = 1;
  float* state = nullptr;
};
bool lstm_seq_supported(int H);
hipError_t launch_lstm_seq(int dt, const void* pre, const float* w_hh, void* out, const void* skip, int B, int H, const LstmSeq& q,
                           hipStream_t s);
hipError_t launch_lstm_layer(int dt, const void* pre, const float* w_hh, void* out, const void* skip, int B, int T,
                             int H, hipStream_t s);
// Cooperative weight-stationary variant for H = 256 / 512 (H/4 workgroups exchange h through `ws`); w_rm is the
// row-major [4H][H] matrix.
bool lstm_coop_eligible(int H);
size_t lstm_coop_ws_bytes(int H);
// host_flag: device-visible mapped word set to 1 when the bounded h-exchange spin times out (the output is then NaN).
// Launched cooperatively: returns hipErrorCooperativeLaunchTooLarge when the H/4 workgroups cannot be co-resident
// (the caller falls back to launch_lstm_layer).
bool lstm_coop_resident(int H);
// the XCD-local form (launch_lstm_coop with coop_launch == 2) fits one XCD
int conv_fused_gn_wgs_per_cu(int dt, bool w8);   // occupancy query: workgroups per CU of the kernels a fused-GroupNorm conv lands on
int lstm_xcd_resident(int H);   // teams of sixteen workgroups one XCD holds
// occupancy query x CU count (with a margin) >= the H/4 workgroups that must be co-resident
hipError_t launch_lstm_coop(int dt, const void* pre, const float* w_rm, void* out, const void* skip, int B, int T, int H,
                            void* ws, unsigned* host_flag, int coop_launch, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// rvq.hip
// ------------------------------------------------------------------------------------------------
// z_cl [B*F][D] fp32 rows; codebooks [n_q][bins][D] fp32 and their squared norms [n_q][bins].
// codes [n_q][B*F] int64 (nullable), quantized_cl [B*F][D] fp32.
hipError_t launch_rvq(const float* z_rows, int rows, int D, const float* codebooks, const float* cb_sqnorm, int bins,
                      int n_q, int64_t* codes, float* quantized_rows, hipStream_t s, int variant = 1);   // variant 0: the round-1 kernel (same codes, bit for bit)
hipError_t launch_rvq_decode(const int64_t* codes, int rows, int D, const float* codebooks, int bins, int n_q,
                             float* quantized_rows, hipStream_t s);
hipError_t launch_sqnorm_rows(const float* x, int rows, int D, float* out, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// train.hip : first slice of the training step (q_sample, L1 objective, Block forward / backward), fp32, [B, C, L]
// ------------------------------------------------------------------------------------------------
// (T = number of timesteps: device-side t is clamped to [0, T) before it indexes a schedule table)
hipError_t launch_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac, const float* sqrt_1mac, int B,
                           int64_t n_per_item, float* out, int T, hipStream_t s);
hipError_t launch_predict_x_start(const float* x_t, const float* eps, const int64_t* t, const float* sqrt_recip_ac, const float* sqrt_recipm1_ac,
                                  int B, int64_t n_per_item, float* out, int T, hipStream_t s);
// per item: clamp(-SD-SDR(est, tgt), min clip_min)  (ClippedSDR over asteroid's MultiSrcNegSDR("sdsdr"), one source)
hipError_t launch_neg_sdsdr(const float* est, const float* tgt, int B, int64_t n_per_item, float clip_min, float* per_item, hipStream_t s);
size_t l1_loss_ws_bytes(int B);
hipError_t launch_l1_loss(const float* pred, const float* target, const int64_t* t, const float* p2w, int B, int64_t n_per_item,
                          float* loss, float* grad, void* ws, int T, hipStream_t s);
size_t train_block_ws_floats(int B, int Cin, int Cout, int L, int groups);
hipError_t launch_train_pw_forward(const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int L, int pre_silu, float* y,
                                   hipStream_t s);
hipError_t launch_train_pw_backward(const float* dy, const float* x, const float* w, int B, int Cin, int Cout, int L, int pre_silu, float* dx,
                                    float* dw, float* db, hipStream_t s);
size_t train_linattn_ws_floats(int B, int H, int D, int N);
hipError_t launch_train_linattn_forward(const float* qkv, int B, int H, int D, int N, float* o, float* ws, hipStream_t s);
hipError_t launch_train_linattn_backward(const float* d_o, const float* qkv, int B, int H, int D, int N, float* ws, float* dqkv, hipStream_t s);
hipError_t launch_train_conv_forward(const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int Lin, int K, int S, int P,
                                     float* y, hipStream_t s);
hipError_t launch_train_conv_backward(const float* dy, const float* x, const float* w, int B, int Cin, int Cout, int Lin, int K, int S, int P,
                                      float* dx, float* dw, float* db, hipStream_t s);
hipError_t launch_train_upsample2(const float* in, int64_t rows, int L, int backward, float* out, hipStream_t s);
hipError_t launch_train_act(const float* x, const float* dy, int64_t n, int kind, float* out, hipStream_t s);
size_t train_attn_ws_floats(int B, int H, int N);
hipError_t launch_train_attn_forward(const float* qkv, int B, int H, int D, int N, float* out, float* ws, hipStream_t s);
hipError_t launch_train_attn_backward(const float* d_o, const float* qkv, int B, int H, int D, int N, float* ws, float* dqkv, hipStream_t s);
hipError_t launch_train_convtr_forward(const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int L, int r, float* y,
                                       hipStream_t s);
hipError_t launch_train_convtr_backward(const float* dy, const float* x, const float* w, int B, int Cin, int Cout, int L, int r, float* dx,
                                        float* dw, float* db, hipStream_t s);
hipError_t launch_train_maxscale(const float* x, const float* dy, int B, int64_t n_per_item, float* out, hipStream_t s);
extern int g_train_bf16;        // LDC_TRAIN_BF16 / option train_bf16: the GEMM shapes with the hi terms only (plain bf16 products, fp32 accumulate)
extern int g_train_fp32_mfma;   // LDC_TRAIN_FP32_MFMA: the round-2 exact-fp32 MFMA GEMMs (convmm_kernel) instead of the split-bf16 ones (train_mm3.hip)
// split-bf16 (3 x bf16 MFMA, fp32-class accuracy) GEMM shapes of a Conv1d under training: train_mm3.hip
hipError_t launch_mm3_forward(const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int Lin, int Lout, int K, int S, int P,
                              float* y, hipStream_t s);
// (bias: per output row = input channel; only the transposed-conv forward, which is this GEMM shape, passes one)
hipError_t launch_mm3_dx(const float* dy, const float* w, int B, int Cin, int Cout, int Lin, int Lout, int K, int S, int P, float* dx, hipStream_t s,
                         const float* bias = nullptr);
// (db: optional bias gradient [Cout] = sum over items and positions of dy, accumulated by the workgroups that stage dy anyway)
hipError_t launch_mm3_dw(const float* dy, const float* x, int B, int Cin, int Cout, int Lin, int Lout, int K, int S, int P, float* dw, hipStream_t s,
                         float* db = nullptr);
extern hipStream_t g_train_side_stream;   // the training side stream (train.hip; nullptr until first used): launches on it take their own workspaces
extern int g_train_dw_side;    // option train_dw_side: the weight-gradient GEMMs of the Blocks on a side stream (set by DiffusionTrainer around its backward pass)
hipStream_t train_side_stream();              // the side stream (created on first use; nullptr on failure)
hipError_t launch_train_join(hipStream_t s);   // `s` waits for everything the side stream holds (no-op when it was not used)
extern int g_train_valu;   // LDC_TRAIN_VALU: the training path's GEMM shapes on the VALU reference kernels instead of the fp32 MFMA ones
hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr, float b1, float b2, float eps, hipStream_t s);
hipError_t launch_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, int* step_dev, float lr, float b1, float b2, float eps, hipStream_t s);
hipError_t launch_train_ln_forward(const float* x, const float* g, int B, int C, int L, float* y, float* stats, hipStream_t s);
hipError_t launch_train_ln_backward(const float* dy, const float* x, const float* g, const float* stats, int B, int C, int L, float* dx,
                                    float* dg, hipStream_t s);
hipError_t launch_train_block_forward(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                                      const float* ss, int B, int Cin, int Cout, int L, int groups, float* y, float* ws, hipStream_t s);
hipError_t launch_train_block_backward(const float* dy, const float* x, const float* gamma, const float* beta, const float* ss, int B,
                                       int Cin, int Cout, int L, int groups, float* ws, float* dx, float* dw, float* db, float* dgamma,
                                       float* dbeta, float* dss, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// bitstream.hip : index packing + range coder (bit-exact with srcs/encodec/binary.py, srcs/quantization/ac.py)
// ------------------------------------------------------------------------------------------------
hipError_t launch_pack_codes(const int64_t* codes, int n_q, int B, int F, int bits, uint8_t* out, int64_t out_stride, hipStream_t s);
hipError_t launch_unpack_codes(const uint8_t* in, int64_t in_stride, int n_q, int B, int F, int bits, int64_t* codes, hipStream_t s);
hipError_t launch_build_cdf(const float* pdf, int rows, int card, int total_range_bits, float roundoff, int min_range, int* cdf,
                            hipStream_t s);
// mode 0: cdf table per (stream, step) [B*S][card]; mode 1: `period` static tables, symbol s uses table s % period
hipError_t launch_ac_encode(const int* symbols, const int* cdf, int B, int S, int card, int mode, int period, int total_range_bits,
                            uint8_t* out, int64_t out_stride, int64_t cap, int64_t* nbytes, hipStream_t s);
hipError_t launch_ac_decode(const uint8_t* in, int64_t in_stride, const int64_t* nbytes, const int* cdf, int B, int S, int card, int mode,
                            int period, int total_range_bits, int* symbols, int* status, hipStream_t s);

}  // namespace ldc
