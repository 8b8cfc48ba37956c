// ldc_internal.h -- what the translation units of the C ABI share: the context and plan structures, error handling, the few
// helpers of ldc_api.cpp (core: context, weights, codec stages, UNet plans, step graphs, samplers, ldc_decode) that
// ldc_api_ext.cpp (bit-stream, resampler and training entry points) and ldc_api_tuning.cpp (primitive KATs, cost / census / host
// statistics, timeline and per-kernel stamps, profiling, microbenchmarks, self-checks) call.  Host-side C++ only.
#pragma once
#include "../../include/ladiffcodec.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "ldc_kernels.h"

// Every device-wide synchronisation of the library goes through this counter (ldc_debug_sync_count): the steady state of the
// decode path -- plans built, graphs captured -- must issue none (tests/test_gpu_parity.py: test_warm_decode_never_waits_for_the_device)
extern long long g_device_syncs;
inline hipError_t counted_device_sync() {
  ++g_device_syncs;
  return hipDeviceSynchronize();
}

using namespace ldc;

// errors: one message per thread, read through ldc_last_error
int fail(int code, const char* fmt, ...);
#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) return fail(LDC_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define LDCCHK(expr)          \
  do {                        \
    int _r = (expr);          \
    if (_r != LDC_OK) return _r; \
  } while (0)

// ------------------------------------------------------------------------------------------------
// small utilities
// ------------------------------------------------------------------------------------------------
struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  bool used = false;
  size_t numel() const {
    size_t n = 1;
    for (auto s : shape) n *= (size_t)s;
    return n;
  }
};

struct Arena {   // bump allocator over one device buffer; base == nullptr measures only
  char* base = nullptr;
  size_t cap = 0, off = 0;
  void* alloc(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    void* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  }
};

struct DevMem {
  std::vector<void*> ptrs;
  ~DevMem() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  int alloc(void** out, size_t bytes) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(LDC_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    ptrs.push_back(p);
    *out = p;
    return LDC_OK;
  }
  template <typename T>
  int upload(T** out, const std::vector<T>& v) {
    void* p = nullptr;
    LDCCHK(alloc(&p, v.size() * sizeof(T)));
    if (!v.empty()) {
      hipError_t e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
      if (e != hipSuccess) return fail(LDC_E_HIP, "hipMemcpy H2D failed: %s", hipGetErrorString(e));
    }
    *out = reinterpret_cast<T*>(p);
    return LDC_OK;
  }
};

// ------------------------------------------------------------------------------------------------
// model descriptions
// ------------------------------------------------------------------------------------------------
struct LstmLayer {
  ConvLayer in_proj;          // k1 GEMM H -> 4H with bias b_ih + b_hh
  float* w_hh = nullptr;      // layout depends on H (see launch_lstm_layer)
  float* w_rm = nullptr;      // row-major copy for the cooperative kernel (H = 256 / 512)
};

struct SeaOp {
  enum Kind { CONV_CIN1, CONV, CONVTR, RES, LSTM } kind = CONV;
  ConvLayer conv;             // CONV / CONVTR / RES first conv (k3, pre-ELU)
  ConvLayer conv2;            // RES second conv (k1, pre-ELU, + shortcut residual)
  ConvLayer shortcut;         // RES shortcut (k1)
  std::vector<LstmLayer> lstm;
  float* w1 = nullptr;        // CONV_CIN1 [Cout][k]
  float* b1 = nullptr;
  int cin = 0, cout = 0, k = 1, stride = 1, hidden = 0;
};

struct Codec {
  std::vector<int> ratios;
  int hop = 1;
  std::vector<SeaOp> enc, dec;
  // RVQ
  int n_q_layers = 0, bins = 1024;
  float* codebooks = nullptr;   // [n_q][bins][D]
  float* cb_sqnorm = nullptr;   // [n_q][bins]
  bool present = false;
};

struct ResnetW {
  ConvLayer c1r;               // block1's conv with res_conv folded in (ConvLayer::wtaps; bf16 / f32 weights only)
  ConvLayer c1, c2, res;
  ConvLayer c2_f8;              // fp8-weight contexts: block2's conv with fp8 INPUTS too (w == null: not eligible)
  bool has_res = false;
  float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
  int cin1 = 0, cin2 = 0, cout = 0;
  int ss_off = 0;               // offset of this block's (scale|shift) in a table row
};
struct LinAttnW {
  float* norm_g = nullptr;
  ConvLayer qkv, out;
  ConvLayer qkv_f8;             // to_qkv with fp8 inputs (see ResnetW::c2_f8)
  ConvLayer qkv_ln;             // to_qkv with the PreNorm LayerNorm folded in (ConvLayer::ln_s; bf16 / f32 weights only)
  ConvLayer qkv_ctx;            // ... with its output columns ordered q | (k_h v_h) x heads for the context fold (ConvCall::qkv_ctx_ws; bf16 engine)
  float* out_g = nullptr;       // null for the bottleneck Attention
  int dim = 0;
};
struct LevelW {
  ResnetW b1, b2;
  LinAttnW attn;
  ConvLayer resample;
  int kind = 0;                 // 0 down (k4 s2 p1), 1 up (nearest x2 + k3 p1), 2 same (k3 p1)
  int cin = 0, cout = 0;
};
struct UnetW {
  int dim = 0, time_dim = 0, groups = 8, heads = 4, dim_head = 32, channels = 128, cond_channels = 128;
  std::vector<int> dims;
  ConvLayer init, final_conv;
  ConvLayer init_c, init_x;     // init_conv split by input: the condition's half (+ bias) is evaluated once per sampler call, x's half every step
  ConvLayer final_conv_f8;      // final_conv with fp8 inputs
  std::vector<LevelW> downs, ups;
  ResnetW mid1, mid2, fin;
  LinAttnW mid_attn;
  std::vector<ConvLayer> upsamplers;
  std::vector<int> up_ratios;
  float* ss_table = nullptr;    // [T][ss_stride] fp32
  float* cur_ss = nullptr;      // [ss_stride]: the row of the step being executed (launch_step_begin)
  int ss_stride = 0;
  int timesteps = 1000;
  double weight_elems = 0;
};

struct Plan {   // one UNet step for a fixed (sub-batch B, L, F); `slot` tells the two halves of a batch apart
  int B = 0, L = 0, F = 0, slot = 0;
  uint64_t last_use = 0;        // LRU tick (ldc_ctx::use_tick)
  long long sk_floats = 0;      // split-K workspace this plan's convs need (sized by a dry run of the launchers)
  long long sk_need_max = 0;
  unsigned long long* kst = nullptr;   // [2048 steps][kKstOps][2] timed-mode stamps of the step's conv launches (ldc_kstamps_enable)
  size_t part_bytes = 0;        // granule regions of the fused GroupNorm applies (sized by the same dry run)
  size_t part_need = 0;
  void* arena_base = nullptr;
  size_t arena_bytes = 0;
  void* zero_ptr = nullptr;     // the step's accumulators (GroupNorm sums, split-K counters, k-max keys, scale_x maxima): cleared by
  size_t zero_bytes = 0;        // launch_step_begin
  void* x_cl = nullptr;         // [B*L][channels]
  void* cond_in_cl = nullptr;   // [B*F][cond_channels]   (raw cond, channels-last)
  void* cond_cl = nullptr;      // [B*L][cond_channels]   (processed)
  void* eps_cl = nullptr;       // [B*L][channels]
  float* maxabs = nullptr;      // [B]
  int* step_state = nullptr;    // device int[2] {t, j} of THIS part: the parts of a batch advance independently
  float* cur_ss = nullptr;      // [ss_stride] timestep-MLP row of the step this part is executing
  std::vector<std::function<hipError_t(hipStream_t)>> cond_ops;   // process_cond (once per denoise)
  std::vector<std::function<hipError_t(hipStream_t)>> step_ops;   // Unet1D.forward after process_cond
  std::vector<int> step_is_conv;                                   // 1 where step_ops[i] is a conv-GEMM launch
  std::vector<int> step_where;                                     // 0 main stream, 1 side stream, 2 fork, 3 join
  std::vector<hipEvent_t> marker_events;                           // one event per fork/join marker (no re-use inside a capture)
  std::vector<double> step_flops;
  std::vector<int> step_class;                                     // LDC_CLASS_* of each op (profiling)
  std::vector<double> step_bytes;                                  // algorithmic HBM bytes of each op
  std::vector<std::string> step_info;                              // human-readable shape (profile dump)
  struct Tap { void* p; int C; int L; };
  std::map<std::string, Tap> taps;
  double flops = 0, act_bytes = 0, conv_bytes = 0;   // conv_bytes: inputs + outputs + packed weights of every conv-GEMM
  std::vector<std::pair<void*, size_t>> zero_once;   // regions cleared when the plan is created (the step state's epoch word)
};

// A batch is decoded as (up to) two independent halves on two streams: utterances do not interact inside
// the UNet, so the halves' kernels overlap and fill each other's tails and launch gaps.
static constexpr int kMaxParts = 4;
struct Halves {                // (the name dates from the two-way split; n parts, n <= kMaxParts)
  int n = 0;
  Plan* p[kMaxParts] = {nullptr, nullptr, nullptr, nullptr};
  int b0[kMaxParts] = {0, 0, 0, 0};   // first item of each part
};

static constexpr int kKstOps = 256;   // stamp slots per step (one per op of the step list)

struct StepGraph {   // per batch part: hipGraph of {step_begin, unet step, p_sample_update, step_advance}
  int B = 0, L = 0, F = 0, n = 0;
  const float* noise = nullptr;
  float* x = nullptr;
  hipStream_t stream = nullptr;
  hipGraphExec_t exec[4] = {nullptr, nullptr, nullptr, nullptr};
  hipGraphExec_t pexec[kMaxParts][2] = {};   // per-part single-stream graphs ([part][0] = K steps, [1] = one step)
  bool per_part = false;
  std::vector<hipEvent_t> part_ev;            // per-part mode: look-ahead events [part][depth]
  uint64_t last_use = 0;
  bool any() const { return exec[0] != nullptr; }
  void destroy() {   // exec[0] aliases pexec[0][0] in per-part mode
    if (per_part) exec[0] = nullptr;
    for (hipEvent_t e : part_ev) (void)hipEventDestroy(e);
    part_ev.clear();
    for (auto& e : exec) { if (e) (void)hipGraphExecDestroy(e); e = nullptr; }
    for (auto& pe : pexec) for (auto& e : pe) { if (e) (void)hipGraphExecDestroy(e); e = nullptr; }
    per_part = false;
  }
};

struct ldc_ctx {
  ldc_config cfg;
  int device = 0;
  int dt = DT_F32;              // UNet compute dtype; the codec (SEANet/LSTM/RVQ/upsampler) is always fp32
  bool w8 = false;              // LDC_BF16_W8: UNet conv weights stored as fp8 e4m3 + per-channel scale (activations bf16)
  bool finalized = false;
  std::map<std::string, HostTensor> raw[2];
  DevMem wmem;                  // weights, tables
  Codec codec[2];
  UnetW unet;
  StepTables sched{};
  // training-side schedule buffers (q_sample, loss weights; ddpm_loss.py:150-168)
  const float* sqrt_alphas_cumprod = nullptr;
  const float* sqrt_one_minus_alphas_cumprod = nullptr;
  const float* p2_loss_weight = nullptr;
  int* step_state = nullptr;    // device int[2]: t, j
  std::vector<std::unique_ptr<Plan>> plans;
  std::vector<StepGraph> graphs;
  Halves last_halves;
  hipStream_t aux_stream[kMaxParts] = {nullptr, nullptr, nullptr, nullptr};   // [0] unused
  hipEvent_t ev_fork = nullptr, ev_join[kMaxParts] = {nullptr, nullptr, nullptr, nullptr};
  static constexpr int kLstmChunks = 4;
  // second stage of the decoder LSTM's layer pipeline of batch part k: aux_stream[2 + k] (k < 2: idle whenever a batch has at most two parts).  NOT streams of
  // their own: HIP maps streams onto 4 hardware queues in creation order, and four more streams in front of aux_stream[1] put the two parts'
  // chains on ONE queue (measured: 143 -> 231 ms per decode)
  hipEvent_t lstm_ev[kMaxParts][kLstmChunks + 1] = {};
  hipStream_t calib_stream = nullptr; bool calibrated = false;   // the caller's stream the part streams were last chosen against (calibrate_part_streams)
  // what the last calibration measured (ldc_stream_info): candidates that overlap with the caller's stream and each other, candidates, the
  // wall time of one 150 us spin alone / of the caller's stream plus every accepted stream spinning together
  int calib_good = -1, calib_cand = 0; double calib_one_ms = 0, calib_all_ms = 0;
  // stream orders already measured, per caller stream (alternating between two caller streams re-ran the calibration -- host-blocking
  // synchronisations -- on every call)
  struct CalibOrder { hipStream_t s; std::vector<hipStream_t> order; int good, cand; double one_ms, all_ms; };
  std::vector<CalibOrder> calib_cache;
  int calib = 1;                // choose the part streams by measured overlap with the caller's stream (LDC_NO_STREAM_CALIB / LDC_AUX_FROM_SIDE turn it off)
  int ends_join = 0;            // ldc_decode joins the parts between front end, denoise loop and back end (LDC_ENDS_JOIN: the structure before the parts ran through)
  int lstm_pipe = 0;            // two-layer register LSTMs as a two-stage pipeline over time chunks (LDC_LSTM_PIPE / option "lstm_pipe"): parity-tested, measured
                                // 2.78 vs 2.63 ms per decoder pass of 16 on the default four hardware queues (the side stream shares one), 2.52 vs 2.62 with eight: off
  int split_batch = 2;
  int merge_advance = 1;        // the step state advances in the step's first kernel (0, LDC_STEP_ADVANCE_LAUNCH: a launch of its own behind p_sample_update)
  int rvq_tiled = 1;            // RVQ search on the LDS-tiled kernel (option "rvq_tiled"; 0: the round-1 kernel, same codes)
  int sea_splitk = 1;           // SEANet few-tile long-K convs split K on the generic kernel (LDC_NO_SEA_SPLITK / option "sea_splitk")
  int split_init = 1;           // init_conv's condition half hoisted out of the denoise loop (LDC_NO_SPLIT_INIT / option "split_init")
  int split_ends = 1;           // ldc_decode: the codec front / back ends per batch part on the parts' streams too (per-utterance normalisation only)
  int fp8_act = 1;              // LDC_FP8_ACT: in an fp8-weight context, tensors whose only consumer is a conv are produced in fp8 and
                                // that conv runs fp8 x fp8 on the block-scaled MFMA (0: bf16 activations x fp8 weights everywhere)
  int part_graphs = 1;          // one single-stream graph per batch part, replays interleaved behind a bounded look-ahead: recorded-AQL replay path,
                                // 3 ms of host time inside hipGraphLaunch per decode instead of 145 (LDC_PART_GRAPHS=0: one fork/join graph, node-by-node path)
  double host_graph_ms = 0, host_wait_ms = 0;   // host time inside hipGraphLaunch / waiting for the look-ahead window (ldc_host_stats)
  long long host_graph_launches = 0;
  hipStream_t side_stream[kMaxParts] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_side_fork[kMaxParts] = {nullptr, nullptr, nullptr, nullptr}, ev_side_join[kMaxParts] = {nullptr, nullptr, nullptr, nullptr};
  int side_streams = 0;
  int fuse_kmax = 1;
  int fuse_ln = 1;              // PreNorm LayerNorm written by the preceding ResnetBlock's last kernel
  int coop_launch = 0;          // LDC_COOP_LAUNCH: hipLaunchCooperativeKernel for the cooperative LSTM (see seanet.hip)
  int lstm_xcd = 1;             // LDC_LSTM_XCD / option "lstm_xcd": the cooperative LSTM of ONE or TWO items on sixteen 1024-thread workgroups of one XCD, the
                                // hidden-state exchange through that XCD's L2 (seanet.hip: lstm_xcd_kernel; configs[0]: 2.6 instead of 3.3 ms per clip); a
                                // device-side failure of it switches the context back to the placement-independent kernel
  int num_cus = 256;
  int xcd_resident[2] = {0, 0};   // teams of the XCD-local LSTM one XCD holds (H = 256, 512)
  int coop_resident[2] = {0, 0};   // [H == 512]: the cooperative LSTM's H/4 workgroups fit the device together (asked at ldc_create)
  int fuse_attn_tail = 1;       // LinearAttention out + to_out conv + LayerNorm + residual in one launch (bf16 engine)
  int fold_ctx = 1;             // round 6, option "fold_ctx": the LinearAttention context accumulated by to_qkv's own epilogue (lean kernel, bf16, folded PreNorm): no context launch, no k column max, k and v never written
  // Flow control of the step-graph replays: with more than ~10-20 multi-thousand-node graph launches outstanding the ROCm 7.2
  // runtime's enqueue path degrades (a decode queued behind a running one took 247 instead of 157 ms), so a replay waits on
  // the host until the replay `flow_depth` launches before it has finished (an event of this context, never a device sync)
  static constexpr int kFlowRing = 32;
  hipEvent_t flow_ev[kFlowRing] = {};
  unsigned long long flow_n = 0;
  int flow_depth = 8;           // LDC_FLOW_DEPTH (0 = unbounded look-ahead)
  int fuse_gn_stats = 1;
  int gn_epi_max_tiles = 1300;  // LDC_GN_EPI_MAXTILES: largest launch (output tiles) that fuses the GroupNorm apply.  A tile holds its slot until every tile
                                // of its item has finished: fine while a launch is one to two rounds of workgroups (c2: 300-1200 tiles), a loss when an
                                // item alone is 150 tiles of a 2400-tile launch (the L = 4800 layout: 501 instead of 450 ms per decode)
  int gn_epi_min_l = 0;         // LDC_GN_EPI_MINL: shortest level (positions per item) whose ResnetBlocks fuse the GroupNorm apply
  int fold_ln = 1;              // PreNorm LayerNorm of the attention blocks folded into to_qkv (LDC_NO_LN_FOLD / option "fold_ln")
                                // Off: measured 3 % slower than two launches (157.3 vs 152.2 ms per decode, profiles/r04_fusion_experiments.md)
  int fold_res = 1;             // res_conv as a second accumulator set of block1's conv (LDC_NO_RES_FOLD / option "fold_res")
  int fuse_gn_epi = 1;          // GroupNorm APPLY in the producing conv's epilogue behind an in-launch per-item wait (LDC_NO_GN_EPI / option "fuse_gn_epi")
  // knobs read from the environment once, at ldc_create (per context, not process-global)
  ConvTune tune;
  int lstm_stream_only = 0;     // LDC_LSTM_STREAM: never use the cooperative LSTM
  int serial_parts = 0;         // LDC_SERIAL: batch parts back to back, eager (diagnostics)
  int graph_steps = 0;          // option "graph_steps": denoise steps per replayed graph (0: by chain count, denoise_loop)
  // plan / graph cache (LRU): a corpus with many distinct lengths must not grow device memory without bound
  uint64_t use_tick = 0, call_tick = 0;
  size_t plan_bytes = 0, plan_bytes_cap = (size_t)48 << 30;   // LDC_PLAN_CACHE_GB
  int plan_count_cap = 24;                                     // LDC_PLAN_CACHE_N
  // device-drawn noise: every sampler call that draws advances the epoch, so no two calls share a realisation
  uint64_t noise_epoch = 0, cur_key = 0;
  // asynchronous device-side failure flag (cooperative LSTM timeout), host-mapped
  unsigned* dev_flag_host = nullptr;
  unsigned* dev_flag_dev = nullptr;
  int enc_final_act = ACT_NONE;
  // device-side timeline of the timed mode (ldc_timeline_enable): [kMaxParts][2048][begin, end] in 100 MHz ticks
  unsigned long long* tl_buf = nullptr;
  bool timeline = false;
  bool kstamps = false;         // ldc_kstamps_enable: per-launch device stamps of the pipelined conv kernel (plans carry the buffers)
  // scratch arena for codec stages and boundary buffers
  char* scratch = nullptr;
  size_t scratch_cap = 0;
  void* outnorm_ws = nullptr;
  size_t outnorm_ws_bytes = 0;
  float* state_buf = nullptr;   // diffusion state of ldc_denoise / ldc_p_sample_loop: a stable address keeps the step graphs valid
  size_t state_bytes = 0;
  hipStream_t own_stream = nullptr;
  // profiling
  bool profile = false;
  double prof_ms = 0, prof_flops = 0;
  int64_t prof_launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  std::vector<double> prof_event_flops;
  std::vector<int> prof_event_class;
  std::vector<double> prof_event_bytes;
  std::vector<std::string> prof_event_info;
  double cls_ms[LDC_N_CLASSES] = {}, cls_flops[LDC_N_CLASSES] = {}, cls_bytes[LDC_N_CLASSES] = {};
  int64_t cls_launches[LDC_N_CLASSES] = {};
};

inline hipStream_t pick_stream(ldc_ctx* c, void* s) { return s ? reinterpret_cast<hipStream_t>(s) : c->own_stream; }
// A kernel that gave up (bounded spin of the cooperative LSTM) raises a host-mapped flag; it is reported by the first
// API call that sees it: synchronous calls (stream == NULL) see their own failures, asynchronous ones the previous call's.
inline int check_dev_flag(ldc_ctx* c) {
  const unsigned v = c->dev_flag_host ? *reinterpret_cast<volatile unsigned*>(c->dev_flag_host) : 0u;
  if (v) {
    *reinterpret_cast<volatile unsigned*>(c->dev_flag_host) = 0u;
    // the bracketed tag is what callers key their fallback on (sample.py: decode_with_retry)
    if (v == 2)
      return fail(LDC_E_HIP, "device-side failure [gn_wait]: the in-launch GroupNorm exchange of a fused conv timed out (its tiles were "
                             "not all resident in time: is the GPU shared with another process?); the outputs of that call are NaN; "
                             "ldc_set_option(ctx, \"fuse_gn_epi\", 0) restores the separate conv + gn_apply launches");
    c->lstm_xcd = 0;   // (if it was the XCD-local form that gave up, the retry takes the placement-independent kernel)
    return fail(LDC_E_HIP, "device-side failure [coop_lstm]: cooperative LSTM: the hidden-state exchange timed out (its workgroups were "
                           "not co-resident: is the GPU shared with another process?); the outputs of that call are NaN; "
                           "ldc_set_option(ctx, \"lstm_stream\", 1) selects the streamed LSTM kernel");
  }
  return LDC_OK;
}
inline int finish_stream(ldc_ctx* c, void* s) {
  if (!s) {   // NULL stream => synchronous call on the context's own stream
    HIPCHK(hipStreamSynchronize(c->own_stream));
    return check_dev_flag(c);
  }
  return LDC_OK;
}
inline uint64_t next_noise_key(ldc_ctx* c, bool draws) {
  c->cur_key = c->cfg.noise_seed ^ (c->noise_epoch * 0x9E3779B97F4A7C15ull);
  if (draws) ++c->noise_epoch;
  return c->cur_key;
}

struct WeightReader {
  ldc_ctx* c;
  int which;
  std::string missing;
  HostTensor* get(const std::string& key, std::initializer_list<int64_t> shape) {
    auto& m = c->raw[which];
    auto it = m.find(key);
    if (it == m.end() && key.rfind("diff_model.", 0) == 0) it = m.find("diffusion.model." + key.substr(11));
    if (it == m.end()) {
      if (missing.size() < 600) missing += key + " ";
      return nullptr;
    }
    HostTensor& t = it->second;
    if (t.shape != std::vector<int64_t>(shape)) {
      if (missing.size() < 600) missing += key + "(shape) ";
      return nullptr;
    }
    t.used = true;
    return &t;
  }
};

struct ConvSpec {
  int dt = DT_F32;
  int cin1 = 0, cin2 = 0, cout = 0, k = 1, stride = 1, dil = 1, pad_left = 0, ups = 0, pad_mode = PAD_ZERO;
  int pre_act = ACT_NONE, post_act = ACT_NONE;
  int no_w8 = 0;               // keep this layer's weights in bf16 even in an fp8-weight context
  int act8 = 0;                // fp8 inputs as well (dt is then DT_FP8): the fp8 x fp8 MFMA path
};

struct SeaRun {   // measures or runs a SEANet stack
  ldc_ctx* c;
  Arena* ar;
  hipStream_t s;
  bool dry;
  int B;
  int side = -1;   // 0 / 1: aux_stream[2 + side] and lstm_ev[side] may carry the second stage of the two-layer LSTM pipeline (run_seanet)
  int teams = 1;   // batch parts whose codec ends may be in flight together: the XCD-local LSTM wants room for twice that many teams on its XCD
};

// ldc_api.cpp
int make_conv(ldc_ctx* c, const ConvSpec& sp, const float* w_oik, const float* bias, ConvLayer* out);
int make_convtr(ldc_ctx* c, int dt, int cin, int cout, int stride, int trim_left, int pre_act, const float* w_iok, const float* bias,
                ConvLayer* out);
int build_lstm(ldc_ctx* c, WeightReader& wr, const std::string& p, int H, int layers, std::vector<LstmLayer>* out);
void drop_plans(ldc_ctx* c);
int ensure_scratch(ldc_ctx* c, size_t bytes, hipStream_t s);
int conv_out_len(const ConvLayer& ly, int L);
int run_seanet(SeaRun& R, const std::vector<SeaOp>& ops, const void* x_in, int L, void** out, int* L_out, int* C_out);
int check_ready(ldc_ctx* c, int which, bool need_cond_codec = false);
int check_unet_args(ldc_ctx* c, int B, int L, int F);
int upsample_factor(const ldc_ctx* c);
int build_plan(ldc_ctx* c, Plan* pl, Arena& ar, int B, int L, int F);
int check_dev(ldc_ctx* c);   // hipSetDevice only: calls that need no weights

// runs `body` twice: once against a measuring arena, then (after sizing the scratch) for real
template <typename F>
inline int with_scratch(ldc_ctx* c, hipStream_t s, F body) {
  Arena measure;
  LDCCHK(body(measure, true));
  LDCCHK(ensure_scratch(c, measure.off + 4096, s));
  Arena real;
  real.base = c->scratch;
  real.cap = c->scratch_cap;
  return body(real, false);
}
