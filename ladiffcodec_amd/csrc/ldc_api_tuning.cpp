// ldc_api_tuning.cpp -- test and tuning entry points of the C ABI: the L1 primitives the parity tests drive (SConv1d,
// SConvTranspose1d, SLSTM on caller-supplied weights), the UNet step's cost model, the XCC census, host statistics, the device-side
// timeline and per-kernel stamps of the timed mode, per-launch event profiling, the conv / GroupNorm microbenchmarks and the
// pipelined-vs-generic conv self-checks.  Nothing here is on the decode path; the context lives in ldc_api.cpp (ldc_internal.h).
#include "ldc_internal.h"

// ------------------------------------------------------------------------------------------------
// L1 primitives for the parity tests
// ------------------------------------------------------------------------------------------------
static int round_up(int v, int m) { return (v + m - 1) / m * m; }

extern "C" int ldc_sconv1d(ldc_ctx* c, const float* x, int B, int Cin, int L, const float* w_host, const float* b_host,
                           int Cout, int k, int stride, int dilation, int causal, int pre_elu, float* y, void* stream) {
  if (!c || !x || !w_host || !y) return fail(LDC_E_INVALID, "null argument");
  if (B <= 0 || Cin <= 0 || L <= 0 || Cout <= 0 || k <= 0 || stride <= 0 || dilation <= 0) return fail(LDC_E_INVALID, "bad sizes");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = pick_stream(c, stream);
  const int cp = round_up(Cin, 16);
  std::vector<float> wp((size_t)Cout * cp * k, 0.f);
  for (int o = 0; o < Cout; ++o)
    for (int i = 0; i < Cin; ++i)
      for (int t = 0; t < k; ++t) wp[((size_t)o * cp + i) * k + t] = w_host[((size_t)o * Cin + i) * k + t];
  DevMem keep;
  std::swap(keep.ptrs, c->wmem.ptrs);   // make_conv allocates from c->wmem; give it a temporary pool
  ConvLayer ly;
  ConvSpec sp;
  sp.dt = DT_F32; sp.cin1 = cp; sp.cout = Cout; sp.k = k; sp.stride = stride; sp.dil = dilation;
  const int padding_total = (k - 1) * dilation - (stride - 1);
  sp.pad_left = causal ? padding_total : padding_total - padding_total / 2;
  sp.pad_mode = PAD_REFLECT; sp.pre_act = pre_elu ? ACT_ELU : ACT_NONE;
  int rc = make_conv(c, sp, wp.data(), b_host, &ly);
  std::swap(keep.ptrs, c->wmem.ptrs);   // `keep` now owns the temporaries and frees them on return
  LDCCHK(rc);
  const int Lout = conv_out_len(ly, L);
  if (L <= sp.pad_left) return fail(LDC_E_INVALID, "input shorter than the reflect padding is not supported (L=%d pad=%d)", L, sp.pad_left);
  void *xc = nullptr, *yc = nullptr;
  LDCCHK(keep.alloc(&xc, (size_t)B * L * cp * 4));
  LDCCHK(keep.alloc(&yc, (size_t)B * Lout * Cout * 4));
  HIPCHK(hipMemsetAsync(xc, 0, (size_t)B * L * cp * 4, s));
  // [B][Cin][L] -> rows [B*L][cp]: transpose into the first Cin columns
  {
    void* tmp = nullptr;
    LDCCHK(keep.alloc(&tmp, (size_t)B * L * Cin * 4));
    HIPCHK(launch_to_cl(DT_F32, x, tmp, B, Cin, L, nullptr, 0, 0.f, s));
    HIPCHK(hipMemcpy2DAsync(xc, (size_t)cp * 4, tmp, (size_t)Cin * 4, (size_t)Cin * 4, (size_t)B * L, hipMemcpyDeviceToDevice, s));
  }
  ConvCall cc;
  cc.B = B; cc.L_in = L; cc.L_rows = Lout; cc.x1 = xc; cc.y = yc; cc.y_ld = Cout;
  HIPCHK(launch_conv(ly, cc, s));
  HIPCHK(launch_from_cl(DT_F32, yc, y, B, Cout, Lout, nullptr, 0, 0.f, s));
  HIPCHK(hipStreamSynchronize(s));
  return LDC_OK;
}

extern "C" int ldc_sconvtr1d(ldc_ctx* c, const float* x, int B, int Cin, int L, const float* w_host, const float* b_host,
                             int Cout, int k, int stride, int causal, float* y, void* stream) {
  if (!c || !x || !w_host || !y) return fail(LDC_E_INVALID, "null argument");
  if (k != 2 * stride) return fail(LDC_E_INVALID, "only kernel_size == 2*stride transposed convs exist on the decode path");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = pick_stream(c, stream);
  const int cp = round_up(Cin, 16);
  std::vector<float> wp((size_t)cp * Cout * k, 0.f);
  memcpy(wp.data(), w_host, (size_t)Cin * Cout * k * 4);
  DevMem keep;
  std::swap(keep.ptrs, c->wmem.ptrs);
  ConvLayer ly;
  const int padding_total = k - stride;
  const int trim_left = causal ? 0 : padding_total - padding_total / 2;
  int rc = make_convtr(c, DT_F32, cp, Cout, stride, trim_left, ACT_NONE, wp.data(), b_host, &ly);
  std::swap(keep.ptrs, c->wmem.ptrs);
  LDCCHK(rc);
  const int Lout = L * stride;
  void *xc = nullptr, *yc = nullptr, *tmp = nullptr;
  LDCCHK(keep.alloc(&xc, (size_t)B * L * cp * 4));
  LDCCHK(keep.alloc(&yc, (size_t)B * Lout * Cout * 4));
  LDCCHK(keep.alloc(&tmp, (size_t)B * L * Cin * 4));
  HIPCHK(hipMemsetAsync(xc, 0, (size_t)B * L * cp * 4, s));
  HIPCHK(launch_to_cl(DT_F32, x, tmp, B, Cin, L, nullptr, 0, 0.f, s));
  HIPCHK(hipMemcpy2DAsync(xc, (size_t)cp * 4, tmp, (size_t)Cin * 4, (size_t)Cin * 4, (size_t)B * L, hipMemcpyDeviceToDevice, s));
  ConvCall cc;
  cc.B = B; cc.L_in = L; cc.L_rows = L + 1; cc.L_final = Lout; cc.x1 = xc; cc.y = yc; cc.y_ld = Cout;
  HIPCHK(launch_conv(ly, cc, s));
  HIPCHK(launch_from_cl(DT_F32, yc, y, B, Cout, Lout, nullptr, 0, 0.f, s));
  HIPCHK(hipStreamSynchronize(s));
  return LDC_OK;
}

extern "C" int ldc_slstm(ldc_ctx* c, const float* x, int B, int H, int T, const float* const* weights_host, int layers,
                         float* y, void* stream) {
  if (!c || !x || !weights_host || !y) return fail(LDC_E_INVALID, "null argument");
  if (H % 16) return fail(LDC_E_INVALID, "H must be a multiple of 16");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = pick_stream(c, stream);
  // stage the weights as a throw-away state dict so that build_lstm can be reused
  ldc_ctx tmpc;
  tmpc.cfg = c->cfg;
  tmpc.device = c->device;
  tmpc.own_stream = c->own_stream;
  for (int n = 0; n < layers; ++n) {
    const char* names[4] = {"weight_ih_l", "weight_hh_l", "bias_ih_l", "bias_hh_l"};
    for (int j = 0; j < 4; ++j) {
      HostTensor t;
      if (j < 2) t.shape = {4 * H, H};
      else t.shape = {4 * H};
      t.data.assign(weights_host[4 * n + j], weights_host[4 * n + j] + t.numel());
      tmpc.raw[0][std::string("p.lstm.") + names[j] + std::to_string(n)] = std::move(t);
    }
  }
  WeightReader wr{&tmpc, 0, ""};
  SeaOp op;
  op.kind = SeaOp::LSTM; op.cin = op.cout = H;
  int rc = build_lstm(&tmpc, wr, "p", H, layers, &op.lstm);
  tmpc.own_stream = nullptr;
  LDCCHK(rc);
  if (!wr.missing.empty()) return fail(LDC_E_INVALID, "internal: %s", wr.missing.c_str());
  std::vector<SeaOp> ops{op};
  int ret = with_scratch(c, s, [&](Arena& ar, bool dry) -> int {
    SeaRun R{c, &ar, s, dry, B};
    void* xc = ar.alloc((size_t)B * T * H * 4);
    if (!dry) HIPCHK(launch_to_cl(DT_F32, x, xc, B, H, T, nullptr, 0, 0.f, s));
    void* o = nullptr;
    int Lo = 0, C = 0;
    LDCCHK(run_seanet(R, ops, xc, T, &o, &Lo, &C));
    if (!dry) HIPCHK(launch_from_cl(DT_F32, o, y, B, H, T, nullptr, 0, 0.f, s));
    return LDC_OK;
  });
  hipError_t e = hipStreamSynchronize(s);
  if (ret != LDC_OK) return ret;
  if (e != hipSuccess) return fail(LDC_E_HIP, "sync failed: %s", hipGetErrorString(e));
  return LDC_OK;   // tmpc.wmem frees the temporaries
}

// ------------------------------------------------------------------------------------------------
// accounting / profiling
// ------------------------------------------------------------------------------------------------
extern "C" int ldc_unet_step_cost(ldc_ctx* c, int B, int L, double* flops, double* bytes) {
  LDCCHK(check_ready(c, LDC_MODEL_MAIN));
  const int F = L / std::max(1, upsample_factor(c));
  LDCCHK(check_unet_args(c, B, L, F));
  Plan tmp;
  Arena measure;
  LDCCHK(build_plan(c, &tmp, measure, B, L, F));
  if (flops) *flops = tmp.flops;
  // algorithmic bytes with perfect intra-block fusion: every conv-boundary activation
  // read + written once, the weights once per step
  // algorithmic bytes of the conv-GEMM launches of one step: every conv reads its input(s) and packed weights once
  // and writes its output once
  if (bytes) *bytes = tmp.conv_bytes;
  return LDC_OK;
}

// Timeline of the TIMED mode.  rocprofv3's kernel trace serialises the batch parts' streams (measured: overlap factor
// 1.01 under the tracer against 1.5 untraced), so the evidence is taken on the device: the first and last kernel of every
// step of every batch part stamp a constant-rate clock.  Toggling drops the captured graphs (the stamp pointer is a
// kernel argument).
// Tuning aid: which XCCs (dies) the workgroups of a launch land on when its stream is created with a CU mask
// (hipExtStreamCreateWithCUMask): `wgs` workgroups each record HW_REG_XCC_ID; hist[x] = workgroups seen on XCC x.  mask == NULL: no mask.
__global__ void xcc_census_kernel(int* out, int spin) {
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11)) & 0xf);
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);   // keep workgroups resident so that the launch spreads
}
extern "C" int ldc_xcc_census(ldc_ctx* c, const uint32_t* mask, int n_words, int wgs, int* hist16) {
  if (!c || !hist16 || wgs < 1 || wgs > (1 << 16)) return fail(LDC_E_INVALID, "bad arguments");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = nullptr;
  if (mask && n_words > 0) HIPCHK(hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask));
  else HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  int* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, (size_t)wgs * sizeof(int)));
  HIPCHK(hipMemsetAsync(d, 0xff, (size_t)wgs * sizeof(int), st));
  hipLaunchKernelGGL(xcc_census_kernel, dim3(wgs), dim3(64), 0, st, d, 64);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  std::vector<int> h((size_t)wgs);
  HIPCHK(hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost));
  for (int i = 0; i < 16; ++i) hist16[i] = 0;
  for (int v : h) if (v >= 0 && v < 16) ++hist16[v];
  (void)hipFree(d);
  (void)hipStreamDestroy(st);
  return LDC_OK;
}

extern "C" int ldc_host_stats(ldc_ctx* c, int reset, double* graph_launch_ms, double* lookahead_wait_ms, int64_t* graph_launches) {
  if (!c) return fail(LDC_E_INVALID, "null context");
  if (graph_launch_ms) *graph_launch_ms = c->host_graph_ms;
  if (lookahead_wait_ms) *lookahead_wait_ms = c->host_wait_ms;
  if (graph_launches) *graph_launches = c->host_graph_launches;
  if (reset) { c->host_graph_ms = 0; c->host_wait_ms = 0; c->host_graph_launches = 0; }
  return LDC_OK;
}

// What the part-stream calibration measured (ldc_api.cpp: calibrate_part_streams): candidate streams that overlap with the caller's stream
// and each other / candidates, wall milliseconds of one 150 us spin alone and of the caller's stream plus all accepted streams spinning
// together (equal when they really run side by side), parts a batch is decoded as.  good = -1: no calibration has run.
extern "C" int ldc_stream_info(ldc_ctx* c, int* good, int* candidates, double* one_spin_ms, double* all_spin_ms, int* parts) {
  if (!c) return fail(LDC_E_INVALID, "null context");
  if (good) *good = c->calib ? c->calib_good : -1;
  if (candidates) *candidates = c->calib_cand;
  if (one_spin_ms) *one_spin_ms = c->calib_one_ms;
  if (all_spin_ms) *all_spin_ms = c->calib_all_ms;
  if (parts) *parts = c->split_batch;
  return LDC_OK;
}

// One sample of the device's two clocks behind everything queued on `stream`: out[0] = the 100 MHz wall clock, out[1] = s_memtime (shader
// cycles).  Two samples around a region give the shader clock the region ran at: (d out[1] / d out[0]) x 100 MHz.  Synchronises the stream.
extern "C" int ldc_clock_sample(ldc_ctx* c, uint64_t* out2, void* stream) {
  if (!c || !out2) return fail(LDC_E_INVALID, "bad arguments");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = stream ? (hipStream_t)stream : c->own_stream;
  unsigned long long* dev = nullptr;
  HIPCHK(hipMalloc((void**)&dev, 16));
  hipError_t e = launch_clock_sample(dev, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  unsigned long long h[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpy(h, dev, 16, hipMemcpyDeviceToHost);
  (void)hipFree(dev);
  HIPCHK(e);
  out2[0] = h[0]; out2[1] = h[1];
  return LDC_OK;
}

extern "C" int ldc_timeline_enable(ldc_ctx* c, int on) {
  if (!c) return fail(LDC_E_INVALID, "null ctx");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(counted_device_sync());
  for (auto& g : c->graphs) g.destroy();
  c->graphs.clear();
  if (on && !c->tl_buf) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, (size_t)kMaxParts * 2048 * 2 * sizeof(unsigned long long)));
    c->tl_buf = (unsigned long long*)p;
  }
  if (on) HIPCHK(hipMemset(c->tl_buf, 0, (size_t)kMaxParts * 2048 * 2 * sizeof(unsigned long long)));
  c->timeline = on != 0;
  return LDC_OK;
}

// Per-launch stamps of the pipelined conv kernel in the timed mode: plans are rebuilt with (or without) a stamp buffer.
extern "C" int ldc_kstamps_enable(ldc_ctx* c, int on) {
  if (!c) return fail(LDC_E_INVALID, "null ctx");
  HIPCHK(hipSetDevice(c->device));
  drop_plans(c);
  c->kstamps = on != 0;
  return LDC_OK;
}
// re-arm the stamp buffers of every cached plan (all-ones: both fields are kept as minima)
extern "C" int ldc_kstamps_reset(ldc_ctx* c) {
  if (!c) return fail(LDC_E_INVALID, "null ctx");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(counted_device_sync());
  for (auto& p : c->plans)
    if (p->kst) HIPCHK(hipMemset(p->kst, 0xff, (size_t)2048 * kKstOps * 2 * 8));
  return LDC_OK;
}
// Plan `idx` (in creation order: the batch parts of the last shape decoded).  ticks: [n_steps][n_ops][2] begin / end in 100 MHz ticks
// (0 / 0 where the op is not a pipelined conv); infos: n_ops strings of `info_cap` bytes (the op descriptions of the profile dump);
// classes: n_ops LDC_CLASS_* codes.  Returns the number of ops of a step through n_ops when ticks == NULL.
extern "C" int ldc_kstamps_read(ldc_ctx* c, int idx, int n_steps, int* n_ops, uint64_t* ticks, char* infos, int info_cap, int* classes) {
  if (!c || idx < 0 || idx >= (int)c->plans.size() || !n_ops) return fail(LDC_E_INVALID, "bad arguments");
  Plan* pl = c->plans[idx].get();
  if (!pl->kst) return fail(LDC_E_STATE, "ldc_kstamps_enable has not been called");
  const int nops = (int)std::min<size_t>(pl->step_ops.size(), kKstOps);
  *n_ops = nops;
  if (!ticks) return LDC_OK;
  if (n_steps < 1 || n_steps > 2048) return fail(LDC_E_INVALID, "n_steps must be in [1, 2048]");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(counted_device_sync());
  std::vector<unsigned long long> h((size_t)n_steps * kKstOps * 2);
  HIPCHK(hipMemcpy(h.data(), pl->kst, h.size() * 8, hipMemcpyDeviceToHost));
  for (int j = 0; j < n_steps; ++j)
    for (int o = 0; o < nops; ++o) {
      const unsigned long long b = h[((size_t)j * kKstOps + o) * 2], e = h[((size_t)j * kKstOps + o) * 2 + 1];
      const bool set = b != ~0ull && e != ~0ull && ~e >= b;
      ticks[((size_t)j * nops + o) * 2] = set ? b : 0;
      ticks[((size_t)j * nops + o) * 2 + 1] = set ? ~e : 0;
    }
  for (int o = 0; o < nops; ++o) {
    if (infos && info_cap > 0) snprintf(infos + (size_t)o * info_cap, info_cap, "%s", pl->step_info[o].c_str());
    if (classes) classes[o] = pl->step_class[o];
  }
  return LDC_OK;
}

// ticks[2 * j] / ticks[2 * j + 1]: begin / end of step j (iteration index of the last sampler call) of batch part `part`,
// in 100 MHz ticks; n <= 2048 steps
extern "C" int ldc_timeline_read(ldc_ctx* c, int part, int n, uint64_t* ticks) {
  if (!c || !ticks || part < 0 || part >= kMaxParts || n < 1 || n > 2048) return fail(LDC_E_INVALID, "bad arguments");
  if (!c->tl_buf) return fail(LDC_E_STATE, "ldc_timeline_enable has not been called");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(counted_device_sync());
  HIPCHK(hipMemcpy(ticks, c->tl_buf + (size_t)part * 2048 * 2, (size_t)n * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return LDC_OK;
}

extern "C" int ldc_profile_enable(ldc_ctx* c, int on) {
  if (!c) return fail(LDC_E_INVALID, "null ctx");
  c->profile = on != 0;
  if (on) {
    for (auto& e : c->prof_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    c->prof_events.clear();
    c->prof_event_flops.clear(); c->prof_event_class.clear(); c->prof_event_bytes.clear(); c->prof_event_info.clear();
    c->prof_ms = 0; c->prof_flops = 0; c->prof_launches = 0;
    for (int k = 0; k < LDC_N_CLASSES; ++k) { c->cls_ms[k] = c->cls_flops[k] = c->cls_bytes[k] = 0; c->cls_launches[k] = 0; }
  }
  return LDC_OK;
}

static int profile_collect(ldc_ctx* c) {
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(counted_device_sync());
  FILE* dump = getenv("LDC_PROFILE_DUMP") && !c->prof_events.empty() ? fopen(getenv("LDC_PROFILE_DUMP"), "a") : nullptr;
  for (size_t i = 0; i < c->prof_events.size(); ++i) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->prof_events[i].first, c->prof_events[i].second));
    const int k = c->prof_event_class[i];
    if (dump) fprintf(dump, "%d %.0f %.0f %.3f %s\n", k, c->prof_event_flops[i], c->prof_event_bytes[i], ms * 1e3, c->prof_event_info[i].c_str());
    c->cls_ms[k] += ms; c->cls_flops[k] += c->prof_event_flops[i]; c->cls_bytes[k] += c->prof_event_bytes[i]; c->cls_launches[k] += 1;
    if (k == LDC_CLASS_CONV) {
      c->prof_ms += ms;
      c->prof_flops += c->prof_event_flops[i];
      c->prof_launches += 1;
    }
    (void)hipEventDestroy(c->prof_events[i].first);
    (void)hipEventDestroy(c->prof_events[i].second);
  }
  if (dump) fclose(dump);
  c->prof_events.clear();
  c->prof_event_flops.clear(); c->prof_event_class.clear(); c->prof_event_bytes.clear(); c->prof_event_info.clear();
  return LDC_OK;
}

extern "C" int ldc_profile_read_classes(ldc_ctx* c, int n, double* ms, int64_t* launches, double* flops, double* bytes) {
  if (!c || n < 1) return fail(LDC_E_INVALID, "bad arguments");
  LDCCHK(profile_collect(c));
  for (int k = 0; k < n && k < LDC_N_CLASSES; ++k) {
    if (ms) ms[k] = c->cls_ms[k];
    if (launches) launches[k] = c->cls_launches[k];
    if (flops) flops[k] = c->cls_flops[k];
    if (bytes) bytes[k] = c->cls_bytes[k];
  }
  return LDC_OK;
}

extern "C" int ldc_profile_read(ldc_ctx* c, double* conv_ms_total, int64_t* conv_launches, double* conv_flops_total) {
  if (!c) return fail(LDC_E_INVALID, "null ctx");
  LDCCHK(profile_collect(c));
  if (conv_ms_total) *conv_ms_total = c->prof_ms;
  if (conv_launches) *conv_launches = c->prof_launches;
  if (conv_flops_total) *conv_flops_total = c->prof_flops;
  return LDC_OK;
}

namespace ldc { extern unsigned long long* g_conv_stamps; }

// uniform [-1, 1) values of the given dtype (host LCG, uploaded)
static int fill_random(void* dev, size_t n, int dt, unsigned seed) {
  const size_t es = dt_size(dt);
  std::vector<char> h(n * es);
  for (size_t i = 0; i < n; ++i) {
    seed = seed * 1664525u + 1013904223u;
    const float v = ((seed >> 8) * (1.0f / 8388608.0f)) - 1.0f;
    if (dt == DT_F32) {
      reinterpret_cast<float*>(h.data())[i] = v;
    } else {
      uint32_t u;
      memcpy(&u, &v, 4);
      reinterpret_cast<uint16_t*>(h.data())[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
  }
  HIPCHK(hipMemcpy(dev, h.data(), h.size(), hipMemcpyHostToDevice));
  return LDC_OK;
}


extern "C" int ldc_conv_microbench(ldc_ctx* c, int dtype, int B, int L, int cin1, int cin2, int cout, int k, int stride,
                                   int ups, int iters, double* ms_per_launch) {
  if (!c || !ms_per_launch || iters < 1) return fail(LDC_E_INVALID, "bad arguments");
  HIPCHK(hipSetDevice(c->device));
  const int dt = dtype == LDC_F32 ? DT_F32 : DT_BF16;
  const bool saved_w8 = c->w8;
  c->w8 = dtype == LDC_BF16_W8;
  const int cin = cin1 + cin2;
  std::vector<float> w((size_t)cout * cin * k), bias(cout, 0.1f);
  unsigned seed = 12345u;
  for (auto& v : w) { seed = seed * 1664525u + 1013904223u; v = ((seed >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.1f; }
  DevMem keep;
  std::swap(keep.ptrs, c->wmem.ptrs);
  ConvLayer ly;
  ConvSpec sp;
  sp.dt = dt; sp.cin1 = cin1; sp.cin2 = cin2; sp.cout = cout; sp.k = k; sp.stride = stride; sp.ups = ups;
  sp.pad_left = (k == 4 && stride == 2) ? 1 : (k - 1) / 2;
  int rc = make_conv(c, sp, w.data(), bias.data(), &ly);
  std::swap(keep.ptrs, c->wmem.ptrs);
  c->w8 = saved_w8;
  LDCCHK(rc);
  const int L_out = ups ? 2 * L : (stride == 2 ? (L + 2 * sp.pad_left - k) / 2 + 1 : L);
  const size_t es = dt_size(dt);
  void *x1 = nullptr, *x2 = nullptr, *y = nullptr;
  LDCCHK(keep.alloc(&x1, (size_t)B * L * cin1 * es));
  if (cin2) LDCCHK(keep.alloc(&x2, (size_t)B * L * cin2 * es));
  LDCCHK(keep.alloc(&y, (size_t)B * L_out * cout * es));
  // full-range pseudo-random operands: constant or zero fills let the chip clock ~15-20 % higher than real data does
  LDCCHK(fill_random(x1, (size_t)B * L * cin1, dt, 777u));
  if (cin2) LDCCHK(fill_random(x2, (size_t)B * L * cin2, dt, 778u));
  ConvCall cc;
  cc.B = B; cc.L_in = L; cc.L_rows = L_out; cc.x1 = x1; cc.x2 = x2; cc.y = y; cc.y_ld = cout;
  {   // split-K workspace as the plan builder provides it
    void *part = nullptr, *cnt = nullptr;
    LDCCHK(keep.alloc(&part, (size_t)(8 << 20) * 4));
    LDCCHK(keep.alloc(&cnt, 1024 * 4));
    HIPCHK(hipMemset(cnt, 0, 1024 * 4));
    cc.sk_part = (float*)part; cc.sk_part_cap = (long long)8 << 20; cc.sk_count = (unsigned*)cnt; cc.sk_count_cap = 1024;
    cc.tune = &c->tune;
  }
  if (getenv("LDC_MB_GN")) {   // tuning aid: the fused GroupNorm statistics of the UNet's block convs (8 groups) ride along
    void* gs = nullptr;
    LDCCHK(keep.alloc(&gs, (size_t)B * 8 * kGnPad * 4));
    HIPCHK(hipMemset(gs, 0, (size_t)B * 8 * kGnPad * 4));
  HIPCHK(hipDeviceSynchronize());
    cc.gn_sum = (float*)gs; cc.gn_groups = 8;
  }
  hipStream_t s = c->own_stream;
  // LDC_MB_GNEPI = 1 | 2 (round 6): the ResnetBlock form of the launch -- GroupNorm apply (8 groups, timestep scale / shift, SiLU) in the epilogue
  // behind the in-launch statistics exchange; 2 = with the residual add and the row statistics for a following PreNorm.  The granule
  // region is cleared in front of every launch (in the decode the step's first kernel does it): the memset is inside the timed loop.
  void* gpart = nullptr;
  size_t gpart_bytes = 0;
  if (const char* ge = getenv("LDC_MB_GNEPI")) {
    const int mode = atoi(ge);
    if (mode >= 1 && cout % 256 == 0 && k == 3) {
      const int mslots = (L_out + 63) / 64 + 1;
      gpart_bytes = (size_t)B * mslots * 4 * (cout / 32) * 16;
      LDCCHK(keep.alloc(&gpart, gpart_bytes));
      void *gam = nullptr, *bet = nullptr, *ss = nullptr, *res = nullptr, *rst = nullptr;
      LDCCHK(keep.alloc(&gam, (size_t)cout * 4)); LDCCHK(keep.alloc(&bet, (size_t)cout * 4)); LDCCHK(keep.alloc(&ss, (size_t)2 * cout * 4));
      LDCCHK(fill_random(gam, cout, DT_F32, 31u)); LDCCHK(fill_random(bet, cout, DT_F32, 32u)); LDCCHK(fill_random(ss, (size_t)2 * cout, DT_F32, 33u));
      cc.gn_part = gpart; cc.gn_mslots = mslots; cc.gn_groups = 8; cc.gn_gamma = (const float*)gam; cc.gn_beta = (const float*)bet; cc.gn_ss = (const float*)ss;
      cc.fail_flag = c->dev_flag_dev;
      if (mode >= 2) {
        LDCCHK(keep.alloc(&res, (size_t)B * L_out * cout * es));
        LDCCHK(fill_random(res, (size_t)B * L_out * cout, dt, 779u));
        LDCCHK(keep.alloc(&rst, (size_t)B * L_out * (cout / 32) * 8));
        cc.residual = res; cc.rowstat_out = (float*)rst; cc.gn_ss = nullptr;
      }
    }
  }
  auto one = [&]() -> int {
    if (gpart) HIPCHK(hipMemsetAsync(gpart, 0, gpart_bytes, s));
    HIPCHK(launch_conv(ly, cc, s));
    return LDC_OK;
  };
  for (int i = 0; i < 3; ++i) LDCCHK(one());
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) LDCCHK(one());
  HIPCHK(hipEventRecord(e1, s));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_launch = ms / iters;
  if (getenv("LDC_CONV_STAMPS")) {
    const int nblk = 1 << 16;
    void* st = nullptr;
    LDCCHK(keep.alloc(&st, (size_t)nblk * 8 * 8));
    HIPCHK(hipMemset(st, 0, (size_t)nblk * 8 * 8));
    ldc::g_conv_stamps = (unsigned long long*)st;
    if (gpart) HIPCHK(hipMemsetAsync(gpart, 0, gpart_bytes, s));
    hipError_t le = launch_conv(ly, cc, s);
    ldc::g_conv_stamps = nullptr;
    HIPCHK(le);
    HIPCHK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h((size_t)nblk * 8);
    HIPCHK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    double pro = 0, loop = 0, epi = 0, tab = 0, dma = 0, gath = 0, g_pub = 0, g_first = 0, g_polls = 0; int n = 0, n_g = 0;
    unsigned long long t_first = ~0ull, t_last = 0;
    int xcc_match = 0, xcc_hist[16] = {0}, xcc_of_class[8][16] = {};
    for (int b = 0; b < nblk; ++b) {
      if (!h[8 * b + 3]) continue;
      xcc_match += ((int)h[8 * b + 6] == (b & 7));
      ++xcc_hist[h[8 * b + 6] & 15];
      ++xcc_of_class[b & 7][h[8 * b + 6] & 15];
      t_first = std::min(t_first, h[8 * b]); t_last = std::max(t_last, h[8 * b + 3]);
      pro += (double)(h[8 * b + 1] - h[8 * b]); loop += (double)(h[8 * b + 2] - h[8 * b + 1]); epi += (double)(h[8 * b + 3] - h[8 * b + 2]);
      tab += (double)(h[8 * b + 4] - h[8 * b]); dma += (double)(h[8 * b + 5] - h[8 * b + 4]);
      if (h[8 * b + 7] >> 32) {   // (the lean kernel's fused GroupNorm epilogue: packed deltas from the loop's end, in units of 4 ticks)
        const unsigned long long w = h[8 * b + 7];
        g_pub += 4.0 * (double)(w & 0xffff); g_first += 4.0 * (double)((w >> 16) & 0xffff); gath += 4.0 * (double)((w >> 32) & 0xffff); g_polls += (double)(w >> 48);
        ++n_g;
      }
      ++n;
    }
    if (n) fprintf(stderr, "  XCC id == workgroup %% 8 for %d of %d workgroups; per-XCC counts %d %d %d %d %d %d %d %d\n", xcc_match, n, xcc_hist[0], xcc_hist[1],
                   xcc_hist[2], xcc_hist[3], xcc_hist[4], xcc_hist[5], xcc_hist[6], xcc_hist[7]);
    if (n) {
      fprintf(stderr, "  workgroup %% 8 -> XCC ids seen (count):");
      for (int cl = 0; cl < 8; ++cl) {
        fprintf(stderr, "  %d:", cl);
        for (int x = 0; x < 16; ++x) if (xcc_of_class[cl][x]) fprintf(stderr, " %d(%d)", x, xcc_of_class[cl][x]);
      }
      fprintf(stderr, "\n");
    }
    if (n) fprintf(stderr, "  stamps (s_memtime ticks per workgroup): blocks=%d prologue=%.1f (tile+copy tables %.1f, first copies issued %.1f, fragment tables %.1f) loop=%.1f epilogue=%.1f | first start -> last end %.0f\n",
                   n, pro / n, tab / n, dma / n, (pro - tab - dma) / n, loop / n, epi / n, (double)(t_last - t_first));
    if (n_g) fprintf(stderr, "  fused GroupNorm epilogue (wave 0): loop end -> published %.1f -> first poll back %.1f -> statistics gathered %.1f (%.2f polls), gathered -> stored %.1f ticks\n",
                     g_pub / n_g, g_first / n_g, gath / n_g, g_polls / n_g, epi / n - gath / n_g);
  }
  return LDC_OK;
}

// Self-check of the pipelined conv-GEMM: the same layer and pseudo-random operands through conv_fast.inc with tile shape
// `tile_cfg` forced (-1: the launcher's own choice) and through the generic kernel (conv_gemm.hip, itself pinned to the
// reference's SConv1d vectors); reports the largest output difference, the largest |reference output|, and the largest relative
// difference of the fused GroupNorm statistics (with_gn) / the fused column maxima (with_colmax).
extern "C" int ldc_conv_compare(ldc_ctx* c, int dtype, int B, int L, int cin1, int cin2, int cout, int k, int stride, int ups,
                                int tile_cfg, int with_gn, int with_colmax, int with_residual, double* max_abs_diff, double* max_abs_ref,
                                double* max_rel_stat) {
  if (!c || !max_abs_diff || !max_abs_ref || !max_rel_stat) return fail(LDC_E_INVALID, "bad arguments");
  HIPCHK(hipSetDevice(c->device));
  const int dt = dtype == LDC_F32 ? DT_F32 : DT_BF16;
  const bool saved_w8 = c->w8;
  c->w8 = dtype == LDC_BF16_W8;
  const int cin = cin1 + cin2;
  std::vector<float> w((size_t)cout * cin * k), bias(cout);
  unsigned seed = 4242u;
  for (auto& v : w) { seed = seed * 1664525u + 1013904223u; v = ((seed >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.2f; }
  for (auto& v : bias) { seed = seed * 1664525u + 1013904223u; v = ((seed >> 8) * (1.0f / 16777216.0f) - 0.5f); }
  DevMem keep;
  std::swap(keep.ptrs, c->wmem.ptrs);
  ConvLayer ly;
  ConvSpec sp;
  sp.dt = dt; sp.cin1 = cin1; sp.cin2 = cin2; sp.cout = cout; sp.k = k; sp.stride = stride; sp.ups = ups;
  sp.pad_left = (k == 4 && stride == 2) ? 1 : (k - 1) / 2;
  int rc = make_conv(c, sp, w.data(), bias.data(), &ly);
  std::swap(keep.ptrs, c->wmem.ptrs);
  c->w8 = saved_w8;
  LDCCHK(rc);
  const int L_out = ups ? 2 * L : (stride == 2 ? (L + 2 * sp.pad_left - k) / 2 + 1 : L);
  const size_t es = dt_size(dt);
  const int groups = 8;
  const size_t n_out = (size_t)B * L_out * cout, stat_n = (size_t)B * groups * kGnPad, cm_n = (size_t)B * cout;
  void *x1 = nullptr, *x2 = nullptr, *res = nullptr, *y[2] = {nullptr, nullptr}, *st[2] = {nullptr, nullptr}, *cm[2] = {nullptr, nullptr};
  LDCCHK(keep.alloc(&x1, (size_t)B * L * cin1 * es));
  LDCCHK(fill_random(x1, (size_t)B * L * cin1, dt, 901u));
  if (cin2) {
    LDCCHK(keep.alloc(&x2, (size_t)B * L * cin2 * es));
    LDCCHK(fill_random(x2, (size_t)B * L * cin2, dt, 902u));
  }
  if (with_residual) {
    LDCCHK(keep.alloc(&res, n_out * es));
    LDCCHK(fill_random(res, n_out, dt, 903u));
  }
  void *part = nullptr, *cnt = nullptr;
  LDCCHK(keep.alloc(&part, (size_t)(8 << 20) * 4));
  LDCCHK(keep.alloc(&cnt, 1024 * 4));
  HIPCHK(hipMemset(cnt, 0, 1024 * 4));
  hipStream_t s = c->own_stream;
  ConvTune tune = c->tune;
  for (int v = 0; v < 2; ++v) {
    LDCCHK(keep.alloc(&y[v], n_out * es));
    LDCCHK(keep.alloc(&st[v], stat_n * 4));
    LDCCHK(keep.alloc(&cm[v], cm_n * 4));
    HIPCHK(hipMemset(y[v], 0, n_out * es));
    HIPCHK(hipMemset(st[v], 0, stat_n * 4));
    HIPCHK(hipMemset(cm[v], 0, cm_n * 4));
    HIPCHK(hipDeviceSynchronize());   // (the memsets went to the null stream, which a non-blocking stream does not wait for)
    // tile_cfg >= 100 (round 6): BOTH passes on the pipelined path with tile shape tile_cfg - 100, pass 0 on conv_fast_kernel (lean off), pass 1
    // on conv_lean_kernel -- same tiles, same accumulation order: the caller expects max_abs_diff == 0
    // (+200 / +300: conv_fast_kernel twice / conv_lean_kernel twice -- run-to-run determinism of either)
    const bool lean_ab = tile_cfg >= 100;
    tune.force_generic = lean_ab ? 0 : (v == 0 ? 1 : 0);
    if (lean_ab) tune.lean = tile_cfg >= 300 ? 1 : (tile_cfg >= 200 ? 0 : v);
    tune.force_tile = lean_ab ? (tile_cfg % 100 == 99 ? -1 : tile_cfg % 100) : tile_cfg;
    ConvCall cc;
    cc.B = B; cc.L_in = L; cc.L_rows = L_out; cc.x1 = x1; cc.x2 = x2; cc.y = y[v]; cc.y_ld = cout; cc.residual = res;
    if (with_gn) { cc.gn_sum = (float*)st[v]; cc.gn_groups = groups; }
    if (with_colmax) { cc.colmax = (unsigned*)cm[v]; cc.colmax_lo = 0; cc.colmax_hi = cout; cc.colmax_stride = cout; }
    cc.sk_part = (float*)part; cc.sk_part_cap = (long long)8 << 20; cc.sk_count = (unsigned*)cnt; cc.sk_count_cap = 1024;
    cc.tune = &tune;
    HIPCHK(launch_conv(ly, cc, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  std::vector<char> h0(n_out * es), h1(n_out * es);
  HIPCHK(hipMemcpy(h0.data(), y[0], h0.size(), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(h1.data(), y[1], h1.size(), hipMemcpyDeviceToHost));
  auto val = [&](const std::vector<char>& h, size_t i) {
    if (dt == DT_F32) return (double)reinterpret_cast<const float*>(h.data())[i];
    const uint32_t u = (uint32_t)reinterpret_cast<const uint16_t*>(h.data())[i] << 16;
    float f;
    memcpy(&f, &u, 4);
    return (double)f;
  };
  double d = 0, m = 0;
  for (size_t i = 0; i < n_out; ++i) {
    const double a = val(h0, i), b = val(h1, i);
    if (!(b == b)) { d = 1e30; break; }
    d = std::max(d, fabs(a - b));
    m = std::max(m, fabs(a));
  }
  *max_abs_diff = d;
  *max_abs_ref = m;
  double rs = 0;
  if (with_gn) {
    std::vector<float> s0(stat_n), s1(stat_n);
    HIPCHK(hipMemcpy(s0.data(), st[0], stat_n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(s1.data(), st[1], stat_n * 4, hipMemcpyDeviceToHost));
    double smax = 0;
    for (size_t i = 0; i < stat_n; ++i) smax = std::max(smax, (double)fabsf(s0[i]));
    for (size_t i = 0; i < stat_n; ++i) rs = std::max(rs, fabs((double)s0[i] - s1[i]) / (smax + 1e-30));
  }
  if (with_colmax) {
    std::vector<unsigned> c0(cm_n), c1(cm_n);
    HIPCHK(hipMemcpy(c0.data(), cm[0], cm_n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(c1.data(), cm[1], cm_n * 4, hipMemcpyDeviceToHost));
    auto unkey = [](unsigned kx) { const unsigned b = (kx & 0x80000000u) ? (kx & 0x7fffffffu) : ~kx; float f; memcpy(&f, &b, 4); return (double)f; };
    for (size_t i = 0; i < cm_n; ++i) rs = std::max(rs, fabs(unkey(c0[i]) - unkey(c1[i])) / (m + 1e-30));
  }
  *max_rel_stat = rs;
  return LDC_OK;
}

// Self-check of the fp8 x fp8 conv (conv_fast_fp8.hip): operands drawn ON the e4m3 grid, so the bf16-activation x fp8-weight
// kernel (same quantised weights, expanded to bf16 in registers, bf16 MFMA) computes exactly the same products; the two results
// may differ by the fp32 summation order only (one bf16 ulp of the output at most).  Reports max |diff| and max |output|.
extern "C" int ldc_conv_compare_fp8(ldc_ctx* c, int B, int L, int cin1, int cin2, int cout, int k, int stride, int ups, int with_gn,
                                    int with_colmax, double* max_abs_diff, double* max_abs_ref, double* max_rel_stat) {
  if (!c || !max_abs_diff || !max_abs_ref || !max_rel_stat) return fail(LDC_E_INVALID, "bad arguments");
  if ((cin1 % 64) || (cin2 % 64)) return fail(LDC_E_INVALID, "fp8 inputs need channel counts that are multiples of 64");
  HIPCHK(hipSetDevice(c->device));
  const int cin = cin1 + cin2;
  std::vector<float> w((size_t)cout * cin * k), bias(cout);
  unsigned seed = 777u;
  for (auto& v : w) { seed = seed * 1664525u + 1013904223u; v = ((seed >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.2f; }
  for (auto& v : bias) { seed = seed * 1664525u + 1013904223u; v = ((seed >> 8) * (1.0f / 16777216.0f) - 0.5f); }
  DevMem keep;
  const bool saved_w8 = c->w8;
  c->w8 = true;
  std::swap(keep.ptrs, c->wmem.ptrs);
  ConvLayer ly[2];
  ConvSpec sp;
  sp.cin1 = cin1; sp.cin2 = cin2; sp.cout = cout; sp.k = k; sp.stride = stride; sp.ups = ups;
  sp.pad_left = (k == 4 && stride == 2) ? 1 : (k - 1) / 2;
  sp.dt = DT_BF16;
  int rc = make_conv(c, sp, w.data(), bias.data(), &ly[0]);            // bf16 activations x fp8 weights
  sp.dt = DT_FP8; sp.act8 = 1;
  if (rc == LDC_OK) rc = make_conv(c, sp, w.data(), bias.data(), &ly[1]);   // fp8 x fp8
  std::swap(keep.ptrs, c->wmem.ptrs);
  c->w8 = saved_w8;
  LDCCHK(rc);
  const int L_out = ups ? 2 * L : (stride == 2 ? (L + 2 * sp.pad_left - k) / 2 + 1 : L);
  const int groups = 8;
  const size_t n_out = (size_t)B * L_out * cout, stat_n = (size_t)B * groups * kGnPad, cm_n = (size_t)B * cout;
  // inputs: e4m3 codes with |value| <= 3.5 (exponent field <= 8), never the NaN code; the same values as bf16
  auto make_input = [&](size_t n, unsigned sd, void** d8, void** d16) -> int {
    std::vector<uint8_t> h8(n);
    std::vector<uint16_t> h16(n);
    for (size_t i = 0; i < n; ++i) {
      sd = sd * 1664525u + 1013904223u;
      uint8_t code = (uint8_t)(sd >> 13);
      if (((code >> 3) & 0xf) > 8) code = (uint8_t)((code & 0x87) | (8 << 3));
      h8[i] = code;
      const float f = host_e4m3_to_f32(code);
      uint32_t u;
      memcpy(&u, &f, 4);
      h16[i] = (uint16_t)(u >> 16);                       // every e4m3 value is exact in bf16
    }
    LDCCHK(keep.alloc(d8, n));
    LDCCHK(keep.alloc(d16, n * 2));
    HIPCHK(hipMemcpy(*d8, h8.data(), n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(*d16, h16.data(), n * 2, hipMemcpyHostToDevice));
    return LDC_OK;
  };
  void *x1[2] = {nullptr, nullptr}, *x2[2] = {nullptr, nullptr};
  LDCCHK(make_input((size_t)B * L * cin1, 31u, &x1[1], &x1[0]));
  if (cin2) LDCCHK(make_input((size_t)B * L * cin2, 32u, &x2[1], &x2[0]));
  void *part = nullptr, *cnt = nullptr;
  LDCCHK(keep.alloc(&part, (size_t)(8 << 20) * 4));
  LDCCHK(keep.alloc(&cnt, 1024 * 4));
  HIPCHK(hipMemset(cnt, 0, 1024 * 4));
  hipStream_t s = c->own_stream;
  void *y[2], *st[2], *cm[2];
  for (int v = 0; v < 2; ++v) {
    LDCCHK(keep.alloc(&y[v], n_out * 2));
    LDCCHK(keep.alloc(&st[v], stat_n * 4));
    LDCCHK(keep.alloc(&cm[v], cm_n * 4));
    HIPCHK(hipMemset(y[v], 0, n_out * 2));
    HIPCHK(hipMemset(st[v], 0, stat_n * 4));
    HIPCHK(hipMemset(cm[v], 0, cm_n * 4));
    HIPCHK(hipDeviceSynchronize());   // (the memsets went to the null stream, which a non-blocking stream does not wait for)
    ConvCall cc;
    cc.B = B; cc.L_in = L; cc.L_rows = L_out; cc.x1 = x1[v]; cc.x2 = x2[v]; cc.y = y[v]; cc.y_ld = cout;
    if (with_gn) { cc.gn_sum = (float*)st[v]; cc.gn_groups = groups; }
    if (with_colmax) { cc.colmax = (unsigned*)cm[v]; cc.colmax_lo = 0; cc.colmax_hi = cout; cc.colmax_stride = cout; }
    cc.sk_part = (float*)part; cc.sk_part_cap = (long long)8 << 20; cc.sk_count = (unsigned*)cnt; cc.sk_count_cap = 1024;
    cc.tune = &c->tune;
    HIPCHK(launch_conv(ly[v], cc, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  std::vector<uint16_t> h0(n_out), h1(n_out);
  HIPCHK(hipMemcpy(h0.data(), y[0], n_out * 2, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(h1.data(), y[1], n_out * 2, hipMemcpyDeviceToHost));
  auto val = [](uint16_t b) { const uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return (double)f; };
  double d = 0, m = 0;
  for (size_t i = 0; i < n_out; ++i) {
    const double a = val(h0[i]), b = val(h1[i]);
    if (!(b == b)) { d = 1e30; break; }
    d = std::max(d, fabs(a - b));
    m = std::max(m, fabs(a));
  }
  *max_abs_diff = d;
  *max_abs_ref = m;
  double rs = 0;
  if (with_gn) {
    std::vector<float> s0(stat_n), s1(stat_n);
    HIPCHK(hipMemcpy(s0.data(), st[0], stat_n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(s1.data(), st[1], stat_n * 4, hipMemcpyDeviceToHost));
    double smax = 0;
    for (size_t i = 0; i < stat_n; ++i) smax = std::max(smax, (double)fabsf(s0[i]));
    for (size_t i = 0; i < stat_n; ++i) rs = std::max(rs, fabs((double)s0[i] - s1[i]) / (smax + 1e-30));
  }
  if (with_colmax) {
    std::vector<unsigned> c0(cm_n), c1(cm_n);
    HIPCHK(hipMemcpy(c0.data(), cm[0], cm_n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(c1.data(), cm[1], cm_n * 4, hipMemcpyDeviceToHost));
    auto unkey = [](unsigned kx) { const unsigned b = (kx & 0x80000000u) ? (kx & 0x7fffffffu) : ~kx; float f; memcpy(&f, &b, 4); return (double)f; };
    for (size_t i = 0; i < cm_n; ++i) rs = std::max(rs, fabs(unkey(c0[i]) - unkey(c1[i])) / (m + 1e-30));
  }
  *max_rel_stat = rs;
  return LDC_OK;
}

extern "C" int ldc_gn_microbench(ldc_ctx* c, int dtype, int B, int L, int C, int with_residual, int iters, double* ms_per_launch) {
  if (!c || !ms_per_launch || iters < 1) return fail(LDC_E_INVALID, "bad arguments");
  HIPCHK(hipSetDevice(c->device));
  const int dt = dtype == LDC_BF16 ? DT_BF16 : DT_F32;
  const size_t es = dt_size(dt), n = (size_t)B * L * C;
  DevMem keep;
  void *x = nullptr, *y = nullptr, *r = nullptr, *st = nullptr, *gb = nullptr;
  LDCCHK(keep.alloc(&x, n * es)); LDCCHK(keep.alloc(&y, n * es)); LDCCHK(keep.alloc(&r, n * es));
  LDCCHK(keep.alloc(&st, (size_t)B * 8 * kGnPad * 4)); LDCCHK(keep.alloc(&gb, (size_t)4 * C * 4));
  HIPCHK(hipMemset(x, 0x3c, n * es)); HIPCHK(hipMemset(r, 0x3c, n * es));
  std::vector<float> hs((size_t)B * 8 * kGnPad, 1.0f), hg((size_t)4 * C, 0.5f);
  for (size_t i = 0; i < hs.size(); i += kGnPad) { hs[i] = 10.f; hs[i + 1] = 1e4f; }
  HIPCHK(hipMemcpy(st, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(gb, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
  float* g = (float*)gb;
  hipStream_t s = c->own_stream;
  void* yln = nullptr;
  const int mode = 0;   // (1: fused LayerNorm output, 2: separate ln_rows launch -- round-2 experiments)
  if (mode) LDCCHK(keep.alloc(&yln, n * es));
  auto go = [&]() {
    hipError_t e = launch_gn_apply(dt, x, y, with_residual ? r : nullptr, B, L, C, 8, (float*)st, g, g + C, g + 2 * C, 0, nullptr, ACT_SILU, s,
                                   mode == 1 ? yln : nullptr, g);
    if (e == hipSuccess && mode == 2) e = launch_ln_rows(dt, y, yln, nullptr, g, B * L, C, s);
    return e;
  };
  for (int i = 0; i < 3; ++i) HIPCHK(go());
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) HIPCHK(go());
  HIPCHK(hipEventRecord(e1, s));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *ms_per_launch = ms / iters;
  return LDC_OK;
}



// Self-check of the folded PreNorm LayerNorm (ConvLayer::ln_s, unet.py:82-101 in front of to_qkv) on rows with a DC offset:
// y_ref = conv1x1(LayerNorm(x) * g) through launch_ln_rows + a plain conv, against the LayerNorm-folded conv reading x itself
// (out[0]) and reading per-row (sum, centred M2) partials per 32-column block as the fused block2 conv leaves them (out[1]).
// x = dc + U(-1, 1).  ADVICE r4: the round-4 single-pass variance cancelled where |mean| >> std.
extern "C" int ldc_ln_fold_compare(ldc_ctx* c, int dtype, int rows, int C, int n_out, double dc, double* max_abs_diff2, double* max_abs_ref) {
  if (!c || !max_abs_diff2 || !max_abs_ref || rows < 1 || C < 32 || C % 32 || n_out < 1) return fail(LDC_E_INVALID, "bad arguments");
  HIPCHK(hipSetDevice(c->device));
  const int dt = dtype == LDC_F32 ? DT_F32 : DT_BF16;
  const size_t es = dt_size(dt);
  auto round_dt = [&](float v) {
    if (dt == DT_F32) return v;
    uint32_t u;
    memcpy(&u, &v, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    float r;
    memcpy(&r, &u, 4);
    return r;
  };
  unsigned seed = 777u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) * (1.0f / 8388608.0f)) - 1.0f; };
  std::vector<float> w((size_t)n_out * C), g(C), w2((size_t)n_out * C), sn(n_out), xh((size_t)rows * C), part((size_t)rows * (C / 32) * 2);
  for (auto& v : w) v = rnd() * 0.1f;
  for (auto& v : g) v = 1.0f + 0.25f * rnd();
  for (auto& v : xh) v = round_dt((float)dc + rnd());
  for (int n = 0; n < n_out; ++n) {
    double acc = 0;
    for (int k = 0; k < C; ++k) { w2[(size_t)n * C + k] = w[(size_t)n * C + k] * g[k]; acc += (double)round_dt(w2[(size_t)n * C + k]); }
    sn[n] = (float)acc;
  }
  for (int r = 0; r < rows; ++r)
    for (int b = 0; b < C / 32; ++b) {
      float s = 0.f, m2 = 0.f;
      for (int k = 0; k < 32; ++k) s += xh[(size_t)r * C + b * 32 + k];
      for (int k = 0; k < 32; ++k) { const float d = xh[(size_t)r * C + b * 32 + k] - s / 32.0f; m2 += d * d; }
      part[((size_t)r * (C / 32) + b) * 2] = s; part[((size_t)r * (C / 32) + b) * 2 + 1] = m2;
    }
  DevMem keep;
  const bool saved_w8 = c->w8;
  c->w8 = false;
  std::swap(keep.ptrs, c->wmem.ptrs);
  ConvLayer plain, folded;
  ConvSpec sp;
  sp.dt = dt; sp.cin1 = C; sp.cout = n_out; sp.k = 1;
  int rc = make_conv(c, sp, w.data(), nullptr, &plain);
  if (rc == LDC_OK) rc = make_conv(c, sp, w2.data(), nullptr, &folded);
  float *d_g = nullptr, *d_sn = nullptr, *d_part = nullptr;
  if (rc == LDC_OK) rc = c->wmem.upload(&d_g, g);
  if (rc == LDC_OK) rc = c->wmem.upload(&d_sn, sn);
  if (rc == LDC_OK) rc = c->wmem.upload(&d_part, part);
  std::swap(keep.ptrs, c->wmem.ptrs);
  c->w8 = saved_w8;
  LDCCHK(rc);
  folded.ln_s = d_sn;
  void *x = nullptr, *xn = nullptr, *y[3] = {nullptr, nullptr, nullptr};
  LDCCHK(keep.alloc(&x, (size_t)rows * C * es));
  LDCCHK(keep.alloc(&xn, (size_t)rows * C * es));
  {
    std::vector<char> hb((size_t)rows * C * es);
    for (size_t i = 0; i < xh.size(); ++i) {
      if (dt == DT_F32) reinterpret_cast<float*>(hb.data())[i] = xh[i];
      else { uint32_t u; memcpy(&u, &xh[i], 4); reinterpret_cast<uint16_t*>(hb.data())[i] = (uint16_t)(u >> 16); }
    }
    HIPCHK(hipMemcpy(x, hb.data(), hb.size(), hipMemcpyHostToDevice));
  }
  hipStream_t s = c->own_stream;
  for (int v = 0; v < 3; ++v) {
    LDCCHK(keep.alloc(&y[v], (size_t)rows * n_out * es));
    HIPCHK(hipMemset(y[v], 0, (size_t)rows * n_out * es));
    HIPCHK(hipDeviceSynchronize());
    ConvCall cc;
    cc.B = 1; cc.L_in = rows; cc.L_rows = rows; cc.y = y[v]; cc.y_ld = n_out; cc.tune = &c->tune;
    if (v == 0) {
      HIPCHK(launch_ln_rows(dt, x, xn, nullptr, d_g, rows, C, s));
      cc.x1 = xn;
      HIPCHK(launch_conv(plain, cc, s));
    } else {
      cc.x1 = x;
      cc.ln_rowstat = v == 2 ? d_part : nullptr;
      HIPCHK(launch_conv(folded, cc, s));
    }
  }
  HIPCHK(hipStreamSynchronize(s));
  std::vector<char> h[3];
  for (int v = 0; v < 3; ++v) { h[v].resize((size_t)rows * n_out * es); HIPCHK(hipMemcpy(h[v].data(), y[v], h[v].size(), hipMemcpyDeviceToHost)); }
  auto val = [&](const std::vector<char>& hv, size_t i) {
    if (dt == DT_F32) return (double)reinterpret_cast<const float*>(hv.data())[i];
    const uint32_t u = (uint32_t)reinterpret_cast<const uint16_t*>(hv.data())[i] << 16;
    float f;
    memcpy(&f, &u, 4);
    return (double)f;
  };
  double m = 0;
  max_abs_diff2[0] = max_abs_diff2[1] = 0;
  for (size_t i = 0; i < (size_t)rows * n_out; ++i) {
    const double a = val(h[0], i);
    m = std::max(m, fabs(a));
    for (int v = 1; v < 3; ++v) {
      const double b = val(h[v], i);
      max_abs_diff2[v - 1] = (b == b) ? std::max(max_abs_diff2[v - 1], fabs(a - b)) : 1e30;
    }
  }
  *max_abs_ref = m;
  return LDC_OK;
}
