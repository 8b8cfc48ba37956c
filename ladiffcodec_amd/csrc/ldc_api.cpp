// ldc_api.cpp -- context, weight folding/packing, execution plans, hipGraph capture and the C ABI
// declared in include/ladiffcodec.h.  Host-side C++ only; every kernel lives in the .hip files.
//
// Mapping to the reference (haiciyang/LaDiffCodec):
//   Ctx::codec[*]    <- DiffAudioRep.encoder / .decoder / .quantizer      (srcs/model.py:52-66)
//   Ctx::unet        <- DiffAudioRep.diff_model = Unet1D                  (srcs/model.py:74, modules/unet.py:250-469)
//   Ctx::sched       <- GaussianDiffusion1D registered buffers            (srcs/losses/ddpm_loss.py:138-168)
//   ldc_denoise      <- GaussianDiffusion1D.halfway_sampling              (srcs/losses/ddpm_loss.py:370-385)
#include "ldc_internal.h"

long long g_device_syncs = 0;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* ldc_last_error(void) { return g_err; }
extern "C" const char* ldc_version(void) { return "ladiffcodec-amd 0.1 (gfx950)"; }


// ------------------------------------------------------------------------------------------------
// weight access + folding
// ------------------------------------------------------------------------------------------------

// w = g * v / ||v||  per dim-0 slice  (torch weight_norm, dim=0; reference conv.py:27-30)
static std::vector<float> fold_weight_norm(const HostTensor& g, const HostTensor& v) {
  const size_t n0 = (size_t)v.shape[0], inner = v.numel() / n0;
  std::vector<float> w(v.numel());
  for (size_t i = 0; i < n0; ++i) {
    double s = 0;
    for (size_t j = 0; j < inner; ++j) s += (double)v.data[i * inner + j] * v.data[i * inner + j];
    const float nrm = (float)sqrt(s);
    const float sc = g.data[i] / nrm;
    for (size_t j = 0; j < inner; ++j) w[i * inner + j] = v.data[i * inner + j] * sc;
  }
  return w;
}

// weight standardisation, fp32 branch: (w - mean) * rsqrt(var + 1e-5), biased var per out channel (unet.py:73-78)
static std::vector<float> fold_weight_std(const HostTensor& wt) {
  const size_t n0 = (size_t)wt.shape[0], inner = wt.numel() / n0;
  std::vector<float> w(wt.numel());
  for (size_t i = 0; i < n0; ++i) {
    double s = 0;
    for (size_t j = 0; j < inner; ++j) s += wt.data[i * inner + j];
    const double mean = s / (double)inner;
    double v = 0;
    for (size_t j = 0; j < inner; ++j) {
      const double d = wt.data[i * inner + j] - mean;
      v += d * d;
    }
    const float fmean = (float)mean;
    const float rs = 1.0f / sqrtf((float)(v / (double)inner) + 1e-5f);
    for (size_t j = 0; j < inner; ++j) w[i * inner + j] = (wt.data[i * inner + j] - fmean) * rs;
  }
  return w;
}


// the column form of gn_apply (norm_act.hip) is the one that can write fp8
static bool gn_apply_fp8_ok(int C) { return C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0 && C <= 2048; }

int make_conv(ldc_ctx* c, const ConvSpec& sp, const float* w_oik, const float* bias, ConvLayer* out) {
  ConvLayer ly;
  ly.dt = sp.dt;
  ly.cin1 = sp.cin1; ly.cin2 = sp.cin2;
  const int bke = 64 / (int)dt_size(sp.dt);
  if (sp.cin1 % bke || sp.cin2 % bke)
    return fail(LDC_E_INVALID, "conv input channels (%d,%d) must be multiples of %d for this dtype", sp.cin1, sp.cin2, bke);
  ly.n = sp.cout;
  ly.bn = conv_pick_bn(sp.cout);
  ly.n_pad = (sp.cout + ly.bn - 1) / ly.bn * ly.bn;
  ly.taps = sp.k; ly.stride = sp.stride; ly.dil = sp.dil; ly.pad_left = sp.pad_left; ly.ups = sp.ups;
  ly.pad_mode = sp.pad_mode; ly.pre_act = sp.pre_act; ly.post_act = sp.post_act;
  ly.flops_per_row = 2.0 * (sp.cin1 + sp.cin2) * sp.k * sp.cout;
  ly.w8 = (c->w8 && sp.dt == DT_BF16 && !sp.no_w8) ? 1 : 0;
  std::vector<char> packed(conv_packed_weight_bytes(ly));
  if (sp.dt == DT_FP8) {
    std::vector<float> scales((size_t)sp.cout);
    pack_conv_weights_fp8act(ly, w_oik, packed.data(), scales.data());
    LDCCHK(c->wmem.upload(&ly.wscale, scales));
  } else if (ly.w8) {
    std::vector<float> scales((size_t)sp.cout);
    pack_conv_weights_fp8(ly, w_oik, packed.data(), scales.data());
    LDCCHK(c->wmem.upload(&ly.wscale, scales));
  } else {
    pack_conv_weights(ly, w_oik, packed.data());
  }
  void* dw = nullptr;
  LDCCHK(c->wmem.alloc(&dw, packed.size()));
  HIPCHK(hipMemcpy(dw, packed.data(), packed.size(), hipMemcpyHostToDevice));
  ly.w = dw;
  if (bias) {
    std::vector<float> b(bias, bias + sp.cout);
    LDCCHK(c->wmem.upload(&ly.bias, b));
  }
  *out = ly;
  return LDC_OK;
}

// ConvTranspose1d(k = 2*stride) as a 2-tap conv over q with N = stride*Cout (see conv_gemm.hip)
int make_convtr(ldc_ctx* c, int dt, int cin, int cout, int stride, int trim_left, int pre_act, const float* w_iok,
                       const float* bias, ConvLayer* out) {
  ConvLayer ly;
  ly.dt = dt;
  ly.cin1 = cin; ly.cin2 = 0;
  const int bke = 64 / (int)dt_size(dt);
  if (cin % bke) return fail(LDC_E_INVALID, "convtr input channels %d must be a multiple of %d", cin, bke);
  ly.n = stride * cout;
  ly.bn = conv_pick_bn(ly.n);
  ly.n_pad = (ly.n + ly.bn - 1) / ly.bn * ly.bn;
  ly.taps = 2; ly.stride = 1; ly.dil = 1; ly.pad_left = 1; ly.ups = 0; ly.pad_mode = PAD_ZERO;
  ly.pre_act = pre_act; ly.post_act = ACT_NONE;
  ly.tr_stride = stride; ly.tr_cout = cout; ly.tr_trim_left = trim_left;
  ly.flops_per_row = 2.0 * cin * 2 * ly.n;
  std::vector<char> packed(conv_packed_weight_bytes(ly));
  pack_convtr_weights(ly, w_iok, cin, cout, stride, packed.data());
  void* dw = nullptr;
  LDCCHK(c->wmem.alloc(&dw, packed.size()));
  HIPCHK(hipMemcpy(dw, packed.data(), packed.size(), hipMemcpyHostToDevice));
  ly.w = dw;
  std::vector<float> b((size_t)ly.n, 0.f);
  if (bias)
    for (int p = 0; p < stride; ++p)
      for (int co = 0; co < cout; ++co) b[(size_t)p * cout + co] = bias[co];
  LDCCHK(c->wmem.upload(&ly.bias, b));
  *out = ly;
  return LDC_OK;
}

// ------------------------------------------------------------------------------------------------
// SEANet construction (reference srcs/modules/seanet.py:108-151, 200-244)
// ------------------------------------------------------------------------------------------------
static int build_wn_conv(ldc_ctx* c, WeightReader& wr, const std::string& p, int cin, int cout, int k, int stride, int dil,
                         int pre_act, ConvLayer* out) {
  HostTensor* b = wr.get(p + ".bias", {cout});
  HostTensor* g = wr.get(p + ".weight_g", {cout, 1, 1});
  HostTensor* v = wr.get(p + ".weight_v", {cout, cin, k});
  if (!b || !g || !v) return LDC_OK;   // reported later through wr.missing
  std::vector<float> w = fold_weight_norm(*g, *v);
  ConvSpec sp;
  sp.dt = DT_F32; sp.cin1 = cin; sp.cout = cout; sp.k = k; sp.stride = stride; sp.dil = dil;
  sp.pad_left = (k - 1) * dil - (stride - 1);   // causal: all padding on the left (conv.py:222-226)
  sp.pad_mode = PAD_REFLECT; sp.pre_act = pre_act;
  return make_conv(c, sp, w.data(), b->data.data(), out);
}

int build_lstm(ldc_ctx* c, WeightReader& wr, const std::string& p, int H, int layers, std::vector<LstmLayer>* out) {
  for (int n = 0; n < layers; ++n) {
    const std::string s = std::to_string(n);
    HostTensor* wih = wr.get(p + ".lstm.weight_ih_l" + s, {4 * H, H});
    HostTensor* whh = wr.get(p + ".lstm.weight_hh_l" + s, {4 * H, H});
    HostTensor* bih = wr.get(p + ".lstm.bias_ih_l" + s, {4 * H});
    HostTensor* bhh = wr.get(p + ".lstm.bias_hh_l" + s, {4 * H});
    if (!wih || !whh || !bih || !bhh) continue;
    LstmLayer L;
    std::vector<float> bias(4 * H);
    for (int i = 0; i < 4 * H; ++i) bias[i] = bih->data[i] + bhh->data[i];
    ConvSpec sp;
    sp.dt = DT_F32; sp.cin1 = H; sp.cout = 4 * H; sp.k = 1;
    LDCCHK(make_conv(c, sp, wih->data.data(), bias.data(), &L.in_proj));
    if (H == 64 || H == 128) {
      LDCCHK(c->wmem.upload(&L.w_hh, whh->data));
    } else {
      // k-major [H/4][4H][4] for the streaming kernel
      std::vector<float> km((size_t)4 * H * H);
      for (int row = 0; row < 4 * H; ++row)
        for (int k = 0; k < H; ++k) km[((size_t)(k / 4) * 4 * H + row) * 4 + (k % 4)] = whh->data[(size_t)row * H + k];
      LDCCHK(c->wmem.upload(&L.w_hh, km));
      if (lstm_coop_eligible(H)) LDCCHK(c->wmem.upload(&L.w_rm, whh->data));
    }
    out->push_back(L);
  }
  return LDC_OK;
}

static int build_res(ldc_ctx* c, WeightReader& wr, const std::string& p, int dim, int dil, SeaOp* op) {
  op->kind = SeaOp::RES;
  op->cin = op->cout = dim;
  op->hidden = dim / 2;
  LDCCHK(build_wn_conv(c, wr, p + ".block.1.conv.conv", dim, dim / 2, 3, 1, dil, ACT_ELU, &op->conv));
  LDCCHK(build_wn_conv(c, wr, p + ".block.3.conv.conv", dim / 2, dim, 1, 1, 1, ACT_ELU, &op->conv2));
  LDCCHK(build_wn_conv(c, wr, p + ".shortcut.conv.conv", dim, dim, 1, 1, 1, ACT_NONE, &op->shortcut));
  return LDC_OK;
}

static int build_codec(ldc_ctx* c, int which, const std::vector<int>& ratios, int n_q_layers, std::string* missing) {
  Codec& cd = c->codec[which];
  cd = Codec();
  cd.ratios = ratios;
  cd.hop = 1;
  for (int r : ratios) cd.hop *= r;
  const int nf = c->cfg.n_filters, D = c->cfg.rep_dims, nres = c->cfg.n_residual_layers, nl = c->cfg.lstm;
  WeightReader wr{c, which, ""};
  // ---- encoder ----
  {
    int idx = 0, mult = 1;
    SeaOp first;
    first.kind = SeaOp::CONV_CIN1; first.cin = 1; first.cout = nf; first.k = 7;
    {
      const std::string p = "encoder.model.0.conv.conv";
      HostTensor* b = wr.get(p + ".bias", {nf});
      HostTensor* g = wr.get(p + ".weight_g", {nf, 1, 1});
      HostTensor* v = wr.get(p + ".weight_v", {nf, 1, 7});
      if (b && g && v) {
        std::vector<float> w = fold_weight_norm(*g, *v);
        LDCCHK(c->wmem.upload(&first.w1, w));
        LDCCHK(c->wmem.upload(&first.b1, b->data));
      }
    }
    cd.enc.push_back(first);
    idx = 1;
    std::vector<int> rev(ratios.rbegin(), ratios.rend());
    for (int ratio : rev) {
      const int ch = mult * nf;
      for (int j = 0; j < nres; ++j) {
        SeaOp op;
        int dil = 1;
        for (int q = 0; q < j; ++q) dil *= 2;
        LDCCHK(build_res(c, wr, "encoder.model." + std::to_string(idx), ch, dil, &op));
        cd.enc.push_back(op);
        ++idx;
      }
      ++idx;   // ELU
      SeaOp op;
      op.kind = SeaOp::CONV; op.cin = ch; op.cout = 2 * ch; op.k = 2 * ratio; op.stride = ratio;
      LDCCHK(build_wn_conv(c, wr, "encoder.model." + std::to_string(idx) + ".conv.conv", ch, 2 * ch, 2 * ratio, ratio, 1,
                           ACT_ELU, &op.conv));
      cd.enc.push_back(op);
      ++idx;
      mult *= 2;
    }
    const int ch = mult * nf;
    bool pre = false;
    if (nl) {
      SeaOp op;
      op.kind = SeaOp::LSTM; op.cin = op.cout = ch;
      LDCCHK(build_lstm(c, wr, "encoder.model." + std::to_string(idx), ch, nl, &op.lstm));
      cd.enc.push_back(op);
      ++idx;
    }
    (void)pre;
    ++idx;   // ELU
    SeaOp op;
    op.kind = SeaOp::CONV; op.cin = ch; op.cout = D; op.k = 7;
    LDCCHK(build_wn_conv(c, wr, "encoder.model." + std::to_string(idx) + ".conv.conv", ch, D, 7, 1, 1, ACT_ELU, &op.conv));
    op.conv.post_act = c->enc_final_act;   // SEANetEncoder(final_activation=...) (seanet.py:144-149): both encoders get it
    cd.enc.push_back(op);
  }
  // ---- decoder ----
  {
    int idx = 0;
    int mult = 1 << ratios.size();
    SeaOp op0;
    op0.kind = SeaOp::CONV; op0.cin = D; op0.cout = mult * nf; op0.k = 7;
    LDCCHK(build_wn_conv(c, wr, "decoder.model.0.conv.conv", D, mult * nf, 7, 1, 1, ACT_NONE, &op0.conv));
    cd.dec.push_back(op0);
    idx = 1;
    if (nl) {
      SeaOp op;
      op.kind = SeaOp::LSTM; op.cin = op.cout = mult * nf;
      LDCCHK(build_lstm(c, wr, "decoder.model.1", mult * nf, nl, &op.lstm));
      cd.dec.push_back(op);
      idx = 2;
    }
    for (int ratio : ratios) {
      const int ch = mult * nf;
      ++idx;   // ELU
      SeaOp op;
      op.kind = SeaOp::CONVTR; op.cin = ch; op.cout = ch / 2; op.k = 2 * ratio; op.stride = ratio;
      {
        const std::string p = "decoder.model." + std::to_string(idx) + ".convtr.convtr";
        HostTensor* b = wr.get(p + ".bias", {ch / 2});
        HostTensor* g = wr.get(p + ".weight_g", {ch, 1, 1});
        HostTensor* v = wr.get(p + ".weight_v", {ch, ch / 2, 2 * ratio});
        if (b && g && v) {
          std::vector<float> w = fold_weight_norm(*g, *v);
          // causal: trim everything (k - s = s) on the right (conv.py:263-268) -> trim_left = 0
          LDCCHK(make_convtr(c, DT_F32, ch, ch / 2, ratio, 0, ACT_ELU, w.data(), b->data.data(), &op.conv));
        }
      }
      cd.dec.push_back(op);
      ++idx;
      for (int j = 0; j < nres; ++j) {
        SeaOp r;
        int dil = 1;
        for (int q = 0; q < j; ++q) dil *= 2;
        LDCCHK(build_res(c, wr, "decoder.model." + std::to_string(idx), ch / 2, dil, &r));
        cd.dec.push_back(r);
        ++idx;
      }
      mult /= 2;
    }
    ++idx;   // ELU
    SeaOp last;
    last.kind = SeaOp::CONV; last.cin = nf; last.cout = 1; last.k = 7;
    LDCCHK(build_wn_conv(c, wr, "decoder.model." + std::to_string(idx) + ".conv.conv", nf, 1, 7, 1, 1, ACT_ELU, &last.conv));
    cd.dec.push_back(last);
  }
  // ---- RVQ ----
  cd.n_q_layers = n_q_layers;
  if (n_q_layers > 0) {
    std::vector<float> all((size_t)n_q_layers * cd.bins * D);
    bool ok = true;
    for (int q = 0; q < n_q_layers; ++q) {
      const std::string p = "quantizer.vq.layers." + std::to_string(q) + "._codebook";
      HostTensor* e = wr.get(p + ".embed", {cd.bins, D});
      wr.get(p + ".inited", {1});
      wr.get(p + ".cluster_size", {cd.bins});
      wr.get(p + ".embed_avg", {cd.bins, D});
      if (!e) { ok = false; continue; }
      memcpy(all.data() + (size_t)q * cd.bins * D, e->data.data(), (size_t)cd.bins * D * sizeof(float));
    }
    if (ok) {
      LDCCHK(c->wmem.upload(&cd.codebooks, all));
      void* p = nullptr;
      LDCCHK(c->wmem.alloc(&p, (size_t)n_q_layers * cd.bins * sizeof(float)));
      cd.cb_sqnorm = reinterpret_cast<float*>(p);
      HIPCHK(launch_sqnorm_rows(cd.codebooks, n_q_layers * cd.bins, D, cd.cb_sqnorm, c->own_stream));
      HIPCHK(hipStreamSynchronize(c->own_stream));
    }
  }
  cd.present = true;
  *missing += wr.missing;
  return LDC_OK;
}

// ------------------------------------------------------------------------------------------------
// Unet1D construction (reference srcs/modules/unet.py:307-377)
// ------------------------------------------------------------------------------------------------
static int build_resnet(ldc_ctx* c, WeightReader& wr, const std::string& p, int cin1, int cin2, int cout, ResnetW* r) {
  const int cin = cin1 + cin2;
  r->cin1 = cin1; r->cin2 = cin2; r->cout = cout;
  HostTensor* w1 = wr.get(p + ".block1.proj.weight", {cout, cin, 3});
  HostTensor* b1 = wr.get(p + ".block1.proj.bias", {cout});
  HostTensor* g1 = wr.get(p + ".block1.norm.weight", {cout});
  HostTensor* be1 = wr.get(p + ".block1.norm.bias", {cout});
  HostTensor* w2 = wr.get(p + ".block2.proj.weight", {cout, cout, 3});
  HostTensor* b2 = wr.get(p + ".block2.proj.bias", {cout});
  HostTensor* g2 = wr.get(p + ".block2.norm.weight", {cout});
  HostTensor* be2 = wr.get(p + ".block2.norm.bias", {cout});
  wr.get(p + ".mlp.1.weight", {2 * cout, c->unet.time_dim});
  wr.get(p + ".mlp.1.bias", {2 * cout});
  r->has_res = cin != cout;
  HostTensor *wr_ = nullptr, *br_ = nullptr;
  if (r->has_res) {
    wr_ = wr.get(p + ".res_conv.weight", {cout, cin, 1});
    br_ = wr.get(p + ".res_conv.bias", {cout});
  }
  if (!w1 || !b1 || !g1 || !be1 || !w2 || !b2 || !g2 || !be2 || (r->has_res && (!wr_ || !br_))) return LDC_OK;
  ConvSpec sp;
  sp.dt = c->dt; sp.cin1 = cin1; sp.cin2 = cin2; sp.cout = cout; sp.k = 3; sp.pad_left = 1;
  {
    std::vector<float> w = fold_weight_std(*w1);
    LDCCHK(make_conv(c, sp, w.data(), b1->data.data(), &r->c1));
  }
  {
    std::vector<float> w = fold_weight_std(*w2);
    ConvSpec s2 = sp;
    s2.cin1 = cout; s2.cin2 = 0;
    LDCCHK(make_conv(c, s2, w.data(), b2->data.data(), &r->c2));
    if (c->w8 && c->fp8_act && c->dt == DT_BF16 && cout % 64 == 0 && gn_apply_fp8_ok(cout)) {
      s2.dt = DT_FP8; s2.act8 = 1;
      LDCCHK(make_conv(c, s2, w.data(), b2->data.data(), &r->c2_f8));
    }
  }
  if (r->has_res) {
    ConvSpec s3 = sp;
    s3.k = 1; s3.pad_left = 0;
    LDCCHK(make_conv(c, s3, wr_->data.data(), br_->data.data(), &r->res));
    if (!c->w8) {
      // res_conv folded into block1's conv (both read x, unet.py:171,189-192): a packed image with four slabs per channel
      // chunk -- the three taps of the standardised block1 weight and the 1x1 res_conv weight (ConvLayer::wtaps)
      const std::vector<float> w = fold_weight_std(*w1);
      std::vector<float> w4((size_t)cout * cin * 4);
      for (size_t oc = 0; oc < (size_t)cout * cin; ++oc) {
        w4[oc * 4 + 0] = w[oc * 3 + 0]; w4[oc * 4 + 1] = w[oc * 3 + 1]; w4[oc * 4 + 2] = w[oc * 3 + 2];
        w4[oc * 4 + 3] = wr_->data[oc];
      }
      ConvSpec s4 = sp;
      s4.k = 4;
      LDCCHK(make_conv(c, s4, w4.data(), b1->data.data(), &r->c1r));
      r->c1r.taps = 3; r->c1r.wtaps = 4; r->c1r.bias2 = r->res.bias;
      r->c1r.flops_per_row = r->c1.flops_per_row + r->res.flops_per_row;
    }
  }
  LDCCHK(c->wmem.upload(&r->g1, g1->data));
  LDCCHK(c->wmem.upload(&r->b1, be1->data));
  LDCCHK(c->wmem.upload(&r->g2, g2->data));
  LDCCHK(c->wmem.upload(&r->b2, be2->data));
  c->unet.weight_elems += (double)w1->numel() + w2->numel() + (wr_ ? wr_->numel() : 0);
  return LDC_OK;
}

static int build_attn(ldc_ctx* c, WeightReader& wr, const std::string& p, int dim, bool linear, LinAttnW* a) {
  const int hidden = c->unet.heads * c->unet.dim_head;
  a->dim = dim;
  HostTensor* ng = wr.get(p + ".fn.norm.g", {1, dim, 1});
  HostTensor* wq = wr.get(p + ".fn.fn.to_qkv.weight", {3 * hidden, dim, 1});
  HostTensor* wo = wr.get(p + (linear ? ".fn.fn.to_out.0.weight" : ".fn.fn.to_out.weight"), {dim, hidden, 1});
  HostTensor* bo = wr.get(p + (linear ? ".fn.fn.to_out.0.bias" : ".fn.fn.to_out.bias"), {dim});
  HostTensor* og = linear ? wr.get(p + ".fn.fn.to_out.1.g", {1, dim, 1}) : nullptr;
  if (!ng || !wq || !wo || !bo || (linear && !og)) return LDC_OK;
  ConvSpec sq;
  sq.dt = c->dt; sq.cin1 = dim; sq.cout = 3 * hidden; sq.k = 1;
  LDCCHK(make_conv(c, sq, wq->data.data(), nullptr, &a->qkv));
  if (c->w8 && c->fp8_act && c->dt == DT_BF16 && dim % 64 == 0 && dim <= 1024) {
    sq.dt = DT_FP8; sq.act8 = 1;
    LDCCHK(make_conv(c, sq, wq->data.data(), nullptr, &a->qkv_f8));
    sq.dt = c->dt; sq.act8 = 0;
  }
  if (!c->w8) {
    // PreNorm folded into to_qkv: W' = W diag(g); ln_s[n] = sum_c W'[n][c] of the weight AS THE MFMA SEES IT (bf16-rounded in the
    // bf16 engine), so that rstd * (W' x - mean * ln_s) equals W' ((x - mean) * rstd) up to fp32 rounding
    const int N3 = 3 * hidden;
    std::vector<float> w2((size_t)N3 * dim), sn((size_t)N3);
    auto as_packed = [&](float v) {
      if (c->dt != DT_BF16) return v;
      uint32_t u;
      memcpy(&u, &v, 4);
      u += 0x7fffu + ((u >> 16) & 1u);       // round to nearest even (finite weights)
      u &= 0xffff0000u;
      float r;
      memcpy(&r, &u, 4);
      return r;
    };
    for (int n = 0; n < N3; ++n) {
      double acc = 0;
      for (int k = 0; k < dim; ++k) {
        const float v = wq->data[(size_t)n * dim + k] * ng->data[k];
        w2[(size_t)n * dim + k] = v;
        acc += (double)as_packed(v);
      }
      sn[n] = (float)acc;
    }
    LDCCHK(make_conv(c, sq, w2.data(), nullptr, &a->qkv_ln));
    LDCCHK(c->wmem.upload(&a->qkv_ln.ln_s, sn));
    if (linear && c->dt == DT_BF16 && hidden == 128 && c->unet.dim_head == 32) {
      // the same layer with its output channels ordered q | (k_h v_h) x heads: a 64-column tile then holds k and v of ONE head (context fold)
      std::vector<float> wp((size_t)N3 * dim), sp((size_t)N3);
      for (int n = 0; n < N3; ++n) {
        int src = n;
        if (n >= hidden) {
          const int t = n - hidden, h = t / 64, w = t % 64;
          src = (w < 32 ? hidden : 2 * hidden) + h * 32 + (w & 31);
        }
        memcpy(&wp[(size_t)n * dim], &w2[(size_t)src * dim], (size_t)dim * sizeof(float));
        sp[n] = sn[src];
      }
      LDCCHK(make_conv(c, sq, wp.data(), nullptr, &a->qkv_ctx));
      LDCCHK(c->wmem.upload(&a->qkv_ctx.ln_s, sp));
    }
  }
  ConvSpec so;
  so.dt = c->dt; so.cin1 = hidden; so.cout = dim; so.k = 1;
  LDCCHK(make_conv(c, so, wo->data.data(), bo->data.data(), &a->out));
  LDCCHK(c->wmem.upload(&a->norm_g, ng->data));
  if (og) LDCCHK(c->wmem.upload(&a->out_g, og->data));
  c->unet.weight_elems += (double)wq->numel() + wo->numel();
  return LDC_OK;
}

static int build_plain_conv(ldc_ctx* c, WeightReader& wr, const std::string& p, int cin1, int cin2, int cout, int k,
                            int stride, int pad, int ups, ConvLayer* out) {
  HostTensor* w = wr.get(p + ".weight", {cout, cin1 + cin2, k});
  HostTensor* b = wr.get(p + ".bias", {cout});
  if (!w || !b) return LDC_OK;
  ConvSpec sp;
  sp.dt = c->dt; sp.cin1 = cin1; sp.cin2 = cin2; sp.cout = cout; sp.k = k; sp.stride = stride; sp.pad_left = pad; sp.ups = ups;
  c->unet.weight_elems += (double)w->numel();
  return make_conv(c, sp, w->data.data(), b->data.data(), out);
}

static int build_time_table(ldc_ctx* c, WeightReader& wr);

static int build_unet(ldc_ctx* c, std::string* missing) {
  UnetW& u = c->unet;
  u = UnetW();
  u.dim = c->cfg.diff_dims;
  u.time_dim = 4 * u.dim;
  u.channels = c->cfg.rep_dims;
  u.cond_channels = 128;
  static const int mults[5] = {1, 2, 2, 4, 4};   // DiffAudioRep fixes dim_mults=(1,2,2,4,4) (model.py:74)
  u.dims.push_back(u.dim);
  for (int m : mults) u.dims.push_back(u.dim * m);
  WeightReader wr{c, LDC_MODEL_MAIN, ""};
  const std::string P = "diff_model";
  LDCCHK(build_plain_conv(c, wr, P + ".init_conv", u.cond_channels, u.channels, u.dim, 7, 1, 3, 0, &u.init));
  {
    // init_conv(cat(cond, x)) = W_c * cond + b  +  W_x * x (unet.py:434): the processed condition does not change over the denoise steps, so its
    // half of the contraction is evaluated once per sampler call (init_c, with the bias) and every step runs only the x half (init_x) and adds it
    HostTensor* w = wr.get(P + ".init_conv.weight", {u.dim, u.cond_channels + u.channels, 7});
    HostTensor* b = wr.get(P + ".init_conv.bias", {u.dim});
    if (w && b) {
      const int Cc = u.cond_channels, Cx = u.channels, Ct = Cc + Cx;
      std::vector<float> wc((size_t)u.dim * Cc * 7), wx((size_t)u.dim * Cx * 7);
      for (int o = 0; o < u.dim; ++o) {
        for (int i = 0; i < Cc; ++i) for (int k = 0; k < 7; ++k) wc[((size_t)o * Cc + i) * 7 + k] = w->data[((size_t)o * Ct + i) * 7 + k];
        for (int i = 0; i < Cx; ++i) for (int k = 0; k < 7; ++k) wx[((size_t)o * Cx + i) * 7 + k] = w->data[((size_t)o * Ct + Cc + i) * 7 + k];
      }
      ConvSpec sc;
      sc.dt = c->dt; sc.cin1 = Cc; sc.cout = u.dim; sc.k = 7; sc.stride = 1; sc.pad_left = 3;
      LDCCHK(make_conv(c, sc, wc.data(), b->data.data(), &u.init_c));
      sc.cin1 = Cx;
      LDCCHK(make_conv(c, sc, wx.data(), nullptr, &u.init_x));
    }
  }
  const int nlev = (int)u.dims.size() - 1;
  int ss_off = 0;
  auto take_ss = [&](ResnetW& r) { r.ss_off = ss_off; ss_off += 2 * r.cout; };
  for (int i = 0; i < nlev; ++i) {
    const int din = u.dims[i], dout = u.dims[i + 1];
    const bool last = i >= nlev - 1;
    LevelW lv;
    const std::string p = P + ".downs." + std::to_string(i);
    LDCCHK(build_resnet(c, wr, p + ".0", din, 0, din, &lv.b1)); take_ss(lv.b1);
    LDCCHK(build_resnet(c, wr, p + ".1", din, 0, din, &lv.b2)); take_ss(lv.b2);
    LDCCHK(build_attn(c, wr, p + ".2", din, true, &lv.attn));
    lv.kind = last ? 2 : 0; lv.cin = din; lv.cout = dout;
    if (last) LDCCHK(build_plain_conv(c, wr, p + ".3", din, 0, dout, 3, 1, 1, 0, &lv.resample));
    else LDCCHK(build_plain_conv(c, wr, p + ".3", din, 0, dout, 4, 2, 1, 0, &lv.resample));
    u.downs.push_back(lv);
  }
  const int mid = u.dims.back();
  LDCCHK(build_resnet(c, wr, P + ".mid_block1", mid, 0, mid, &u.mid1)); take_ss(u.mid1);
  LDCCHK(build_attn(c, wr, P + ".mid_attn", mid, false, &u.mid_attn));
  LDCCHK(build_resnet(c, wr, P + ".mid_block2", mid, 0, mid, &u.mid2)); take_ss(u.mid2);
  for (int i = 0; i < nlev; ++i) {
    const int din = u.dims[nlev - 1 - i], dout = u.dims[nlev - i];
    const bool last = i == nlev - 1;
    LevelW lv;
    const std::string p = P + ".ups." + std::to_string(i);
    LDCCHK(build_resnet(c, wr, p + ".0", dout, din, dout, &lv.b1)); take_ss(lv.b1);
    LDCCHK(build_resnet(c, wr, p + ".1", dout, din, dout, &lv.b2)); take_ss(lv.b2);
    LDCCHK(build_attn(c, wr, p + ".2", dout, true, &lv.attn));
    lv.kind = last ? 2 : 1; lv.cin = dout; lv.cout = din;
    if (last) LDCCHK(build_plain_conv(c, wr, p + ".3", dout, 0, din, 3, 1, 1, 0, &lv.resample));
    else LDCCHK(build_plain_conv(c, wr, p + ".3.1", dout, 0, din, 3, 1, 1, 1, &lv.resample));
    u.ups.push_back(lv);
  }
  LDCCHK(build_resnet(c, wr, P + ".final_res_block", u.dim, u.dim, u.dim, &u.fin)); take_ss(u.fin);
  LDCCHK(build_plain_conv(c, wr, P + ".final_conv", u.dim, 0, u.channels, 1, 1, 0, 0, &u.final_conv));
  if (c->w8 && c->fp8_act && c->dt == DT_BF16 && u.dim % 64 == 0) {
    HostTensor* w = wr.get(P + ".final_conv.weight", {u.channels, u.dim, 1});
    HostTensor* b = wr.get(P + ".final_conv.bias", {u.channels});
    if (w && b) {
      ConvSpec sp;
      sp.dt = DT_FP8; sp.act8 = 1; sp.cin1 = u.dim; sp.cout = u.channels; sp.k = 1;
      LDCCHK(make_conv(c, sp, w->data.data(), b->data.data(), &u.final_conv_f8));
    }
  }
  u.ss_stride = ss_off;
  // cond upsampler: non-causal SConvTranspose1d(k=2r, s=r), no weight-norm (unet.py:372-377, conv.py:270-273)
  for (int i = 0; i < c->cfg.n_upsampling_ratios; ++i) {
    const int r = c->cfg.upsampling_ratios[i];
    const std::string p = P + ".upsampling_layers." + std::to_string(i) + ".convtr.convtr";
    HostTensor* w = wr.get(p + ".weight", {u.cond_channels, u.cond_channels, 2 * r});
    HostTensor* b = wr.get(p + ".bias", {u.cond_channels});
    u.up_ratios.push_back(r);
    if (!w || !b) continue;
    ConvLayer ly;
    const int padding_total = r, right = padding_total / 2, left = padding_total - right;
    LDCCHK(make_convtr(c, DT_F32, u.cond_channels, u.cond_channels, r, left, ACT_NONE, w->data.data(), b->data.data(), &ly));
    u.upsamplers.push_back(ly);
  }
  // schedule buffers come from the checkpoint (they are registered buffers, ddpm_loss.py:138-168)
  static const char* names[13] = {"betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                                  "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
                                  "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                                  "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
                                  "p2_loss_weight"};
  std::map<std::string, float*> dev;
  for (const char* n : names) {
    HostTensor* t = wr.get(std::string("diffusion.") + n, {u.timesteps});
    if (!t) continue;
    float* d = nullptr;
    LDCCHK(c->wmem.upload(&d, t->data));
    dev[n] = d;
  }
  c->sched.sqrt_recip_alphas_cumprod = dev["sqrt_recip_alphas_cumprod"];
  c->sched.sqrt_recipm1_alphas_cumprod = dev["sqrt_recipm1_alphas_cumprod"];
  c->sched.posterior_mean_coef1 = dev["posterior_mean_coef1"];
  c->sched.posterior_mean_coef2 = dev["posterior_mean_coef2"];
  c->sched.posterior_log_variance_clipped = dev["posterior_log_variance_clipped"];
  c->sqrt_alphas_cumprod = dev["sqrt_alphas_cumprod"];
  c->sqrt_one_minus_alphas_cumprod = dev["sqrt_one_minus_alphas_cumprod"];
  c->p2_loss_weight = dev["p2_loss_weight"];
  if (wr.missing.empty()) LDCCHK(build_time_table(c, wr));
  *missing += wr.missing;
  return LDC_OK;
}

// SinusoidalPosEmb + time_mlp + every ResnetBlock.mlp evaluated for all t = 0..T-1 at load time
// (they depend on t only; reference unet.py:109-116, 327-332, 162-165, 183-186).  fp32 on the GPU,
// through the same conv-GEMM kernel (a Linear is a k=1 conv over T "positions").
static int build_time_table(ldc_ctx* c, WeightReader& wr) {
  UnetW& u = c->unet;
  const int T = u.timesteps, dim = u.dim, td = u.time_dim;
  const std::string P = "diff_model";
  HostTensor* w1 = wr.get(P + ".time_mlp.1.weight", {td, dim});
  HostTensor* b1 = wr.get(P + ".time_mlp.1.bias", {td});
  HostTensor* w2 = wr.get(P + ".time_mlp.3.weight", {td, td});
  HostTensor* b2 = wr.get(P + ".time_mlp.3.bias", {td});
  if (!w1 || !b1 || !w2 || !b2) return LDC_OK;
  const int half = dim / 2;
  std::vector<float> emb((size_t)T * dim);
  const float lg = logf(10000.0f) / (float)(half - 1);
  for (int t = 0; t < T; ++t)
    for (int k = 0; k < half; ++k) {
      const float f = expf((float)k * -lg);
      const float a = (float)t * f;
      emb[(size_t)t * dim + k] = sinf(a);
      emb[(size_t)t * dim + half + k] = cosf(a);
    }
  DevMem tmp;
  float *d_emb = nullptr, *d_h1 = nullptr, *d_h2 = nullptr;
  LDCCHK(tmp.upload(&d_emb, emb));
  void* p = nullptr;
  LDCCHK(tmp.alloc(&p, (size_t)T * td * 4)); d_h1 = (float*)p;
  LDCCHK(tmp.alloc(&p, (size_t)T * td * 4)); d_h2 = (float*)p;
  LDCCHK(c->wmem.alloc(&p, (size_t)T * u.ss_stride * 4));
  u.ss_table = (float*)p;
  LDCCHK(c->wmem.alloc(&p, (size_t)u.ss_stride * 4));
  u.cur_ss = (float*)p;
  hipStream_t s = c->own_stream;
  ConvLayer l1, l2;
  ConvSpec sp;
  sp.dt = DT_F32; sp.cin1 = dim; sp.cout = td; sp.k = 1; sp.post_act = ACT_GELU;
  LDCCHK(make_conv(c, sp, w1->data.data(), b1->data.data(), &l1));
  sp.cin1 = td; sp.post_act = ACT_SILU;   // SiLU of ResnetBlock.mlp[0] folded here: every block consumes SiLU(temb)
  LDCCHK(make_conv(c, sp, w2->data.data(), b2->data.data(), &l2));
  ConvCall cc;
  cc.B = 1; cc.L_in = T; cc.L_rows = T;
  cc.x1 = d_emb; cc.y = d_h1; cc.y_ld = td;
  HIPCHK(launch_conv(l1, cc, s));
  cc.x1 = d_h1; cc.y = d_h2;
  HIPCHK(launch_conv(l2, cc, s));
  std::vector<ResnetW*> blocks;
  for (auto& lv : u.downs) { blocks.push_back(&lv.b1); blocks.push_back(&lv.b2); }
  blocks.push_back(&u.mid1); blocks.push_back(&u.mid2);
  for (auto& lv : u.ups) { blocks.push_back(&lv.b1); blocks.push_back(&lv.b2); }
  blocks.push_back(&u.fin);
  std::vector<std::string> prefixes;
  for (size_t i = 0; i < u.downs.size(); ++i) { prefixes.push_back(P + ".downs." + std::to_string(i) + ".0"); prefixes.push_back(P + ".downs." + std::to_string(i) + ".1"); }
  prefixes.push_back(P + ".mid_block1"); prefixes.push_back(P + ".mid_block2");
  for (size_t i = 0; i < u.ups.size(); ++i) { prefixes.push_back(P + ".ups." + std::to_string(i) + ".0"); prefixes.push_back(P + ".ups." + std::to_string(i) + ".1"); }
  prefixes.push_back(P + ".final_res_block");
  for (size_t i = 0; i < blocks.size(); ++i) {
    ResnetW* r = blocks[i];
    HostTensor* wm = wr.get(prefixes[i] + ".mlp.1.weight", {2 * r->cout, td});
    HostTensor* bm = wr.get(prefixes[i] + ".mlp.1.bias", {2 * r->cout});
    if (!wm || !bm) continue;
    ConvLayer lm;
    ConvSpec sm;
    sm.dt = DT_F32; sm.cin1 = td; sm.cout = 2 * r->cout; sm.k = 1;
    LDCCHK(make_conv(c, sm, wm->data.data(), bm->data.data(), &lm));
    ConvCall cm;
    cm.B = 1; cm.L_in = T; cm.L_rows = T; cm.x1 = d_h2; cm.y = u.ss_table + r->ss_off; cm.y_ld = u.ss_stride;
    HIPCHK(launch_conv(lm, cm, s));
    u.weight_elems += (double)wm->numel();
  }
  HIPCHK(hipStreamSynchronize(s));
  return LDC_OK;
}

// ------------------------------------------------------------------------------------------------
// lifecycle
// ------------------------------------------------------------------------------------------------
extern "C" int ldc_create(const ldc_config* cfg, int device, ldc_ctx** out) {
  if (!cfg || !out) return fail(LDC_E_INVALID, "null argument");
  if (cfg->compute_dtype != LDC_F32 && cfg->compute_dtype != LDC_BF16 && cfg->compute_dtype != LDC_BF16_W8)
    return fail(LDC_E_INVALID, "compute_dtype must be LDC_F32, LDC_BF16 or LDC_BF16_W8");
  if (cfg->n_enc_ratios < 1 || cfg->n_enc_ratios > LDC_MAX_RATIOS || cfg->n_upsampling_ratios < 0 ||
      cfg->n_upsampling_ratios > LDC_MAX_RATIOS)
    return fail(LDC_E_INVALID, "bad ratio counts");
  if (cfg->rep_dims != 128) return fail(LDC_E_INVALID, "rep_dims must be 128 (RVQ / UNet channel width of the reference checkpoints)");
  if (cfg->diff_dims % 32 || cfg->diff_dims <= 0) return fail(LDC_E_INVALID, "diff_dims must be a positive multiple of 32");
  if (cfg->n_filters % 32) return fail(LDC_E_INVALID, "n_filters must be a multiple of 32");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(LDC_E_INVALID, "device %d out of range (%d visible)", device, ndev);
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(LDC_E_INVALID, "this library is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
  std::unique_ptr<ldc_ctx> c(new ldc_ctx());
  c->cfg = *cfg;
  c->device = device;
  c->num_cus = prop.multiProcessorCount;
  c->dt = cfg->compute_dtype == LDC_F32 ? DT_F32 : DT_BF16;
  c->w8 = cfg->compute_dtype == LDC_BF16_W8;
  HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  for (int k = 0; k < kMaxParts; ++k)
    for (int e = 0; e <= ldc_ctx::kLstmChunks; ++e) HIPCHK(hipEventCreateWithFlags(&c->lstm_ev[k][e], hipEventDisableTiming));
  if (const char* ex = getenv("LDC_TEST_EXTRA_STREAMS")) {   // test hook: shift the stream -> hardware-queue mapping the way other libraries' streams would
    for (int i = 0; i < atoi(ex); ++i) { hipStream_t dummy = nullptr; HIPCHK(hipStreamCreateWithFlags(&dummy, hipStreamNonBlocking)); }
  }
  HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  for (auto& e : c->flow_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (int k = 1; k < kMaxParts; ++k) {
    HIPCHK(hipStreamCreateWithFlags(&c->aux_stream[k], hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->ev_join[k], hipEventDisableTiming));
  }
  for (int k = 0; k < kMaxParts; ++k) {
    HIPCHK(hipStreamCreateWithFlags(&c->side_stream[k], hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->ev_side_fork[k], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_side_join[k], hipEventDisableTiming));
  }
  // Every knob of the library is an OPTION (one table: find_option below; tools/README.md lists them): ldc_set_option at run time, or
  // LDC_OPTIONS="name=value,name=value" in the environment at ldc_create (round 6: ~50 separate LDC_* variables before).
  c->xcd_resident[0] = lstm_xcd_resident(256);
  c->xcd_resident[1] = lstm_xcd_resident(512);
  c->coop_resident[0] = lstm_coop_resident(256) ? 1 : 0;
  c->coop_resident[1] = lstm_coop_resident(512) ? 1 : 0;
  c->plan_bytes_cap = (size_t)48 << 30;
  c->plan_count_cap = 24;
  if (const char* mp = getenv("LDC_AUX_FROM_SIDE")) {   // diagnostics (hardware-queue mapping): part stream k = side stream digit k of the value ('-' keeps it)
    c->calib = 0;
    for (int k = 1; k < kMaxParts && mp[k - 1]; ++k)
      if (mp[k - 1] >= '0' && mp[k - 1] < '0' + kMaxParts) std::swap(c->aux_stream[k], c->side_stream[mp[k - 1] - '0']);
  }
  if (const char* ov = getenv("LDC_OPTIONS")) {
    std::string all(ov);
    size_t pos = 0;
    while (pos < all.size()) {
      size_t end = all.find_first_of(",; ", pos);
      if (end == std::string::npos) end = all.size();
      const std::string kv = all.substr(pos, end - pos);
      pos = end + 1;
      if (kv.empty()) continue;
      const size_t eq = kv.find('=');
      const std::string name = kv.substr(0, eq);
      const int value = eq == std::string::npos ? 1 : atoi(kv.c_str() + eq + 1);
      std::unique_ptr<ldc_ctx>& cc = c;
      const int rc = ldc_set_option(cc.get(), name.c_str(), value);
      if (rc != LDC_OK) return fail(LDC_E_INVALID, "LDC_OPTIONS: %s", ldc_last_error());
    }
  }
  switch (cfg->final_activation) {
    case LDC_ACT_NONE: c->enc_final_act = ACT_NONE; break;
    case LDC_ACT_TANH: c->enc_final_act = ACT_TANH; break;
    case LDC_ACT_SIGMOID: c->enc_final_act = ACT_SIGMOID; break;
    case LDC_ACT_ELU: c->enc_final_act = ACT_ELU; break;
    case LDC_ACT_SILU: c->enc_final_act = ACT_SILU; break;
    case LDC_ACT_GELU: c->enc_final_act = ACT_GELU; break;
    case LDC_ACT_RELU: c->enc_final_act = ACT_RELU; break;
    default: return fail(LDC_E_INVALID, "final_activation code %d is not one of LDC_ACT_*", cfg->final_activation);
  }
  {
    void* hp = nullptr;
    HIPCHK(hipHostMalloc(&hp, 64, hipHostMallocMapped));
    memset(hp, 0, 64);
    c->dev_flag_host = (unsigned*)hp;
    void* dp = nullptr;
    HIPCHK(hipHostGetDevicePointer(&dp, hp, 0));
    c->dev_flag_dev = (unsigned*)dp;
  }
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, 4 * sizeof(int)));
  c->step_state = (int*)p;
  *out = c.release();
  return LDC_OK;
}

void drop_plans(ldc_ctx* c) {
  (void)counted_device_sync();   // nothing captured or planned may still be running when it is destroyed
  for (auto& g : c->graphs) g.destroy();
  c->graphs.clear();
  for (auto& pl : c->plans) {
    for (hipEvent_t e : pl->marker_events) (void)hipEventDestroy(e);
    if (pl->arena_base) (void)hipFree(pl->arena_base);
  }
  c->plans.clear();
  c->plan_bytes = 0;
  c->last_halves = Halves();
}

extern "C" int ldc_destroy(ldc_ctx* c) {
  if (!c) return LDC_OK;
  (void)hipSetDevice(c->device);
  (void)counted_device_sync();
  drop_plans(c);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->outnorm_ws) (void)hipFree(c->outnorm_ws);
  if (c->state_buf) (void)hipFree(c->state_buf);
  if (c->step_state) (void)hipFree(c->step_state);
  if (c->dev_flag_host) (void)hipHostFree(c->dev_flag_host);
  if (c->tl_buf) (void)hipFree(c->tl_buf);
  for (auto& e : c->prof_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  for (int k = 0; k < kMaxParts; ++k) {
    for (int e = 0; e <= ldc_ctx::kLstmChunks; ++e)
      if (c->lstm_ev[k][e]) (void)hipEventDestroy(c->lstm_ev[k][e]);
  }
  for (auto& e : c->flow_ev) if (e) (void)hipEventDestroy(e);
  for (int k = 1; k < kMaxParts; ++k) {
    if (c->ev_join[k]) (void)hipEventDestroy(c->ev_join[k]);
    if (c->aux_stream[k]) (void)hipStreamDestroy(c->aux_stream[k]);
  }
  for (int k = 0; k < kMaxParts; ++k) {
    if (c->ev_side_fork[k]) (void)hipEventDestroy(c->ev_side_fork[k]);
    if (c->ev_side_join[k]) (void)hipEventDestroy(c->ev_side_join[k]);
    if (c->side_stream[k]) (void)hipStreamDestroy(c->side_stream[k]);
  }
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return LDC_OK;
}

// OCP e4m3 quantisation exactly as the fp8 weight packer applies it (host function: testable without a GPU)
extern "C" void ldc_quantize_e4m3(const float* in, int64_t n, uint8_t* out_codes, float* out_values) {
  for (int64_t i = 0; i < n; ++i) {
    const uint8_t q = host_f32_to_e4m3(in[i]);
    if (out_codes) out_codes[i] = q;
    if (out_values) out_values[i] = host_e4m3_to_f32(q);
  }
}

// ---- options: ONE table for ldc_set_option and LDC_OPTIONS (no os.environ mutation around ldc_create; VERDICT r5: ~70 LDC_* switches) ----
// replan: cached plans / graphs depend on the value and are dropped when it changes.  tools/README.md documents every name.
struct OptRef { int* p; bool replan; int lo, hi; };
static bool find_option(ldc_ctx* c, const std::string& n, OptRef* r) {
#define LDC_OPT(NAME, FIELD, REPLAN, LO, HI) if (n == NAME) { r->p = &(FIELD); r->replan = REPLAN; r->lo = LO; r->hi = HI; return true; }
  // launch structure of a decode
  LDC_OPT("split", c->split_batch, true, 1, kMaxParts)            // independent chains a batch is decoded as
  LDC_OPT("part_graphs", c->part_graphs, false, 0, 2)             // 0 one fork / join graph | 1 one single-stream graph per part (two parts) | 2 also for 3 / 4 parts
  LDC_OPT("graph_steps", c->graph_steps, false, 0, 1000)          // steps per replayed graph (0: by chain count)
  LDC_OPT("flow_depth", c->flow_depth, false, 0, (int)ldc_ctx::kFlowRing - 1)
  LDC_OPT("split_ends", c->split_ends, false, 0, 1)               // codec front / back ends per batch part on the parts' streams
  LDC_OPT("ends_join", c->ends_join, false, 0, 1)
  LDC_OPT("split_init", c->split_init, true, 0, 1)                // init_conv's condition half once per sampler call
  LDC_OPT("side_streams", c->side_streams, true, 0, 1)            // res_conv on a side stream (single chain only)
  LDC_OPT("serial_parts", c->serial_parts, false, 0, 1)           // diagnostics: parts back to back, eager
  LDC_OPT("stream_calib", c->calib, false, 0, 1)                  // choose the part streams by measured overlap
  LDC_OPT("merge_advance", c->merge_advance, true, 0, 1)          // the step state advances in the step's first kernel
  // fusions of the UNet step
  LDC_OPT("fuse_gn_stats", c->fuse_gn_stats, true, 0, 1)
  LDC_OPT("fuse_gn_epi", c->fuse_gn_epi, true, 0, 1)              // GroupNorm apply in the conv epilogue; 0 is the fallback after a [gn_wait] failure
  LDC_OPT("gn_epi_min_l", c->gn_epi_min_l, true, 0, 1 << 30)
  LDC_OPT("gn_epi_max_tiles", c->gn_epi_max_tiles, true, 0, 1 << 30)
  LDC_OPT("fold_res", c->fold_res, true, 0, 1)
  LDC_OPT("fold_ln", c->fold_ln, true, 0, 1)
  LDC_OPT("fuse_kmax", c->fuse_kmax, true, 0, 1)
  LDC_OPT("fuse_ln", c->fuse_ln, true, 0, 1)
  LDC_OPT("fuse_attn_tail", c->fuse_attn_tail, true, 0, 1)
  LDC_OPT("fold_ctx", c->fold_ctx, true, 0, 1)                     // LinearAttention context inside to_qkv's epilogue (lean kernel, bf16)
  // conv launchers (ConvTune)
  LDC_OPT("conv_lean", c->tune.lean, true, 0, 1)                  // conv_lean_kernel where its shapes allow; 0 = conv_fast_kernel (A/B)
  LDC_OPT("conv_generic", c->tune.force_generic, true, 0, 1)      // every conv on the generic kernel
  LDC_OPT("conv_small_tiles", c->tune.small_max, true, 0, 1 << 30)
  LDC_OPT("conv_splitk", c->tune.splitk, true, 0, 4)
  LDC_OPT("sk_tiles", c->tune.sk_tiles, true, 0, 1 << 30)
  LDC_OPT("sk_u2", c->tune.sk_u2, true, 0, 1 << 30)
  LDC_OPT("sk_u3", c->tune.sk_u3, true, 0, 1 << 30)
  LDC_OPT("conv_mfast", c->tune.m_fastest, true, 0, 2)
  LDC_OPT("conv_debug", c->tune.debug, true, 0, 255)              // ablation bits (conv_lean.inc / conv_fast.inc)
  LDC_OPT("conv_tile", c->tune.force_tile, true, -1, 1)
  LDC_OPT("conv_xcd_order", c->tune.xcd_order, true, 0, 8)
  LDC_OPT("gn_nap", c->tune.gn_nap, true, 0, 4096)
  LDC_OPT("gn_nap0", c->tune.gn_nap0, true, 0, 4096)
  // codec ends
  LDC_OPT("lstm_stream", c->lstm_stream_only, false, 0, 1)        // never the cooperative LSTM kernel
  LDC_OPT("lstm_xcd", c->lstm_xcd, false, 0, 1)
  LDC_OPT("lstm_pipe", c->lstm_pipe, false, 0, 1)
  LDC_OPT("coop_launch", c->coop_launch, false, 0, 1)
  LDC_OPT("sea_splitk", c->sea_splitk, false, 0, 1)
  LDC_OPT("rvq_tiled", c->rvq_tiled, false, 0, 1)
  // process-wide training switches
  LDC_OPT("train_bf16", g_train_bf16, false, 0, 1)
  LDC_OPT("train_fp32_mfma", g_train_fp32_mfma, false, 0, 1)
  LDC_OPT("train_valu", g_train_valu, false, 0, 1)
  LDC_OPT("train_dw_side", g_train_dw_side, false, 0, 1)
#undef LDC_OPT
  return false;
}

extern "C" int ldc_set_option(ldc_ctx* c, const char* name, int value) {
  if (!c || !name) return fail(LDC_E_INVALID, "null context or option name");
  const std::string n(name);
  if (n == "fp8_act") {   // fp8-weight contexts: fp8 x fp8 MFMA where a tensor's only consumer is a conv (decided when the weights are packed)
    if (c->finalized) return fail(LDC_E_STATE, "fp8_act must be set before ldc_finalize_weights");
    c->fp8_act = value ? 1 : 0;
    return LDC_OK;
  }
  if (n == "plan_cache_gb") { c->plan_bytes_cap = (size_t)std::max(1, value) << 30; return LDC_OK; }
  if (n == "plan_cache_n") { c->plan_count_cap = std::max(2 * kMaxParts, value); return LDC_OK; }
  OptRef r;
  if (!find_option(c, n, &r)) return fail(LDC_E_INVALID, "unknown option '%s' (tools/README.md lists the options)", name);
  if (value < r.lo || value > r.hi) return fail(LDC_E_INVALID, "option %s must be in %d..%d", name, r.lo, r.hi);
  if (n == "side_streams" && value && c->split_batch != 1) return fail(LDC_E_INVALID, "side streams need a single chain (split 1)");
  if (value == *r.p) return LDC_OK;
  if (r.replan && !c->plans.empty()) { HIPCHK(hipSetDevice(c->device)); drop_plans(c); }   // plans of one shape differ by the value: start clean
  *r.p = value;
  if (n == "split" && value != 1) c->side_streams = 0;
  return LDC_OK;
}

// device-wide synchronisations issued by this library in this process so far (documented cold paths only: plan eviction, re-capture,
// option changes, profiling / tuning reads)
extern "C" long long ldc_debug_sync_count(void) { return g_device_syncs; }

// debug hook: raise the device-side failure flag as a kernel that gave up would (1 = cooperative LSTM, 2 = fused GroupNorm wait)
extern "C" int ldc_debug_raise_failure(ldc_ctx* c, int code) {
  if (!c || !c->dev_flag_host || (code != 1 && code != 2)) return fail(LDC_E_INVALID, "bad arguments");
  *reinterpret_cast<volatile unsigned*>(c->dev_flag_host) = (unsigned)code;
  return LDC_OK;
}

extern "C" int ldc_reseed(ldc_ctx* c, uint64_t seed) {
  if (!c) return fail(LDC_E_INVALID, "null ctx");
  c->cfg.noise_seed = seed;
  c->noise_epoch = 0;
  return LDC_OK;
}

extern "C" int ldc_set_weight(ldc_ctx* c, int which, const char* key, const float* data, const int64_t* shape, int ndim) {
  if (!c || !key || !data || (ndim > 0 && !shape)) return fail(LDC_E_INVALID, "null argument");
  if (which != LDC_MODEL_MAIN && which != LDC_MODEL_COND) return fail(LDC_E_INVALID, "which must be LDC_MODEL_MAIN or LDC_MODEL_COND");
  if (c->finalized) return fail(LDC_E_STATE, "weights already finalized");
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.assign(data, data + t.numel());
  c->raw[which][key] = std::move(t);
  return LDC_OK;
}

extern "C" int ldc_finalize_weights(ldc_ctx* c, int strict) {
  if (!c) return fail(LDC_E_INVALID, "null ctx");
  if (c->finalized) return fail(LDC_E_STATE, "already finalized");
  HIPCHK(hipSetDevice(c->device));
  std::string missing;
  std::vector<int> main_ratios(c->cfg.enc_ratios, c->cfg.enc_ratios + c->cfg.n_enc_ratios);
  LDCCHK(build_codec(c, LDC_MODEL_MAIN, main_ratios, 0, &missing));
  LDCCHK(build_unet(c, &missing));
  if (c->cfg.has_cond_model) {
    // quirk Q1: the cond codec is always built with the default ratios [8,5,4,2] (sample.py:63, model.py:34)
    const int frame_rate_ceil = 50;
    const int n_q = (int)floor(1000.0 * c->cfg.cond_bandwidth / (frame_rate_ceil * 10));   // model.py:65
    LDCCHK(build_codec(c, LDC_MODEL_COND, {8, 5, 4, 2}, n_q, &missing));
  }
  if (!missing.empty()) return fail(LDC_E_MISSING, "missing or mis-shaped keys in state_dict: %s", missing.c_str());
  if (strict) {
    std::string unexpected;
    for (int w = 0; w < 2; ++w)
      for (auto& kv : c->raw[w])
        if (!kv.second.used) {
          // the UNet is registered twice (diff_model.* and diffusion.model.*): either alias satisfies the other
          const std::string& k = kv.first;
          std::string alias;
          if (k.rfind("diffusion.model.", 0) == 0) alias = "diff_model." + k.substr(16);
          else if (k.rfind("diff_model.", 0) == 0) alias = "diffusion.model." + k.substr(11);
          auto it = alias.empty() ? c->raw[w].end() : c->raw[w].find(alias);
          if (it != c->raw[w].end() && it->second.used && it->second.shape == kv.second.shape) continue;
          if (unexpected.size() < 600) unexpected += k + " ";
        }
    if (!unexpected.empty()) return fail(LDC_E_MISSING, "unexpected keys in state_dict: %s", unexpected.c_str());
  }
  c->raw[0].clear();
  c->raw[1].clear();
  c->finalized = true;
  return LDC_OK;
}

// ------------------------------------------------------------------------------------------------
// scratch
// ------------------------------------------------------------------------------------------------
// Growth of a context-owned workspace is the one place a stage call waits for the device (the old buffer may still be
// in use by earlier asynchronous calls); steady-state calls never do (include/ladiffcodec.h documents this).
int ensure_scratch(ldc_ctx* c, size_t bytes, hipStream_t s) {
  if (bytes <= c->scratch_cap) return LDC_OK;
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(counted_device_sync());
  if (c->scratch) HIPCHK(hipFree(c->scratch));
  c->scratch = nullptr;
  c->scratch_cap = 0;
  void* p = nullptr;
  const size_t want = bytes + bytes / 4;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) return fail(LDC_E_NOMEM, "hipMalloc(%zu) for scratch failed: %s", want, hipGetErrorString(e));
  c->scratch = (char*)p;
  c->scratch_cap = want;
  return LDC_OK;
}

// ------------------------------------------------------------------------------------------------
// SEANet execution (codec dtype = fp32, channels-last)
// ------------------------------------------------------------------------------------------------
int conv_out_len(const ConvLayer& ly, int L) {
  // SConv1d output length with the reference's extra right padding (conv.py:56-63): ceil(L / stride)
  const int eff_k = (ly.taps - 1) * ly.dil + 1;
  const int padding_total = eff_k - ly.stride;
  const double n_frames = (double)(L - eff_k + padding_total) / ly.stride + 1.0;
  return (int)ceil(n_frames);
}


static int sea_conv(SeaRun& R, const ConvLayer& ly, const void* x, const void* residual, int L_in, void** y, int* L_out,
                    int cout) {
  ConvCall cc;
  cc.B = R.B; cc.L_in = L_in; cc.x1 = x; cc.residual = residual; cc.tune = &R.c->tune;
  if (ly.tr_stride) {
    cc.L_rows = L_in + 1;
    cc.L_final = L_in * ly.tr_stride;
    *L_out = cc.L_final;
    cc.y_ld = ly.tr_cout;
  } else {
    cc.L_rows = conv_out_len(ly, L_in);
    *L_out = cc.L_rows;
    cc.y_ld = cout;
  }
  *y = R.ar->alloc((size_t)R.B * (*L_out) * cc.y_ld * 4);
  cc.y = *y;
  if (R.c->sea_splitk) {   // few-tile long-K layers: split-K partial sums in the run's arena
    const long long fl = conv_generic_splitk_floats(ly, cc);
    if (fl > 0) {
      cc.sk_part = (float*)R.ar->alloc((size_t)fl * 4);
      cc.sk_part_cap = fl;
      cc.generic_split = 1;
    }
  }
  if (!R.dry) HIPCHK(launch_conv(ly, cc, R.s));
  return LDC_OK;
}

int run_seanet(SeaRun& R, const std::vector<SeaOp>& ops, const void* x_in, int L, void** out, int* L_out, int* C_out) {
  const void* x = x_in;
  int C = 0;
  for (const SeaOp& op : ops) {
    void* y = nullptr;
    int Ln = L;
    switch (op.kind) {
      case SeaOp::CONV_CIN1: {
        if (L <= op.k - 1) return fail(LDC_E_INVALID, "input shorter than the first conv's receptive field");
        y = R.ar->alloc((size_t)R.B * L * op.cout * 4);
        if (!R.dry) HIPCHK(launch_conv_cin1(DT_F32, (const float*)x, y, op.w1, op.b1, R.B, L, op.cout, op.k, R.s));
        C = op.cout;
        break;
      }
      case SeaOp::CONV:
      case SeaOp::CONVTR:
        LDCCHK(sea_conv(R, op.conv, x, nullptr, L, &y, &Ln, op.cout));
        C = op.cout;
        break;
      case SeaOp::RES: {
        void *sc = nullptr, *h = nullptr;
        int l1 = L, l2 = L, l3 = L;
        LDCCHK(sea_conv(R, op.shortcut, x, nullptr, L, &sc, &l1, op.cout));
        LDCCHK(sea_conv(R, op.conv, x, nullptr, L, &h, &l2, op.hidden));
        LDCCHK(sea_conv(R, op.conv2, h, sc, l2, &y, &l3, op.cout));
        Ln = l3;
        C = op.cout;
        break;
      }
      case SeaOp::LSTM: {
        const int H = op.cout;
        const void* in = x;
        // Two-layer register LSTM (the main codec's decoder: 1 200 steps x 2 layers, one workgroup per item) as a TWO-STAGE PIPELINE over
        // time chunks: layer 0 writes its chunk time-major, the side stream runs layer 1's input GEMM over exactly those rows (contiguous
        // in that layout) and layer 1's chunk while layer 0 is on its next one; the recurrent state travels through [B][2H] buffers.
        // 5 chunk times instead of 8 on paper; measured (tools/lstm_pipe_time.py) it does not pay on this runtime -- see ldc_ctx::lstm_pipe -- and is off.
        const bool pipe = R.c->lstm_pipe && R.side >= 0 && R.side < 2 && R.c->aux_stream[2 + R.side] && op.lstm.size() == 2 && lstm_seq_supported(H) && !op.lstm[0].w_rm && !op.lstm[1].w_rm &&
                          L >= 64 * ldc_ctx::kLstmChunks;
        if (pipe) {
          constexpr int NCH = ldc_ctx::kLstmChunks;
          void* pre0 = R.ar->alloc((size_t)R.B * L * 4 * H * 4);
          void* o0 = R.ar->alloc((size_t)R.B * L * H * 4);         // time-major [L][B][H]
          void* pre1 = R.ar->alloc((size_t)R.B * L * 4 * H * 4);   // time-major [L][B][4H]
          void* o1 = R.ar->alloc((size_t)R.B * L * H * 4);
          float* st0 = (float*)R.ar->alloc((size_t)R.B * 2 * H * 4);
          float* st1 = (float*)R.ar->alloc((size_t)R.B * 2 * H * 4);
          if (!R.dry) {
            hipStream_t s2 = R.c->aux_stream[2 + R.side];
            hipEvent_t* ev = R.c->lstm_ev[R.side];
            ConvCall cc;
            cc.B = R.B; cc.L_in = L; cc.L_rows = L; cc.x1 = in; cc.y = pre0; cc.y_ld = 4 * H; cc.tune = &R.c->tune;
            HIPCHK(launch_conv(op.lstm[0].in_proj, cc, R.s));
            for (int k = 0; k < NCH; ++k) {
              const int t0 = (int)((long long)L * k / NCH), t1 = (int)((long long)L * (k + 1) / NCH);
              LstmSeq q0;
              q0.t0 = t0; q0.t1 = t1; q0.pre_bs = L; q0.pre_ts = 1; q0.out_bs = 1; q0.out_ts = R.B; q0.state = st0;
              HIPCHK(launch_lstm_seq(DT_F32, pre0, op.lstm[0].w_hh, o0, nullptr, R.B, H, q0, R.s));
              HIPCHK(hipEventRecord(ev[k], R.s));
              HIPCHK(hipStreamWaitEvent(s2, ev[k], 0));
              ConvCall c1;
              c1.B = 1; c1.L_in = (t1 - t0) * R.B; c1.L_rows = c1.L_in; c1.tune = &R.c->tune; c1.y_ld = 4 * H;
              c1.x1 = (const char*)o0 + (size_t)t0 * R.B * H * 4;
              c1.y = (char*)pre1 + (size_t)t0 * R.B * 4 * H * 4;
              HIPCHK(launch_conv(op.lstm[1].in_proj, c1, s2));
              LstmSeq q1;
              q1.t0 = t0; q1.t1 = t1; q1.pre_bs = 1; q1.pre_ts = R.B; q1.out_bs = L; q1.out_ts = 1; q1.skip_bs = L; q1.skip_ts = 1; q1.state = st1;
              HIPCHK(launch_lstm_seq(DT_F32, pre1, op.lstm[1].w_hh, o1, x, R.B, H, q1, s2));
            }
            HIPCHK(hipEventRecord(ev[NCH], s2));
            HIPCHK(hipStreamWaitEvent(R.s, ev[NCH], 0));
          }
          y = o1;
          C = H;
          break;
        }
        for (size_t n = 0; n < op.lstm.size(); ++n) {
          void* pre = R.ar->alloc((size_t)R.B * L * 4 * H * 4);
          void* o = R.ar->alloc((size_t)R.B * L * H * 4);
          const bool coop = op.lstm[n].w_rm && !R.c->lstm_stream_only && R.c->coop_resident[H == 512 ? 1 : 0];
          void* lws = coop ? R.ar->alloc(lstm_coop_ws_bytes(H)) : nullptr;
          if (!R.dry) {
            ConvCall cc;
            cc.B = R.B; cc.L_in = L; cc.L_rows = L; cc.x1 = in; cc.y = pre; cc.y_ld = 4 * H; cc.tune = &R.c->tune;
            HIPCHK(launch_conv(op.lstm[n].in_proj, cc, R.s));
            const bool lastl = n + 1 == op.lstm.size();
            hipError_t le = hipErrorCooperativeLaunchTooLarge;
            if (coop) le = launch_lstm_coop(DT_F32, pre, op.lstm[n].w_rm, o, lastl ? x : nullptr, R.B, L, H, lws, R.c->dev_flag_dev, (R.c->lstm_xcd && R.c->xcd_resident[H == 512 ? 1 : 0] >= 2 * R.teams) ? 2 : R.c->coop_launch, R.s);
            if (le == hipErrorCooperativeLaunchTooLarge) {   // (or not eligible): one workgroup per item, W_hh streamed from L2
              (void)hipGetLastError();
              le = launch_lstm_layer(DT_F32, pre, op.lstm[n].w_hh, o, lastl ? x : nullptr, R.B, L, H, R.s);
            }
            HIPCHK(le);
          }
          in = o;
          y = o;
        }
        C = H;
        break;
      }
    }
    x = y;
    L = Ln;
  }
  *out = const_cast<void*>(x);
  *L_out = L;
  *C_out = C;
  return LDC_OK;
}


int check_ready(ldc_ctx* c, int which, bool need_cond_codec) {
  if (!c) return fail(LDC_E_INVALID, "null ctx");
  if (!c->finalized) return fail(LDC_E_STATE, "ldc_finalize_weights has not been called");
  if (which != LDC_MODEL_MAIN && which != LDC_MODEL_COND) return fail(LDC_E_INVALID, "bad model selector");
  if ((which == LDC_MODEL_COND || need_cond_codec) && !c->codec[LDC_MODEL_COND].present)
    return fail(LDC_E_STATE, "no cond model configured (has_cond_model = 0)");
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail(LDC_E_HIP, "hipSetDevice failed: %s", hipGetErrorString(e));
  return check_dev_flag(c);
}

extern "C" int ldc_seanet_encode(ldc_ctx* c, int which, const float* wav, int B, int T, float* z_out, void* stream) {
  LDCCHK(check_ready(c, which));
  if (!wav || !z_out || B <= 0 || T <= 0) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  const Codec& cd = c->codec[which];
  LDCCHK(with_scratch(c, s, [&](Arena& ar, bool dry) -> int {
    SeaRun R{c, &ar, s, dry, B};
    void* z = nullptr;
    int L = 0, C = 0;
    LDCCHK(run_seanet(R, cd.enc, wav, T, &z, &L, &C));
    if (!dry) HIPCHK(launch_from_cl(DT_F32, z, z_out, B, C, L, nullptr, 0, 0.f, s));
    return LDC_OK;
  }));
  return finish_stream(c, stream);
}

extern "C" int ldc_seanet_decode(ldc_ctx* c, int which, const float* z, int B, int L, float* wav_out, void* stream) {
  LDCCHK(check_ready(c, which));
  if (!z || !wav_out || B <= 0 || L <= 0) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  const Codec& cd = c->codec[which];
  const int D = c->cfg.rep_dims;
  LDCCHK(with_scratch(c, s, [&](Arena& ar, bool dry) -> int {
    SeaRun R{c, &ar, s, dry, B};
    R.side = 0;
    void* zc = ar.alloc((size_t)B * L * D * 4);
    if (!dry) HIPCHK(launch_to_cl(DT_F32, z, zc, B, D, L, nullptr, 0, 0.f, s));
    void* y = nullptr;
    int Lo = 0, C = 0;
    LDCCHK(run_seanet(R, cd.dec, zc, L, &y, &Lo, &C));
    // last conv has Cout = 1: channels-last [B*T][1] is already [B,1,T]
    if (!dry) HIPCHK(hipMemcpyAsync(wav_out, y, (size_t)B * Lo * 4, hipMemcpyDeviceToDevice, s));
    return LDC_OK;
  }));
  return finish_stream(c, stream);
}

// ------------------------------------------------------------------------------------------------
// RVQ
// ------------------------------------------------------------------------------------------------
static int rvq_rows(ldc_ctx* c, const float* z_rows, int rows, int n_q, int64_t* codes, float* q_rows, Arena& ar, bool dry,
                    hipStream_t s) {
  const Codec& cd = c->codec[LDC_MODEL_COND];
  int64_t* cw = codes;
  if (!cw) cw = (int64_t*)ar.alloc((size_t)n_q * rows * sizeof(int64_t));
  if (!dry) HIPCHK(launch_rvq(z_rows, rows, c->cfg.rep_dims, cd.codebooks, cd.cb_sqnorm, cd.bins, n_q, cw, q_rows, s, c->rvq_tiled));
  return LDC_OK;
}

static int n_q_for_bandwidth(ldc_ctx* c, float bandwidth) {
  // vq.py:86-98 with frame_rate = 16000/320 = 50: bw_per_q = log2(1024)*50/1000 = 0.5
  const Codec& cd = c->codec[LDC_MODEL_COND];
  const double bw = bandwidth > 0 ? bandwidth : c->cfg.cond_bandwidth;
  const double bw_per_q = log2((double)cd.bins) * (16000.0 / cd.hop) / 1000.0;
  int n_q = cd.n_q_layers;
  if (bw > 0) n_q = (int)std::max(1.0, floor(bw / bw_per_q));
  return std::min(n_q, cd.n_q_layers);
}

extern "C" int ldc_rvq_encode(ldc_ctx* c, const float* z, int B, int F, int n_q, int64_t* codes_out, float* quantized_out,
                              void* stream) {
  LDCCHK(check_ready(c, LDC_MODEL_COND));
  const Codec& cd = c->codec[LDC_MODEL_COND];
  if (!z || !quantized_out || B <= 0 || F <= 0 || n_q < 1 || n_q > cd.n_q_layers)
    return fail(LDC_E_INVALID, "bad arguments (n_q must be in [1,%d])", cd.n_q_layers);
  hipStream_t s = pick_stream(c, stream);
  const int D = c->cfg.rep_dims;
  LDCCHK(with_scratch(c, s, [&](Arena& ar, bool dry) -> int {
    float* zr = (float*)ar.alloc((size_t)B * F * D * 4);
    float* qr = (float*)ar.alloc((size_t)B * F * D * 4);
    if (!dry) HIPCHK(launch_to_cl(DT_F32, z, zr, B, D, F, nullptr, 0, 0.f, s));
    LDCCHK(rvq_rows(c, zr, B * F, n_q, codes_out, qr, ar, dry, s));
    if (!dry) HIPCHK(launch_from_cl(DT_F32, qr, quantized_out, B, D, F, nullptr, 0, 0.f, s));
    return LDC_OK;
  }));
  return finish_stream(c, stream);
}

extern "C" int ldc_rvq_decode(ldc_ctx* c, const int64_t* codes, int B, int F, int n_q, float* quantized_out, void* stream) {
  LDCCHK(check_ready(c, LDC_MODEL_COND));
  const Codec& cd = c->codec[LDC_MODEL_COND];
  if (!codes || !quantized_out || B <= 0 || F <= 0 || n_q < 1 || n_q > cd.n_q_layers) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  const int D = c->cfg.rep_dims;
  LDCCHK(with_scratch(c, s, [&](Arena& ar, bool dry) -> int {
    float* qr = (float*)ar.alloc((size_t)B * F * D * 4);
    if (!dry) {
      HIPCHK(launch_rvq_decode(codes, B * F, D, cd.codebooks, cd.bins, n_q, qr, s));
      HIPCHK(launch_from_cl(DT_F32, qr, quantized_out, B, D, F, nullptr, 0, 0.f, s));
    }
    return LDC_OK;
  }));
  return finish_stream(c, stream);
}

// encoder -> RVQ, rows stay channels-last in between; returns the quantized rows pointer (scratch)
static int get_cond_rows(ldc_ctx* c, const float* wav, int B, int T, float bandwidth, Arena& ar, bool dry, hipStream_t s,
                         float** q_rows, int* F_out, int64_t* codes_out, int teams = 1) {
  const Codec& cd = c->codec[LDC_MODEL_COND];
  SeaRun R{c, &ar, s, dry, B};
  R.teams = teams;
  void* z = nullptr;
  int F = 0, C = 0;
  LDCCHK(run_seanet(R, cd.enc, wav, T, &z, &F, &C));
  float* qr = (float*)ar.alloc((size_t)B * F * C * 4);
  LDCCHK(rvq_rows(c, (const float*)z, B * F, n_q_for_bandwidth(c, bandwidth), codes_out, qr, ar, dry, s));
  *q_rows = qr;
  *F_out = F;
  return LDC_OK;
}

extern "C" int ldc_get_cond(ldc_ctx* c, const float* wav, int B, int T, float bandwidth, float* cond_out, int64_t* codes_out,
                            void* stream) {
  LDCCHK(check_ready(c, LDC_MODEL_COND));
  if (!wav || !cond_out || B <= 0 || T <= 0) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  LDCCHK(with_scratch(c, s, [&](Arena& ar, bool dry) -> int {
    float* qr = nullptr;
    int F = 0;
    LDCCHK(get_cond_rows(c, wav, B, T, bandwidth, ar, dry, s, &qr, &F, codes_out));
    if (!dry) HIPCHK(launch_from_cl(DT_F32, qr, cond_out, B, c->cfg.rep_dims, F, nullptr, 0, 0.f, s));
    return LDC_OK;
  }));
  return finish_stream(c, stream);
}

// ------------------------------------------------------------------------------------------------
// cond upsampler (fp32): rows [B*F][C] -> rows [B*L][C]
// ------------------------------------------------------------------------------------------------
static int upsample_rows(ldc_ctx* c, const void* rows_in, int B, int F, Arena& ar, bool dry, hipStream_t s, void** out,
                         int* L_out) {
  const UnetW& u = c->unet;
  const void* x = rows_in;
  int L = F;
  for (const ConvLayer& ly : u.upsamplers) {
    ConvCall cc;
    cc.B = B; cc.L_in = L; cc.L_rows = L + 1; cc.L_final = L * ly.tr_stride; cc.y_ld = ly.tr_cout; cc.x1 = x; cc.tune = &c->tune;
    void* y = ar.alloc((size_t)B * cc.L_final * ly.tr_cout * 4);
    cc.y = y;
    if (!dry) HIPCHK(launch_conv(ly, cc, s));
    x = y;
    L = cc.L_final;
  }
  *out = const_cast<void*>(x);
  *L_out = L;
  return LDC_OK;
}

int upsample_factor(const ldc_ctx* c) {
  int f = 1;
  for (int r : c->unet.up_ratios) f *= r;
  return f;
}

extern "C" int ldc_cond_upsample(ldc_ctx* c, const float* cond, int B, int F, int normalise, float* img_out, void* stream) {
  LDCCHK(check_ready(c, LDC_MODEL_MAIN));
  if (!cond || !img_out || B <= 0 || F <= 0 || normalise < 0 || normalise > 2) return fail(LDC_E_INVALID, "bad arguments");
  hipStream_t s = pick_stream(c, stream);
  const int C = c->unet.cond_channels;
  LDCCHK(with_scratch(c, s, [&](Arena& ar, bool dry) -> int {
    void* rows = ar.alloc((size_t)B * F * C * 4);
    float* mx = (float*)ar.alloc((size_t)B * 4);
    if (!dry) HIPCHK(launch_to_cl(DT_F32, cond, rows, B, C, F, nullptr, 0, 0.f, s));
    void* up = nullptr;
    int L = 0;
    LDCCHK(upsample_rows(c, rows, B, F, ar, dry, s, &up, &L));
    if (!dry) {
      if (normalise) {
        HIPCHK(hipMemsetAsync(mx, 0, (size_t)B * 4, s));
        HIPCHK(launch_maxabs(DT_F32, up, B, (int64_t)L * C, normalise == 2, mx, s));
        HIPCHK(launch_from_cl(DT_F32, up, img_out, B, C, L, mx, normalise == 2, 1e-8f, s));
      } else {
        HIPCHK(launch_from_cl(DT_F32, up, img_out, B, C, L, nullptr, 0, 0.f, s));
      }
    }
    return LDC_OK;
  }));
  return finish_stream(c, stream);
}

// ------------------------------------------------------------------------------------------------
// UNet plan
// ------------------------------------------------------------------------------------------------
struct PlanBuilder {
  ldc_ctx* c;
  Plan* pl;
  Arena* ar;
  int B;
  size_t es;   // element size of the UNet dtype
  float* stats_pool = nullptr;   // [n_gn][B][groups][2]
  int stats_used = 0;
  char* part_pool = nullptr;     // granule regions of the fused GroupNorm applies (inside the region the step's first kernel clears)
  size_t part_used = 0, part_cap = 0;
  float* linattn_ws = nullptr;
  int linattn_used = 0;
  float* sk_part = nullptr;        // split-K workspace shared by the plan's convs (they run one after another)
  unsigned* sk_count = nullptr;    // arrival counters, inside the region the step's memset clears
  long long sk_part_cap = 0;
  int sk_count_cap = 0;

  void* act(int rows, int C) {
    pl->act_bytes += (double)rows * C * es;
    return ar->alloc((size_t)rows * C * es);
  }
  int where = 0;   // stream selector for the ops being added (0 main, 1 side)
  void mark(int kind) {   // 2 = fork (side waits for main), 3 = join (main waits for side)
    pl->step_ops.push_back([](hipStream_t) { return hipSuccess; });
    pl->step_is_conv.push_back(0);
    pl->step_where.push_back(kind);
    pl->step_flops.push_back(0);
    pl->step_class.push_back(LDC_CLASS_OTHER);
    pl->step_bytes.push_back(0);
    pl->step_info.push_back("-");
  }
  void add(std::function<hipError_t(hipStream_t)> f, bool is_conv = false, double flops = 0, int cls = LDC_CLASS_OTHER,
           double bytes = 0) {
    pl->step_ops.push_back(std::move(f));
    pl->step_where.push_back(where);
    pl->step_is_conv.push_back(is_conv ? 1 : 0);
    pl->step_flops.push_back(flops);
    pl->step_class.push_back(is_conv ? LDC_CLASS_CONV : cls);
    pl->step_bytes.push_back(bytes);
    pl->step_info.push_back(info.empty() ? std::string("-") : info);
    info.clear();
    pl->flops += flops;
  }
  std::string info;   // description of the next op added
  void* y2_next = nullptr;   // second output of the next conv added (a layer with a folded 1x1 conv)
  float* qkv_ctx_next = nullptr;   // context workspace of the next conv added (to_qkv with the context fold)
  int qkv_ctx_stride_next = 0;
  const float* ln_rowstat_next = nullptr;   // row-statistics partials of the next (LayerNorm-folded) conv's input
  bool want_rowstat = false;                // the next resnet()'s fused block2 conv leaves those partials of its output ...
  float* last_rowstat = nullptr;            // ... here (null when it could not)
  // fused GroupNorm apply of a conv (see ConvCall::gn_cnt)
  struct GnEpi {
    void* part = nullptr;          // granule region of this conv
    int mslots = 0;
    float* rowstat = nullptr;      // with a residual: per-row (sum, sumsq) partials of the output for a LayerNorm folded into the consumer
    const float* gamma = nullptr; const float* beta = nullptr; const float* ss = nullptr;
    int out = 0;
  };
  // rows per tile the pipelined kernel would use for this conv (0: generic kernel) -- the fused apply needs L_out >= that
  // (out[0] = rows per tile, out[1] = wave rows, out[2] = split-K factor)
  void conv_bm(const ConvLayer& ly, int L_in, int L_out, bool with_stats, int* out) {   // out: int[4]
    ConvCall d;
    d.B = B; d.L_in = L_in; d.L_rows = L_out; d.y_ld = ly.n; d.tune = &c->tune;
    d.sk_part = sk_part; d.sk_count = sk_count; d.sk_part_cap = sk_part_cap; d.sk_count_cap = sk_count_cap;
    if (with_stats) { d.gn_sum = stats_pool; d.gn_groups = c->unet.groups; }
    long long need = 0;
    out[0] = out[1] = 0; out[2] = 1; out[3] = 0;
    d.sk_need = &need; d.bm_out = out;
    // (a layer with a folded second conv is refused without its second output -- conv_kargs -- and the refusal reads as "generic kernel, no
    // fused epilogue": since round 5 asked for the folded layer's own tile shape, the nine folded block1 convs of a step had silently gone
    // back to conv + gn_apply pairs.  Nothing is written in a dry run.)
    if (ly.wtaps) d.y2 = reinterpret_cast<void*>(16);
    (void)launch_conv(ly, d, nullptr);
  }
  // a granule region for a fused conv with tile height bm and wm wave rows; null when the pool is exhausted (first planning pass: sizes only)
  bool take_part(GnEpi* ge, int L, int bm, int wm, int n) {
    ge->mslots = (L + bm - 1) / bm + 1;
    const size_t bytes = (size_t)B * ge->mslots * wm * (n / 32) * 16;
    pl->part_need += bytes;
    if (!part_pool || part_used + bytes > part_cap) return false;
    ge->part = part_pool + part_used;
    part_used += bytes;
    return true;
  }
  void conv(const ConvLayer& ly, const void* x1, const void* x2, void* y, const void* residual, int L_in, int L_out,
            float* gn_sum = nullptr, unsigned* colmax = nullptr, int cm_lo = 0, int cm_hi = 0, int cm_stride = 0,
            const GnEpi* ge = nullptr) {
    ConvCall cc;
    cc.B = B; cc.L_in = L_in; cc.L_rows = L_out; cc.x1 = x1; cc.x2 = x2; cc.y = y; cc.residual = residual; cc.y_ld = ly.n;
    cc.gn_sum = gn_sum; cc.gn_groups = gn_sum ? c->unet.groups : 0;
    cc.y2 = y2_next; y2_next = nullptr;
    cc.qkv_ctx_ws = qkv_ctx_next; cc.qkv_ctx_stride = qkv_ctx_stride_next; qkv_ctx_next = nullptr; qkv_ctx_stride_next = 0;
    cc.ln_rowstat = ly.ln_s ? ln_rowstat_next : nullptr; ln_rowstat_next = nullptr;
    if (pl->kst && pl->step_ops.size() < (size_t)kKstOps) {
      cc.kst = pl->kst + pl->step_ops.size() * 2; cc.kst_stride = kKstOps * 2; cc.kst_step = pl->step_state;
    }
    if (ge && ge->part) {
      cc.gn_groups = c->unet.groups;
      cc.rowstat_out = residual ? ge->rowstat : nullptr;
      cc.gn_part = ge->part; cc.gn_mslots = ge->mslots; cc.gn_gamma = ge->gamma; cc.gn_beta = ge->beta; cc.gn_ss = ge->ss; cc.gn_out = ge->out;
      cc.fail_flag = c->dev_flag_dev;
    }
    cc.colmax = colmax; cc.colmax_lo = cm_lo; cc.colmax_hi = cm_hi; cc.colmax_stride = cm_stride;
    cc.sk_part = sk_part; cc.sk_count = sk_count; cc.sk_part_cap = sk_part_cap; cc.sk_count_cap = sk_count_cap;
    cc.tune = &c->tune;
    {   // dry run of the launcher: how much split-K workspace would this conv use?
      ConvCall d = cc;
      long long need = 0;
      d.sk_need = &need;
      (void)launch_conv(ly, d, nullptr);
      pl->sk_need_max = std::max(pl->sk_need_max, need);
    }
    const ConvLayer* lp = &ly;
    {
      char buf[96];
      snprintf(buf, sizeof(buf), "k%d%s_s%d_u%d_c%d+%d->%d_L%d%s%s", ly.taps, ly.wtaps ? "+res" : "", ly.stride, ly.ups, ly.cin1, ly.cin2, ly.n, L_out,
               (ge && ge->part) ? (residual ? "_gnapply+res" : "_gnapply") : (gn_sum ? "_gn" : ""), colmax ? "_kmax" : (cc.qkv_ctx_ws ? "_ctx" : ""));
      info = buf;
    }
    const double cbytes = ((double)B * L_in * (ly.cin1 + ly.cin2) + (double)B * L_out * ly.n * (((ge && ge->part && residual) ? 2 : 1) + (ly.wtaps ? 1 : 0))) * es + (double)conv_packed_weight_bytes(ly);
    pl->conv_bytes += cbytes;
    add([lp, cc](hipStream_t s) { return launch_conv(*lp, cc, s); }, true, ly.flops_per_row * (double)B * L_out, LDC_CLASS_CONV, cbytes);
  }
  // zeroed-every-step bytes from the granule pool (nullptr while the pool is being sized)
  void* take_raw(size_t bytes) {
    bytes = (bytes + 63) / 64 * 64;
    pl->part_need += bytes;
    if (!part_pool || part_used + bytes > part_cap) return nullptr;
    void* p = part_pool + part_used;
    part_used += bytes;
    return p;
  }
  float* next_stats() {
    const int g = c->unet.groups;
    return stats_pool + (size_t)(stats_used++) * B * g * kGnPad;
  }
  // ResnetBlock.forward (unet.py:176-192)
  // ln_g != null: the block's last kernel also writes LayerNorm(out) * ln_g to *xn_out (the PreNorm of the attention
  // block that consumes `out`)
  // out_mode (final ResnetBlock only): bit 2 = write tanh(out) (unet.py:467), bit 0 = in fp8 (its only consumer is final_conv)
  void* resnet(const ResnetW& r, const void* x1, const void* x2, int L, const float* ln_g = nullptr, void** xn_out = nullptr,
               bool xn_fp8 = false, int out_mode = 0) {
    const int rows = B * L, dt = c->dt, g = c->unet.groups, Bn = B;
    const UnetW* u = &c->unet;
    const float* cur_ss = pl->cur_ss;
    // fp8-weight context: block1's output feeds block2's conv alone, so it is produced in fp8 and that conv runs fp8 x fp8
    const bool f8 = c->w8 && c->fp8_act && r.c2_f8.w != nullptr;
    void* b = f8 ? ar->alloc((size_t)rows * r.cout) : act(rows, r.cout);
    void* out = (out_mode & 1) ? ar->alloc((size_t)rows * r.cout) : act(rows, r.cout);
    float* st1 = next_stats();
    float* st2 = next_stats();
    const int cpg = r.cout / g;
    const bool fuse_stats = c->fuse_gn_stats && cpg >= 4 && (cpg & (cpg - 1)) == 0;
    const bool rowstat_wanted = want_rowstat;
    want_rowstat = false;
    last_rowstat = nullptr;
    // GroupNorm apply inside the conv epilogue (in-launch per-item wait): the tile height must not exceed twice an item's rows
    // (a tile then straddles at most three items), and the output must be in the UNet dtype (fp8 outputs keep gn_apply)
    const bool epi_ok = c->fuse_gn_epi && cpg % 32 == 0 && r.cout % 32 == 0 && L >= c->gn_epi_min_l;
    bool epi1 = false, epi2 = false;
    GnEpi ge1, ge2;
    if (epi_ok) {
      int t1[4], t2[4];
      conv_bm(r.c1, L, L, false, t1);
      // (the tile shape of the layer that will be LAUNCHED: with res_conv folded in -- decided below by the same rule -- that is c1r,
      // which the 256 x 64 tiles do not take)
      if (r.has_res && c->fold_res && r.c1r.w && t1[0] > 0 && t1[2] == 1) conv_bm(r.c1r, L, L, false, t1);
      // Two gates on the tiles the dry run reports (BM x BN): (i) the launch as a whole stays within gn_epi_max_tiles (a performance gate:
      // beyond two rounds of workgroups the waiting tiles cost more than the gn_apply launch they replace); (ii) a SAFETY gate per item --
      // every tile spins until all tiles of its item(s) have published, and the dispatch order interleaves the items of a group of 8 M tiles,
      // so an item's tiles plus two dispatch groups (the one it sits in, the one a straddling tile reaches into) must be resident together
      // even while the other batch parts' launches hold their share of the chip (three workgroups of these kernels fit a CU: 41-51 KB of
      // ring, <= 168 registers; the parts run concurrently).  A long single file (sample.py's whole-file mode: B = 1, L = samples / hop)
      // fails (ii) and takes the conv + gn_apply pair instead of stalling every fused launch for its time-out.
      // The slots come from the occupancy query of the kernels these convs land on x the device's CUs (round 6; before: a constant 3 x 256).
      const int concurrent = std::max(2, c->split_batch);
      const long resident_slots = (long)(c->w8 ? std::min(conv_fused_gn_wgs_per_cu(DT_BF16, true), conv_fused_gn_wgs_per_cu(DT_FP8, false)) : conv_fused_gn_wgs_per_cu(dt, false)) *
                                  c->num_cus / concurrent * 8 / 10;
      auto few_tiles = [&](const int* t) {
        if (t[0] <= 0 || t[3] <= 0) return false;
        const long ntn = (r.cout + t[3] - 1) / t[3];
        const long launch_tiles = (long)((rows + t[0] - 1) / t[0]) * ntn;
        const long item_tiles = (long)((L + t[0] - 1) / t[0] + 1) * ntn;
        return launch_tiles <= c->gn_epi_max_tiles && (launch_tiles <= resident_slots || item_tiles + 16 * ntn <= resident_slots);
      };
      epi1 = (!f8 || dt == DT_BF16) && few_tiles(t1) && t1[0] <= 2 * L && take_part(&ge1, L, t1[0], t1[1], r.cout);   // f8: block1's output in fp8
      conv_bm(f8 ? r.c2_f8 : r.c2, L, L, false, t2);
      epi2 = (!(out_mode & 1) || dt == DT_BF16) && few_tiles(t2) && t2[0] <= 2 * L && take_part(&ge2, L, t2[0], t2[1], r.cout);
    }
    void* a = epi1 ? nullptr : act(rows, r.cout);   // un-normalised conv outputs exist only on the unfused path
    void* d = epi2 ? nullptr : act(rows, r.cout);
    // res_conv: folded into block1's conv where that conv runs on the pipelined kernel with an unsplit K (one launch and one
    // read of x less); otherwise its own launch (on the side stream when there is one: it only feeds the final add)
    const void* res = x1;
    bool folded = false;
    void* rr = r.has_res ? act(rows, r.cout) : nullptr;
    if (r.has_res && c->fold_res && r.c1r.w) {
      int t[4];
      conv_bm(r.c1, L, L, false, t);
      folded = t[0] > 0 && t[2] == 1;
    }
    if (r.has_res && !folded) {
      mark(2);
      where = 1;
      conv(r.res, x1, x2, rr, nullptr, L, L);
      where = 0;
    }
    if (r.has_res) res = rr;
    const ConvLayer& c1 = folded ? r.c1r : r.c1;
    const ResnetW* rp = &r;
    if (epi1) {   // block1: conv -> GroupNorm -> (scale + 1, shift) -> SiLU, one launch, one store
      ge1.gamma = r.g1; ge1.beta = r.b1; ge1.ss = cur_ss + r.ss_off; ge1.out = f8 ? 1 : 0;
      if (folded) y2_next = rr;
      conv(c1, x1, x2, b, nullptr, L, L, nullptr, nullptr, 0, 0, 0, &ge1);
    } else {
      if (folded) y2_next = rr;
      conv(c1, x1, x2, a, nullptr, L, L, fuse_stats ? st1 : nullptr);
      if (!fuse_stats) add([=](hipStream_t s) { return launch_gn_stats(dt, a, Bn, L, rp->cout, g, st1, s); });
      add([=](hipStream_t s) {
        return launch_gn_apply(dt, a, b, nullptr, Bn, L, rp->cout, g, st1, rp->g1, rp->b1, cur_ss + rp->ss_off,
                               0, nullptr, ACT_SILU, s, nullptr, nullptr, f8 ? 1 : 0);
      }, false, 0, LDC_CLASS_GN_APPLY, (f8 ? 1.5 : 2.0) * Bn * L * rp->cout * es);
    }
    void* xn = nullptr;
    if (ln_g && xn_out && c->fuse_ln && gn_apply_ln_fusable(r.cout)) {
      xn = xn_fp8 ? ar->alloc((size_t)rows * r.cout) : act(rows, r.cout);
      *xn_out = xn;
    }
    if (epi2) {   // block2: conv -> GroupNorm -> SiLU -> + res (-> tanh), one launch; the PreNorm LayerNorm of a following attention block reads `out`
      ge2.gamma = r.g2; ge2.beta = r.b2; ge2.ss = nullptr; ge2.out = out_mode & 5;
      if (rowstat_wanted && !f8) ge2.rowstat = last_rowstat = (float*)ar->alloc((size_t)rows * (r.cout / 32) * 8);
      if (r.has_res && !folded) mark(3);
      conv(f8 ? r.c2_f8 : r.c2, b, nullptr, out, res, L, L, nullptr, nullptr, 0, 0, 0, &ge2);
      if (xn) {
        const int C = r.cout;
        add([=](hipStream_t s) { return launch_ln_rows(dt, out, xn, nullptr, ln_g, rows, C, s, xn_fp8 ? 1 : 0); }, false, 0, LDC_CLASS_LAYERNORM,
            (xn_fp8 ? 1.5 : 2.0) * rows * C * es);
      }
      return out;
    }
    conv(f8 ? r.c2_f8 : r.c2, b, nullptr, d, nullptr, L, L, fuse_stats ? st2 : nullptr);
    if (!fuse_stats) add([=](hipStream_t s) { return launch_gn_stats(dt, d, Bn, L, rp->cout, g, st2, s); });
    if (r.has_res && !folded) mark(3);
    const int out8_ln = ((xn && xn_fp8) ? 2 : 0) | (gn_apply_fp8_ok(r.cout) ? out_mode : 0);
    add([=](hipStream_t s) {
      return launch_gn_apply(dt, d, out, res, Bn, L, rp->cout, g, st2, rp->g2, rp->b2, nullptr, 0, nullptr, ACT_SILU, s, xn, ln_g, out8_ln);
    }, false, 0, LDC_CLASS_GN_APPLY, (xn ? 4.0 : 3.0) * Bn * L * rp->cout * es);
    return out;
  }
  bool qkv_fp8(const LinAttnW& a) const { return c->w8 && c->fp8_act && a.qkv_f8.w != nullptr; }
  // the attention block's PreNorm can ride inside to_qkv: the folded layer exists, runs on the pipelined kernel, and the ResnetBlock
  // in front fuses its GroupNorm apply (else its gn_apply launch writes the LayerNorm output on the way at no extra launch)
  bool ln_foldable(const LinAttnW& a, int L) {
    if (!c->fold_ln || !c->fuse_gn_epi || c->w8 || !a.qkv_ln.w || L < c->gn_epi_min_l) return false;
    int t[4];
    conv_bm(a.qkv_ln, L, L, false, t);
    return t[0] > 0;
  }
  // Residual(PreNorm(LinearAttention)) (unet.py:208-222) / Residual(PreNorm(Attention)) (:234-246)
  void* attention(const LinAttnW& a, const void* x, int L, bool linear, void* xn_pre = nullptr) {
    const int rows = B * L, dt = c->dt, Bn = B;
    const int H = c->unet.heads, Dh = c->unet.dim_head, hid = H * Dh;
    const bool f8 = qkv_fp8(a);
    const bool lnf = !xn_pre && ln_foldable(a, L);     // LayerNorm inside to_qkv: the conv reads x itself
    const float* rowstat = lnf ? last_rowstat : nullptr;   // (sum, sumsq) partials of x's rows left by the ResnetBlock in front, if it could
    last_rowstat = nullptr;
    void* xn = lnf ? const_cast<void*>(x) : (xn_pre ? xn_pre : (f8 ? ar->alloc((size_t)rows * a.dim) : act(rows, a.dim)));
    const ConvLayer& qkv_ly = lnf ? a.qkv_ln : (f8 ? a.qkv_f8 : a.qkv);
    void* qkv = act(rows, 3 * hid);
    void* o = act(rows, hid);
    void* out = act(rows, a.dim);
    const LinAttnW* ap = &a;
    // one workspace per LinearAttention layer when the k column-max is fused: all of them are zeroed by the
    // step's single memset (they sit behind the GroupNorm statistics)
    float* ws = (linear && c->fuse_kmax) ? linattn_ws + (size_t)(linattn_used++) * B * linattn_ws_floats_per_item(H, Dh) : linattn_ws;
    if (!xn_pre && !lnf)
      add([=](hipStream_t s) { return launch_ln_rows(dt, x, xn, nullptr, ap->norm_g, rows, ap->dim, s, f8 ? 1 : 0); }, false, 0, LDC_CLASS_LAYERNORM,
          2.0 * rows * a.dim * es);
    if (linear) {
      const size_t wss = linattn_ws_floats_per_item(H, Dh);
      if (c->fuse_kmax && c->fuse_attn_tail && !a.out.w8 && linattn_tail_supported(dt, H, Dh, a.dim)) {
        // three launches: qkv conv (+ k column max) -> context -> tail (out, to_out conv, LayerNorm, + x); round 6: TWO where to_qkv
        // accumulates the context itself (lean kernel, bf16, folded PreNorm: ConvCall::qkv_ctx_ws)
        ln_rowstat_next = rowstat;
        if (lnf && c->fold_ctx && c->tune.lean && !c->tune.force_generic && dt == DT_BF16 && a.qkv_ctx.w && hid == 128 && a.dim % 64 == 0) {
          qkv_ctx_next = ws; qkv_ctx_stride_next = (int)wss;
          conv(a.qkv_ctx, xn, nullptr, qkv, nullptr, L, L);
        } else {
          conv(qkv_ly, xn, nullptr, qkv, nullptr, L, L, nullptr, reinterpret_cast<unsigned*>(ws), hid, 2 * hid, (int)wss);
          add([=](hipStream_t s) { return launch_linattn_ctx(dt, qkv, ws, Bn, L, H, Dh, s); }, false, 0, LDC_CLASS_LINATTN, 2.0 * rows * hid * es);
        }
        const int dim = a.dim;
        info = "tail_c" + std::to_string(dim) + "_L" + std::to_string(L) + "_B" + std::to_string(B);
        add([=](hipStream_t s) {
          return launch_linattn_tail(dt, qkv, ws, ap->out.w, ap->out.n_pad, ap->out.bias, ap->out_g, x, out, Bn, L, H, Dh, dim, s);
        }, false, 2.0 * rows * hid * dim, LDC_CLASS_LINATTN, (1.0 * hid + 2.0 * dim) * rows * es);
        return out;
      }
      if (c->fuse_kmax) {
        ln_rowstat_next = rowstat;
        conv(qkv_ly, xn, nullptr, qkv, nullptr, L, L, nullptr, reinterpret_cast<unsigned*>(ws), hid, 2 * hid, (int)wss);
        add([=](hipStream_t s) { return launch_linattn(dt, qkv, o, ws, Bn, L, H, Dh, true, s); }, false, 0, LDC_CLASS_LINATTN,
            4.0 * rows * hid * es);
      } else {
        ln_rowstat_next = rowstat;
        conv(qkv_ly, xn, nullptr, qkv, nullptr, L, L);
        add([=](hipStream_t s) { return launch_linattn(dt, qkv, o, ws, Bn, L, H, Dh, false, s); }, false, 0, LDC_CLASS_LINATTN,
            5.0 * rows * hid * es);
      }
      void* t = act(rows, a.dim);
      conv(a.out, o, nullptr, t, nullptr, L, L);
      add([=](hipStream_t s) { return launch_ln_rows(dt, t, out, x, ap->out_g, rows, ap->dim, s); }, false, 0, LDC_CLASS_LAYERNORM,
          3.0 * rows * a.dim * es);
    } else {
      ln_rowstat_next = rowstat;
      conv(qkv_ly, xn, nullptr, qkv, nullptr, L, L);
      add([=](hipStream_t s) { return launch_attn_full(dt, qkv, o, Bn, L, H, Dh, s); }, false, 0, LDC_CLASS_ATTN_FULL, 4.0 * rows * hid * es);
      conv(a.out, o, nullptr, out, x, L, L);   // + x in the epilogue
    }
    return out;
  }
};

int build_plan(ldc_ctx* c, Plan* pl, Arena& ar, int B, int L, int F) {
  const UnetW& u = c->unet;
  pl->B = B; pl->L = L; pl->F = F;
  pl->cond_ops.clear(); pl->step_ops.clear(); pl->step_is_conv.clear(); pl->step_where.clear(); pl->step_flops.clear(); pl->step_class.clear(); pl->step_bytes.clear(); pl->step_info.clear(); pl->taps.clear();
  pl->flops = 0; pl->act_bytes = 0; pl->conv_bytes = 0; pl->sk_need_max = 0; pl->part_need = 0;
  const int dt = c->dt;
  const size_t es = dt_size(dt);
  const int Cc = u.cond_channels, Cx = u.channels;
  PlanBuilder pb{c, pl, &ar, B, es};
  pl->zero_once.clear();
  const int n_gn = 2 * (int)(2 * u.downs.size() + 2 + 2 * u.ups.size() + 1);
  const size_t gn_bytes = (size_t)n_gn * B * u.groups * kGnPad * 4;
  const size_t n_lin = c->fuse_kmax ? u.downs.size() + u.ups.size() : 1;
  const size_t lin_bytes = n_lin * B * linattn_ws_floats_per_item(u.heads, u.dim_head) * 4;
  const int sk_tiles_cap = 1024;
  const size_t part_bytes = (pl->part_bytes + 63) / 64 * 64;   // granule regions of the fused GroupNorm applies (sized by the dry planning pass)
  const size_t sk_count_bytes = (size_t)sk_tiles_cap * 4 + part_bytes;
  const size_t sx_bytes = c->cfg.unet_scale_x ? (size_t)B * 4 : 0;   // per-item max|cat(cond, x)| of --unet_scale_x
  pb.stats_pool = (float*)ar.alloc(gn_bytes + sk_count_bytes + lin_bytes + sx_bytes + 64);
  pb.sk_count = reinterpret_cast<unsigned*>(pb.stats_pool + gn_bytes / 4);
  pb.part_pool = reinterpret_cast<char*>(pb.sk_count + sk_tiles_cap);
  pb.part_cap = part_bytes;
  pb.sk_count_cap = sk_tiles_cap;
  pb.linattn_ws = pb.stats_pool + (gn_bytes + sk_count_bytes) / 4;
  // the region the step's ONE memset clears: GroupNorm sums, split-K counters, fused k-max keys, scale_x maxima
  float* sx_max = pb.linattn_ws + (c->fuse_kmax ? lin_bytes / 4 : 0);
  const size_t stats_bytes = gn_bytes + sk_count_bytes + (c->fuse_kmax ? lin_bytes : 0) + sx_bytes;
  pb.sk_part_cap = pl->sk_floats;   // sized from the convs' own needs by a dry planning pass (get_plan)
  pb.sk_part = (float*)ar.alloc((size_t)std::max<long long>(pb.sk_part_cap, 4) * 4);
  pl->maxabs = (float*)ar.alloc((size_t)B * 4);
  pl->step_state = (int*)ar.alloc(64);
  if (ar.base) pl->zero_once.push_back({pl->step_state, 64});   // [4] = the epoch of the UNet pass (launch_step_begin counts it up)
  pl->kst = c->kstamps ? (unsigned long long*)ar.alloc((size_t)2048 * kKstOps * 2 * 8) : nullptr;
  pl->cur_ss = (float*)ar.alloc((size_t)std::max(1, u.ss_stride) * 4);
  pl->x_cl = ar.alloc((size_t)B * L * Cx * es);
  pl->eps_cl = ar.alloc((size_t)B * L * Cx * es);
  pl->cond_cl = ar.alloc((size_t)B * L * Cc * es);
  // ---- process_cond (unet.py:407-420): fp32 upsampler, per-item max-abs, cast to the UNet dtype ----
  void* cond_in = ar.alloc((size_t)B * F * Cc * 4);
  pl->cond_in_cl = cond_in;
  {
    const void* x = cond_in;
    int Lc = F;
    for (const ConvLayer& ly : u.upsamplers) {
      ConvCall cc;
      cc.B = B; cc.L_in = Lc; cc.L_rows = Lc + 1; cc.L_final = Lc * ly.tr_stride; cc.y_ld = ly.tr_cout; cc.x1 = x; cc.tune = &c->tune;
      void* y = ar.alloc((size_t)B * cc.L_final * ly.tr_cout * 4);
      cc.y = y;
      const ConvLayer* lp = &ly;
      pl->cond_ops.push_back([lp, cc](hipStream_t s) { return launch_conv(*lp, cc, s); });
      x = y;
      Lc = cc.L_final;
    }
    if (Lc != L) return fail(LDC_E_INVALID, "upsampled condition length %d != latent length %d (F=%d, upsampling product %d)", Lc, L, F, upsample_factor(c));
    float* mx = pl->maxabs;
    void* cond_cl = pl->cond_cl;
    const bool scale = c->cfg.unet_scale_cond != 0;
    const int Bn = B;
    const int64_t npi = (int64_t)L * Cc;
    // fp32 rows -> (scaled) rows in the UNet dtype.  A [rows][C] fp32 -> dt copy is a "transpose" of a
    // [B][C=1][L*C] tensor: reuse to_cl with C=1 (pure cast + scale).
    if (scale) {
      pl->cond_ops.push_back([=](hipStream_t s) { return hipMemsetAsync(mx, 0, (size_t)Bn * 4, s); });
      pl->cond_ops.push_back([=](hipStream_t s) { return launch_maxabs(DT_F32, x, Bn, npi, 1, mx, s); });
      pl->cond_ops.push_back([=](hipStream_t s) { return launch_to_cl(dt, (const float*)x, cond_cl, Bn, 1, (int)npi, mx, 1, 1e-20f, s); });
    } else {
      pl->cond_ops.push_back([=](hipStream_t s) { return launch_to_cl(dt, (const float*)x, cond_cl, Bn, 1, (int)npi, nullptr, 0, 0.f, s); });
    }
  }
  pl->taps["cond_proc"] = {pl->cond_cl, Cc, L};
  // ---- Unet1D.forward (unet.py:430-469) ----
  pl->zero_ptr = pb.stats_pool;            // cleared by the step's first kernel (launch_step_begin): no memset node of its own
  pl->zero_bytes = stats_bytes;
  const void* in_cond = pl->cond_cl;
  const void* in_x = pl->x_cl;
  if (c->cfg.unet_scale_x) {
    // unet.py:432-433: x = cat(cond, x) / (max|.| per item + 1e-20).  The max runs over both halves of the concatenation;
    // the two scaled copies feed init_conv as its two inputs.
    void* cs = ar.alloc((size_t)B * L * Cc * es);
    void* xs = ar.alloc((size_t)B * L * Cx * es);
    const void* cin = pl->cond_cl; const void* xin = pl->x_cl;
    const int Bn = B;
    const int64_t nc = (int64_t)L * Cc, nx = (int64_t)L * Cx;
    pb.add([=](hipStream_t s) { return launch_maxabs(dt, cin, Bn, nc, 1, sx_max, s); });
    pb.add([=](hipStream_t s) { return launch_maxabs(dt, xin, Bn, nx, 1, sx_max, s); });
    pb.add([=](hipStream_t s) { return launch_scale_copy(dt, cin, cs, Bn, nc, sx_max, 1e-20f, s); }, false, 0, LDC_CLASS_ELEMENTWISE, 2.0 * B * nc * es);
    pb.add([=](hipStream_t s) { return launch_scale_copy(dt, xin, xs, Bn, nx, sx_max, 1e-20f, s); }, false, 0, LDC_CLASS_ELEMENTWISE, 2.0 * B * nx * es);
    in_cond = cs; in_x = xs;
  }
  void* x0 = pb.act(B * L, u.dim);
  if (c->split_init && !c->w8 && !c->cfg.unet_scale_x && u.init_c.w && u.init_x.w) {
    // the condition's half of init_conv once per sampler call (behind process_cond), the x half + that tensor every step
    void* pc = pb.act(B * L, u.dim);
    {
      ConvCall cc;
      cc.B = B; cc.L_in = L; cc.L_rows = L; cc.x1 = pl->cond_cl; cc.y = pc; cc.y_ld = u.init_c.n; cc.tune = &c->tune;
      const ConvLayer* lp = &u.init_c;
      pl->cond_ops.push_back([lp, cc](hipStream_t s) { return launch_conv(*lp, cc, s); });
    }
    pb.conv(u.init_x, in_x, nullptr, x0, pc, L, L);
  } else {
    pb.conv(u.init, in_cond, in_x, x0, nullptr, L, L);
  }
  pl->taps["init"] = {x0, u.dim, L};
  const void* x = x0;
  int Lc = L;
  std::vector<std::pair<const void*, int>> hs;
  for (size_t i = 0; i < u.downs.size(); ++i) {
    const LevelW& lv = u.downs[i];
    x = pb.resnet(lv.b1, x, nullptr, Lc); hs.push_back({x, Lc});
    void* xn = nullptr;
    if (pb.ln_foldable(lv.attn, Lc)) { pb.want_rowstat = true; x = pb.resnet(lv.b2, x, nullptr, Lc); }
    else x = pb.resnet(lv.b2, x, nullptr, Lc, lv.attn.norm_g, &xn, pb.qkv_fp8(lv.attn));
    x = pb.attention(lv.attn, x, Lc, true, xn); hs.push_back({x, Lc});
    int Ln = Lc;
    if (lv.kind == 0) Ln = (Lc + 2 - 4) / 2 + 1;
    void* y = pb.act(B * Ln, lv.cout);
    pb.conv(lv.resample, x, nullptr, y, nullptr, Lc, Ln);
    x = y; Lc = Ln;
    pl->taps["down" + std::to_string(i)] = {y, lv.cout, Lc};
  }
  {
    void* xn = nullptr;
    if (pb.ln_foldable(u.mid_attn, Lc)) { pb.want_rowstat = true; x = pb.resnet(u.mid1, x, nullptr, Lc); }
    else x = pb.resnet(u.mid1, x, nullptr, Lc, u.mid_attn.norm_g, &xn, pb.qkv_fp8(u.mid_attn));
    x = pb.attention(u.mid_attn, x, Lc, false, xn);
  }
  x = pb.resnet(u.mid2, x, nullptr, Lc);
  pl->taps["mid"] = {const_cast<void*>(x), u.dims.back(), Lc};
  for (size_t i = 0; i < u.ups.size(); ++i) {
    const LevelW& lv = u.ups[i];
    if (hs.back().second != Lc) return fail(LDC_E_INVALID, "latent length %d is not divisible by 2^%zu", L, u.downs.size() - 1);
    x = pb.resnet(lv.b1, x, hs.back().first, Lc); hs.pop_back();
    void* xn = nullptr;
    if (pb.ln_foldable(lv.attn, Lc)) { pb.want_rowstat = true; x = pb.resnet(lv.b2, x, hs.back().first, Lc); }
    else x = pb.resnet(lv.b2, x, hs.back().first, Lc, lv.attn.norm_g, &xn, pb.qkv_fp8(lv.attn));
    hs.pop_back();
    x = pb.attention(lv.attn, x, Lc, true, xn);
    const int Ln = lv.kind == 1 ? 2 * Lc : Lc;
    void* y = pb.act(B * Ln, lv.cout);
    pb.conv(lv.resample, x, nullptr, y, nullptr, Lc, Ln);
    x = y; Lc = Ln;
    pl->taps["up" + std::to_string(i)] = {y, lv.cout, Lc};
  }
  if (Lc != L) return fail(LDC_E_INVALID, "latent length %d does not survive the down/up path (got %d)", L, Lc);
  {
    // final ResnetBlock -> tanh -> final_conv (unet.py:465-469): the tanh is applied by the block's last kernel (one launch and
    // two passes over the tensor less), in fp8 when final_conv runs fp8 x fp8
    const bool f8 = c->w8 && c->fp8_act && u.final_conv_f8.w != nullptr;
    const bool fuse_tanh = gn_apply_fp8_ok(u.dim);
    const void* th = nullptr;
    if (fuse_tanh) {
      th = pb.resnet(u.fin, x, x0, L, nullptr, nullptr, false, 4 | (f8 ? 1 : 0));
    } else {
      x = pb.resnet(u.fin, x, x0, L);
      void* tb = f8 ? ar.alloc((size_t)B * L * u.dim) : pb.act(B * L, u.dim);
      const void* xin = x;
      const int64_t n = (int64_t)B * L * u.dim;
      pb.add([=](hipStream_t s) { return launch_act(dt, xin, tb, n, ACT_TANH, s, f8 ? 1 : 0); }, false, 0, LDC_CLASS_ELEMENTWISE, (f8 ? 1.5 : 2.0) * n * es);
      th = tb;
    }
    pb.conv(f8 ? u.final_conv_f8 : u.final_conv, th, nullptr, pl->eps_cl, nullptr, L, L);
  }
  return LDC_OK;
}

// Drop plan `idx` (and every captured graph of its (L, F)): waits for the device first, nothing of it may be running.
static void evict_plan(ldc_ctx* c, size_t idx) {
  (void)counted_device_sync();
  Plan* pl = c->plans[idx].get();
  for (size_t g = 0; g < c->graphs.size();) {
    if (c->graphs[g].L == pl->L && c->graphs[g].F == pl->F) {
      c->graphs[g].destroy();
      c->graphs.erase(c->graphs.begin() + g);
    } else {
      ++g;
    }
  }
  for (int k = 0; k < c->last_halves.n; ++k)
    if (c->last_halves.p[k] == pl) c->last_halves = Halves();
  for (hipEvent_t e : pl->marker_events) (void)hipEventDestroy(e);
  if (pl->arena_base) (void)hipFree(pl->arena_base);
  c->plan_bytes -= pl->arena_bytes;
  c->plans.erase(c->plans.begin() + idx);
}

static int get_plan(ldc_ctx* c, int B, int L, int F, int slot, hipStream_t s, Plan** out) {
  for (auto& p : c->plans)
    if (p->B == B && p->L == L && p->F == F && p->slot == slot) {
      p->last_use = ++c->use_tick;
      *out = p.get();
      return LDC_OK;
    }
  std::unique_ptr<Plan> pl(new Plan());
  {   // pass 1: what do the convs of this plan need as split-K workspace?
    Arena dry;
    LDCCHK(build_plan(c, pl.get(), dry, B, L, F));
    pl->sk_floats = pl->sk_need_max;
    pl->part_bytes = pl->part_need;
    // once more with those sizes: convs that fuse their GroupNorm apply only once the granule pool exists ask for more of it, and
    // the split-K decisions feed back into what is folded
    Arena dry2;
    LDCCHK(build_plan(c, pl.get(), dry2, B, L, F));
    pl->sk_floats = std::max(pl->sk_floats, pl->sk_need_max);
    pl->part_bytes = std::max(pl->part_bytes, pl->part_need);
  }
  Arena measure;
  LDCCHK(build_plan(c, pl.get(), measure, B, L, F));
  const size_t want = measure.off + 4096;
  // LRU eviction: plans of the call being served (last_use > call_tick) are never candidates
  for (;;) {
    const bool over = c->plan_bytes + want > c->plan_bytes_cap || (int)c->plans.size() >= c->plan_count_cap;
    if (!over) break;
    size_t victim = c->plans.size();
    for (size_t i = 0; i < c->plans.size(); ++i)
      if (c->plans[i]->last_use <= c->call_tick && (victim == c->plans.size() || c->plans[i]->last_use < c->plans[victim]->last_use)) victim = i;
    if (victim == c->plans.size()) break;   // everything cached belongs to this call: let hipMalloc decide
    evict_plan(c, victim);
  }
  void* base = nullptr;
  hipError_t e = hipMalloc(&base, want);
  if (e != hipSuccess) return fail(LDC_E_NOMEM, "hipMalloc(%zu) for the UNet workspace failed: %s", want, hipGetErrorString(e));
  Arena real;
  real.base = (char*)base;
  real.cap = want;
  int rc = build_plan(c, pl.get(), real, B, L, F);
  if (rc != LDC_OK) { (void)hipFree(base); return rc; }
  for (const auto& z : pl->zero_once) {   // the step state's epoch word: cleared once, before the plan's first use
    hipError_t ez = hipMemsetAsync(z.first, 0, z.second, s);
    if (ez != hipSuccess) { (void)hipFree(base); return fail(LDC_E_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(ez)); }
  }
  pl->arena_base = base;
  pl->arena_bytes = want;
  pl->slot = slot;
  pl->last_use = ++c->use_tick;
  c->plan_bytes += want;
  *out = pl.get();
  c->plans.push_back(std::move(pl));
  (void)s;
  return LDC_OK;
}

// The batch parts run as chains on streams of this context next to the caller's stream s.  HIP serves a process's streams from four
// hardware queues and two streams that share one run their graphs back to back (measured: 229 instead of 143 ms per decode with part 1 on
// such a stream; which streams share depends on what else the process created before -- torch's pool, RCCL, other contexts).  So the part
// streams are CHOSEN, once per caller stream: a candidate is accepted when a 150 us single-workgroup spin on it and on every stream
// accepted so far takes one spin, not two.  ~2 ms, the first time a batch is decoded in parts on a stream; never inside a capture.
static int calibrate_part_streams(ldc_ctx* c, hipStream_t s) {
  if (!c->calib || (c->calibrated && c->calib_stream == s)) return LDC_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return LDC_OK; }
  std::vector<hipStream_t*> cand;
  for (int k = 1; k < kMaxParts; ++k) cand.push_back(&c->aux_stream[k]);
  for (int k = 0; k < kMaxParts; ++k) cand.push_back(&c->side_stream[k]);
  for (const auto& e : c->calib_cache)
    if (e.s == s && e.order.size() == cand.size()) {   // measured before against this caller stream: re-apply, no synchronisation
      for (size_t i = 0; i < cand.size(); ++i) *cand[i] = e.order[i];
      c->calib_good = e.good; c->calib_cand = e.cand; c->calib_one_ms = e.one_ms; c->calib_all_ms = e.all_ms;
      c->calib_stream = s; c->calibrated = true;
      return LDC_OK;
    }
  std::vector<hipStream_t> chosen{s};
  auto timed = [&](const std::vector<hipStream_t>& set, double* ms) -> int {
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {   // (best of four: host wall time, a loaded host inflates single samples)
      for (hipStream_t q : set) HIPCHK(hipStreamSynchronize(q));
      const auto t0 = std::chrono::steady_clock::now();
      for (hipStream_t q : set) HIPCHK(launch_spin_us(150, q));
      for (hipStream_t q : set) HIPCHK(hipStreamSynchronize(q));
      best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    *ms = best;
    return LDC_OK;
  };
  double base = 0;
  LDCCHK(timed(chosen, &base));   // (also loads the kernel)
  LDCCHK(timed(chosen, &base));
  std::vector<hipStream_t> good, rest;
  for (hipStream_t* p : cand) {
    if ((int)good.size() >= kMaxParts - 1) { rest.push_back(*p); continue; }
    std::vector<hipStream_t> set = chosen;
    set.push_back(*p);
    double ms = 0;
    LDCCHK(timed(set, &ms));
    if (ms < base + 0.09) { good.push_back(*p); chosen.push_back(*p); }   // one 0.15 ms spin more = a shared queue
    else rest.push_back(*p);
  }
  if (getenv("LDC_VERBOSE")) fprintf(stderr, "[ldc] part streams: %zu of %zu candidates overlap with the caller's stream and each other (one spin: %.3f ms)\n", good.size(), cand.size(), base);
  std::vector<hipStream_t> order = good;
  order.insert(order.end(), rest.begin(), rest.end());
  for (size_t i = 0; i < cand.size(); ++i) *cand[i] = order[i];
  double all_ms = base;
  if (!good.empty()) LDCCHK(timed(chosen, &all_ms));   // the caller's stream and every accepted stream spinning together: one spin when they overlap
  c->calib_good = (int)good.size(); c->calib_cand = (int)cand.size(); c->calib_one_ms = base; c->calib_all_ms = all_ms;
  if (c->calib_cache.size() >= 8) c->calib_cache.erase(c->calib_cache.begin());
  c->calib_cache.push_back({s, order, c->calib_good, c->calib_cand, base, all_ms});
  c->calib_stream = s;
  c->calibrated = true;
  return LDC_OK;
}

static int get_halves(ldc_ctx* c, int B, int L, int F, hipStream_t s, Halves* h) {
  *h = Halves();
  c->call_tick = c->use_tick;   // plans touched from here on belong to the call being served (not evictable)
  // Independent chains the batch is decoded as (utterances never interact inside the UNet).  A chain is a
  // dependent sequence of ~155 launches per step whose cost is mostly per-launch floor, so chains side by side hide each
  // other's floors.  Measured on one box (32 x 2.4 s): 2 x 16 items 503 audio-s/s, 3 chains 505, 4 x 8 items 510 -- but
  // the convs of a 4 x 8 decode run at 209 instead of 344 TFLOP/s per launch, so two chains stay the default (LDC_SPLIT).
  h->n = std::max(1, std::min(c->split_batch, B));
  if (h->n >= 2) LDCCHK(calibrate_part_streams(c, s));
  for (int k = 0; k < h->n; ++k) {
    const int lo = (int)((long long)B * k / h->n), hi = (int)((long long)B * (k + 1) / h->n);
    h->b0[k] = lo;
    LDCCHK(get_plan(c, hi - lo, L, F, k, s, &h->p[k]));
  }
  c->last_halves = *h;
  return LDC_OK;
}

static int run_ops(ldc_ctx* c, Plan* pl, const std::vector<std::function<hipError_t(hipStream_t)>>& ops, bool is_step,
                   hipStream_t s) {
  const bool use_side = is_step && !c->profile && c->side_streams;
  hipStream_t side = c->side_stream[pl->slot];
  if (use_side && pl->marker_events.empty()) {
    size_t nm = 0;
    for (int w : pl->step_where) nm += (w >= 2);
    for (size_t k = 0; k < nm; ++k) {
      hipEvent_t e = nullptr;
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      pl->marker_events.push_back(e);
    }
  }
  size_t marker = 0;
  for (size_t i = 0; i < ops.size(); ++i) {
    const bool prof = c->profile && is_step;
    const int where = is_step ? pl->step_where[i] : 0;
    if (where == 2) {
      if (use_side) {
        hipEvent_t e = pl->marker_events[marker];
        HIPCHK(hipEventRecord(e, s));
        HIPCHK(hipStreamWaitEvent(side, e, 0));
      }
      ++marker;
      continue;
    }
    if (where == 3) {
      if (use_side) {
        hipEvent_t e = pl->marker_events[marker];
        HIPCHK(hipEventRecord(e, side));
        HIPCHK(hipStreamWaitEvent(s, e, 0));
      }
      ++marker;
      continue;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof) {
      HIPCHK(hipEventCreate(&e0));
      HIPCHK(hipEventCreate(&e1));
      HIPCHK(hipEventRecord(e0, s));
    }
    {
      const hipError_t oe = ops[i]((use_side && where == 1) ? side : s);
      if (oe != hipSuccess) {
        (void)hipGetLastError();
        return fail(LDC_E_HIP, "%s op %zu (%s) failed: %s", is_step ? "step" : "cond", i, (is_step && i < pl->step_info.size()) ? pl->step_info[i].c_str() : "-", hipGetErrorString(oe));
      }
    }
    if (prof) {
      HIPCHK(hipEventRecord(e1, s));
      c->prof_events.push_back({e0, e1});
      c->prof_event_flops.push_back(pl->step_flops[i]);
      c->prof_event_class.push_back(pl->step_class[i]);
      c->prof_event_bytes.push_back(pl->step_bytes[i]);
      c->prof_event_info.push_back(pl->step_info[i] + "_B" + std::to_string(pl->B));
    }
  }
  return LDC_OK;
}

static int load_cond(ldc_ctx* c, const Halves& h, const float* cond, hipStream_t s) {
  for (int k = 0; k < h.n; ++k) {
    Plan* pl = h.p[k];
    const float* src = cond + (size_t)h.b0[k] * c->unet.cond_channels * pl->F;
    HIPCHK(launch_to_cl(DT_F32, src, pl->cond_in_cl, pl->B, c->unet.cond_channels, pl->F, nullptr, 0, 0.f, s));
    LDCCHK(run_ops(c, pl, pl->cond_ops, false, s));
  }
  return LDC_OK;
}

static int load_x(ldc_ctx* c, const Halves& h, const float* x, hipStream_t s) {
  for (int k = 0; k < h.n; ++k) {
    Plan* pl = h.p[k];
    HIPCHK(launch_to_cl(c->dt, x + (size_t)h.b0[k] * c->unet.channels * pl->L, pl->x_cl, pl->B, c->unet.channels, pl->L, nullptr,
                        0, 0.f, s));
  }
  return LDC_OK;
}

int check_unet_args(ldc_ctx* c, int B, int L, int F) {
  if (B <= 0 || L <= 0 || F <= 0) return fail(LDC_E_INVALID, "bad sizes");
  // upsampling_ratios=None (unet.py:411): process_cond only scales, so the condition must already have the latent length
  if (F * upsample_factor(c) != L) return fail(LDC_E_INVALID, "L (%d) must equal F (%d) x prod(upsampling_ratios) (%d)", L, F, upsample_factor(c));
  return LDC_OK;
}

extern "C" int ldc_unet_forward(ldc_ctx* c, const float* x, int t, const float* cond, int B, int L, int F, float* eps_out,
                                void* stream) {
  LDCCHK(check_ready(c, LDC_MODEL_MAIN));
  if (!x || !cond || !eps_out) return fail(LDC_E_INVALID, "null tensor");
  if (t < 0 || t >= c->unet.timesteps) return fail(LDC_E_INVALID, "t out of range");
  LDCCHK(check_unet_args(c, B, L, F));
  hipStream_t s = pick_stream(c, stream);
  Halves h;
  LDCCHK(get_halves(c, B, L, F, s, &h));
  LDCCHK(load_cond(c, h, cond, s));
  LDCCHK(load_x(c, h, x, s));
  for (int k = 0; k < h.n; ++k) {
    Plan* pl = h.p[k];
    HIPCHK(launch_step_set(pl->step_state, t, 0, c->cur_key, s));
    HIPCHK(launch_step_begin(c->unet.ss_table, c->unet.ss_stride, pl->step_state, pl->cur_ss, nullptr, s, pl->zero_ptr, pl->zero_bytes));
    LDCCHK(run_ops(c, pl, pl->step_ops, true, s));
    HIPCHK(launch_from_cl(c->dt, pl->eps_cl, eps_out + (size_t)h.b0[k] * c->unet.channels * L, pl->B, c->unet.channels, L, nullptr,
                          0, 0.f, s));
  }
  return finish_stream(c, stream);
}

extern "C" int ldc_unet_debug_tap(ldc_ctx* c, const char* name, float* out, int64_t capacity, void* stream) {
  LDCCHK(check_ready(c, LDC_MODEL_MAIN));
  if (!name || !out) return fail(LDC_E_INVALID, "null argument");
  const Halves& h = c->last_halves;
  if (h.n == 0) return fail(LDC_E_STATE, "no UNet call has been made yet");
  int64_t need = 0;
  for (int k = 0; k < h.n; ++k) {
    auto it = h.p[k]->taps.find(name);
    if (it == h.p[k]->taps.end()) return fail(LDC_E_INVALID, "unknown tap '%s'", name);
    need += (int64_t)h.p[k]->B * it->second.C * it->second.L;
  }
  if (capacity < need) return fail(LDC_E_INVALID, "tap '%s' needs %lld elements", name, (long long)need);
  hipStream_t s = pick_stream(c, stream);
  for (int k = 0; k < h.n; ++k) {
    const Plan::Tap& tp = h.p[k]->taps.find(name)->second;
    HIPCHK(launch_from_cl(c->dt, tp.p, out + (size_t)h.b0[k] * tp.C * tp.L, h.p[k]->B, tp.C, tp.L, nullptr, 0, 0.f, s));
  }
  return finish_stream(c, stream);
}

// one reverse-diffusion step of batch part k on stream s: select the timestep row, run the UNet, update the state,
// advance this part's step counter.  Parts never interact, so each is a self-contained chain.
static int half_step(ldc_ctx* c, const Halves& h, int k, float* x, const float* noise, int64_t noise_stride, hipStream_t s) {
  Plan* pl = h.p[k];
  const size_t off = (size_t)h.b0[k] * c->unet.channels * pl->L;
  unsigned long long* tl = c->timeline ? c->tl_buf + (size_t)k * 2048 * 2 : nullptr;
  // (the step state moves on in the step's FIRST kernel: the callers start a loop from (t + 1, iteration - 1), set_steps)
  HIPCHK(launch_step_begin(c->unet.ss_table, c->unet.ss_stride, pl->step_state, pl->cur_ss, tl, s, pl->zero_ptr, pl->zero_bytes, c->merge_advance));
  LDCCHK(run_ops(c, pl, pl->step_ops, true, s));
  HIPCHK(launch_p_sample_update(c->dt, x + off, pl->eps_cl, noise ? noise + off : nullptr, noise_stride, pl->x_cl, pl->B,
                                c->unet.channels, pl->L, c->sched, pl->step_state, (uint64_t)off, s));
  if (!c->merge_advance) HIPCHK(launch_step_advance(pl->step_state, tl, s));   // (LDC_STEP_ADVANCE_LAUNCH: the round-4 structure, for A/B runs)
  return LDC_OK;
}
// timeline runs: the end stamp of a loop's last step (every other step's is taken by the next step's first kernel)
static int stamp_loop_end(ldc_ctx* c, const Halves& h, int k, hipStream_t s) {
  if (!c->timeline || !c->merge_advance) return LDC_OK;
  HIPCHK(launch_step_advance(h.p[k]->step_state, c->tl_buf + (size_t)k * 2048 * 2, s));
  return LDC_OK;
}

static bool parts_parallel(ldc_ctx* c, const Halves& h) {
  return h.n >= 2 && !c->profile && !c->serial_parts;   // LDC_SERIAL (diagnostics): parts back to back, eager
}
// the parts' streams pick up after everything queued on s / s waits for every part
static int fork_parts(ldc_ctx* c, const Halves& h, hipStream_t s) {
  HIPCHK(hipEventRecord(c->ev_fork, s));
  for (int k = 1; k < h.n; ++k) HIPCHK(hipStreamWaitEvent(c->aux_stream[k], c->ev_fork, 0));
  return LDC_OK;
}
static int join_parts(ldc_ctx* c, const Halves& h, hipStream_t s) {
  for (int k = 1; k < h.n; ++k) {
    HIPCHK(hipEventRecord(c->ev_join[k], c->aux_stream[k]));
    HIPCHK(hipStreamWaitEvent(s, c->ev_join[k], 0));
  }
  return LDC_OK;
}
// the state BEFORE the first step of a loop that starts at timestep t, iteration j (half_step's first kernel advances)
static int set_steps(ldc_ctx* c, const Halves& h, int t, int j, hipStream_t s) {
  for (int k = 0; k < h.n; ++k) HIPCHK(launch_step_set(h.p[k]->step_state, t + c->merge_advance, j - c->merge_advance, c->cur_key, s));
  return LDC_OK;
}

// one step of every part, eagerly: part 0 on s, the others on the auxiliary streams, joined at the end
static int one_step(ldc_ctx* c, const Halves& h, float* x, const float* noise, int64_t noise_stride, hipStream_t s) {
  if (parts_parallel(c, h)) {
    LDCCHK(fork_parts(c, h, s));
    for (int k = 0; k < h.n; ++k) LDCCHK(half_step(c, h, k, x, noise, noise_stride, k == 0 ? s : c->aux_stream[k]));
    LDCCHK(join_parts(c, h, s));
  } else {
    for (int k = 0; k < h.n; ++k) LDCCHK(half_step(c, h, k, x, noise, noise_stride, s));
  }
  return LDC_OK;
}

extern "C" int ldc_p_sample(ldc_ctx* c, float* x, int t, const float* cond, const float* noise, int B, int L, int F,
                            void* stream) {
  LDCCHK(check_ready(c, LDC_MODEL_MAIN));
  if (!x || !cond) return fail(LDC_E_INVALID, "null tensor");
  if (t < 0 || t >= c->unet.timesteps) return fail(LDC_E_INVALID, "t out of range");
  LDCCHK(check_unet_args(c, B, L, F));
  hipStream_t s = pick_stream(c, stream);
  Halves h;
  LDCCHK(get_halves(c, B, L, F, s, &h));
  LDCCHK(load_cond(c, h, cond, s));
  LDCCHK(load_x(c, h, x, s));
  next_noise_key(c, noise == nullptr && t > 0);
  LDCCHK(set_steps(c, h, t, 0, s));
  LDCCHK(one_step(c, h, x, noise, 0, s));
  return finish_stream(c, stream);
}

// the denoise loop on prepared plans (cond already processed, x_cl already set)
// left_forked (optional): the caller continues per part on the parts' streams; the per-part replay path then leaves them un-joined and says so
static int denoise_loop(ldc_ctx* c, const Halves& h, int B, float* x, const float* noise, int n_steps, hipStream_t s, bool* left_forked = nullptr) {
  if (left_forked) *left_forked = false;
  const int L = h.p[0]->L, F = h.p[0]->F;
  const int64_t stride = (int64_t)B * c->unet.channels * L;
  LDCCHK(set_steps(c, h, n_steps - 1, 0, s));
  if (c->profile || c->serial_parts || n_steps < 3) {
    if (left_forked && h.n >= 2) LDCCHK(join_parts(c, h, s));   // (the caller's per-part work is on the auxiliary streams; these steps may all run on s)
    for (int i = 0; i < n_steps; ++i) LDCCHK(one_step(c, h, x, noise, stride, s));
    for (int k = 0; k < h.n; ++k) LDCCHK(stamp_loop_end(c, h, k, s));
    return LDC_OK;
  }
  StepGraph* sg = nullptr;
  for (auto& g : c->graphs)
    if (g.B == B && g.L == L && g.F == F) sg = &g;
  if (!sg) {
    // graphs of shapes whose plans are gone were dropped with them (evict_plan); additionally keep at most 16 alive
    if (c->graphs.size() >= 16) {
      size_t victim = 0;
      for (size_t i = 1; i < c->graphs.size(); ++i)
        if (c->graphs[i].last_use < c->graphs[victim].last_use) victim = i;
      HIPCHK(counted_device_sync());
      c->graphs[victim].destroy();
      c->graphs.erase(c->graphs.begin() + victim);
    }
    c->graphs.push_back(StepGraph());
    sg = &c->graphs.back();
    sg->B = B; sg->L = L; sg->F = F;
  }
  sg->last_use = ++c->use_tick;
  const bool par = parts_parallel(c, h);
  // steps per replayed graph: the parts fork at the head of the graph and join at its tail, so K > 1 lets them
  // drift apart for K steps (concurrent replays of SEPARATE graphs on different streams were measured: the ROCm
  // 7.2 runtime serialises them, 198 vs 188 ms)
  // ~1 500 kernel nodes per graph measured best in both arrangements: 5 steps of two chains (3 / 4 / 5 / 7 steps within 0.2 %,
  // 10 / 25 steps 2 % / 16 % slower), 10 steps of a single chain (5 steps 2.8 % slower; 8 / 10 / 25 equal).  A remainder of
  // single-step replays is slow (13 or 17 steps per graph: 154 instead of 124 ms), so the defaults divide the usual 50.
  int k_want = c->graph_steps > 0 ? c->graph_steps : (h.n == 1 ? 10 : 5);
  if (c->graph_steps == 0) {   // prefer a graph length that divides the step count (within the flat part of the measured range)
    const int lo = h.n == 1 ? 8 : 3, hi = h.n == 1 ? 25 : 7;
    int best = 0;
    for (int k = lo; k <= hi; ++k)
      if (n_steps % k == 0 && (best == 0 || std::abs(k - k_want) < std::abs(best - k_want))) best = k;
    if (best) k_want = best;
  }
  const int K = std::min(k_want, std::max(1, n_steps - 1));
  int done = 0;
  if (!sg->any() || sg->noise != noise || sg->x != x || sg->stream != s || sg->n != h.n * 100 + K + ((par && c->part_graphs && (h.n == 2 || c->part_graphs >= 2)) ? 100000 : 0)) {
    if (sg->any()) {
      // replays of the old executable graphs may still be in flight: drain before destroying (a re-capture is one of
      // the documented places where a call waits for the device)
      HIPCHK(counted_device_sync());
      sg->destroy();
    }
    // first step eagerly: loads code objects / sets function attributes outside of the capture
    LDCCHK(one_step(c, h, x, noise, stride, s));
    done = 1;
    const bool per_part = par && c->part_graphs && (h.n == 2 || c->part_graphs >= 2);   // (LDC_PART_GRAPHS=2: also for three / four parts)   // (three parts on per-part graphs measured 45 % slower than the fork/join graph)
    if (per_part) {
      // ONE SINGLE-STREAM graph per batch part, captured and replayed on the part's own stream.  ROCm 7.2 replays a
      // single-stream graph from AQL packets recorded at instantiation (~0.4 ms of host time for 1 500 kernel nodes); a graph
      // that forks and joins streams inside takes the node-by-node path (~9 us per node: 140 ms of host time per 50-step decode
      // of two parts, which is then what bounds the decode).  The parts fork once in front of the loop and join once behind it.
      for (int k = 0; k < h.n; ++k) {
        hipStream_t sk = k == 0 ? s : c->aux_stream[k];
        for (int which = 0; which < 2; ++which) {
          const int steps = which == 0 ? K : 1;
          if (which == 1 && K == 1) break;
          hipGraph_t g = nullptr;
          HIPCHK(hipStreamBeginCapture(sk, hipStreamCaptureModeRelaxed));
          int r = LDC_OK;
          for (int i = 0; i < steps && r == LDC_OK; ++i) r = half_step(c, h, k, x, noise, stride, sk);
          hipError_t e = hipStreamEndCapture(sk, &g);
          if (r != LDC_OK) { if (g) (void)hipGraphDestroy(g); return r; }
          if (e != hipSuccess) return fail(LDC_E_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
          e = hipGraphInstantiate(&sg->pexec[k][which], g, nullptr, nullptr, 0);
          (void)hipGraphDestroy(g);
          if (e != hipSuccess) { sg->pexec[k][which] = nullptr; return fail(LDC_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
        }
      }
      sg->exec[0] = sg->pexec[0][0];   // marks the entry as built (any()); never launched through exec[] in this mode
      sg->per_part = true;
    } else {
    // exec[0]: K steps of every part; exec[1]: one step (remainder); the device-side step counters make every
    // replay continue where the last one stopped
    for (int which = 0; which < 2; ++which) {
      const int steps = which == 0 ? K : 1;
      if (which == 1 && K == 1) break;
      hipGraph_t g = nullptr;
      HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
      int r = LDC_OK;
      if (par) r = fork_parts(c, h, s);
      for (int k = 0; k < h.n && r == LDC_OK; ++k)
        for (int i = 0; i < steps && r == LDC_OK; ++i) r = half_step(c, h, k, x, noise, stride, (k == 0 || !par) ? s : c->aux_stream[k]);
      if (par && r == LDC_OK) r = join_parts(c, h, s);
      hipError_t e = hipStreamEndCapture(s, &g);
      if (r != LDC_OK) { if (g) (void)hipGraphDestroy(g); return r; }
      if (e != hipSuccess) return fail(LDC_E_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
      e = hipGraphInstantiate(&sg->exec[which], g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if (e != hipSuccess) { sg->exec[which] = nullptr; return fail(LDC_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
    }
      sg->per_part = false;
    }
    sg->noise = noise; sg->x = x; sg->stream = s; sg->n = h.n * 100 + K + (per_part ? 100000 : 0);
  }
  // host time inside hipGraphLaunch is accounted (ldc_host_stats): with ~1 500 kernel nodes per replay it is what caps one
  // process once the kernels get faster
  auto timed_launch = [&](hipGraphExec_t ge) -> hipError_t {
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipGraphLaunch(ge, s);
    c->host_graph_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ++c->host_graph_launches;
    return e;
  };
  auto replay = [&](hipGraphExec_t ge) -> int {
    if (c->flow_depth > 0) {
      if (c->flow_n >= (unsigned long long)c->flow_depth) {
        const auto t0 = std::chrono::steady_clock::now();
        HIPCHK(hipEventSynchronize(c->flow_ev[(c->flow_n - c->flow_depth) % ldc_ctx::kFlowRing]));
        c->host_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      }
      HIPCHK(timed_launch(ge));
      HIPCHK(hipEventRecord(c->flow_ev[c->flow_n % ldc_ctx::kFlowRing], s));
      ++c->flow_n;
    } else {
      HIPCHK(timed_launch(ge));
    }
    return LDC_OK;
  };
  int i = done;
  if (sg->per_part) {
    // One single-stream graph per part, replayed from recorded AQL packets.  The parts' replays are issued INTERLEAVED (graph j
    // of every part before graph j + 1 of any) behind a bounded look-ahead per part: through round 3 all of part 0's replays
    // were issued first, hipGraphLaunch blocked on the full hardware queue for most of the decode (76-105 ms "inside
    // hipGraphLaunch") and part 1 started late.  Waiting happens on events of this context, outside hipGraphLaunch.
    LDCCHK(fork_parts(c, h, s));
    const int depth = std::max(1, std::min(c->flow_depth > 0 ? c->flow_depth : 3, 3));
    std::vector<hipEvent_t>& evs = sg->part_ev;
    const int n_big = (n_steps - done) / K, n_rep = n_big + ((n_steps - done) - n_big * K);
    while ((int)evs.size() < h.n * depth) {
      hipEvent_t e = nullptr;
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      evs.push_back(e);
    }
    // (a failure between fork and join must still join: the auxiliary streams would otherwise be left with work that is unordered
    // against the caller's stream -- ADVICE r4.  Note for asynchronous callers: the look-ahead wait below blocks the HOST for up to
    // `depth` replays of a part, on events of this context.)
    hipError_t err = hipSuccess;
    const char* what = "";
    for (int j = 0; j < n_rep && err == hipSuccess; ++j)
      for (int k = 0; k < h.n && err == hipSuccess; ++k) {
        hipStream_t sk = k == 0 ? s : c->aux_stream[k];
        hipEvent_t ev = evs[(size_t)k * depth + j % depth];
        if (j >= depth) {
          const auto t0 = std::chrono::steady_clock::now();
          err = hipEventSynchronize(ev);   // replay j - depth of this part has finished
          c->host_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
          if (err != hipSuccess) { what = "hipEventSynchronize"; break; }
        }
        const auto t0 = std::chrono::steady_clock::now();
        err = hipGraphLaunch(sg->pexec[k][(j < n_big || K == 1) ? 0 : 1], sk);
        c->host_graph_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        ++c->host_graph_launches;
        if (err != hipSuccess) { what = "hipGraphLaunch"; break; }
        err = hipEventRecord(ev, sk);
        if (err != hipSuccess) what = "hipEventRecord";
      }
    for (int k = 0; k < h.n && err == hipSuccess; ++k)
      if (stamp_loop_end(c, h, k, k == 0 ? s : c->aux_stream[k]) != LDC_OK) { err = hipErrorUnknown; what = "the loop-end stamp"; }
    if (err == hipSuccess && left_forked) {   // the caller's per-part work follows on the same streams and joins behind it
      *left_forked = true;
      return LDC_OK;
    }
    const int jr = join_parts(c, h, s);
    if (err != hipSuccess) return fail(LDC_E_HIP, "%s failed in the per-part graph replay: %s", what, hipGetErrorString(err));
    return jr;
  }
  // The fork / join graph is launched on s ALONE: its part branches are graph nodes, not aux_stream[k].  A caller that arrives with the parts
  // forked (ldc_decode's per-part front ends, left_forked != null) still has work of parts k >= 1 on the auxiliary streams that nothing
  // orders in front of the replay unless the eager first step above ran (it joins): from the second decode of a shape on it does not
  // (the replayed step kernels raced with part k's front end writing cond / x_cl / x).
  if (left_forked && par) LDCCHK(join_parts(c, h, s));
  for (; i + K <= n_steps; i += K) LDCCHK(replay(sg->exec[0]));
  for (; i < n_steps; ++i) LDCCHK(replay(sg->exec[K == 1 ? 0 : 1]));
  for (int k = 0; k < h.n; ++k) LDCCHK(stamp_loop_end(c, h, k, s));
  return LDC_OK;
}

static int ensure_state(ldc_ctx* c, size_t bytes) {
  if (bytes <= c->state_bytes) return LDC_OK;
  HIPCHK(counted_device_sync());
  if (c->state_buf) HIPCHK(hipFree(c->state_buf));
  c->state_buf = nullptr; c->state_bytes = 0;
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, bytes));
  c->state_buf = (float*)p; c->state_bytes = bytes;
  return LDC_OK;
}

extern "C" int ldc_denoise(ldc_ctx* c, float* img, const float* cond, const float* noise, int n_steps, int B, int L, int F,
                           void* stream) {
  LDCCHK(check_ready(c, LDC_MODEL_MAIN));
  if (!img || !cond) return fail(LDC_E_INVALID, "null tensor");
  if (n_steps < 1 || n_steps > c->unet.timesteps) return fail(LDC_E_INVALID, "n_steps must be in [1,%d]", c->unet.timesteps);
  LDCCHK(check_unet_args(c, B, L, F));
  hipStream_t s = pick_stream(c, stream);
  Halves h;
  LDCCHK(get_halves(c, B, L, F, s, &h));
  LDCCHK(load_cond(c, h, cond, s));
  // the loop runs on a context-owned copy of the state: the captured step graphs hold its address, so callers may
  // pass a different tensor every time without forcing a re-capture
  const size_t nbytes = (size_t)B * c->unet.channels * L * 4;
  LDCCHK(ensure_state(c, nbytes));
  HIPCHK(hipMemcpyAsync(c->state_buf, img, nbytes, hipMemcpyDeviceToDevice, s));
  LDCCHK(load_x(c, h, c->state_buf, s));
  next_noise_key(c, noise == nullptr);
  LDCCHK(denoise_loop(c, h, B, c->state_buf, noise, n_steps, s));
  HIPCHK(hipMemcpyAsync(img, c->state_buf, nbytes, hipMemcpyDeviceToDevice, s));
  return finish_stream(c, stream);
}

// GaussianDiffusion1D.p_sample_loop (ddpm_loss.py:253-266): ancestral sampling over ALL timesteps from a standard
// normal start.  img == NULL on entry is not allowed: the caller passes the buffer; fill_start != 0 draws the start
// image on the device (Philox stream 0xffffffff), otherwise the buffer's contents are the start image (parity runs).
extern "C" int ldc_p_sample_loop(ldc_ctx* c, float* img, const float* cond, const float* noise, int fill_start, int B, int L,
                                 int F, void* stream) {
  LDCCHK(check_ready(c, LDC_MODEL_MAIN));
  if (!img || !cond) return fail(LDC_E_INVALID, "null tensor");
  LDCCHK(check_unet_args(c, B, L, F));
  hipStream_t s = pick_stream(c, stream);
  next_noise_key(c, noise == nullptr || fill_start);
  if (fill_start) HIPCHK(launch_random_fill(img, (int64_t)B * c->unet.channels * L, 0, c->cur_key, 0xffffffffu, s));
  Halves h;
  LDCCHK(get_halves(c, B, L, F, s, &h));
  LDCCHK(load_cond(c, h, cond, s));
  const size_t nbytes = (size_t)B * c->unet.channels * L * 4;
  LDCCHK(ensure_state(c, nbytes));
  HIPCHK(hipMemcpyAsync(c->state_buf, img, nbytes, hipMemcpyDeviceToDevice, s));
  LDCCHK(load_x(c, h, c->state_buf, s));
  LDCCHK(denoise_loop(c, h, B, c->state_buf, noise, c->unet.timesteps, s));
  HIPCHK(hipMemcpyAsync(img, c->state_buf, nbytes, hipMemcpyDeviceToDevice, s));
  return finish_stream(c, stream);
}

// GaussianDiffusion1D.infilling (ddpm_loss.py:331-367): for t = midway_t-1 .. 0
//   img <- p_sample(img, t); img <- (1-lam) img + lam infill; infill <- p_sample(infill, t); img <- (1-lam) img + lam infill
// `noise` (optional, parity runs): [2*midway_t][B][C][L], draw 2*i for img and 2*i+1 for infill at iteration i.
// fill_start != 0 draws the uniform [0,1) start image of ddpm_loss.py:336 on the device.
extern "C" int ldc_infilling(ldc_ctx* c, float* img, float* infill_img, const float* cond, int midway_t, const float* noise,
                             float lam, int fill_start, int B, int L, int F, void* stream) {
  LDCCHK(check_ready(c, LDC_MODEL_MAIN));
  if (!img || !infill_img || !cond) return fail(LDC_E_INVALID, "null tensor");
  if (midway_t < 1 || midway_t > c->unet.timesteps) return fail(LDC_E_INVALID, "midway_t must be in [1,%d]", c->unet.timesteps);
  LDCCHK(check_unet_args(c, B, L, F));
  hipStream_t s = pick_stream(c, stream);
  const int64_t n = (int64_t)B * c->unet.channels * L;
  next_noise_key(c, noise == nullptr || fill_start);
  if (fill_start) HIPCHK(launch_random_fill(img, n, 1, c->cur_key, 0xfffffffeu, s));
  Halves h;
  LDCCHK(get_halves(c, B, L, F, s, &h));
  LDCCHK(load_cond(c, h, cond, s));
  int draw = 0;
  for (int t = midway_t - 1; t >= 0; --t) {
    LDCCHK(load_x(c, h, img, s));
    LDCCHK(set_steps(c, h, t, draw++, s));
    LDCCHK(one_step(c, h, img, noise, n, s));
    HIPCHK(launch_axpby(img, infill_img, 1.0f - lam, lam, n, s));
    LDCCHK(load_x(c, h, infill_img, s));
    LDCCHK(set_steps(c, h, t, draw++, s));
    LDCCHK(one_step(c, h, infill_img, noise, n, s));
    HIPCHK(launch_axpby(img, infill_img, 1.0f - lam, lam, n, s));
  }
  return finish_stream(c, stream);
}

static int ensure_outnorm(ldc_ctx* c, int B) {
  const size_t need = output_normalise_ws_bytes(B);
  if (need <= c->outnorm_ws_bytes) return LDC_OK;
  HIPCHK(counted_device_sync());
  if (c->outnorm_ws) HIPCHK(hipFree(c->outnorm_ws));
  c->outnorm_ws = nullptr;
  HIPCHK(hipMalloc(&c->outnorm_ws, need * 2));
  c->outnorm_ws_bytes = need * 2;
  return LDC_OK;
}

extern "C" int ldc_output_normalise(ldc_ctx* c, float* wav, int B, int T, int per_item, void* stream) {
  if (!c) return fail(LDC_E_INVALID, "null ctx");
  if (!wav || B <= 0 || T <= 0) return fail(LDC_E_INVALID, "bad arguments");
  HIPCHK(hipSetDevice(c->device));
  LDCCHK(ensure_outnorm(c, B));
  hipStream_t s = pick_stream(c, stream);
  HIPCHK(launch_output_normalise(wav, B, T, per_item ? 1 : 0, c->outnorm_ws, s));
  return finish_stream(c, stream);
}

// synthesis() body for one resident batch (sample.py:94-134)
extern "C" int ldc_decode(ldc_ctx* c, const float* wav, int B, int T, int n_steps, const float* noise, int per_item,
                          float* wav_out, float* latents_out, float* cond_out, int64_t* codes_out, void* stream) {
  LDCCHK(check_ready(c, LDC_MODEL_MAIN, true));
  if (!wav || !wav_out || B <= 0 || T <= 0) return fail(LDC_E_INVALID, "bad arguments");
  if (n_steps < 1 || n_steps > c->unet.timesteps) return fail(LDC_E_INVALID, "n_steps must be in [1,%d]", c->unet.timesteps);
  const Codec& cc = c->codec[LDC_MODEL_COND];
  const Codec& mc = c->codec[LDC_MODEL_MAIN];
  if (T % cc.hop || T % mc.hop) return fail(LDC_E_INVALID, "T must be a multiple of %d and %d (sample.py:87 trims to 640)", cc.hop, mc.hop);
  const int F = T / cc.hop, L = T / mc.hop, D = c->cfg.rep_dims;
  LDCCHK(check_unet_args(c, B, L, F));
  LDCCHK(ensure_outnorm(c, B));
  hipStream_t s = pick_stream(c, stream);
  Halves h;
  LDCCHK(get_halves(c, B, L, F, s, &h));
  // The codec front end (cond encoder -> RVQ -> upsampler -> start image) and back end (decoder) are chains of ~40 latency-bound
  // launches each (fp32 convs on a fraction of the CUs, LSTM recurrences): with per-utterance normalisation nothing couples the
  // items, so every batch part runs its own front / back end on its own stream, like its denoise steps, and the parts' chains fill
  // each other's gaps (round 5; `split_ends` 0 / LDC_NO_SPLIT_ENDS restores the whole-batch ends).  The RVQ codes are written
  // [n_q][B][F]: a caller that wants them gets the whole-batch path.
  const bool split_ends = c->split_ends && per_item && !codes_out && h.n >= 2 && parts_parallel(c, h);
  LDCCHK(with_scratch(c, s, [&](Arena& ar, bool dry) -> int {
    float* x = latents_out ? latents_out : (float*)ar.alloc((size_t)B * D * L * 4);
    float* mx = (float*)ar.alloc((size_t)B * 4);
    if (!dry && split_ends) LDCCHK(fork_parts(c, h, s));
    const int n_front = split_ends ? h.n : 1;
    // From here to the final join the parts' streams carry work that only join_parts orders against the caller's stream (and against the
    // scratch arena the next call reuses): the forked regions collect their first error and the call joins before it reports it.
    bool parts_open = !dry && split_ends;
    auto bail = [&](int rc) -> int {
      if (parts_open) (void)join_parts(c, h, s);
      parts_open = false;
      return rc;
    };
    auto front_end = [&]() -> int {
    for (int k = 0; k < n_front; ++k) {
      hipStream_t sk = (split_ends && k > 0) ? c->aux_stream[k] : s;
      const int b0 = split_ends ? h.b0[k] : 0, Bk = split_ends ? h.p[k]->B : B;
      float* qr = nullptr;
      int Fq = 0;
      LDCCHK(get_cond_rows(c, wav + (size_t)b0 * T, Bk, T, 0.f, ar, dry, sk, &qr, &Fq, codes_out, n_front));
      if (!dry && Fq != F) return fail(LDC_E_INVALID, "internal: encoder produced %d frames, expected %d", Fq, F);
      // start image: upsample, /= max|.|+1e-8 (sample.py:125-129)
      void* up = nullptr;
      int Lu = 0;
      LDCCHK(upsample_rows(c, qr, Bk, F, ar, dry, sk, &up, &Lu));
      if (dry) continue;
      if (cond_out) HIPCHK(launch_from_cl(DT_F32, qr, cond_out + (size_t)b0 * D * F, Bk, D, F, nullptr, 0, 0.f, sk));
      HIPCHK(hipMemsetAsync(mx + b0, 0, (size_t)Bk * 4, sk));
      HIPCHK(launch_maxabs(DT_F32, up, Bk, (int64_t)L * D, per_item ? 1 : 0, mx + b0, sk));
      HIPCHK(launch_from_cl(DT_F32, up, x + (size_t)b0 * D * L, Bk, D, L, mx + b0, per_item ? 1 : 0, 1e-8f, sk));
      // process_cond for the UNet (rows are already channels-last fp32)
      for (int kk = 0; kk < h.n; ++kk) {
        if (split_ends && kk != k) continue;
        Plan* pl = h.p[kk];
        HIPCHK(hipMemcpyAsync(pl->cond_in_cl, qr + (size_t)(h.b0[kk] - b0) * F * D, (size_t)pl->B * F * D * 4, hipMemcpyDeviceToDevice, sk));
        LDCCHK(run_ops(c, pl, pl->cond_ops, false, sk));
        HIPCHK(launch_to_cl(c->dt, x + (size_t)h.b0[kk] * c->unet.channels * pl->L, pl->x_cl, pl->B, c->unet.channels, pl->L, nullptr, 0, 0.f, sk));
      }
    }
    return LDC_OK;
    };
    { const int fr = front_end(); if (fr != LDC_OK) return bail(fr); }
    if (!dry) {
      // The parts stay on their streams from the front end to the back end (no join in front of the loop: it forks the part streams behind
      // part 0's front end itself; none behind it when the per-part replay path ran): they de-phase by the front ends' serialised pieces
      // (the cooperative LSTMs), and one part's latency-bound back end then runs under the other's last denoise steps.
      // (c->ends_join / LDC_ENDS_JOIN: both joins as before, for A/B runs)
      if (split_ends && c->ends_join) { parts_open = false; LDCCHK(join_parts(c, h, s)); }
      next_noise_key(c, noise == nullptr);
      bool forked = false;
      const int dr = denoise_loop(c, h, B, x, noise, n_steps, s, (split_ends && !c->ends_join) ? &forked : nullptr);
      if (dr != LDC_OK) return bail(dr);
      if (split_ends && !forked) { parts_open = false; LDCCHK(fork_parts(c, h, s)); }   // (the loop joined, or never left s)
      parts_open = split_ends;
    }
    // decoder (quirk Q3: no x18 un-scaling on this path, sample.py:131)
    int Lo_all = T;
    auto back_end = [&]() -> int {
    for (int k = 0; k < n_front; ++k) {
      hipStream_t sk = (split_ends && k > 0) ? c->aux_stream[k] : s;
      const int b0 = split_ends ? h.b0[k] : 0, Bk = split_ends ? h.p[k]->B : B;
      SeaRun R{c, &ar, sk, dry, Bk};
      R.side = h.n <= 2 ? k : -1;   // (aux_stream[2], [3] are free when the batch has at most two parts)
      R.teams = h.n;
      void* zc = ar.alloc((size_t)Bk * L * D * 4);
      if (!dry) HIPCHK(launch_to_cl(DT_F32, x + (size_t)b0 * D * L, zc, Bk, D, L, nullptr, 0, 0.f, sk));
      void* y = nullptr;
      int Lo = 0, Cd = 0;
      LDCCHK(run_seanet(R, mc.dec, zc, L, &y, &Lo, &Cd));
      if (!dry) HIPCHK(hipMemcpyAsync(wav_out + (size_t)b0 * Lo, y, (size_t)Bk * Lo * 4, hipMemcpyDeviceToDevice, sk));
      Lo_all = Lo;
    }
    return LDC_OK;
    };
    { const int br = back_end(); if (br != LDC_OK) return bail(br); }
    if (!dry) {
      if (split_ends) { parts_open = false; LDCCHK(join_parts(c, h, s)); }
      HIPCHK(launch_output_normalise(wav_out, B, Lo_all, per_item ? 1 : 0, c->outnorm_ws, s));
    }
    return LDC_OK;
  }));
  return finish_stream(c, stream);
}


// calls that need no weights (bit-stream layer, resampler, tuning aids): any context of the device will do
int check_dev(ldc_ctx* c) {
  if (!c) return fail(LDC_E_INVALID, "null ctx");
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail(LDC_E_HIP, "hipSetDevice failed: %s", hipGetErrorString(e));
  return LDC_OK;
}
