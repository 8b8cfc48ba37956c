"""CPU oracle for the LaDiffCodec decode path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker / the timed CPU baseline.  The shipped path (`ladiffcodec_amd`)
never routes through it.

What it is: a plain-PyTorch fp32 CPU restatement (functional, state-dict driven, no nn.Module) of
the reference's arithmetic for every stage on the hot path.  Each
function cites the reference file:line it follows.  The reference itself is pure Python/PyTorch
(no native code), so the restatement uses the same ATen primitives (conv1d, conv_transpose1d,
group_norm, softmax, LSTM cell arithmetic) the reference reaches through torch.nn.

Pinning: the reference's own tests hold no vectors for this path, so the oracle is
pinned against outputs of the reference itself, generated in the build container by
`tools/gen_golden.py` (which imports /root/reference) and committed under `tests/golden/`;
`tests/test_oracle_golden.py` checks every stage against them.

Tensors use the reference layout: activations [B, C, L] fp32, codes [n_q, B, F] int64.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from ladiffcodec_amd import spec
from ladiffcodec_amd.spec import CodecConfig, UnetConfig

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------------------------
# L1 primitives: padding rules, weight-norm, streaming convs  (reference srcs/modules/conv.py)
# ----------------------------------------------------------------------------------------------

def fold_weight_norm(g: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """w = g * v / ||v||, norm over all dims but 0 (torch weight_norm dim=0; conv.py:27-30)."""
    n = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v * (g / n)


def extra_padding_for_conv1d(length: int, kernel_size: int, stride: int, padding_total: int) -> int:
    """conv.py:56-63."""
    n_frames = (length - kernel_size + padding_total) / stride + 1
    ideal = (math.ceil(n_frames) - 1) * stride + (kernel_size - padding_total)
    return ideal - length


def pad1d_reflect(x: torch.Tensor, left: int, right: int) -> torch.Tensor:
    """conv.py:81-98: reflect pad that tolerates inputs shorter than the pad."""
    length = x.shape[-1]
    max_pad = max(left, right)
    extra = 0
    if length <= max_pad:
        extra = max_pad - length + 1
        x = F.pad(x, (0, extra))
    y = F.pad(x, (left, right), mode="reflect")
    return y[..., : y.shape[-1] - extra]


def sconv1d(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], stride: int = 1, dilation: int = 1,
            causal: bool = True) -> torch.Tensor:
    """SConv1d.forward, conv.py:217-232 (pad_mode='reflect')."""
    k = w.shape[-1]
    padding_total = (k - 1) * dilation - (stride - 1)
    extra = extra_padding_for_conv1d(x.shape[-1], k, stride, padding_total)
    if causal:
        x = pad1d_reflect(x, padding_total, extra)
    else:
        right = padding_total // 2
        x = pad1d_reflect(x, padding_total - right, right + extra)
    return F.conv1d(x, w, b, stride=stride, dilation=dilation)


def sconvtr1d(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], stride: int, causal: bool) -> torch.Tensor:
    """SConvTranspose1d.forward, conv.py:252-274 (trim_right_ratio = 1)."""
    k = w.shape[-1]
    padding_total = k - stride
    y = F.conv_transpose1d(x, w, b, stride=stride)
    if causal:
        right = math.ceil(padding_total * 1.0)
        left = padding_total - right
    else:
        right = padding_total // 2
        left = padding_total - right
    return y[..., left: y.shape[-1] - right]


def _wn(sd: SD, prefix: str) -> Tuple[torch.Tensor, torch.Tensor]:
    return fold_weight_norm(sd[prefix + ".weight_g"], sd[prefix + ".weight_v"]), sd[prefix + ".bias"]


def lstm_skip(x: torch.Tensor, sd: SD, prefix: str, layers: int) -> torch.Tensor:
    """SLSTM.forward, lstm.py:22-28: nn.LSTM over time (gate order i,f,g,o) then + input."""
    y = x.permute(2, 0, 1)                      # [T, B, C]
    inp = y
    for n in range(layers):
        w_ih, w_hh = sd[f"{prefix}.lstm.weight_ih_l{n}"], sd[f"{prefix}.lstm.weight_hh_l{n}"]
        bias = sd[f"{prefix}.lstm.bias_ih_l{n}"] + sd[f"{prefix}.lstm.bias_hh_l{n}"]
        hdim = w_hh.shape[1]
        pre = inp @ w_ih.t() + bias             # [T, B, 4H]
        h = x.new_zeros(x.shape[0], hdim)
        c = x.new_zeros(x.shape[0], hdim)
        outs = []
        for t in range(pre.shape[0]):
            gates = pre[t] + h @ w_hh.t()
            i, f, g, o = gates.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        inp = torch.stack(outs, 0)
    return (inp + y).permute(1, 2, 0)


# ----------------------------------------------------------------------------------------------
# L2 SEANet encoder / decoder  (reference srcs/modules/seanet.py)
# ----------------------------------------------------------------------------------------------

def _resblock(x: torch.Tensor, sd: SD, p: str, ly: spec.SeanetLayer) -> torch.Tensor:
    """SEANetResnetBlock.forward, seanet.py:62-63 with true_skip=False: shortcut conv + [ELU,k3,ELU,k1]."""
    w1, b1 = _wn(sd, p + ".block.1.conv.conv")
    w2, b2 = _wn(sd, p + ".block.3.conv.conv")
    ws, bs = _wn(sd, p + ".shortcut.conv.conv")
    h = sconv1d(F.elu(x), w1, b1, dilation=ly.dilation)
    h = sconv1d(F.elu(h), w2, b2)
    return sconv1d(x, ws, bs) + h


def seanet_encode(sd: SD, c: CodecConfig, wav: torch.Tensor, prefix: str = "encoder") -> torch.Tensor:
    """SEANetEncoder.forward, seanet.py:153 over the Sequential built at :108-151.  [B,1,T]->[B,D,T/hop]."""
    x = wav
    pending_elu = False
    for ly in spec.seanet_encoder_layers(c):
        p = f"{prefix}.model.{ly.index}"
        if ly.kind == "elu":
            pending_elu = True
            continue
        if pending_elu:
            x = F.elu(x)
            pending_elu = False
        if ly.kind == "conv":
            w, b = _wn(sd, p + ".conv.conv")
            x = sconv1d(x, w, b, stride=ly.stride)
        elif ly.kind == "res":
            x = _resblock(x, sd, p, ly)
        elif ly.kind == "lstm":
            x = lstm_skip(x, sd, p, ly.layers)
    if c.final_activation is not None:            # seanet.py:144-149: getattr(nn, final_activation)() after the last conv
        x = getattr(torch.nn, c.final_activation)()(x)
    return x


def seanet_decode(sd: SD, c: CodecConfig, z: torch.Tensor, prefix: str = "decoder") -> torch.Tensor:
    """SEANetDecoder.forward, seanet.py:246-248 over the Sequential built at :200-244.  [B,D,L]->[B,1,L*hop]."""
    x = z
    pending_elu = False
    for ly in spec.seanet_decoder_layers(c):
        p = f"{prefix}.model.{ly.index}"
        if ly.kind == "elu":
            pending_elu = True
            continue
        if pending_elu:
            x = F.elu(x)
            pending_elu = False
        if ly.kind == "conv":
            w, b = _wn(sd, p + ".conv.conv")
            x = sconv1d(x, w, b)
        elif ly.kind == "convtr":
            w, b = _wn(sd, p + ".convtr.convtr")
            x = sconvtr1d(x, w, b, ly.stride, causal=True)
        elif ly.kind == "res":
            x = _resblock(x, sd, p, ly)
        elif ly.kind == "lstm":
            x = lstm_skip(x, sd, p, ly.layers)
    return x


# ----------------------------------------------------------------------------------------------
# L2 RVQ  (reference srcs/quantization/core_vq.py, vq.py)
# ----------------------------------------------------------------------------------------------

def rvq_nearest(x: torch.Tensor, embed: torch.Tensor) -> torch.Tensor:
    """EuclideanCodebook.quantize, core_vq.py:174-182: argmax of -(|x|^2 - 2 x.E^T + |E|^2), first max wins."""
    e = embed.t()
    dist = -(x.pow(2).sum(1, keepdim=True) - 2 * x @ e + e.pow(2).sum(0, keepdim=True))
    return dist.max(dim=-1).indices


def rvq_forward(sd: SD, x: torch.Tensor, n_q: int, prefix: str = "quantizer") -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """ResidualVectorQuantization.forward (eval), core_vq.py:324-342 via VectorQuantization.forward :292-311.

    x [B, D, F] -> quantized [B, D, F], codes [n_q, B, F] int64, margins [n_q, B, F] (top-1 minus
    top-2 score; test aid only: a code is compared bit-exactly only where its margin is safe).
    """
    B, D, Fr = x.shape
    residual = x
    out = torch.zeros_like(x)
    codes, margins = [], []
    for q in range(n_q):
        embed = sd[f"{prefix}.vq.layers.{q}._codebook.embed"]
        flat = residual.permute(0, 2, 1).reshape(-1, D)
        e = embed.t()
        dist = -(flat.pow(2).sum(1, keepdim=True) - 2 * flat @ e + e.pow(2).sum(0, keepdim=True))
        top2 = dist.topk(2, dim=-1).values
        idx = dist.max(dim=-1).indices
        quant = F.embedding(idx, embed).view(B, Fr, D).permute(0, 2, 1)
        residual = residual - quant
        out = out + quant
        codes.append(idx.view(B, Fr))
        margins.append((top2[:, 0] - top2[:, 1]).view(B, Fr))
    return out, torch.stack(codes), torch.stack(margins)


def get_cond(sd: SD, c: CodecConfig, wav: torch.Tensor, bandwidth: Optional[float] = None):
    """DiffAudioRep.get_cond, model.py:223-231: encoder -> RVQ(eval) -> .quantized (codes discarded there)."""
    z = seanet_encode(sd, c, wav)
    n_q = c.n_q_for_bandwidth(bandwidth)
    quantized, codes, margins = rvq_forward(sd, z, n_q)
    return quantized, codes, margins, z


# ----------------------------------------------------------------------------------------------
# L2 Unet1D  (reference srcs/modules/unet.py)
# ----------------------------------------------------------------------------------------------

WS_PREFOLDED = False    # tests of the fp8-weight engine hand the oracle weights that are already standardised (and quantised)
ACT_FP8 = False         # tests of the fp8 x fp8 engine: the tensors that engine produces in fp8 (block1's output, the PreNorm output
                        # in front of to_qkv, tanh(x) in front of final_conv; channel counts that are multiples of 64) are rounded to
                        # OCP e4m3 here too (saturating at +-448), so that the comparison isolates the kernels from the format


def _q8(x: torch.Tensor) -> torch.Tensor:
    """what an fp8-producing kernel stores (norm_act.hip store8_fp8), for tensors the fp8 x fp8 path quantises"""
    if not ACT_FP8 or x.shape[1] % 64 != 0:
        return x
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def ws_fold(w: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """WeightStandardizedConv2d.forward weight path, unet.py:73-78 (fp32 -> eps 1e-5, biased var)."""
    if WS_PREFOLDED:
        return w
    flat = w.flatten(1)
    mean = flat.mean(1).view(-1, 1, 1)
    var = flat.var(1, unbiased=False).view(-1, 1, 1)
    return (w - mean) * (var + eps).rsqrt()


def channel_layernorm(x: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """LayerNorm.forward, unet.py:87-91: over channels, biased var, eps 1e-5, gain only."""
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) * (var + 1e-5).rsqrt() * g


def time_embedding(sd: SD, u: UnetConfig, t: torch.Tensor, prefix: str) -> torch.Tensor:
    """SinusoidalPosEmb + time_mlp, unet.py:109-116, 327-332: [B] -> [B, 4*dim]."""
    half = u.dim // 2
    f = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    emb = t[:, None] * f[None, :]
    emb = torch.cat((emb.sin(), emb.cos()), dim=-1)
    h = F.linear(emb, sd[prefix + ".time_mlp.1.weight"], sd[prefix + ".time_mlp.1.bias"])
    h = F.gelu(h)
    return F.linear(h, sd[prefix + ".time_mlp.3.weight"], sd[prefix + ".time_mlp.3.bias"])


def block_scale_shift(sd: SD, r: spec.ResnetSpec, temb: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """ResnetBlock.mlp = SiLU -> Linear(time_dim, 2*dim_out), chunk (scale, shift); unet.py:162-165,183-186."""
    e = F.linear(F.silu(temb), sd[r.prefix + ".mlp.1.weight"], sd[r.prefix + ".mlp.1.bias"])
    scale, shift = e[:, :, None].chunk(2, dim=1)
    return scale, shift


def resnet_block(sd: SD, r: spec.ResnetSpec, x: torch.Tensor, temb: torch.Tensor, groups: int = 8) -> torch.Tensor:
    """ResnetBlock.forward, unet.py:176-192 with Block.forward :145-154."""
    scale, shift = block_scale_shift(sd, r, temb)
    p = r.prefix
    h = F.conv1d(x, ws_fold(sd[p + ".block1.proj.weight"]), sd[p + ".block1.proj.bias"], padding=1)
    h = F.group_norm(h, groups, sd[p + ".block1.norm.weight"], sd[p + ".block1.norm.bias"])
    h = _q8(F.silu(h * (scale + 1) + shift))
    h = F.conv1d(h, ws_fold(sd[p + ".block2.proj.weight"]), sd[p + ".block2.proj.bias"], padding=1)
    h = F.group_norm(h, groups, sd[p + ".block2.norm.weight"], sd[p + ".block2.norm.bias"])
    h = F.silu(h)
    res = F.conv1d(x, sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"]) if r.has_res_conv else x
    return h + res


def linear_attention(sd: SD, p: str, x: torch.Tensor, heads: int, dim_head: int) -> torch.Tensor:
    """Residual(PreNorm(LinearAttention)), unet.py:50-56, 93-101, 208-222."""
    b, c, n = x.shape
    xn = _q8(channel_layernorm(x, sd[p + ".fn.norm.g"])) if c <= 1024 else channel_layernorm(x, sd[p + ".fn.norm.g"])
    qkv = F.conv1d(xn, sd[p + ".fn.fn.to_qkv.weight"]).chunk(3, dim=1)
    q, k, v = (t.reshape(b, heads, dim_head, n) for t in qkv)
    q = q.softmax(dim=-2) * dim_head ** -0.5
    k = k.softmax(dim=-1)
    context = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", context, q).reshape(b, heads * dim_head, n)
    out = F.conv1d(out, sd[p + ".fn.fn.to_out.0.weight"], sd[p + ".fn.fn.to_out.0.bias"])
    out = channel_layernorm(out, sd[p + ".fn.fn.to_out.1.g"])
    return out + x


def full_attention(sd: SD, p: str, x: torch.Tensor, heads: int, dim_head: int) -> torch.Tensor:
    """Residual(PreNorm(Attention)), unet.py:234-246."""
    b, c, n = x.shape
    xn = _q8(channel_layernorm(x, sd[p + ".fn.norm.g"])) if c <= 1024 else channel_layernorm(x, sd[p + ".fn.norm.g"])
    qkv = F.conv1d(xn, sd[p + ".fn.fn.to_qkv.weight"]).chunk(3, dim=1)
    q, k, v = (t.reshape(b, heads, dim_head, n) for t in qkv)
    q = q * dim_head ** -0.5
    sim = torch.einsum("bhdi,bhdj->bhij", q, k)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhdj->bhid", attn, v)           # [b,h,n,d]
    out = out.permute(0, 1, 3, 2).reshape(b, heads * dim_head, n)
    out = F.conv1d(out, sd[p + ".fn.fn.to_out.weight"], sd[p + ".fn.fn.to_out.bias"])
    return out + x


def cond_upsample(sd: SD, u: UnetConfig, cond: torch.Tensor, prefix: str = "diff_model") -> torch.Tensor:
    """Unet1D.upsampling_layers applied in order (unet.py:372-377; called from sample.py:125-128):
    non-causal SConvTranspose1d(k=2r, s=r), no weight-norm."""
    x = cond
    for i, r in enumerate(u.upsampling_ratios or ()):
        p = f"{prefix}.upsampling_layers.{i}.convtr.convtr"
        x = sconvtr1d(x, sd[p + ".weight"], sd[p + ".bias"], r, causal=False)
    return x


def process_cond(sd: SD, u: UnetConfig, cond: torch.Tensor, prefix: str = "diff_model") -> torch.Tensor:
    """Unet1D.process_cond, unet.py:407-420: upsample, then per-item max-abs scaling (scaling(), :401-403)."""
    x = cond_upsample(sd, u, cond, prefix) if u.upsampling_ratios is not None else cond
    if u.unet_scale_cond:
        B = x.shape[0]
        scale = x.reshape(B, -1).abs().max(dim=1, keepdim=True).values.unsqueeze(-1)
        x = x / (scale + 1e-20)
    return x


def unet_forward(sd: SD, u: UnetConfig, x: torch.Tensor, t: torch.Tensor, cond: torch.Tensor,
                 prefix: str = "diff_model", taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """Unet1D.forward, unet.py:422-469 (other_cond=True, use_film=False, self_condition=False)."""
    g = spec.unet_graph(u, prefix)
    xc = process_cond(sd, u, cond, prefix)
    x = torch.cat((xc, x), dim=1)
    if u.unet_scale_x:
        B = x.shape[0]
        s = x.reshape(B, -1).abs().max(dim=1, keepdim=True).values.unsqueeze(-1)
        x = x / (s + 1e-20)
    x = F.conv1d(x, sd[prefix + ".init_conv.weight"], sd[prefix + ".init_conv.bias"], padding=3)
    r = x.clone()
    temb = time_embedding(sd, u, t.to(torch.float32), prefix)
    if taps is not None:
        taps["cond_proc"] = xc
        taps["init"] = x
        taps["temb"] = temb
    hs: List[torch.Tensor] = []
    for i, lv in enumerate(g.downs):
        x = resnet_block(sd, lv.block1, x, temb, u.groups); hs.append(x)
        x = resnet_block(sd, lv.block2, x, temb, u.groups)
        x = linear_attention(sd, lv.attn_prefix, x, u.heads, u.dim_head); hs.append(x)
        w, b = sd[lv.resample_prefix + ".weight"], sd[lv.resample_prefix + ".bias"]
        x = F.conv1d(x, w, b, stride=2, padding=1) if lv.resample_kind == "down" else F.conv1d(x, w, b, padding=1)
        if taps is not None:
            taps[f"down{i}"] = x
    x = resnet_block(sd, g.mid1, x, temb, u.groups)
    x = full_attention(sd, prefix + ".mid_attn", x, u.heads, u.dim_head)
    x = resnet_block(sd, g.mid2, x, temb, u.groups)
    if taps is not None:
        taps["mid"] = x
    for i, lv in enumerate(g.ups):
        x = resnet_block(sd, lv.block1, torch.cat((x, hs.pop()), dim=1), temb, u.groups)
        x = resnet_block(sd, lv.block2, torch.cat((x, hs.pop()), dim=1), temb, u.groups)
        x = linear_attention(sd, lv.attn_prefix, x, u.heads, u.dim_head)
        w, b = sd[lv.resample_prefix + ".weight"], sd[lv.resample_prefix + ".bias"]
        if lv.resample_kind == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = F.conv1d(x, w, b, padding=1)
        if taps is not None:
            taps[f"up{i}"] = x
    x = resnet_block(sd, g.final, torch.cat((x, r), dim=1), temb, u.groups)
    x = _q8(torch.tanh(x))
    return F.conv1d(x, sd[prefix + ".final_conv.weight"], sd[prefix + ".final_conv.bias"])


# ----------------------------------------------------------------------------------------------
# L3 diffusion  (reference srcs/losses/ddpm_loss.py)
# ----------------------------------------------------------------------------------------------

def p_sample_update(sd: SD, x: torch.Tensor, eps: torch.Tensor, t: int, noise: Optional[torch.Tensor]) -> torch.Tensor:
    """The arithmetic of p_sample after the model call, ddpm_loss.py:175-179 (x0 from eps), :237-238
    (clamp), :199-206 (posterior mean), :249-250 (add noise unless t == 0)."""
    x0 = sd["diffusion.sqrt_recip_alphas_cumprod"][t] * x - sd["diffusion.sqrt_recipm1_alphas_cumprod"][t] * eps
    x0 = x0.clamp(-1.0, 1.0)
    mean = sd["diffusion.posterior_mean_coef1"][t] * x0 + sd["diffusion.posterior_mean_coef2"][t] * x
    if t > 0:
        assert noise is not None
        return mean + (0.5 * sd["diffusion.posterior_log_variance_clipped"][t]).exp() * noise
    return mean


def p_sample(sd: SD, u: UnetConfig, x: torch.Tensor, t: int, cond: torch.Tensor, noise: Optional[torch.Tensor]) -> torch.Tensor:
    """GaussianDiffusion1D.p_sample, ddpm_loss.py:244-251."""
    tt = torch.full((x.shape[0],), t, dtype=torch.long)
    eps = unet_forward(sd, u, x, tt, cond, prefix="diffusion.model" if "diffusion.model.init_conv.weight" in sd else "diff_model")
    return p_sample_update(sd, x, eps, t, noise)


def halfway_sampling(sd: SD, u: UnetConfig, img: torch.Tensor, cond: torch.Tensor, n_steps: int,
                     noises: Optional[torch.Tensor]) -> torch.Tensor:
    """GaussianDiffusion1D.halfway_sampling, ddpm_loss.py:370-385: t = n_steps-1 ... 0.
    `noises[j]` is the draw consumed at loop iteration j (t = n_steps-1-j); the last (t=0) is unused."""
    if img.shape == cond.shape:
        img = cond_upsample(sd, u, img)
    for j, t in enumerate(reversed(range(n_steps))):
        img = p_sample(sd, u, img, t, cond, None if t == 0 else noises[j])
    return img


def _draw(noises, j):
    return noises(j) if callable(noises) else noises[j]


def p_sample_loop(sd: SD, u: UnetConfig, img: torch.Tensor, cond: torch.Tensor, noises) -> torch.Tensor:
    """GaussianDiffusion1D.p_sample_loop, ddpm_loss.py:253-266: t = T-1 ... 0 over ALL timesteps, starting from `img`
    (the reference draws it ~ N(0,1), :257).  `noises[j]` (or noises(j)) is the draw of loop iteration j."""
    T = int(sd["diffusion.betas"].shape[0])
    for j, t in enumerate(reversed(range(T))):
        img = p_sample(sd, u, img, t, cond, None if t == 0 else _draw(noises, j))
    return img


def infilling(sd: SD, u: UnetConfig, img: torch.Tensor, infill_img: torch.Tensor, cond: torch.Tensor, midway_t: int,
              noises, lam: float = 0.8):
    """GaussianDiffusion1D.infilling, ddpm_loss.py:331-367 (self_condition False).  `img` is the start image (the
    reference draws it ~ U[0,1), :336); draws 2i / 2i+1 feed the two p_sample calls of iteration i.  Returns
    (img, infill_img); the reference returns img."""
    for i, t in enumerate(reversed(range(midway_t))):
        img = p_sample(sd, u, img, t, cond, None if t == 0 else _draw(noises, 2 * i))            # :349
        img = (1 - lam) * img + lam * infill_img                                                   # :357
        infill_img = p_sample(sd, u, infill_img, t, cond, None if t == 0 else _draw(noises, 2 * i + 1))   # :359
        img = (1 - lam) * img + lam * infill_img                                                   # :361
    return img, infill_img


# ----------------------------------------------------------------------------------------------
# L5 harness arithmetic  (reference srcs/sample.py:94-134)
# ----------------------------------------------------------------------------------------------

def start_image(sd: SD, u: UnetConfig, cond: torch.Tensor, per_item: bool = False) -> torch.Tensor:
    """sample.py:125-129: upsample cond, divide by max|.|+1e-8 over the whole tensor (one file = one
    batch in the reference).  per_item=True applies it per utterance (batch of independent files)."""
    img = cond_upsample(sd, u, cond)
    if per_item:
        m = img.reshape(img.shape[0], -1).abs().max(dim=1).values.view(-1, 1, 1)
    else:
        m = img.abs().max()
    return img / (m + 1e-8)


def output_normalise(x: torch.Tensor, per_item: bool = False) -> torch.Tensor:
    """sample.py:133-134: x /= std(x)+1e-8 (unbiased, whole tensor); x /= max|x|+1e-8."""
    if per_item:
        flat = x.reshape(x.shape[0], -1)
        x = x / (flat.std(dim=1).view(-1, 1, 1) + 1e-8)
        flat = x.reshape(x.shape[0], -1)
        return x / (flat.abs().max(dim=1).values.view(-1, 1, 1) + 1e-8)
    x = x / (torch.std(x.flatten()) + 1e-8)
    return x / (torch.max(torch.abs(x.flatten())) + 1e-8)


def decode_utterances(sd_cond: SD, cc: CodecConfig, sd_main: SD, mc: CodecConfig, u: UnetConfig, wav: torch.Tensor,
                      n_steps: int, noises: Optional[torch.Tensor], cond_bandwidth: Optional[float] = None,
                      per_item: bool = False) -> Dict[str, torch.Tensor]:
    """synthesis() per-file body, sample.py:87-134, with the stage tensors of SURVEY.md §8b(3)."""
    cond, codes, margins, z = get_cond(sd_cond, cc, wav, cond_bandwidth)
    img0 = start_image(sd_main, u, cond, per_item)
    latents = halfway_sampling(sd_main, u, img0, cond, n_steps, noises)
    wav_raw = seanet_decode(sd_main, mc, latents)              # quirk Q3: no x18 un-scaling here
    return {"z": z, "cond": cond, "codes": codes, "margins": margins, "img0": img0, "latents": latents,
            "wav_raw": wav_raw, "wav": output_normalise(wav_raw, per_item)}
