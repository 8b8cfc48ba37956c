"""Checkpoint key/shape specification of the two models on the decode path.

The drop-in boundary includes the `.amlt` checkpoint format: `torch.save(state_dict)` of
`DiffAudioRep` (reference srcs/utils.py:85-108, key families listed in SURVEY.md §8b).  This module
enumerates, from the CLI-level configuration alone, every key and shape such a state dict holds, in
the reference's registration order.  It is used by

  * the loader (`checkpoint.py`) for the strict key/shape check that `load_state_dict(strict=True)`
    performs in the reference (srcs/sample.py:58),
  * the synthetic-checkpoint generator (`synth.py`),
  * the oracle and the tests, to walk the layer graph without any `nn.Module`.

Layer numbering follows the `nn.Sequential` positions of SEANetEncoder/Decoder
(reference srcs/modules/seanet.py:108-151, 202-244) and the ModuleList layout of Unet1D
(reference srcs/modules/unet.py:307-377).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

Shape = Tuple[int, ...]

SCHEDULE_BUFFERS = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
    "posterior_mean_coef1", "posterior_mean_coef2", "p2_loss_weight",
)


@dataclass
class CodecConfig:
    """Arguments of DiffAudioRep that shape the SEANet encoder/decoder + RVQ (model.py:34-66)."""
    rep_dims: int = 128
    n_filters: int = 32
    n_residual_layers: int = 1
    lstm: int = 2
    enc_ratios: Sequence[int] = (8, 5, 4, 2)
    dilation_base: int = 2
    quantization: bool = False
    bandwidth: float = 3.0
    sample_rate: int = 16000
    bins: int = 1024
    final_activation: Optional[str] = None     # nn module name applied after the encoder's last conv (seanet.py:144-149)

    @property
    def hop_length(self) -> int:
        h = 1
        for r in self.enc_ratios:
            h *= r
        return h

    @property
    def frame_rate(self) -> float:
        return self.sample_rate / self.hop_length

    @property
    def n_q_layers(self) -> int:
        """Number of codebooks the quantizer is BUILT with (model.py:64-66)."""
        import math
        return int(1000 * self.bandwidth // (math.ceil(self.frame_rate) * 10))

    def n_q_for_bandwidth(self, bandwidth: Optional[float] = None) -> int:
        """Number of codebooks USED in forward (vq.py:86-98)."""
        import math
        bw = self.bandwidth if bandwidth is None else bandwidth
        bw_per_q = math.log2(self.bins) * self.frame_rate / 1000
        n_q = self.n_q_layers
        if bw and bw > 0.0:
            n_q = int(max(1, math.floor(bw / bw_per_q)))
        return n_q


@dataclass
class UnetConfig:
    """Arguments of Unet1D as DiffAudioRep builds it (model.py:74)."""
    dim: int = 256
    dim_mults: Sequence[int] = (1, 2, 2, 4, 4)
    inp_channels: int = 128
    cond_channels: int = 128
    upsampling_ratios: Optional[Sequence[int]] = (5, 4, 2)
    unet_scale_cond: bool = True
    unet_scale_x: bool = False
    groups: int = 8
    heads: int = 4
    dim_head: int = 32
    timesteps: int = 1000

    @property
    def time_dim(self) -> int:
        return self.dim * 4

    @property
    def dims(self) -> List[int]:
        return [self.dim] + [self.dim * m for m in self.dim_mults]

    @property
    def in_out(self) -> List[Tuple[int, int]]:
        d = self.dims
        return list(zip(d[:-1], d[1:]))


# ----------------------------------------------------------------------------------------------
# SEANet layer walk
# ----------------------------------------------------------------------------------------------

@dataclass
class SeanetLayer:
    kind: str                 # 'conv' | 'convtr' | 'res' | 'lstm' | 'elu'
    index: int                # position in nn.Sequential
    cin: int = 0
    cout: int = 0
    kernel: int = 1
    stride: int = 1
    dilation: int = 1
    hidden: int = 0           # res only
    layers: int = 0           # lstm only


def seanet_encoder_layers(c: CodecConfig) -> List[SeanetLayer]:
    """seanet.py:108-151 (ratios are consumed reversed, :101)."""
    out: List[SeanetLayer] = []
    idx = 0
    mult = 1
    out.append(SeanetLayer("conv", idx, 1, mult * c.n_filters, 7)); idx += 1
    for ratio in reversed(list(c.enc_ratios)):
        ch = mult * c.n_filters
        for j in range(c.n_residual_layers):
            out.append(SeanetLayer("res", idx, ch, ch, 3, 1, c.dilation_base ** j, hidden=ch // 2)); idx += 1
        out.append(SeanetLayer("elu", idx)); idx += 1
        out.append(SeanetLayer("conv", idx, ch, ch * 2, ratio * 2, ratio)); idx += 1
        mult *= 2
    ch = mult * c.n_filters
    if c.lstm:
        out.append(SeanetLayer("lstm", idx, ch, ch, layers=c.lstm)); idx += 1
    out.append(SeanetLayer("elu", idx)); idx += 1
    out.append(SeanetLayer("conv", idx, ch, c.rep_dims, 7)); idx += 1
    return out


def seanet_decoder_layers(c: CodecConfig) -> List[SeanetLayer]:
    """seanet.py:200-244."""
    out: List[SeanetLayer] = []
    idx = 0
    mult = int(2 ** len(c.enc_ratios))
    out.append(SeanetLayer("conv", idx, c.rep_dims, mult * c.n_filters, 7)); idx += 1
    if c.lstm:
        out.append(SeanetLayer("lstm", idx, mult * c.n_filters, mult * c.n_filters, layers=c.lstm)); idx += 1
    for ratio in c.enc_ratios:
        ch = mult * c.n_filters
        out.append(SeanetLayer("elu", idx)); idx += 1
        out.append(SeanetLayer("convtr", idx, ch, ch // 2, ratio * 2, ratio)); idx += 1
        for j in range(c.n_residual_layers):
            out.append(SeanetLayer("res", idx, ch // 2, ch // 2, 3, 1, c.dilation_base ** j, hidden=ch // 4)); idx += 1
        mult //= 2
    out.append(SeanetLayer("elu", idx)); idx += 1
    out.append(SeanetLayer("conv", idx, c.n_filters, 1, 7)); idx += 1
    return out


def _wn_conv_keys(prefix: str, cout: int, cin: int, k: int) -> List[Tuple[str, Shape]]:
    # old-style weight_norm registers bias, weight_g, weight_v in this order (conv.py:29-30)
    return [(prefix + ".bias", (cout,)), (prefix + ".weight_g", (cout, 1, 1)), (prefix + ".weight_v", (cout, cin, k))]


def _seanet_keys(prefix: str, layers: List[SeanetLayer]) -> List[Tuple[str, Shape]]:
    keys: List[Tuple[str, Shape]] = []
    for ly in layers:
        p = f"{prefix}.model.{ly.index}"
        if ly.kind == "conv":
            keys += _wn_conv_keys(p + ".conv.conv", ly.cout, ly.cin, ly.kernel)
        elif ly.kind == "convtr":
            # ConvTranspose1d weight is [Cin, Cout, k]; weight_g is per INPUT channel
            keys += [(p + ".convtr.convtr.bias", (ly.cout,)),
                     (p + ".convtr.convtr.weight_g", (ly.cin, 1, 1)),
                     (p + ".convtr.convtr.weight_v", (ly.cin, ly.cout, ly.kernel))]
        elif ly.kind == "res":
            keys += _wn_conv_keys(p + ".block.1.conv.conv", ly.hidden, ly.cin, ly.kernel)
            keys += _wn_conv_keys(p + ".block.3.conv.conv", ly.cout, ly.hidden, 1)
            keys += _wn_conv_keys(p + ".shortcut.conv.conv", ly.cout, ly.cin, 1)
        elif ly.kind == "lstm":
            h = ly.cout
            for n in range(ly.layers):
                keys += [(f"{p}.lstm.weight_ih_l{n}", (4 * h, h)), (f"{p}.lstm.weight_hh_l{n}", (4 * h, h)),
                         (f"{p}.lstm.bias_ih_l{n}", (4 * h,)), (f"{p}.lstm.bias_hh_l{n}", (4 * h,))]
    return keys


def codec_keys(c: CodecConfig) -> List[Tuple[str, Shape]]:
    keys = _seanet_keys("encoder", seanet_encoder_layers(c)) + _seanet_keys("decoder", seanet_decoder_layers(c))
    if c.quantization:
        for q in range(c.n_q_layers):
            p = f"quantizer.vq.layers.{q}._codebook"
            keys += [(p + ".inited", (1,)), (p + ".cluster_size", (c.bins,)),
                     (p + ".embed", (c.bins, c.rep_dims)), (p + ".embed_avg", (c.bins, c.rep_dims))]
    return keys


# ----------------------------------------------------------------------------------------------
# Unet1D layer walk
# ----------------------------------------------------------------------------------------------

@dataclass
class ResnetSpec:
    prefix: str
    cin: int
    cout: int

    @property
    def has_res_conv(self) -> bool:
        return self.cin != self.cout


@dataclass
class UnetLevel:
    block1: ResnetSpec
    block2: ResnetSpec
    attn_prefix: str
    attn_dim: int
    resample_prefix: str      # key prefix of the conv weight
    resample_kind: str        # 'down' (k4 s2 p1) | 'up' (nearest x2 + k3 p1) | 'same' (k3 p1)
    resample_cin: int
    resample_cout: int


@dataclass
class UnetGraph:
    downs: List[UnetLevel] = field(default_factory=list)
    ups: List[UnetLevel] = field(default_factory=list)
    mid1: Optional[ResnetSpec] = None
    mid2: Optional[ResnetSpec] = None
    final: Optional[ResnetSpec] = None

    def resnet_blocks(self) -> List[ResnetSpec]:
        """All ResnetBlocks in execution order (unet.py:440-466)."""
        out: List[ResnetSpec] = []
        for lv in self.downs:
            out += [lv.block1, lv.block2]
        out += [self.mid1, self.mid2]
        for lv in self.ups:
            out += [lv.block1, lv.block2]
        out.append(self.final)
        return out


def unet_graph(u: UnetConfig, prefix: str = "diff_model") -> UnetGraph:
    g = UnetGraph()
    in_out = u.in_out
    n = len(in_out)
    for i, (din, dout) in enumerate(in_out):
        last = i >= n - 1
        p = f"{prefix}.downs.{i}"
        g.downs.append(UnetLevel(ResnetSpec(p + ".0", din, din), ResnetSpec(p + ".1", din, din), p + ".2", din,
                                 p + ".3", "same" if last else "down", din, dout))
    mid = u.dims[-1]
    g.mid1 = ResnetSpec(prefix + ".mid_block1", mid, mid)
    g.mid2 = ResnetSpec(prefix + ".mid_block2", mid, mid)
    for i, (din, dout) in enumerate(reversed(in_out)):
        last = i == n - 1
        p = f"{prefix}.ups.{i}"
        g.ups.append(UnetLevel(ResnetSpec(p + ".0", dout + din, dout), ResnetSpec(p + ".1", dout + din, dout),
                               p + ".2", dout, p + (".3" if last else ".3.1"), "same" if last else "up", dout, din))
    g.final = ResnetSpec(prefix + ".final_res_block", u.dim * 2, u.dim)
    return g


def _resnet_keys(r: ResnetSpec, time_dim: int) -> List[Tuple[str, Shape]]:
    p = r.prefix
    keys = [(p + ".mlp.1.weight", (2 * r.cout, time_dim)), (p + ".mlp.1.bias", (2 * r.cout,)),
            (p + ".block1.proj.weight", (r.cout, r.cin, 3)), (p + ".block1.proj.bias", (r.cout,)),
            (p + ".block1.norm.weight", (r.cout,)), (p + ".block1.norm.bias", (r.cout,)),
            (p + ".block2.proj.weight", (r.cout, r.cout, 3)), (p + ".block2.proj.bias", (r.cout,)),
            (p + ".block2.norm.weight", (r.cout,)), (p + ".block2.norm.bias", (r.cout,))]
    if r.has_res_conv:
        keys += [(p + ".res_conv.weight", (r.cout, r.cin, 1)), (p + ".res_conv.bias", (r.cout,))]
    return keys


def _linattn_keys(p: str, dim: int, hidden: int) -> List[Tuple[str, Shape]]:
    return [(p + ".fn.fn.to_qkv.weight", (3 * hidden, dim, 1)), (p + ".fn.fn.to_out.0.weight", (dim, hidden, 1)),
            (p + ".fn.fn.to_out.0.bias", (dim,)), (p + ".fn.fn.to_out.1.g", (1, dim, 1)), (p + ".fn.norm.g", (1, dim, 1))]


def unet_keys(u: UnetConfig, prefix: str = "diff_model") -> List[Tuple[str, Shape]]:
    g = unet_graph(u, prefix)
    hidden = u.heads * u.dim_head
    cin0 = u.inp_channels + u.cond_channels
    keys: List[Tuple[str, Shape]] = [
        (prefix + ".init_conv.weight", (u.dim, cin0, 7)), (prefix + ".init_conv.bias", (u.dim,)),
        (prefix + ".time_mlp.1.weight", (u.time_dim, u.dim)), (prefix + ".time_mlp.1.bias", (u.time_dim,)),
        (prefix + ".time_mlp.3.weight", (u.time_dim, u.time_dim)), (prefix + ".time_mlp.3.bias", (u.time_dim,)),
    ]
    for lv in g.downs:
        keys += _resnet_keys(lv.block1, u.time_dim) + _resnet_keys(lv.block2, u.time_dim)
        keys += _linattn_keys(lv.attn_prefix, lv.attn_dim, hidden)
        k = 4 if lv.resample_kind == "down" else 3
        keys += [(lv.resample_prefix + ".weight", (lv.resample_cout, lv.resample_cin, k)),
                 (lv.resample_prefix + ".bias", (lv.resample_cout,))]
    for lv in g.ups:
        keys += _resnet_keys(lv.block1, u.time_dim) + _resnet_keys(lv.block2, u.time_dim)
        keys += _linattn_keys(lv.attn_prefix, lv.attn_dim, hidden)
        keys += [(lv.resample_prefix + ".weight", (lv.resample_cout, lv.resample_cin, 3)),
                 (lv.resample_prefix + ".bias", (lv.resample_cout,))]
    keys += _resnet_keys(g.mid1, u.time_dim)
    mp = prefix + ".mid_attn"
    keys += [(mp + ".fn.fn.to_qkv.weight", (3 * hidden, u.dims[-1], 1)), (mp + ".fn.fn.to_out.weight", (u.dims[-1], hidden, 1)),
             (mp + ".fn.fn.to_out.bias", (u.dims[-1],)), (mp + ".fn.norm.g", (1, u.dims[-1], 1))]
    keys += _resnet_keys(g.mid2, u.time_dim)
    keys += _resnet_keys(g.final, u.time_dim)
    keys += [(prefix + ".final_conv.weight", (u.inp_channels, u.dim, 1)), (prefix + ".final_conv.bias", (u.inp_channels,))]
    if u.upsampling_ratios is not None:
        for i, r in enumerate(u.upsampling_ratios):
            p = f"{prefix}.upsampling_layers.{i}.convtr.convtr"
            keys += [(p + ".weight", (u.cond_channels, u.cond_channels, 2 * r)), (p + ".bias", (u.cond_channels,))]
    return keys


def ladiff_keys(c: CodecConfig, u: UnetConfig) -> List[Tuple[str, Shape]]:
    """Full key set of the diffusion checkpoint: codec (no quantizer) + the UNet under BOTH
    `diff_model.*` and `diffusion.model.*` (same tensors registered twice, model.py:74,106) + 13
    schedule buffers (ddpm_loss.py:138-168)."""
    keys = codec_keys(c)
    keys += unet_keys(u, "diff_model")
    keys += [(f"diffusion.{b}", (u.timesteps,)) for b in SCHEDULE_BUFFERS]
    keys += unet_keys(u, "diffusion.model")
    return keys


def as_dict(keys: List[Tuple[str, Shape]]) -> Dict[str, Shape]:
    return dict(keys)
