"""Generate tests/golden/train256.npz: ONE optimisation-step's worth of expectations of the diffusion UNet at the size
BASELINE configs[3] names (diff_dims 256, seq_length 1200, enc_ratios 8 4), from the REFERENCE under torch autograd:
DiffAudioRep.forward (srcs/model.py:146-209) on two 2.4 s utterances -- frozen encoder, x_rep / 18, GaussianDiffusion1D.p_losses
(srcs/losses/ddpm_loss.py:404-441: q_sample, the no-grad predicted_x_start, the l1 objective with p2 weights), loss.backward(),
decoder(predicted_x_start * 18) and the SD-SDR monitoring loss `neg_loss` (model.py:181-196).

asteroid (requirements.txt: asteroid==0.6.0) is absent from the image: ClippedSDR (srcs/losses/losses_fn.py:56-66) wraps
asteroid.losses.sdr.MultiSrcNegSDR("sdsdr"), whose published algorithm is restated in `neg_sdsdr` below (zero-mean, target
scaling by <est, tgt> / |tgt|^2, e_noise = est - tgt, 10 log10, EPS 1e-8, mean over sources, negated).

Run in the build container only:   python tools/gen_golden_train256.py
Gradients are stored as strided samples of selected parameters (a k = 7 conv, a k = 4 stride-2 downsample, k = 3 convs incl. a
concatenated-input one, an upsample conv, 1x1 / Linear layers, norms, both transposed-conv upsamplers) plus the float64 sum of |g|
of EVERY parameter.
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")

from ref_import import import_reference  # noqa: E402
from gen_golden import NoiseTape, build_cond_model, build_main_model, np32  # noqa: E402
from ladiffcodec_amd import synth  # noqa: E402
from ladiffcodec_amd.spec import CodecConfig, UnetConfig  # noqa: E402
from helpers import sub, sub_stride  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "train256.npz")
SAMPLED = ["init_conv.weight", "init_conv.bias", "time_mlp.1.weight", "time_mlp.3.bias", "downs.0.0.block1.proj.weight", "downs.0.0.mlp.1.weight",
           "downs.0.0.block2.norm.weight", "downs.0.2.fn.norm.g", "downs.0.2.fn.fn.to_qkv.weight", "downs.0.2.fn.fn.to_out.0.weight",
           "downs.0.2.fn.fn.to_out.1.g", "downs.0.3.weight", "downs.2.0.block1.proj.weight", "downs.4.3.weight",
           "mid_block1.block2.proj.weight", "mid_attn.fn.fn.to_qkv.weight", "mid_attn.fn.fn.to_out.weight", "ups.0.0.block1.proj.weight",
           "ups.0.0.res_conv.weight", "ups.2.1.res_conv.weight", "ups.1.3.1.weight", "ups.4.1.block2.proj.bias", "final_res_block.block1.proj.weight", "final_conv.weight",
           "final_conv.bias", "upsampling_layers.0.convtr.convtr.weight", "upsampling_layers.1.convtr.convtr.weight",
           "upsampling_layers.1.convtr.convtr.bias"]


def neg_sdsdr(est_targets, targets, eps=1e-8):
    """asteroid 0.6.0 MultiSrcNegSDR('sdsdr', zero_mean=True, take_log=True): inputs [batch, n_src, time] -> [batch]"""
    est_targets = est_targets - est_targets.mean(dim=2, keepdim=True)
    targets = targets - targets.mean(dim=2, keepdim=True)
    dot = torch.sum(est_targets * targets, dim=2, keepdim=True)
    energy = torch.sum(targets ** 2, dim=2, keepdim=True) + eps
    scaled = dot * targets / energy
    e_noise = est_targets - targets
    losses = torch.sum(scaled ** 2, dim=2) / (torch.sum(e_noise ** 2, dim=2) + eps)
    losses = 10 * torch.log10(losses + eps)
    return -losses.mean(dim=-1)


def main():
    torch.set_num_threads(8)
    ref = import_reference()
    import srcs.losses.ddpm_loss as ref_ddpm
    mc = CodecConfig(enc_ratios=(8, 4), quantization=False)
    u = UnetConfig(dim=256, upsampling_ratios=(5, 2), unet_scale_cond=True)
    cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0)
    model = build_main_model(ref, mc, u, seed=1)
    cond_model = build_cond_model(ref, cc, seed=11)
    wav = torch.from_numpy(synth.synthetic_wav(2, 38400, seed=5))
    t = torch.tensor([37, 812])
    noise = torch.randn(2, 128, 1200, generator=torch.Generator().manual_seed(6))
    out = {"t": t.numpy().astype(np.int64)}
    with torch.no_grad():
        cond = cond_model.get_cond(wav)
        x_rep = model.encoder(wav)
    x_rep, scale = model.scaling(x_rep, global_max=18.0)                  # model.py:165 (--scaling_global)
    model.diffusion.seq_length = x_rep.shape[-1]                          # --seq_length 1200 (train.py: the latent length)
    tape = NoiseTape([noise])
    ref_ddpm.torch.randn_like, saved = tape, ref_ddpm.torch.randn_like
    try:
        loss, predicted_x_start, x_t, _t = model.diffusion(x_rep.detach(), cond, t=t)   # ddpm_loss.py:443-450 -> p_losses
    finally:
        ref_ddpm.torch.randn_like = saved
    assert tape.i == 1
    loss.backward()
    with torch.no_grad():
        x_hat = model.decoder(predicted_x_start * scale)
        neg = torch.clamp(neg_sdsdr(wav, x_hat), min=-30.0)                # ClippedSDR(est_targets = x, targets = x_hat), model.py:194
        neg_loss = neg.mean()
    out["loss"] = np32(loss.reshape(1))
    out["neg_loss"] = np32(neg_loss.reshape(1))
    out["neg_per_item"] = np32(neg)
    for key, ten in (("x_rep", x_rep), ("x_t", x_t), ("predicted_x_start", predicted_x_start), ("x_hat", x_hat)):
        a = np32(ten)
        st = sub_stride(a.size, 40_000)
        out[key], out[key + ".stride"] = sub(a, st), np.array(st, np.int64)
    names, sums = [], []
    for name, prm in model.diff_model.named_parameters():
        g = prm.grad
        assert g is not None, name
        names.append(name)
        sums.append(float(g.double().abs().sum()))
        if name in SAMPLED:
            a = np32(g).reshape(-1)
            st = max(1, a.size // 20_000)
            st = st + 1 if st > 1 and st % 2 == 0 else st              # odd strides: no aliasing with the power-of-two tensor dims
            out["g." + name] = np.ascontiguousarray(a[::st])
            out["g." + name + ".stride"] = np.array(st, np.int64)
    missing = [n for n in SAMPLED if "g." + n not in out]
    assert not missing, missing
    out["names"] = np.array(names)
    out["abs_sums"] = np.array(sums, np.float64)
    np.savez_compressed(OUT, **out)
    print("loss", float(loss), "neg_loss", float(neg_loss), "params", len(names), "->", OUT, os.path.getsize(OUT) / 1e6, "MB")


if __name__ == "__main__":
    main()
