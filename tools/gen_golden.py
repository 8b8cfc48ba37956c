"""Generate tests/golden/*.npz by running the REFERENCE (/root/reference, imported read-only with
placeholder modules for its absent deps) on seeded synthetic checkpoints and inputs.

Run in the build container only:   python tools/gen_golden.py
The fixtures hold inputs and expected outputs (data); the weights are regenerated from
(config, seed) by ladiffcodec_amd.synth at test time, so no checkpoint is committed.

Loading the synthetic state dicts into the reference modules with strict=True doubles as the check
that ladiffcodec_amd/spec.py enumerates exactly the reference's key set and shapes.
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
warnings.filterwarnings("ignore")

from ref_import import import_reference  # noqa: E402
from ladiffcodec_amd import synth  # noqa: E402
from ladiffcodec_amd.spec import CodecConfig, UnetConfig  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def build_cond_model(ref, cc: CodecConfig, seed: int):
    # exactly as reference srcs/sample.py:63 (the `ratios=` kwarg is swallowed: quirk Q1)
    m = ref.DiffAudioRep(rep_dims=cc.rep_dims, emb_dims=128, n_residual_layers=cc.n_residual_layers,
                         n_filters=cc.n_filters, lstm=cc.lstm, quantization=True, bandwidth=cc.bandwidth,
                         ratios=[8, 5, 4, 2], final_activation=cc.final_activation)
    sd = synth.to_torch(synth.codec_state_dict(cc, seed))
    m.load_state_dict(sd, strict=True)
    return m.eval()


def build_main_model(ref, mc: CodecConfig, u: UnetConfig, seed: int):
    # as reference srcs/sample.py:56 with the README flags (--run_diff --scaling_global --unet_scale_cond)
    m = ref.DiffAudioRep(other_cond=True, rep_dims=mc.rep_dims, emb_dims=128, diff_dims=u.dim, n_filters=mc.n_filters,
                         lstm=mc.lstm, n_residual_layers=mc.n_residual_layers, enc_ratios=list(mc.enc_ratios),
                         upsampling_ratios=list(u.upsampling_ratios) if u.upsampling_ratios is not None else None,
                         run_diff=True, model_type="unet",
                         scaling_global=True, unet_scale_cond=u.unet_scale_cond, unet_scale_x=u.unet_scale_x,
                         sampling_timesteps=1000, quantization=False, bandwidth=3.0, cond_global=3.0, seq_length=16000)
    sd = synth.to_torch(synth.ladiff_state_dict(mc, u, seed))
    m.load_state_dict(sd, strict=True)
    return m.eval()


class NoiseTape:
    """Replaces torch.randn_like inside the reference's ddpm_loss so the noise draws are known."""

    def __init__(self, noises):
        self.noises = noises
        self.i = 0

    def __call__(self, x):
        n = self.noises[self.i]
        self.i += 1
        assert n.shape == x.shape
        return n


def main():
    os.makedirs(OUT, exist_ok=True)
    only = os.environ.get("GOLDEN_ONLY")   # regenerate a single fixture family (the zip containers carry timestamps)
    if only:
        real_save = np.savez_compressed

        def filtered(path, **kw):
            if only in os.path.basename(path):
                real_save(path, **kw)
        np.savez_compressed = filtered
    ref = import_reference()
    import srcs.losses.ddpm_loss as ref_ddpm
    from srcs.modules.conv import SConv1d, SConvTranspose1d
    from srcs.modules.lstm import SLSTM

    # ---------------------------------------------------------------- schedule tables (a9)
    diff = ref_ddpm.GaussianDiffusion1D(type("M", (torch.nn.Module,), {"channels": 128, "self_condition": False})(),
                                        seq_length=16, sampling_timesteps=1000)
    np.savez_compressed(os.path.join(OUT, "schedule.npz"), **{k: np32(v) for k, v in diff.state_dict().items()})

    # ---------------------------------------------------------------- primitive KATs (L1)
    g = torch.Generator().manual_seed(7)
    torch.manual_seed(7)      # the modules below are default-initialised from the global generator: seed it (regenerable fixture)
    kat = {}
    cases = [  # name, cin, cout, k, stride, dilation, causal, length
        ("c_k7", 8, 16, 7, 1, 1, True, 37), ("c_k4s2", 8, 16, 4, 2, 1, True, 37), ("c_k10s5", 8, 8, 10, 5, 1, True, 43),
        ("c_k16s8", 4, 8, 16, 8, 1, True, 100), ("c_k3d2", 8, 8, 3, 1, 2, True, 21), ("c_k3_nc", 8, 8, 3, 1, 1, False, 21),
        ("c_k7_short", 4, 4, 7, 1, 1, True, 5),
    ]
    for name, cin, cout, k, s, d, causal, length in cases:
        m = SConv1d(cin, cout, k, stride=s, dilation=d, causal=causal, norm="weight_norm")
        with torch.no_grad():
            m.conv.conv.weight_g.mul_(torch.rand(cout, 1, 1, generator=g) + 0.5)
        x = torch.randn(2, cin, length, generator=g)
        with torch.no_grad():
            y = m(x)
        kat[name + ".x"] = np32(x); kat[name + ".y"] = np32(y)
        kat[name + ".g"] = np32(m.conv.conv.weight_g); kat[name + ".v"] = np32(m.conv.conv.weight_v)
        kat[name + ".b"] = np32(m.conv.conv.bias)
        kat[name + ".cfg"] = np.array([k, s, d, int(causal)], np.int64)
    tcases = [("t_k16s8_c", 8, 4, 16, 8, True, 9, "weight_norm"), ("t_k4s2_c", 8, 8, 4, 2, True, 13, "weight_norm"),
              ("t_k10s5_nc", 8, 8, 10, 5, False, 11, "none"), ("t_k4s2_nc", 8, 8, 4, 2, False, 11, "none"),
              ("t_k8s4_nc", 8, 8, 8, 4, False, 7, "none")]
    for name, cin, cout, k, s, causal, length, norm in tcases:
        m = SConvTranspose1d(cin, cout, k, stride=s, causal=causal, norm=norm, trim_right_ratio=1.0)
        x = torch.randn(2, cin, length, generator=g)
        with torch.no_grad():
            y = m(x)
        kat[name + ".x"] = np32(x); kat[name + ".y"] = np32(y)
        if norm == "weight_norm":
            kat[name + ".g"] = np32(m.convtr.convtr.weight_g); kat[name + ".v"] = np32(m.convtr.convtr.weight_v)
        else:
            kat[name + ".w"] = np32(m.convtr.convtr.weight)
        kat[name + ".b"] = np32(m.convtr.convtr.bias)
        kat[name + ".cfg"] = np.array([k, s, 1, int(causal)], np.int64)
    lstm = SLSTM(16, num_layers=2)
    x = torch.randn(3, 16, 11, generator=g)
    with torch.no_grad():
        y = lstm(x)
    kat["lstm.x"] = np32(x); kat["lstm.y"] = np32(y)
    for k_, v_ in lstm.state_dict().items():
        kat["lstm.sd." + k_] = np32(v_)
    np.savez_compressed(os.path.join(OUT, "primitives.npz"), **kat)

    # ---------------------------------------------------------------- codec round trip (config C1 shape; a3-a5, a16)
    cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0)
    cond_model = build_cond_model(ref, cc, seed=11)
    wav = torch.from_numpy(synth.synthetic_wav(2, 6400, seed=1234)) * 0.5
    with torch.no_grad():
        z = cond_model.encoder(wav)
        q = cond_model.quantizer(z, sample_rate=cond_model.frame_rate, bandwidth=cond_model.bandwidth)
        q15 = cond_model.quantizer(z, sample_rate=cond_model.frame_rate, bandwidth=1.5)
        cond = cond_model.get_cond(wav)
        dec = cond_model.decoder(q.quantized)
    assert torch.equal(cond, q.quantized)
    np.savez_compressed(os.path.join(OUT, "codec_c1.npz"), wav=np32(wav), z=np32(z), quantized=np32(q.quantized),
                        codes=q.codes.numpy().astype(np.int64), quantized_1p5=np32(q15.quantized),
                        codes_1p5=q15.codes.numpy().astype(np.int64), decoded=np32(dec),
                        meta=np.array([11, 6400], np.int64))
    print("codec_c1: z", tuple(z.shape), "codes", tuple(q.codes.shape), "decoded", tuple(dec.shape))

    # ---------------------------------------------------------------- UNet single step + chain + e2e, [8,4] layout (C2 shape)
    def ladiff_case(tag, mc, u, T, n_chain, seed_w, seed_in):
        main = build_main_model(ref, mc, u, seed=seed_w)
        gg = torch.Generator().manual_seed(seed_in)
        wav = torch.from_numpy(synth.synthetic_wav(2, T, seed=seed_in)) * 0.5
        L = T // mc.hop_length
        out = {"wav": np32(wav), "meta": np.array([seed_w, T, n_chain], np.int64)}
        with torch.no_grad():
            cond = cond_model.get_cond(wav)
            out["cond"] = np32(cond)
            # single UNet call at two timesteps, random x (a10-a15)
            x = torch.randn(2, 128, L, generator=gg) * 0.7
            for t in (0, 37):
                tt = torch.full((2,), t, dtype=torch.long)
                out[f"eps_t{t}"] = np32(main.diff_model(x, tt, cond))
            out["x"] = np32(x)
            out["cond_proc"] = np32(main.diff_model.process_cond(cond))
            # start image (a6): sample.py:125-129
            img = cond
            for layer in main.diff_model.upsampling_layers:
                img = layer(img)
            out["img_up"] = np32(img)
            img = img / (torch.max(torch.abs(img.flatten())) + 1e-8)
            out["img0"] = np32(img)
            # chain (a7, a8) with recorded noise
            noises = torch.randn(n_chain, 2, 128, L, generator=gg)
            tape = NoiseTape(list(noises))
            ref_ddpm.torch.randn_like, saved = tape, ref_ddpm.torch.randn_like
            try:
                # NB: ddpm_loss uses `torch.randn_like`; patching the attribute on the module object `torch`
                # is global, so restore it right after.
                lat = main.diffusion.halfway_sampling(img=img.clone(), condition=cond, t=n_chain)
                one, _ = (main.diffusion.p_sample(x.clone(), 5, cond))
            finally:
                ref_ddpm.torch.randn_like = saved
            assert tape.i == n_chain  # n_chain-1 draws in the chain (t>0) + 1 in the single p_sample
            out["noises"] = np32(noises)
            out["latents"] = np32(lat)
            out["p_sample_t5"] = np32(one)       # used noises[n_chain-1]
            dec = main.decoder(lat)
            out["wav_raw"] = np32(dec)
            y = dec / (torch.std(dec.flatten()) + 1e-8)
            y = y / (torch.max(torch.abs(y.flatten())) + 1e-8)
            out["wav_out"] = np32(y)
        np.savez_compressed(os.path.join(OUT, f"ladiff_{tag}.npz"), **out)
        print(f"ladiff_{tag}: L={L} eps", out["eps_t0"].shape, "latents absmax", float(np.abs(out["latents"]).max()))

    # ---------------------------------------------------------------- p_sample_loop and infilling drivers
    def drivers_case(tag, mc, u, T, seed_w, seed_in, midway_t):
        main = build_main_model(ref, mc, u, seed=seed_w)
        wav = torch.from_numpy(synth.synthetic_wav(1, T, seed=seed_in)) * 0.5
        L = T // mc.hop_length
        n_t = main.diffusion.num_timesteps
        out = {"wav": np32(wav), "meta": np.array([seed_w, T, seed_in, midway_t, n_t], np.int64)}
        with torch.no_grad():
            cond = cond_model.get_cond(wav)
            out["cond"] = np32(cond)
            # p_sample_loop: start image torch.randn(shape) from the global generator, per-step draws from a tape
            # regenerated at test time from the same seeds (the tape itself would be 82 MB)
            torch.manual_seed(seed_in)
            img0 = torch.randn(1, 128, L)
            gg = torch.Generator().manual_seed(seed_in + 1)
            tape = NoiseTape([torch.randn(1, 128, L, generator=gg) for _ in range(n_t)])
            ref_ddpm.torch.randn_like, saved = tape, ref_ddpm.torch.randn_like
            ref_ddpm.tqdm, saved_tqdm = (lambda it, **k: it), ref_ddpm.tqdm
            try:
                torch.manual_seed(seed_in)
                res = main.diffusion.p_sample_loop((1, 128, L), condition=cond)
                assert tape.i == n_t - 1
                out["loop_img0"] = np32(img0)
                out["loop_out"] = np32(res)
                # infilling: start image torch.rand(...) from the global generator; `noise` given so that the
                # reference's unused default draw (:352) does not consume a tape entry
                main.diffusion.seq_length = L
                torch.manual_seed(seed_in + 2)
                fill0 = torch.rand(1, 128, L)
                infill = cond
                for layer in main.diff_model.upsampling_layers:
                    infill = layer(infill)
                infill = infill / (torch.max(torch.abs(infill.flatten())) + 1e-8)
                g2 = torch.Generator().manual_seed(seed_in + 3)
                tape2 = NoiseTape([torch.randn(1, 128, L, generator=g2) for _ in range(2 * midway_t)])
                ref_ddpm.torch.randn_like = tape2
                torch.manual_seed(seed_in + 2)
                filled = main.diffusion.infilling(infill.clone(), cond, midway_t=midway_t, noise=torch.zeros(1), lam=0.8)
                assert tape2.i == 2 * (midway_t - 1)
                out["fill_img0"] = np32(fill0)
                out["fill_infill0"] = np32(infill)
                out["fill_out"] = np32(filled)
            finally:
                ref_ddpm.torch.randn_like = saved
                ref_ddpm.tqdm = saved_tqdm
        np.savez_compressed(os.path.join(OUT, f"drivers_{tag}.npz"), **out)
        print(f"drivers_{tag}: loop_out absmax", float(np.abs(out["loop_out"]).max()), "fill_out absmax", float(np.abs(out["fill_out"]).max()))

    # ---------------------------------------------------------------- flag variants on the path: --unet_scale_x
    # (unet.py:432-433), upsampling_ratios=None (unet.py:411), --final_activation (seanet.py:144-149)
    def variants_case():
        out = {}
        gg = torch.Generator().manual_seed(2024)
        mc = CodecConfig(enc_ratios=(8, 4), quantization=False)
        with torch.no_grad():
            u = UnetConfig(dim=32, upsampling_ratios=(5, 2), unet_scale_cond=True, unet_scale_x=True)
            m = build_main_model(ref, mc, u, seed=31)
            x = torch.randn(2, 128, 160, generator=gg) * 1.7
            cond = torch.randn(2, 128, 16, generator=gg)
            out["sx.x"], out["sx.cond"] = np32(x), np32(cond)
            out["sx.eps_t37"] = np32(m.diff_model(x, torch.full((2,), 37, dtype=torch.long), cond))
            u = UnetConfig(dim=32, upsampling_ratios=None, unet_scale_cond=True)
            m = build_main_model(ref, mc, u, seed=32)
            x = torch.randn(2, 128, 160, generator=gg) * 0.7
            cond = torch.randn(2, 128, 160, generator=gg)
            out["nu.x"], out["nu.cond"] = np32(x), np32(cond)
            out["nu.eps_t37"] = np32(m.diff_model(x, torch.full((2,), 37, dtype=torch.long), cond))
            # (halfway_sampling itself raises AttributeError for this configuration -- ddpm_loss.py:376-378 touches
            # model.upsampling_layers, which does not exist when upsampling_ratios is None; p_sample works)
            one, _ = m.diffusion.p_sample(x.clone(), 0, cond)
            out["nu.p_sample_t0"] = np32(one)
            ccf = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0, final_activation="Tanh")
            cm = build_cond_model(ref, ccf, seed=11)
            wav = torch.from_numpy(synth.synthetic_wav(2, 3200, seed=55)) * 0.5
            z = cm.encoder(wav)
            q = cm.quantizer(z, sample_rate=cm.frame_rate, bandwidth=cm.bandwidth)
            out["fa.wav"], out["fa.z"], out["fa.codes"] = np32(wav), np32(z), q.codes.numpy().astype(np.int64)
            out["fa.quantized"] = np32(q.quantized)
        out["meta"] = np.array([31, 32, 11], np.int64)
        np.savez_compressed(os.path.join(OUT, "variants.npz"), **out)
        print("variants: sx eps absmax", float(np.abs(out["sx.eps_t37"]).max()), "fa z absmax", float(np.abs(out["fa.z"]).max()))

    # ---------------------------------------------------------------- training step, first slice
    def train_case():
        from srcs.modules.unet import Block
        out = {}
        torch.manual_seed(4040)
        gg = torch.Generator().manual_seed(4041)
        for tag, cin, cout, L, with_ss in (("a", 32, 64, 160, True), ("b", 64, 64, 75, False)):
            blk = Block(cin, cout, groups=8)
            with torch.no_grad():
                blk.norm.weight.copy_(torch.rand(cout, generator=gg) + 0.5)
                blk.norm.bias.copy_(torch.randn(cout, generator=gg) * 0.1)
            x = torch.randn(2, cin, L, generator=gg, requires_grad=True)
            ss = None
            if with_ss:
                scale = (torch.randn(2, cout, 1, generator=gg) * 0.3).requires_grad_()
                shift = (torch.randn(2, cout, 1, generator=gg) * 0.3).requires_grad_()
                ss = (scale, shift)
            y = blk(x, scale_shift=ss)
            dy = torch.randn(y.shape, generator=gg)
            y.backward(dy)
            out.update({f"{tag}.x": np32(x), f"{tag}.w": np32(blk.proj.weight), f"{tag}.b": np32(blk.proj.bias),
                        f"{tag}.gamma": np32(blk.norm.weight), f"{tag}.beta": np32(blk.norm.bias), f"{tag}.y": np32(y), f"{tag}.dy": np32(dy),
                        f"{tag}.dx": np32(x.grad), f"{tag}.dw": np32(blk.proj.weight.grad), f"{tag}.db": np32(blk.proj.bias.grad),
                        f"{tag}.dgamma": np32(blk.norm.weight.grad), f"{tag}.dbeta": np32(blk.norm.bias.grad)})
            if with_ss:
                out.update({f"{tag}.scale": np32(scale), f"{tag}.shift": np32(shift), f"{tag}.dscale": np32(scale.grad),
                            f"{tag}.dshift": np32(shift.grad)})
        # q_sample and the objective of p_losses (ddpm_loss.py:386-392, 404-438) with the model output as the leaf
        model_out = (torch.randn(3, 128, 80, generator=gg) * 0.8).requires_grad_()

        class Fixed(torch.nn.Module):
            channels, self_condition = 128, False

            def forward(self, x, t, c=None):
                return model_out
        diff = ref_ddpm.GaussianDiffusion1D(Fixed(), seq_length=80, sampling_timesteps=1000)
        x0 = torch.randn(3, 128, 80, generator=gg).clamp(-1, 1)
        noise = torch.randn(3, 128, 80, generator=gg)
        t = torch.tensor([3, 500, 999])
        loss, _, x_t = diff.p_losses(x0, t, cond=None, noise=noise)
        loss.backward()
        out.update({"q.x0": np32(x0), "q.noise": np32(noise), "q.t": t.numpy().astype(np.int64), "q.x_t": np32(x_t),
                    "q.model_out": np32(model_out), "q.loss": np32(loss.reshape(1)), "q.grad": np32(model_out.grad)})
        # channel LayerNorm (unet.py:82-101) under autograd
        from srcs.modules.unet import LayerNorm
        ln = LayerNorm(64)
        with torch.no_grad():
            ln.g.copy_(torch.rand(1, 64, 1, generator=gg) + 0.5)
        xl = (torch.randn(2, 64, 90, generator=gg) * 1.3 + 0.2).requires_grad_()
        yl = ln(xl)
        dyl = torch.randn(yl.shape, generator=gg)
        yl.backward(dyl)
        out.update({"ln.x": np32(xl), "ln.g": np32(ln.g), "ln.y": np32(yl), "ln.dy": np32(dyl), "ln.dx": np32(xl.grad), "ln.dg": np32(ln.g.grad)})
        # three steps of the reference's optimiser (train.py:365-371: optim.Adam(params, lr)) on a flat parameter
        pa = torch.nn.Parameter(torch.randn(5000, generator=gg) * 0.2)
        opt = torch.optim.Adam([pa], lr=3e-4)
        out["adam.p0"] = np32(pa)
        for k in range(3):
            gk = torch.randn(5000, generator=gg) * (0.5 if k != 1 else 1e-4)
            pa.grad = gk.clone()
            opt.step()
            out[f"adam.g{k}"] = np32(gk)
            out[f"adam.p{k + 1}"] = np32(pa)
        # a whole ResnetBlock with time embedding (unet.py:157-192) under autograd: with and without res_conv
        from srcs.modules.unet import ResnetBlock
        for tag, cin, cout, L in (("rb1", 32, 64, 96), ("rb2", 64, 64, 50)):
            rb = ResnetBlock(cin, cout, time_emb_dim=128, groups=8)
            with torch.no_grad():
                for blk in (rb.block1, rb.block2):
                    blk.norm.weight.copy_(torch.rand(cout, generator=gg) + 0.5)
                    blk.norm.bias.copy_(torch.randn(cout, generator=gg) * 0.1)
            x = torch.randn(2, cin, L, generator=gg, requires_grad=True)
            temb = torch.randn(2, 128, generator=gg, requires_grad=True)
            y = rb(x, temb)
            dy = torch.randn(y.shape, generator=gg)
            y.backward(dy)
            out.update({f"{tag}.x": np32(x), f"{tag}.temb": np32(temb), f"{tag}.y": np32(y), f"{tag}.dy": np32(dy), f"{tag}.dx": np32(x.grad),
                        f"{tag}.dtemb": np32(temb.grad)})
            for name, prm in rb.named_parameters():
                out[f"{tag}.p.{name}"] = np32(prm)
                out[f"{tag}.g.{name}"] = np32(prm.grad)
        # Residual(PreNorm(dim, LinearAttention(dim))) under autograd (unet.py:103-116,194-222)
        from srcs.modules.unet import LinearAttention, PreNorm, Residual
        att = Residual(PreNorm(64, LinearAttention(64)))
        with torch.no_grad():
            att.fn.norm.g.copy_(torch.rand(1, 64, 1, generator=gg) + 0.5)
            att.fn.fn.to_out[1].g.copy_(torch.rand(1, 64, 1, generator=gg) + 0.5)
        xa = (torch.randn(2, 64, 150, generator=gg) * 1.5).requires_grad_()
        ya = att(xa)
        dya = torch.randn(ya.shape, generator=gg)
        ya.backward(dya)
        out.update({"la.x": np32(xa), "la.y": np32(ya), "la.dy": np32(dya), "la.dx": np32(xa.grad)})
        for name, prm in att.fn.named_parameters():          # norm.g, fn.to_qkv.weight, fn.to_out.0.weight, fn.to_out.0.bias, fn.to_out.1.g
            key = name[3:] if name.startswith("fn.") else name
            out[f"la.p.{key}"] = np32(prm)
            out[f"la.g.{key}"] = np32(prm.grad)
        np.savez_compressed(os.path.join(OUT, "train_block.npz"), **out)
        print("train_block: loss", float(loss), "dw absmax", float(np.abs(out["a.dw"]).max()), "ln dg absmax", float(np.abs(out["ln.dg"]).max()))

    def train_unet_case():
        """The assembled Unet1D (unet.py:248-470) under autograd at a small width: two levels (dim 16, dim_mults (1, 2)), 8 + 8 input
        channels (`other_cond`: the condition is concatenated in front of x; upsampling_ratios None and unet_scale_cond False, so
        process_cond is the identity), forward output and the gradient of EVERY parameter and of both inputs."""
        from srcs.modules.unet import Unet1D
        torch.manual_seed(5150)
        gg = torch.Generator().manual_seed(5151)
        net = Unet1D(16, dim_mults=(1, 2), inp_channels=8, other_cond=True, cond_channels=8, upsampling_ratios=None, unet_scale_cond=False)
        with torch.no_grad():
            for name, prm in net.named_parameters():
                if name.endswith(".g") or name.endswith("norm.weight"):
                    prm.copy_(torch.rand(prm.shape, generator=gg) + 0.5)
        x = torch.randn(2, 8, 32, generator=gg, requires_grad=True)
        xc = torch.randn(2, 8, 32, generator=gg, requires_grad=True)
        time = torch.tensor([17, 803])
        y = net(x, time, xc)
        dy = torch.randn(y.shape, generator=gg)
        y.backward(dy)
        out = {"x": np32(x), "xc": np32(xc), "time": time.numpy().astype(np.int64), "y": np32(y), "dy": np32(dy), "dx": np32(x.grad), "dxc": np32(xc.grad)}
        for name, prm in net.named_parameters():
            out["p." + name] = np32(prm)
            out["g." + name] = np32(prm.grad) if prm.grad is not None else np.zeros(prm.shape, np.float32)
        # the same net with process_cond in the path: upsampling_ratios [5, 2] (two SConvTranspose1d) and unet_scale_cond
        torch.manual_seed(5252)
        net2 = Unet1D(16, dim_mults=(1, 2), inp_channels=8, other_cond=True, cond_channels=8, upsampling_ratios=[5, 2], unet_scale_cond=True)
        x2 = torch.randn(2, 8, 40, generator=gg, requires_grad=True)
        c2 = torch.randn(2, 8, 4, generator=gg, requires_grad=True)
        y2 = net2(x2, torch.tensor([250, 3]), c2)
        dy2 = torch.randn(y2.shape, generator=gg)
        y2.backward(dy2)
        out.update({"u.x": np32(x2), "u.cond": np32(c2), "u.time": np.array([250, 3], np.int64), "u.y": np32(y2), "u.dy": np32(dy2),
                    "u.dx": np32(x2.grad), "u.dcond": np32(c2.grad)})
        for name, prm in net2.named_parameters():
            out["u.p." + name] = np32(prm)
            out["u.g." + name] = np32(prm.grad)
        np.savez_compressed(os.path.join(OUT, "train_unet.npz"), **out)
        print("train_unet:", len([k for k in out if k.startswith("p.")]), "parameters,", sum(v.size for k, v in out.items() if k.startswith("p.")), "elements")

    if os.environ.get("GOLDEN_ONLY") == "train_unet":
        train_unet_case()
        return
    train_case()
    train_unet_case()
    if os.environ.get("GOLDEN_ONLY") == "train":
        return
    variants_case()
    mc84 = CodecConfig(enc_ratios=(8, 4), quantization=False)
    u84 = UnetConfig(dim=32, upsampling_ratios=(5, 2), unet_scale_cond=True)
    if os.environ.get("GOLDEN_ONLY") == "variants":
        return
    if os.environ.get("GOLDEN_ONLY") == "drivers":
        drivers_case("r84", mc84, u84, T=2560, seed_w=21, seed_in=777, midway_t=5)
        return
    drivers_case("r84", mc84, u84, T=2560, seed_w=21, seed_in=777, midway_t=5)
    ladiff_case("r84", mc84, u84, T=5120, n_chain=4, seed_w=21, seed_in=4321)
    mc8 = CodecConfig(enc_ratios=(8,), quantization=False)
    u8 = UnetConfig(dim=32, upsampling_ratios=(5, 4, 2), unet_scale_cond=False)
    ladiff_case("r8", mc8, u8, T=2560, n_chain=3, seed_w=22, seed_in=999)


if __name__ == "__main__":
    main()
