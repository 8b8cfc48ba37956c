"""Generate tests/golden/bench256.npz: the dim-256 expectations of tests/test_gpu_bench_shape.py, produced by running the
REFERENCE (/root/reference, imported read-only through tools/ref_import.py) on the seeded synthetic checkpoints and inputs the
tests regenerate.  Until round 3 the GPU tests recomputed these with the CPU oracle on the GPU box (most of a 1 000 s pytest
step); now they are data, and the tests are reference-pinned instead of oracle-pinned.

Run in the build container only (about ten minutes on 8 cores):   python tools/gen_golden_bench.py

Large tensors are stored as strided samples plus float64 checksums of the whole tensor (`put()` below; tests/helpers.py
`BenchGolden.compare()` applies the same sampling to the GPU result).
Reference entry points exercised: srcs/sample.py:125-134 (start image, halfway_sampling, decoder, normalise),
srcs/losses/ddpm_loss.py:370-385, srcs/modules/unet.py:422-469, srcs/model.py:223-231 (get_cond).
"""
import os
import sys
import time
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
from helpers import CASES, load_golden, sub, sub_stride  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "bench256.npz")
torch.set_num_threads(8)


def put(out, key, t, cap=80_000):
    """strided sample (at most ~cap elements) + checksums of the whole tensor"""
    a = np32(t)
    stride = sub_stride(a.size, cap)
    out[key] = sub(a, stride)
    out[key + ".stride"] = np.array(stride, np.int64)
    out[key + ".sum"] = np.array([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()], np.float64)
    out[key + ".shape"] = np.array(a.shape, np.int64)


def decode_like_sample_py(ref_ddpm, cond_model, main, wav, n_steps, noises):
    """srcs/sample.py:125-134 for one file (B = number of channels = 1): returns codes, latents, normalised wav"""
    with torch.no_grad():
        z = cond_model.encoder(wav)
        q = cond_model.quantizer(z, sample_rate=cond_model.frame_rate, bandwidth=cond_model.bandwidth)
        cond = cond_model.get_cond(wav)
        img = cond
        for layer in main.diff_model.upsampling_layers:
            img = layer(img)
        img = img / (torch.max(torch.abs(img.flatten())) + 1e-8)
        tape = NoiseTape(list(noises))
        ref_ddpm.torch.randn_like, saved = tape, ref_ddpm.torch.randn_like
        try:
            lat = main.diffusion.halfway_sampling(img=img, condition=cond, t=n_steps)
        finally:
            ref_ddpm.torch.randn_like = saved
        assert tape.i == n_steps - 1
        dec = main.decoder(lat)
        y = dec / (torch.std(dec.flatten()) + 1e-8)
        y = y / (torch.max(torch.abs(y.flatten())) + 1e-8)
    return q.codes, lat, y


def main():
    ref = import_reference()
    import srcs.losses.ddpm_loss as ref_ddpm
    out = {}
    t00 = time.time()

    mc = CodecConfig(enc_ratios=(8, 4), quantization=False)
    u = UnetConfig(dim=256, upsampling_ratios=(5, 2), unet_scale_cond=True)
    main_c2 = build_main_model(ref, mc, u, seed=1)

    # ---- eps + interior taps at the bench grid (test_bench_grid_unet_eps_and_taps): items 3 / 29 of B = 32, t = 37 / 499
    B, Lz, F = 32, 1200, 120
    g = torch.Generator().manual_seed(41)
    x = torch.randn(B, 128, Lz, generator=g) * 0.7
    cond = torch.randn(B, 128, F, generator=g)
    dm = main_c2.diff_model
    taps = {}
    hooks = [dm.downs[4][3].register_forward_hook(lambda m, i, o: taps.__setitem__("down4", o)),
             dm.mid_block2.register_forward_hook(lambda m, i, o: taps.__setitem__("mid", o)),
             dm.ups[0][3].register_forward_hook(lambda m, i, o: taps.__setitem__("up0", o)),
             dm.ups[4][3].register_forward_hook(lambda m, i, o: taps.__setitem__("up4", o))]
    with torch.no_grad():
        for i in (3, 29):
            for t in (37, 499):
                eps = dm(x[i:i + 1], torch.full((1,), t, dtype=torch.long), cond[i:i + 1])
                put(out, f"eps.{i}.{t}", eps)
                for n in ("down4", "mid", "up0", "up4"):
                    put(out, f"tap.{n}.{i}.{t}", taps[n], cap=40_000)
    for h in hooks:
        h.remove()
    print(f"eps/taps done {time.time() - t00:.0f} s", flush=True)

    # ---- the timed workload (test_bench_decode_50_steps_against_oracle): items 5 / 22 of the bench batch, all 50 steps
    cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0)
    cond_model = build_cond_model(ref, cc, seed=11)
    wav = torch.from_numpy(synth.synthetic_wav(32, 38400, seed=1234))
    for i in (5, 22):
        noises = torch.randn(50, 1, 128, Lz, generator=torch.Generator().manual_seed(100 + i))
        codes, lat, y = decode_like_sample_py(ref_ddpm, cond_model, main_c2, wav[i:i + 1], 50, noises)
        out[f"dec50.{i}.codes"] = codes.numpy().astype(np.int64)
        put(out, f"dec50.{i}.latents", lat)
        put(out, f"dec50.{i}.wav", y)
        print(f"dec50 item {i} done {time.time() - t00:.0f} s", flush=True)

    # ---- configs[2] (test_c3_1p5kbps_200_steps): 1.5 kbps condition, 200 steps, item 2 of a 4-utterance batch
    cc15 = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=1.5)
    cond15 = build_cond_model(ref, cc15, seed=11)
    wav3 = torch.from_numpy(synth.synthetic_wav(4, 38400, seed=77))
    mine = torch.randn(200, 1, 128, Lz, generator=torch.Generator().manual_seed(9))
    codes, lat, y = decode_like_sample_py(ref_ddpm, cond15, main_c2, wav3[2:3], 200, mine)
    out["c3.codes"] = codes.numpy().astype(np.int64)
    put(out, "c3.latents", lat)
    put(out, "c3.wav", y)
    print(f"c3 done {time.time() - t00:.0f} s", flush=True)
    del main_c2

    # ---- the released checkpoints' layout (test_c8_full_width_unet_and_decoder)
    mc8 = CodecConfig(enc_ratios=(8,), quantization=False)
    u8 = UnetConfig(dim=256, upsampling_ratios=(5, 4, 2), unet_scale_cond=True)
    main_c8 = build_main_model(ref, mc8, u8, seed=1)
    g = torch.Generator().manual_seed(43)
    x8 = torch.randn(2, 128, 4800, generator=g) * 0.7
    c8 = torch.randn(2, 128, 120, generator=g)
    with torch.no_grad():
        put(out, "c8.eps", main_c8.diff_model(x8, torch.full((2,), 211, dtype=torch.long), c8))
        lat8 = torch.tanh(torch.randn(2, 128, 4800, generator=g))
        put(out, "c8.dec", main_c8.decoder(lat8))
    del main_c8
    print(f"c8 done {time.time() - t00:.0f} s", flush=True)

    # ---- 250-step chains from t = 249 on the small fixtures (test_chain_from_high_t_and_dead_unet_guard)
    for tag in ("r84", "r8"):
        gd = load_golden("ladiff_" + tag)
        mcs, us, seed_w = CASES[tag]
        m = build_main_model(ref, mcs, us, seed=seed_w)
        noises = torch.randn(250, *gd["x"].shape, generator=torch.Generator().manual_seed(77))
        tape = NoiseTape(list(noises))
        ref_ddpm.torch.randn_like, saved = tape, ref_ddpm.torch.randn_like
        try:
            with torch.no_grad():
                lat = m.diffusion.halfway_sampling(img=torch.from_numpy(gd["img0"]).clone(), condition=torch.from_numpy(gd["cond"]), t=250)
        finally:
            ref_ddpm.torch.randn_like = saved
        out[f"chain250.{tag}"] = np32(lat)
        print(f"chain250 {tag} done {time.time() - t00:.0f} s", flush=True)

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) / 1e6, "MB")


if __name__ == "__main__":
    main()
