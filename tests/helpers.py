"""Shared builders for the parity tests: the synthetic checkpoints the golden fixtures were made on."""
import os

import numpy as np
import torch

from ladiffcodec_amd import synth
from ladiffcodec_amd.spec import CodecConfig, UnetConfig

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

COND_CFG = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0)
COND_SEED = 11

CASES = {
    # tag: (main codec cfg, unet cfg, weight seed)   -- must match tools/gen_golden.py
    "r84": (CodecConfig(enc_ratios=(8, 4), quantization=False),
            UnetConfig(dim=32, upsampling_ratios=(5, 2), unet_scale_cond=True), 21),
    "r8": (CodecConfig(enc_ratios=(8,), quantization=False),
           UnetConfig(dim=32, upsampling_ratios=(5, 4, 2), unet_scale_cond=False), 22),
}


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def cond_sd_np():
    return synth.codec_state_dict(COND_CFG, COND_SEED)


def main_sd_np(tag):
    mc, u, seed = CASES[tag]
    return synth.ladiff_state_dict(mc, u, seed)


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def driver_noises(g):
    """The noise tapes of tests/golden/drivers_*.npz, regenerated from the recorded seed exactly as tools/gen_golden.py
    drew them (the tapes themselves would be 40 MB): (loop draws [T,1,128,L], infilling draws [2*midway_t,1,128,L])."""
    _, _, seed_in, midway_t, n_t = (int(v) for v in g["meta"])
    L = g["loop_img0"].shape[2]
    gg = torch.Generator().manual_seed(seed_in + 1)
    loop = torch.stack([torch.randn(1, 128, L, generator=gg) for _ in range(n_t)])
    g2 = torch.Generator().manual_seed(seed_in + 3)
    fill = torch.stack([torch.randn(1, 128, L, generator=g2) for _ in range(2 * midway_t)])
    return loop, fill, midway_t


def sub_stride(size, cap):
    return max(1, -(-int(size) // int(cap)))


def sub(a, stride):
    """Strided sample of a large tensor along its last axis: the full-width fixtures of tests/golden/bench256.npz keep a
    sample of every large tensor (its stride is stored beside it) plus float64 checksums of the whole; the same function
    samples the GPU result."""
    return np.ascontiguousarray(np.asarray(a)[..., ::int(stride)])


def checksums(a):
    a = np.asarray(a, np.float64)
    return np.array([a.sum(), np.abs(a).sum()], np.float64)


class BenchGolden:
    """tests/golden/bench256.npz (tools/gen_golden_bench.py): expectations of the dim-256 GPU tests, from the reference."""

    def __init__(self):
        self.g = load_golden("bench256")

    def compare(self, key, got):
        """max |got - ref| / max |ref| on the stored sample; also checks the shape"""
        got = np.asarray(got)
        assert tuple(got.shape) == tuple(int(v) for v in self.g[key + ".shape"]), (key, got.shape, self.g[key + ".shape"])
        ref = self.g[key]
        return rel_err(sub(got, self.g[key + ".stride"]), ref)

    def checksum_err(self, key, got):
        """relative error of (sum, sum |.|) over the WHOLE tensor: catches damage outside the strided sample"""
        c = checksums(got)
        r = self.g[key + ".sum"]
        return float(np.abs(c - r).max() / (np.abs(r[1]) + 1e-30))


# ---- inputs of the two 50-step CLI tests (tests/test_gpu_frontend.py) and of their fixture generator (tools/gen_golden_cli.py) ----
def cli50_default_audio(k, n=5120):
    from ladiffcodec_amd import synth
    return (synth.synthetic_wav(1, n, seed=300 + k)[0, 0] * 0.5).astype(np.float32)


def cli50_default_tape(i, Lz, steps=50):
    return torch.randn(steps, 1, 128, Lz, generator=torch.Generator().manual_seed(9000 + i))


def cli50_c5_audio(chunk=38400):
    from ladiffcodec_amd import synth
    n = 12 * chunk + chunk // 2 + 100                              # 30 s (and a few samples more)
    return (synth.synthetic_wav(1, n, seed=515)[0, 0] * 0.5).astype(np.float32)


def cli50_c5_plan(n, chunk=38400):
    tail = (n - 12 * chunk) // 2560 * 2560                          # 1.12 s: whole 2560-sample quanta (cond frames x UNet halvings)
    return [(k * chunk, chunk) for k in range(12)] + [(12 * chunk, tail)], chunk


def cli50_c5_tape(k, Lz, steps=50):
    return torch.randn(steps, 1, 128, Lz, generator=torch.Generator().manual_seed(7000 + k))


def fake_quantise_unet(sd_np, u):
    """What the library does to every UNet conv weight in an fp8-weight context: weight-standardise the Block convs
    (unet.py:73-78), then per output channel scale = max|w| / 448 and OCP e4m3 round-to-nearest-even; returns the state
    dict with the dequantised values (Block convs stay standardised: the oracle runs with WS_PREFOLDED)."""
    from oracle import ldc_oracle as O
    out = {}
    for k, v in sd_np.items():
        t = torch.from_numpy(np.ascontiguousarray(v))
        is_conv = k.startswith("diff_model.") and t.dim() == 3 and "upsampling_layers" not in k and not k.endswith(".g")
        if is_conv:
            if ".block1.proj.weight" in k or ".block2.proj.weight" in k:
                t = O.ws_fold(t)
            amax = t.abs().amax(dim=(1, 2), keepdim=True)
            sc = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
            t = (t / sc).to(torch.float8_e4m3fn).float() * sc
        out[k] = t
    return out


def libri_tree(root):
    """A LibriSpeech-shaped tree of int16 wavs (tests/test_dataset_cpu.py, tools/gen_golden_dataset.py): train-clean-100 with files
    of different lengths, one silent, one shorter than a 0.5 s crop, one with a -32768 sample and a long silent stretch; dev-clean
    with two files (one shorter than the crop)."""
    import scipy.io.wavfile as wavfile
    rng = np.random.default_rng(77)

    def put(split, spk, chap, k, x):
        d = os.path.join(root, split, str(spk), str(chap))
        os.makedirs(d, exist_ok=True)
        wavfile.write(os.path.join(d, f"{spk}-{chap}-{k:04d}.wav"), 16000, x.astype(np.int16))

    def tone(n, f, amp):
        t = np.arange(n) / 16000.0
        return amp * np.sin(2 * np.pi * f * t) + 0.05 * amp * rng.standard_normal(n)

    put("train-clean-100", 103, 1240, 0, tone(20000, 220.0, 9000))
    put("train-clean-100", 103, 1240, 1, np.zeros(12000))                              # silent: skipped
    put("train-clean-100", 103, 1241, 0, tone(5000, 330.0, 12000))                     # shorter than the crop: skipped
    x = tone(30000, 150.0, 20000)
    x[4000:26000] = 0                                                                   # silent stretch: crops there are redrawn
    x[100] = -32768                                                                     # int16 abs wraps
    put("train-clean-100", 1034, 121119, 0, x)
    put("train-clean-100", 1034, 121119, 1, tone(8000, 440.0, 3000))                   # exactly the crop length
    put("dev-clean", 84, 121123, 0, tone(16000, 200.0, 15000))
    put("dev-clean", 84, 121123, 1, tone(3000, 500.0, 7000))                           # shorter than the crop
