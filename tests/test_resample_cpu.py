"""The front-end resampler's oracle (oracle/resample_oracle.py, torchaudio 0.13's published sinc_interp_hann algorithm; parity
UNPINNED -- torchaudio is absent here) against properties and a direct float64 evaluation of the interpolation formula."""
import math

import numpy as np

from oracle import resample_oracle as RO


def test_kernel_bank_shape_and_dc_gain():
    k, width, orig, new = RO.sinc_resample_kernel(44100, 16000)
    assert (orig, new) == (441, 160) and width == math.ceil(6 * 441 / (160 * 0.99)) and k.shape == (160, 2 * width + 441)
    assert np.allclose(k.sum(axis=1), 1.0, atol=2e-3)          # every phase passes DC with unit gain
    k2, w2, o2, n2 = RO.sinc_resample_kernel(8000, 16000)
    assert (o2, n2, w2) == (1, 2, math.ceil(6 / 0.99)) and abs(k2[0, w2] - 0.99) < 1e-6    # phase 0 is centred on the sample


def test_identity_length_and_tone():
    x = np.random.default_rng(0).standard_normal((2, 1000)).astype(np.float32)
    assert RO.resample(x, 16000, 16000) is not None and np.array_equal(RO.resample(x, 16000, 16000), x)
    for sr in (8000, 22050, 44100, 48000):
        n = 4000
        t = np.arange(n) / sr
        tone = np.sin(2 * np.pi * 440.0 * t).astype(np.float32)[None]
        y = RO.resample(tone, sr, 16000)
        assert y.shape[1] == math.ceil(16000 // math.gcd(sr, 16000) * n / (sr // math.gcd(sr, 16000)))
        want = np.sin(2 * np.pi * 440.0 * np.arange(y.shape[1]) / 16000.0)
        mid = slice(200, y.shape[1] - 200)                      # away from the zero-padded edges
        assert np.abs(y[0, mid] - want[mid]).max() < 2e-2, sr


def test_direct_evaluation_of_the_interpolation_formula():
    """y[j] = sum_s x[s] * h(j / new - s / orig) with h the Hann-windowed sinc, in float64."""
    rng = np.random.default_rng(1)
    x = rng.standard_normal(300).astype(np.float32)
    orig, new, lpw, roll = 3, 2, 6, 0.99
    y = RO.resample(x[None], 48000, 32000)[0]
    base = min(orig, new) * roll
    for j in (0, 5, 77, len(y) - 1):
        acc = 0.0
        for s in range(len(x)):
            t = (s / orig - j / new) * base
            if abs(t) >= lpw:
                continue
            w = math.cos(t * math.pi / lpw / 2) ** 2
            acc += x[s] * (1.0 if t == 0 else math.sin(t * math.pi) / (t * math.pi)) * w * base / orig
        assert abs(acc - y[j]) < 1e-4, j
