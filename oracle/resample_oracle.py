"""CPU restatement of `torchaudio.functional.resample` as the reference calls it (srcs/sample.py:84:
`torchaudio.functional.resample(wav, orig_freq=sr, new_freq=16000)`, defaults lowpass_filter_width=6, rolloff=0.99,
resampling_method="sinc_interp_hann") -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: torchaudio (pinned by the reference at 0.13.1, requirements.txt:96) is a third-party dependency that is
not vendored in /root/reference and not installed in this image, so no golden vector could be generated from it.  The
algorithm below is torchaudio 0.13's published `_get_sinc_resample_kernel` / `_apply_sinc_resample_kernel`
(torchaudio/functional/functional.py): a bank of `new` windowed-sinc filters (Hann window, kernel built in float64, cut-off
rolloff * min(orig, new) / 2 after reducing both rates by their gcd) applied as a strided convolution; it is anchored on the
reference's call site and on properties (identity, tone preservation, direct evaluation of the interpolation formula).
"""
import math

import numpy as np


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t *= base_freq
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base_freq / orig
    with np.errstate(invalid="ignore", divide="ignore"):
        kernels = np.where(t == 0, 1.0, np.sin(t) / t)
    kernels *= window * scale
    return kernels.astype(np.float32), width, orig, new          # [new, 2*width + orig]


def resample(wav: np.ndarray, orig_freq: int, new_freq: int) -> np.ndarray:
    """wav [..., T] float32 -> [..., ceil(new * T / orig)]."""
    wav = np.asarray(wav, np.float32)
    if orig_freq == new_freq:
        return wav
    k, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
    T = wav.shape[-1]
    flat = wav.reshape(-1, T)
    padded = np.pad(flat, ((0, 0), (width, width + orig)))
    frames = (padded.shape[1] - k.shape[1]) // orig + 1
    out = np.zeros((flat.shape[0], frames, new), np.float32)
    for i in range(frames):
        seg = padded[:, i * orig: i * orig + k.shape[1]]            # [C, K]
        out[:, i, :] = (seg.astype(np.float32) @ k.T.astype(np.float32))
    target = int(math.ceil(new * T / orig))
    return out.reshape(flat.shape[0], -1)[:, :target].reshape(wav.shape[:-1] + (target,))
