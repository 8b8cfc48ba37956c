"""CPU restatement of `torchaudio.functional.resample` as the reference calls it (srcs/sample.py:84:
`torchaudio.functional.resample(wav, orig_freq=sr, new_freq=16000)`, defaults lowpass_filter_width=6, rolloff=0.99,
resampling_method="sinc_interpolation" / "sinc_interp_hann") -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: torchaudio (pinned by the reference at 0.13.1, requirements.txt:96) is a third-party dependency that is
not vendored in /root/reference and not installed in this image (`pip download torchaudio==0.13.1`: no index, no wheel in the
offline wheelhouse -- DESIGN.md section 2), so no golden vector could be generated from it.  The algorithm below is torchaudio
0.13's published `_get_sinc_resample_kernel` / `_apply_sinc_resample_kernel` (torchaudio/functional/functional.py): a bank of
`new` windowed-sinc filters (Hann window, cut-off rolloff * min(orig, new) / 2 after reducing both rates by their gcd) applied
as a strided convolution.  `functional.resample` passes the WAVEFORM's dtype to the kernel builder: for the fp32 tensors
`torchaudio.load` returns the bank is built with fp32 torch tensor operations (float64 is what `transforms.Resample` does with
dtype=None) -- restated here with the same torch operations in the same order (round 5; rounds 3-4 built it in float64).  It is
anchored on the reference's call site and on properties (identity, tone preservation, direct evaluation of the interpolation
formula).
"""
import math

import numpy as np
import torch


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99, dtype=torch.float32):
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=dtype)[None, :] / orig
    t = torch.arange(0, -new, -1, dtype=dtype)[:, None] / new + idx
    t *= base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0).to(t), t.sin() / t)
    kernels *= window * scale
    return kernels.to(torch.float32).numpy(), width, orig, new          # [new, 2*width + orig]


def resample(wav: np.ndarray, orig_freq: int, new_freq: int) -> np.ndarray:
    """wav [..., T] float32 -> [..., ceil(new * T / orig)]."""
    wav = np.asarray(wav, np.float32)
    if orig_freq == new_freq:
        return wav
    k, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
    T = wav.shape[-1]
    flat = torch.from_numpy(np.ascontiguousarray(wav.reshape(-1, T)))
    padded = torch.nn.functional.pad(flat, (width, width + orig))
    res = torch.nn.functional.conv1d(padded[:, None], torch.from_numpy(k)[:, None, :], stride=orig)   # [C, new, frames]
    res = res.transpose(1, 2).reshape(flat.shape[0], -1)
    target = int(math.ceil(new * T / orig))
    return res[:, :target].numpy().reshape(wav.shape[:-1] + (target,))
