"""Seeded synthetic checkpoints with exactly the reference key set.

The released checkpoints are external downloads (reference README.md:29) and there is no network,
so parity tests, the bench and the golden-vector generator all run on synthetic weights.  Values
come from numpy's PCG64 (`default_rng(seed)`), drawn key by key in the order `spec.py` lists them,
so the same (config, seed) gives bit-identical weights in the build container (where the reference
is run on them) and on the GPU box (where only this package exists).

Scales follow PyTorch's default initialisers so activations stay O(1) through the deep stacks;
weight-norm gains and norm affines are perturbed away from their identity initial values so the
folding code paths are exercised.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np

from . import spec
from .spec import CodecConfig, UnetConfig


def cosine_schedule_buffers(timesteps: int = 1000, s: float = 0.008) -> "OrderedDict[str, np.ndarray]":
    """The 13 registered buffers of GaussianDiffusion1D, float64 math then fp32 cast.

    Restates reference srcs/losses/ddpm_loss.py:50-60 (cosine betas) and :116-168 (derived tables).
    """
    steps = timesteps + 1
    t = np.linspace(0, timesteps, steps, dtype=np.float64) / timesteps
    ac = np.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = np.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)
    alphas = 1.0 - betas
    alphas_cumprod = np.cumprod(alphas)
    alphas_cumprod_prev = np.concatenate([[1.0], alphas_cumprod[:-1]])
    posterior_variance = betas * (1.0 - alphas_cumprod_prev) / (1.0 - alphas_cumprod)
    out = OrderedDict()
    out["betas"] = betas
    out["alphas_cumprod"] = alphas_cumprod
    out["alphas_cumprod_prev"] = alphas_cumprod_prev
    out["sqrt_alphas_cumprod"] = np.sqrt(alphas_cumprod)
    out["sqrt_one_minus_alphas_cumprod"] = np.sqrt(1.0 - alphas_cumprod)
    out["log_one_minus_alphas_cumprod"] = np.log(1.0 - alphas_cumprod)
    out["sqrt_recip_alphas_cumprod"] = np.sqrt(1.0 / alphas_cumprod)
    out["sqrt_recipm1_alphas_cumprod"] = np.sqrt(1.0 / alphas_cumprod - 1)
    out["posterior_variance"] = posterior_variance
    out["posterior_log_variance_clipped"] = np.log(np.clip(posterior_variance, 1e-20, None))
    out["posterior_mean_coef1"] = betas * np.sqrt(alphas_cumprod_prev) / (1.0 - alphas_cumprod)
    out["posterior_mean_coef2"] = (1.0 - alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - alphas_cumprod)
    out["p2_loss_weight"] = (1 + alphas_cumprod / (1 - alphas_cumprod)) ** -0.0
    return OrderedDict((k, v.astype(np.float32)) for k, v in out.items())


def _draw(rng: np.random.Generator, key: str, shape: Tuple[int, ...], v_norms: Dict[str, np.ndarray]) -> np.ndarray:
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "inited":
        return np.ones(shape, np.float32)            # quirk Q9: an un-inited codebook k-means-es on first forward
    if leaf == "cluster_size":
        return np.ones(shape, np.float32)
    if leaf in ("embed", "embed_avg"):
        return rng.standard_normal(shape).astype(np.float32)
    if leaf == "g":                                    # channel LayerNorm gain (unet.py:85)
        return (1.0 + 0.2 * rng.standard_normal(shape)).astype(np.float32)
    if ".norm." in key and leaf == "weight":          # GroupNorm gamma
        return (1.0 + 0.2 * rng.standard_normal(shape)).astype(np.float32)
    if ".norm." in key and leaf == "bias":            # GroupNorm beta
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if leaf == "weight_g":
        # weight_norm initialises g = ||v||; scale it so the fold is not a no-op.  The matching
        # weight_v is drawn right after (bias, weight_g, weight_v order) and patches this value.
        return rng.uniform(0.6, 1.4, size=shape).astype(np.float32)
    if leaf.startswith("bias"):
        return rng.uniform(-0.05, 0.05, size=shape).astype(np.float32)
    # conv / linear / lstm weights: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    if "lstm" in key:
        fan_in = shape[-1]
    elif len(shape) == 3 and ".convtr." in key:
        fan_in = shape[0] * shape[2] // 2            # each output sample sees ~k/stride = 2 taps
    elif len(shape) == 3:
        fan_in = shape[1] * shape[2]
    else:
        fan_in = shape[-1]
    b = 1.0 / math.sqrt(max(fan_in, 1))
    return rng.uniform(-b, b, size=shape).astype(np.float32)


def _generate(keys: List[Tuple[str, Tuple[int, ...]]], seed: int) -> "OrderedDict[str, np.ndarray]":
    rng = np.random.default_rng(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape in keys:
        sd[key] = _draw(rng, key, shape, {})
    # weight_g = (drawn gain) * ||weight_v|| per dim-0 slice, as weight_norm(dim=0) defines it
    for key in list(sd):
        if key.endswith("weight_g"):
            v = sd[key[:-1] + "v"].astype(np.float64)
            nrm = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(1)).reshape(-1, 1, 1)
            sd[key] = (sd[key].astype(np.float64) * nrm).astype(np.float32)
    return sd


def codec_state_dict(c: CodecConfig, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Synthetic EnCodec-style checkpoint (the `--model_for_cond` file)."""
    return _generate(spec.codec_keys(c), seed)


def ladiff_state_dict(c: CodecConfig, u: UnetConfig, seed: int = 1) -> "OrderedDict[str, np.ndarray]":
    """Synthetic LaDiffCodec checkpoint (the `--model_path` file): autoencoder + UNet (stored under
    both `diff_model.*` and `diffusion.model.*`, as the reference's state_dict does) + schedule."""
    keys = spec.codec_keys(c) + spec.unet_keys(u, "diff_model")
    sd = _generate(keys, seed)
    for name, val in cosine_schedule_buffers(u.timesteps).items():
        sd[f"diffusion.{name}"] = val
    for key, _ in spec.unet_keys(u, "diff_model"):
        sd["diffusion.model." + key[len("diff_model."):]] = sd[key]
    # reorder to the reference's registration order
    order = [k for k, _ in spec.ladiff_keys(c, u)]
    return OrderedDict((k, sd[k]) for k in order)


def to_torch(sd: "OrderedDict[str, np.ndarray]"):
    import torch
    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in sd.items())


def save_amlt(sd: "OrderedDict[str, np.ndarray]", path: str, ddp_prefix: bool = False) -> None:
    """Write `torch.save(state_dict)` exactly as reference srcs/utils.py:91 does (optionally with the
    `module.` prefix a DDP-wrapped model would have produced)."""
    import torch
    t = to_torch(sd)
    if ddp_prefix:
        t = OrderedDict(("module." + k, v) for k, v in t.items())
    torch.save(t, path)


def synthetic_wav(batch: int, length: int, seed: int = 1234) -> np.ndarray:
    """LibriSpeech-shaped synthetic audio: 16 kHz mono, sum of a few harmonics with a slow envelope
    plus low-level noise, peak-normalised like reference srcs/dataset_libri.py:48-52."""
    rng = np.random.default_rng(seed)
    t = np.arange(length, dtype=np.float64) / 16000.0
    out = np.zeros((batch, 1, length), np.float64)
    for b in range(batch):
        f0 = rng.uniform(90, 240)
        sig = np.zeros(length)
        for h in range(1, 9):
            sig += rng.uniform(0.1, 1.0) / h * np.sin(2 * math.pi * f0 * h * t + rng.uniform(0, 2 * math.pi))
        env = 0.55 + 0.45 * np.sin(2 * math.pi * rng.uniform(1.0, 4.0) * t + rng.uniform(0, 2 * math.pi))
        sig = sig * env + 0.02 * rng.standard_normal(length)
        out[b, 0] = sig / (np.abs(sig).max() + 1e-8)
    return out.astype(np.float32)
