"""Decode time of the bench batch with the batch split into 2 / 3 / 4 chains, queued back to back vs drained per decode.
usage: LDC_SPLIT=n python tools/split_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ladiffcodec_amd import lib as L, synth
from ladiffcodec_amd.spec import CodecConfig, UnetConfig
from ladiffcodec_amd.model import Engine
cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0)
mc = CodecConfig(enc_ratios=(8, 4), quantization=False)
u = UnetConfig(dim=256, upsampling_ratios=(5, 2), unet_scale_cond=True)
full = synth.ladiff_state_dict(mc, u, seed=1)
e = Engine(mc, u, cc, dtype="bf16", device=0)
e.load_state_dict(L.MODEL_MAIN, {k: v for k, v in full.items() if not k.startswith("diffusion.model.")})
e.load_state_dict(L.MODEL_COND, synth.codec_state_dict(cc, seed=0)); e.finalize(strict=True)
B, T, N = int(os.environ.get("B", "32")), 38400, 50
wav = torch.from_numpy(synth.synthetic_wav(B, T, seed=1234)).cuda()
e.decode(wav, N, per_item=True); torch.cuda.synchronize()
for mode in ("queued", "drained", "queued"):
    t0 = time.perf_counter()
    for _ in range(4):
        th = time.perf_counter()
        e.decode(wav, N, per_item=True)
        host_ms = 1000 * (time.perf_counter() - th)
        if mode == "drained":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"split={os.environ.get('LDC_SPLIT', '2')} {mode}: {1000 * (time.perf_counter() - t0) / 4:.1f} ms per decode (host time of the last call {host_ms:.1f} ms)")
