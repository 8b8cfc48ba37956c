"""Probe: two engines (two contexts) decoding alternate batches on two torch streams vs one engine, same GPU.
Usage: python tools/pipeline_probe.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ladiffcodec_amd import lib as L, synth, spec
from ladiffcodec_amd.spec import CodecConfig, UnetConfig
from ladiffcodec_amd.model import Engine

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0)
mc = CodecConfig(enc_ratios=(8, 4), quantization=False)
u = UnetConfig(dim=256, upsampling_ratios=(5, 2), unet_scale_cond=True)
full = synth.ladiff_state_dict(mc, u, seed=1)
sd_main = {k: v for k, v in full.items() if not k.startswith("diffusion.model.")}
sd_cond = synth.codec_state_dict(cc, seed=0)
dev = torch.device("cuda", 0)
B, T, N = 32, 38400, 50
wav = torch.from_numpy(synth.synthetic_wav(B, T, seed=1234)).to(dev)


def make():
    e = Engine(mc, u, cc, dtype="bf16", device=0, noise_seed=4321)
    e.load_state_dict(L.MODEL_MAIN, sd_main); e.load_state_dict(L.MODEL_COND, sd_cond); e.finalize(strict=True)
    return e


engs = [make(), make()]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
for e, s in zip(engs, streams):
    with torch.cuda.stream(s):
        e.decode(wav, N, per_item=True)
torch.cuda.synchronize()
for mode in ("single", "pipelined", "single", "pipelined"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K):
        k = i % 2 if mode == "pipelined" else 0
        with torch.cuda.stream(streams[k]):
            out = engs[k].decode(wav, N, per_item=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{mode}: {1000 * dt / K:.2f} ms per batch, {B * T / 16000 * K / dt:.1f} audio-s/s")
