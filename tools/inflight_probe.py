"""Two (or more) batches of the bench shape in flight: one engine (context) per slot, each on its own stream, batches dealt
round-robin.  usage: [LDC_NO_SPLIT=1] python tools/inflight_probe.py [slots] [batches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ladiffcodec_amd import lib as L, synth
from ladiffcodec_amd.spec import CodecConfig, UnetConfig
from ladiffcodec_amd.model import Engine
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0)
mc = CodecConfig(enc_ratios=(8, 4), quantization=False)
u = UnetConfig(dim=256, upsampling_ratios=(5, 2), unet_scale_cond=True)
full = synth.ladiff_state_dict(mc, u, seed=1)
main = {k: v for k, v in full.items() if not k.startswith("diffusion.model.")}
cond = synth.codec_state_dict(cc, seed=0)
engs, streams = [], []
for i in range(slots):
    e = Engine(mc, u, cc, dtype="bf16", device=0, noise_seed=100 + i)
    e.load_state_dict(L.MODEL_MAIN, main); e.load_state_dict(L.MODEL_COND, cond); e.finalize(strict=True)
    engs.append(e); streams.append(torch.cuda.current_stream() if os.environ.get("PROBE_DEFAULT_STREAM") else torch.cuda.Stream())
B, T, N = 32, 38400, 50
wav = torch.from_numpy(synth.synthetic_wav(B, T, seed=1234)).cuda()
torch.cuda.synchronize()
for i in range(slots):
    with torch.cuda.stream(streams[i]):
        engs[i].decode(wav, N, per_item=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
outs = []
depth = int(os.environ.get("PROBE_DEPTH", "0"))      # > 0: at most this many decodes of a slot may be outstanding when the next is submitted
evs = [[] for _ in range(slots)]
for k in range(nb):
    i = k % slots
    if depth and len(evs[i]) >= depth:
        evs[i][-depth].synchronize()
    with torch.cuda.stream(streams[i]):
        outs.append(engs[i].decode(wav, N, per_item=True))
        ev = torch.cuda.Event(); ev.record(); evs[i].append(ev)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert all(bool(torch.isfinite(o).all()) for o in outs)
print(f"slots={slots} split={'1' if os.environ.get('LDC_NO_SPLIT') else os.environ.get('LDC_SPLIT', '2')}: {1000 * dt / nb:.1f} ms per batch of {B}, {nb * B * T / 16000.0 / dt:.1f} audio-s/s")
