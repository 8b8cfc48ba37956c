"""Two optimisation steps of the full-width diffusion UNet (BASELINE config 4 shape: diff_dims 256, seq_length 1200, upsampling 5 2),
the workload of tools/run_r03.sh's `train` profile; then (HOST=1) the host's share: time until step_from_wav RETURNS (everything
enqueued) against time until the GPU is through.  usage: [HOST=1] python tools/train_step_time.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ladiffcodec_amd import lib as L, synth, train as TR
from ladiffcodec_amd.spec import CodecConfig, UnetConfig
from ladiffcodec_amd.model import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0)
mc = CodecConfig(enc_ratios=(8, 4), quantization=False)
u = UnetConfig(dim=256, upsampling_ratios=(5, 2), unet_scale_cond=True)
full = synth.ladiff_state_dict(mc, u, seed=1)
e = Engine(mc, u, cc, dtype="f32", device=0)
e.load_state_dict(L.MODEL_MAIN, {k: v for k, v in full.items() if not k.startswith("diffusion.model.")})
e.load_state_dict(L.MODEL_COND, synth.codec_state_dict(cc, seed=0)); e.finalize(strict=True)
sd = {k[len("diff_model."):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in full.items() if k.startswith("diff_model.")}
tr = TR.DiffusionTrainer(e, sd, dim=u.dim, dim_mults=u.dim_mults, lr=1e-4, upsampling_ratios=u.upsampling_ratios, unet_scale_cond=True)
wav = torch.from_numpy(synth.synthetic_wav(B, 38400, seed=5))
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = tr.step_from_wav(wav, generator=torch.Generator().manual_seed(it))
    torch.cuda.synchronize()
    print(f"step {it}: loss {float(loss.cpu()[0]):.4f}, {time.perf_counter() - t0:.2f} s for B = {B} x 2.4 s (L = 1200, {sum(v.numel() for v in sd.values()) / 1e6:.1f} M parameters)")

if os.environ.get("HOST"):
    wav = wav.cuda()
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = tr.step_from_wav(wav)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"step {it}: host returned after {1e3 * (t1 - t0):.1f} ms, GPU done after {1e3 * (t2 - t0):.1f} ms")
