"""Wall-clock split of one C2 decode: cond encode+RVQ, denoise loop (50 steps), decoder+normalise, whole ldc_decode."""
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
B, T, N = 32, 38400, 50
wav = torch.from_numpy(synth.synthetic_wav(B, T, seed=1234)).cuda()


def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return 1000 * (time.perf_counter() - t0) / reps, r


t_cond, cond = timed(lambda: e.get_cond(wav))
img = e.cond_upsample(cond, 0)
img = img / (img.abs().amax() + 1e-8)
t_den, lat = timed(lambda: e.denoise(img, cond, N))
t_dec, _ = timed(lambda: e.output_normalise(e.decode_latents(L.MODEL_MAIN, lat), per_item=True))
t_all, _ = timed(lambda: e.decode(wav, N, per_item=True))
print(f"get_cond {t_cond:.2f} ms | denoise x{N} {t_den:.2f} ms ({t_den / N:.3f} per step) | decoder+norm {t_dec:.2f} ms | ldc_decode {t_all:.2f} ms")
