"""Where the device-to-device copies and ATen glue kernels of an optimisation step come from: torch.profiler with Python stacks over one
step of tools/train_step_time.py's workload; prints the train.py line each `Memcpy DtoD` / aten::add / aten::copy_ was issued from."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
from ladiffcodec_amd import lib as L, synth, train as TR
from ladiffcodec_amd.spec import CodecConfig, UnetConfig
from ladiffcodec_amd.model import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0)
mc = CodecConfig(enc_ratios=(8, 4), quantization=False)
u = UnetConfig(dim=256, upsampling_ratios=(5, 2), unet_scale_cond=True)
full = synth.ladiff_state_dict(mc, u, seed=1)
e = Engine(mc, u, cc, dtype="f32", device=0)
e.load_state_dict(L.MODEL_MAIN, {k: v for k, v in full.items() if not k.startswith("diffusion.model.")})
e.load_state_dict(L.MODEL_COND, synth.codec_state_dict(cc, seed=0)); e.finalize(strict=True)
sd = {k[len("diff_model."):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in full.items() if k.startswith("diff_model.")}
tr = TR.DiffusionTrainer(e, sd, dim=u.dim, dim_mults=u.dim_mults, lr=1e-4, upsampling_ratios=u.upsampling_ratios, unet_scale_cond=True)
wav = torch.from_numpy(synth.synthetic_wav(B, 38400, seed=5)).cuda()
for it in range(2):
    tr.step_from_wav(wav)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    tr.step_from_wav(wav)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::contiguous", "aten::clone", "aten::mul", "aten::zeros", "aten::zero_", "aten::fill_", "aten::sub", "aten::div"):
        where = "?"
        for fr in ev.stack:
            if "train.py" in fr or "model.py" in fr or "parallel.py" in fr:
                where = fr.split("/")[-1]
                break
        cnt[(ev.name, where)] += 1
for (name, where), n in cnt.most_common(60):
    print(f"{n:5d}  {name:18s} {where}")
