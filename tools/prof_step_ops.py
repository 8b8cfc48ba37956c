"""Which torch ops a full-width optimisation step issues (torch.profiler, counts and times): python tools/prof_step_ops.py  (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HOST"] = ""
import runpy
import torch
sys.argv = [sys.argv[0], "32"]
g = runpy.run_path(os.path.join(ROOT, "tools", "train_step_time.py"), run_name="not_main")
tr, wav = g["tr"], g["wav"].cuda()
tr.step_from_wav(wav); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    tr.step_from_wav(wav); torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(ev, key=lambda e: -e.count)
for e in rows[:40]:
    print(f"{e.key[:70]:70s} n={e.count:5d} cpu={e.cpu_time_total/1e3:8.2f} ms cuda={getattr(e,'device_time_total',0)/1e3:8.2f} ms")
