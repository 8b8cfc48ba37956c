"""Generate tests/golden/cli50.npz: what the two 50-step CLI tests of tests/test_gpu_frontend.py expect, computed ONCE here by the CPU
oracle (oracle/ldc_oracle.py, itself pinned to the reference by tests/test_oracle_golden.py) instead of on the GPU box's host cores in
every run of the suite (286 of its 445 s on a slow box).  The inputs -- seeded audio, seeded noise tapes, the dim-32 synthetic
checkpoints -- are rebuilt by the tests with the same helpers (tests/helpers.py: cli50_*).

  default.<k>   waveform the default-mode test expects for file u<k>.wav (bf16 engine, 50 steps, its own noise tape)
  c5.whole      the 30 s recording of the configs[4] test: 13 chunks decoded by the oracle on the fake-quantised (e4m3) weights and
                activations, raw decoder outputs joined and normalised over the whole recording (sample.py:133-134)

Run in the build container (about ten minutes on 8 cores):   python tools/gen_golden_cli.py"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ladiffcodec_amd import synth  # noqa: E402
from oracle import ldc_oracle as O  # noqa: E402
from helpers import (CASES, COND_CFG, T, cli50_c5_audio, cli50_c5_plan, cli50_c5_tape, cli50_default_audio, cli50_default_tape,  # noqa: E402
                     cond_sd_np, fake_quantise_unet, main_sd_np)

OUT = os.path.join(ROOT, "tests", "golden", "cli50.npz")
torch.set_num_threads(8)


def main():
    t0 = time.time()
    out = {}
    mc, u, _ = CASES["r84"]
    sdc, sdm = synth.to_torch(cond_sd_np()), synth.to_torch(main_sd_np("r84"))
    steps = 50
    for k in range(7):
        x = cli50_default_audio(k)
        ref = O.decode_utterances(sdc, COND_CFG, sdm, mc, u, T(x).reshape(1, 1, -1), steps, cli50_default_tape(k, x.size // mc.hop_length, steps))
        out[f"default.{k}"] = ref["wav"].numpy().reshape(-1).astype(np.float32)
        print(f"default {k} done {time.time() - t0:.0f} s", flush=True)
    x = cli50_c5_audio()
    plan, chunk = cli50_c5_plan(x.size)
    sd_q = fake_quantise_unet(main_sd_np("r84"), u)
    O.WS_PREFOLDED, O.ACT_FP8 = True, True
    try:
        xb = torch.stack([T(x[st:st + chunk]) for st, _ in plan[:12]]).reshape(12, 1, chunk)
        nb = torch.cat([cli50_c5_tape(k, chunk // mc.hop_length, steps) for k in range(12)], dim=1)
        full = O.decode_utterances(sdc, COND_CFG, sd_q, mc, u, xb, steps, nb, per_item=True)["wav_raw"]
        st, ln = plan[12]
        last = O.decode_utterances(sdc, COND_CFG, sd_q, mc, u, T(x[st:st + ln]).reshape(1, 1, ln), steps,
                                   cli50_c5_tape(12, ln // mc.hop_length, steps), per_item=True)["wav_raw"]
    finally:
        O.WS_PREFOLDED, O.ACT_FP8 = False, False
    raws = [full[k:k + 1] for k in range(12)] + [last]
    out["c5.whole"] = O.output_normalise(torch.cat(raws, dim=-1)).numpy().reshape(-1).astype(np.float32)
    print(f"c5 done {time.time() - t0:.0f} s", flush=True)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) / 1e6, "MB")


if __name__ == "__main__":
    main()
