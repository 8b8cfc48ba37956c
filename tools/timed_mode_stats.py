"""Per-kernel durations of the mode the bench TIMES (hipGraph replay, two batch parts on their own streams), from device stamps:
every launch of the pipelined conv kernel records its earliest workgroup start and latest workgroup end on the 100 MHz wall clock
(ldc_kstamps_*); rocprofv3's kernel trace serialises the two streams, so its durations are isolated ones.  Writes a markdown table:
    python tools/timed_mode_stats.py > profiles/r04_timed_mode_kernel_stats.md
Workload: BASELINE configs[1] (32 x 2.4 s, 50 steps, bf16), the bench's own engine set-up."""
import collections
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ladiffcodec_amd import lib as L, synth  # noqa: E402
from ladiffcodec_amd.model import Engine  # noqa: E402
from ladiffcodec_amd.spec import CodecConfig, UnetConfig  # noqa: E402


def main():
    B, T, N = 32, 38400, 50
    mc = CodecConfig(enc_ratios=(8, 4), quantization=False)
    u = UnetConfig(dim=256, upsampling_ratios=(5, 2), unet_scale_cond=True)
    cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=3.0)
    sd = synth.ladiff_state_dict(mc, u, seed=1)
    e = Engine(mc, u, cc, dtype="bf16", device=0, noise_seed=4321)
    e.load_state_dict(L.MODEL_MAIN, {k: v for k, v in sd.items() if not k.startswith("diffusion.model.")})
    e.load_state_dict(L.MODEL_COND, synth.codec_state_dict(cc, seed=0))
    e.finalize(strict=True)
    wav = torch.from_numpy(synth.synthetic_wav(B, T, seed=1234)).cuda()
    # reference timing without stamps
    for _ in range(2):
        e.decode(wav, N, noise=None, per_item=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        e.decode(wav, N, noise=None, per_item=True)
    torch.cuda.synchronize()
    plain_ms = (time.perf_counter() - t0) / 3 * 1e3
    e.kstamps_enable(True)
    e.timeline_enable(True)
    for _ in range(2):
        e.decode(wav, N, noise=None, per_item=True)
    torch.cuda.synchronize()
    e.kstamps_reset()
    t0 = time.perf_counter()
    e.decode(wav, N, noise=None, per_item=True)
    torch.cuda.synchronize()
    stamped_ms = (time.perf_counter() - t0) * 1e3
    tl = e.timeline(N, 2)
    rows = collections.OrderedDict()
    per_step_conv = []
    n_conv_ops = 0
    for part in range(2):
        t, names, classes = e.kstamps(part, N)
        n_ops = len(names)
        dur = t[:, :, 1] - t[:, :, 0]
        ok = (t[:, :, 0] > 0) & (t[:, :, 1] > 0)
        # the first steps of a decode run eagerly / inside the capture: keep the replayed ones
        for o in range(n_ops):
            m = ok[5:, o]
            if not m.any():
                continue
            n_conv_ops += 1 if part == 0 else 0
            d = dur[5:, o][m]
            r = rows.setdefault(names[o], [])
            r.extend(d.tolist())
        per_step_conv.append(np.where(ok, dur, 0.0).sum(axis=1)[5:])
        # gaps: from the end of one stamped conv to the start of the next stamped conv of the same part and step
    step_ms = [float((a[5:, 1] - a[5:, 0]).mean()) / 1e3 for a in tl]
    print("# Per-kernel durations in the TIMED mode (graph replay, two batch parts on two streams), from device stamps")
    print()
    print(f"`python tools/timed_mode_stats.py` on MI355X: BASELINE configs[1] (32 x 2.4 s, 50 steps, bf16 UNet).  Every launch of the pipelined conv")
    print("kernel stamps the start of its first workgroup and the latest end among every eighth workgroup and the last eight (100 MHz device clock, `ldc_kstamps_*`); steps 5..49 of one")
    print("replayed decode, both parts.  Durations therefore include what the other part's kernels, running on the same CUs at the same time,")
    print("cost this launch -- the figure rocprofv3 cannot show (its kernel trace serialises the streams).")
    print()
    print(f"* decode wall time: {plain_ms:.1f} ms without stamps, {stamped_ms:.1f} ms with them (one store per launch, one atomic per eighth workgroup)")
    print(f"* mean denoise step per part (timeline stamps): {step_ms[0]:.3f} / {step_ms[1]:.3f} ms; the parts run concurrently")
    cs = [float(a.mean()) / 1e3 for a in per_step_conv]
    print(f"* sum of the stamped conv launches per step and part: {cs[0]:.3f} / {cs[1]:.3f} ms = {100 * cs[0] / step_ms[0]:.0f} % / {100 * cs[1] / step_ms[1]:.0f} % of the part's step")
    print(f"  ({n_conv_ops} pipelined-conv launches per step and part; the rest of a step is the launches in between -- attention, LayerNorm, step bookkeeping -- and the launch boundaries)")
    print()
    print("| conv launch (k, inputs -> outputs, positions, fused epilogue) | launches | mean us | p10 us | p90 us | share of stamped time |")
    print("|---|---|---|---|---|---|")
    tot = sum(sum(v) for v in rows.values())
    for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        a = np.array(v)
        print(f"| `{name}` | {len(v)} | {a.mean():.1f} | {np.percentile(a, 10):.1f} | {np.percentile(a, 90):.1f} | {100 * a.sum() / tot:.1f} % |")
    print()
    print(f"total stamped conv time: {tot / 1e3:.1f} ms over both parts and 45 steps = {tot / 1e3 / 45 / 2:.3f} ms per step and part")


if __name__ == "__main__":
    main()
