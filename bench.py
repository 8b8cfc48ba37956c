"""Throughput bench of the LaDiffCodec decode path on MI355X.

metric  : audio-seconds decoded per wall-second (BASELINE.json), 16 kHz, 3 kbps condition, 50-step DDPM
workload: BASELINE.json configs[1] ("C2" in SURVEY.md section 8): LaDiffCodec diff_dims=256, enc_ratios 8 4,
          batch = 32 x 2.4 s utterances per GPU, bf16 UNet on MFMA (codec stages exact fp32).
step    : one pass of the whole hot path over one resident batch: cond encode -> RVQ -> upsample ->
          N denoise steps (hipGraph replay) -> SEANet decode -> normalise  (ldc_decode).
N > 1   : one process per GPU (torch.distributed.run), every rank decodes its own batch of
          independent utterances (weak scaling, no data-path collective); RCCL only broadcasts the
          checkpoint once and gathers the decoded waveforms at the end of every step.

Prints ONE JSON line on rank 0.  Synthetic seeded weights and audio (no network for checkpoints/datasets).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ladiffcodec_amd import lib as L, parallel, spec, synth  # noqa: E402
from ladiffcodec_amd.spec import CodecConfig, UnetConfig  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E ~8 TB/s
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "fp8": 2500.0}   # fp8 WEIGHTS are expanded to bf16 in registers: the MFMA is the bf16 one     # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md


def fp8_label():
    """what the MFMAs of the fp8 engine are (BASELINE configs[4]): the convs fed by a conv-only tensor (block2 of every ResnetBlock,
    to_qkv, final_conv: 34 % of the step's flops) run fp8 x fp8 on the block-scaled fp8 MFMA, the others bf16 x fp8 weights"""
    if os.environ.get("LDC_FP8_ACT", "1") != "0":
        return ("fp8 e4m3 x fp8 e4m3 on v_mfma_scale_f32_32x32x64_f8f6f4 for the convs whose input is produced for them alone (34 % of the "
                "UNet's flops), bf16 activations x fp8 e4m3 weights (bf16 MFMA) for the others; fp32 accumulate")
    return "bf16 activations x fp8 (e4m3) weights, bf16 MFMA"


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    # cgroup CPU quota (containers): honour it, an oversubscribed OpenMP pool is pathologically slow
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = min(n, max(1, int(int(q[0]) / int(q[1]))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(cc, mc, u, sd_cond, sd_main, n_steps, seconds, batch, budget_s=20.0):
    """The CPU oracle (a port of the reference's arithmetic, oracle/ldc_oracle.py) timed on this host, on a
    bounded sample: front end (encode, RVQ, upsample) and decoder in full, as many of the N DDPM steps as
    fit in ~budget_s (at least 2), the rest extrapolated from the measured per-step mean."""
    from oracle import ldc_oracle as O
    cores = host_threads()
    torch.set_num_threads(cores)
    T = int(seconds * 16000) // 640 * 640
    wav = torch.from_numpy(synth.synthetic_wav(batch, T, seed=1234))
    a, b = synth.to_torch(sd_cond), synth.to_torch(sd_main)
    g = torch.Generator().manual_seed(4321)
    with torch.no_grad():
        t0 = time.perf_counter()
        cond, _, _, _ = O.get_cond(a, cc, wav)
        img = O.start_image(b, u, cond, per_item=True)
        t_front = time.perf_counter() - t0
        step_times = []
        t = n_steps - 1
        while t >= 0 and (len(step_times) < 2 or sum(step_times) + step_times[-1] < budget_s):
            noise = torch.randn(img.shape, generator=g)
            t1 = time.perf_counter()
            img = O.p_sample(b, u, img, t, cond, noise)
            step_times.append(time.perf_counter() - t1)
            t -= 1
        t2 = time.perf_counter()
        O.output_normalise(O.seanet_decode(b, mc, img), per_item=True)
        t_back = time.perf_counter() - t2
    timed = len(step_times)
    per_step = sum(step_times[1:]) / max(1, timed - 1) if timed > 1 else step_times[0]
    total = t_front + t_back + per_step * n_steps
    return {"value": batch * T / 16000.0 / total, "unit": "audio-s/wall-s", "cores": cores, "kind": "port",
            "sample": f"{batch} x {T / 16000.0:.1f} s utterance(s), fp32 oracle: front/back ends in full "
                      f"({t_front + t_back:.2f} s), {timed} of {n_steps} DDPM steps timed ({per_step:.3f} s/step), "
                      f"rest extrapolated"}


def bench_training(args, eng, mc, u, cc, sd_main, sd_cond, wav, T, rank, world, dev, result_fd):
    """BASELINE configs[3]: one optimisation step of the diffusion UNet per bench step (srcs/train.py --run_diff: frozen encoders,
    q_sample, UNet forward, l1 objective, UNet backward, gradient averaging over ranks, Adam) on B utterances per GPU.  The training
    path is csrc/train.hip + csrc/train_mm3.hip (fp32 tensors; the GEMM shapes on the bf16 MFMA with every operand split into two bf16 --
    three MFMAs per product, 2^-16-class -- or, with LDC_TRAIN_FP32_MFMA=1, on the exact-fp32 MFMA), pinned to the reference's autograd."""
    from ladiffcodec_amd import train as TR
    B = wav.shape[0]
    sd = {k[len("diff_model."):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_main.items() if k.startswith("diff_model.")}
    # a second engine with the same frozen codec weights: the encoders of the NEXT batch run on a side stream under this batch's UNet
    # (every step still runs its encoders inside the timed region, once; LDC_TRAIN_NO_PREFETCH=1 keeps them in line)
    front = None
    if not os.environ.get("LDC_TRAIN_NO_PREFETCH"):
        from ladiffcodec_amd.model import Engine
        front = Engine(mc, u, cc, dtype="f32", device=dev.index if dev.index is not None else 0, noise_seed=99)
        front.load_state_dict(L.MODEL_MAIN, sd_main)
        front.load_state_dict(L.MODEL_COND, sd_cond)
        front.finalize(strict=True)
    tr = TR.DiffusionTrainer(eng, sd, dim=u.dim, dim_mults=u.dim_mults, lr=1e-4, upsampling_ratios=u.upsampling_ratios,
                             unet_scale_cond=u.unet_scale_cond, frontend=front)
    if args.no_dw_side:
        tr.dw_side = False      # A/B: the Blocks' weight-gradient GEMMs in line instead of on the side stream
    for i in range(args.warmup):
        tr.step_from_wav(wav, next_wav=wav)
        torch.cuda.synchronize(dev)
        log(f"warmup {i} done")
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.step_from_wav(wav, next_wav=wav)
    torch.cuda.synchronize(dev)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, device=dev)
    assert bool(torch.isfinite(loss).all()), "non-finite loss"
    log(f"timed region: {elapsed:.3f} s for {args.steps} optimisation step(s), last loss {float(loss.cpu()[0]):.4f}")
    n_par = sum(v.numel() for v in sd.values())
    fwd_flops, _ = eng.unet_step_cost(B, T // mc.hop_length)
    ach = 3.0 * fwd_flops * args.steps / elapsed / 1e12           # forward + dX + dW of every conv-shaped layer
    exact = bool(os.environ.get("LDC_TRAIN_FP32_MFMA"))
    plain = bool(os.environ.get("LDC_TRAIN_BF16")) and not exact       # opt-in: hi terms only, one bf16 MFMA per product (autocast-class numerics)
    result = {
        "metric": "audio-sec per wall-sec through ONE optimisation step of the diffusion UNet (training, --run_diff)",
        "value": world * B * (T / 16000.0) * args.steps / elapsed, "unit": "audio-s/wall-s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if exact else ("bf16 (fp32 tensors; GEMM operands rounded to bf16, one MFMA per product, fp32 accumulate: LDC_TRAIN_BF16)" if plain else
                                      "bf16x3 (fp32 tensors; GEMM operands split into bf16 hi + lo, three MFMAs per product, fp32 accumulate)"),
        "data": "synthetic (seeded weights with the reference key set, synthetic 16 kHz audio, device-drawn t / noise)",
        "config": {"workload": f"diffusion training step, diff_dims={u.dim}, seq_length {T // mc.hop_length}, batch={B}x{T / 16000.0:.1f} s per GPU, "
                               f"Adam over {n_par / 1e6:.1f} M parameters", "name": "c4", "global_batch": world * B,
                   "frozen_encoders": "next batch's, on a side stream under this step's UNet" if front is not None else "in line",
                   "rccl_ranks": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                   "parallelism": f"dp{world} (flat fp32 gradient reduce-scatter + all-gather per step: {4 * n_par / 1e6:.0f} MB)"},
        "roofline": ({"bound": "mfma", "kernel": "convmm_kernel<0|1|2>: forward, dX and dW of every conv / pointwise layer on the exact-fp32 MFMA (csrc/train.hip)",
                      "achieved": ach, "peak": MFMA_PEAK_TFLOPS["f32"], "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS["f32"], "traffic": None}
                     if exact else
                     {"bound": "mfma", "kernel": "mm3_kernel<0|1|2>: forward, dX and dW with ONE bf16 MFMA per product (hi terms only, LDC_TRAIN_BF16; csrc/train_mm3.hip)",
                      "achieved": ach, "peak": MFMA_PEAK_TFLOPS["bf16"], "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS["bf16"], "traffic": None}
                     if plain else
                     {"bound": "mfma", "kernel": "mm3_kernel<0|1|2>: forward, dX and dW of every conv / pointwise / Linear / transposed-conv layer as three "
                                                 "bf16 MFMAs per fp32 product (csrc/train_mm3.hip)",
                      "achieved": ach, "peak": MFMA_PEAK_TFLOPS["bf16"] / 3.0, "unit": "TFLOP/s", "frac": 3.0 * ach / MFMA_PEAK_TFLOPS["bf16"], "traffic": None,
                      "peak_note": "dense bf16 MFMA peak / 3: `achieved` counts the algorithmic (fp32-equivalent) flops, each of which is three bf16 MFMA flops"}),
    }
    result["roofline"]["note"] = ("whole-step average: 3 x the UNet's forward conv flops / step time (frozen encoders, norms, attention cores and "
                                  "Adam included in the time)")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ldc_oracle as O, train_oracle as TO
        cores = host_threads()
        torch.set_num_threads(cores)
        nb = min(2, B)
        a, b = synth.to_torch(sd_cond), synth.to_torch(sd_main)
        w = wav[:nb].cpu()
        params = {k: v.clone().requires_grad_() for k, v in b.items() if k.startswith("diff_model.")}
        sched = {k: v for k, v in b.items() if k.startswith("diffusion.")}
        opt = torch.optim.Adam(list(params.values()), lr=1e-4)
        g = torch.Generator().manual_seed(1)
        times = []
        for _ in range(2):
            t1 = time.perf_counter()
            with torch.no_grad():
                cond = O.get_cond(a, cc, w)[0]
                x0 = O.seanet_encode(b, mc, w) / 18.0
            t = torch.randint(0, 1000, (nb,), generator=g)
            noise = torch.randn(x0.shape, generator=g)
            opt.zero_grad()
            lo = TO.p_losses_objective(sched, O.unet_forward(params, u, TO.q_sample(sched, x0, t, noise), t, cond), noise, t)
            lo.backward()
            opt.step()
            times.append(time.perf_counter() - t1)
        result["cpu_baseline"] = {"value": nb * (T / 16000.0) / times[-1], "unit": "audio-s/wall-s", "cores": cores, "kind": "port",
                                  "sample": f"{nb} x {T / 16000.0:.1f} s utterance(s), fp32 oracle under torch autograd + torch.optim.Adam, second of two steps ({times[-1]:.2f} s)"}
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    os.close(result_fd)


def seanet_flops(cc, T):
    """algorithmic conv + LSTM-input-GEMM flops of one encode + one decode of a T-sample clip (spec.seanet_*_layers)"""
    total, L = 0.0, T
    for ly in spec.seanet_encoder_layers(cc):
        if ly.kind == "conv":
            L = L // ly.stride
            total += 2.0 * L * ly.cin * ly.cout * ly.kernel
        elif ly.kind == "res":
            total += 2.0 * L * (ly.cin * ly.hidden * ly.kernel + ly.hidden * ly.cout + ly.cin * ly.cout)
        elif ly.kind == "lstm":
            total += 2.0 * L * ly.layers * 8 * ly.cin * ly.cin
    for ly in spec.seanet_decoder_layers(cc):
        if ly.kind == "conv":
            total += 2.0 * L * ly.cin * ly.cout * ly.kernel
        elif ly.kind == "convtr":
            total += 2.0 * L * ly.cin * ly.cout * ly.kernel      # every input position meets every tap once
            L = L * ly.stride
        elif ly.kind == "res":
            total += 2.0 * L * (ly.cin * ly.hidden * ly.kernel + ly.hidden * ly.cout + ly.cin * ly.cout)
        elif ly.kind == "lstm":
            total += 2.0 * L * ly.layers * 8 * ly.cin * ly.cin
    return total


def bench_c1(args, eng, cc, sd_cond, wav, T, rank, world, dev, result_fd):
    """BASELINE configs[0]: EnCodec-style round trip without diffusion -- SEANet encode -> RVQ (3 kbps) -> cond-codec decode of
    2.4 s clips (enc_ratios 8 5 4 2; model.py:223-231 + seanet.py:157-248), all exact fp32.  The reference's own case is ONE clip on
    the CPU; B = 1 is the default here too (latency-bound: LSTM recurrences over 120 frames), --batch N gives the batched rate."""
    B = wav.shape[0]

    def step():
        cond = eng.get_cond(wav)
        out = eng.decode_latents(L.MODEL_COND, cond)
        parallel.gather_results(out, world)
        return out
    for _ in range(max(1, args.warmup)):
        step()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(dev)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, device=dev)
    assert bool(torch.isfinite(out).all()) and out.shape[-1] == T
    flops = seanet_flops(cc, T) * B
    ach = flops * args.steps / elapsed / 1e12
    result = {
        "metric": "audio-sec round-tripped / wall-sec, 16kHz EnCodec encode -> RVQ 3kbps -> decode (no diffusion)",
        "value": world * B * (T / 16000.0) * args.steps / elapsed, "unit": "audio-s/wall-s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (seeded weights with the reference key set, synthetic 16 kHz audio)",
        "config": {"workload": f"EnCodec 16kHz encode->RVQ->decode only (no diffusion), {B}x{T / 16000.0:.1f} s clip(s), enc_ratios 8 5 4 2, bandwidth=3",
                   "name": "c1", "global_batch": world * B, "rccl_ranks": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                   "parallelism": f"dp{world} (utterance-sharded, no data-path collective)"},
        "roofline": {"bound": "mfma", "kernel": "conv_gemm_kernel<float> (SEANet convs, transposed convs and LSTM input GEMMs on the exact-fp32 MFMA) -- "
                                                "whole round trip incl. the sequential LSTM recurrences and RVQ",
                     "achieved": ach, "peak": MFMA_PEAK_TFLOPS["f32"], "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS["f32"], "traffic": None,
                     "algorithmic_gflop_per_clip": flops / B / 1e9,
                     "note": "latency-bound at this size: four LSTM layers of 120 sequential steps and ~40 small launches per clip"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ldc_oracle as O
        cores = host_threads()
        torch.set_num_threads(cores)
        a = synth.to_torch(sd_cond)
        w = wav[:1].cpu()
        times = []
        with torch.no_grad():
            for _ in range(3):
                t1 = time.perf_counter()
                q = O.get_cond(a, cc, w)[0]
                O.seanet_decode(a, cc, q)
                times.append(time.perf_counter() - t1)
        result["cpu_baseline"] = {"value": (T / 16000.0) / min(times[1:]), "unit": "audio-s/wall-s", "cores": cores, "kind": "port",
                                  "sample": f"1 x {T / 16000.0:.1f} s clip, fp32 oracle (encode, RVQ, decode), best of two after one warm-up ({min(times[1:]):.3f} s)"}
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    os.close(result_fd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c8", "c5", "c4"],
                    help="c1 = BASELINE configs[0]: EnCodec encode -> RVQ -> decode of one 2.4 s clip, no diffusion; c2 = BASELINE configs[1] (the metric's config: 3 kbps, enc_ratios 8 4, 50 steps); c3 = configs[2] per GPU "
                         "(1.5 kbps condition, 200 steps); c8 = the released checkpoints' layout (enc_ratios 8, latent L = 4800, "
                         "upsampling 5 4 2; README.md:30,35), 3 kbps, 50 steps; c5 = configs[4]: one 30 s recording as 13 chunks of "
                         "2.4 s (batch items), fp8 UNet weights; c4 = configs[3] per GPU: one optimisation step of the diffusion UNet per bench step (fp32 correctness path of the training row)")
    ap.add_argument("--batch", type=int, default=0, help="utterances per GPU (0 = the config's own: 32; c5: 13)")
    ap.add_argument("--seconds", type=float, default=2.4)
    ap.add_argument("--denoise-steps", type=int, default=0, help="0 = the config's own (50; c3: 200)")
    ap.add_argument("--diff-dims", type=int, default=256)
    ap.add_argument("--dtype", default="", choices=["", "bf16", "f32", "fp8"], help="UNet dtype (default bf16; c5: fp8 weights)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4, help="utterances in the CPU-oracle sample (SURVEY 8d: B = 4)")
    ap.add_argument("--no-dw-side", action="store_true", help="c4: weight-gradient GEMMs in line (A/B of DiffusionTrainer.dw_side)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the supplementary two-batches-in-flight measurement")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="batches decoded concurrently (default 1 = the metric's reading: one batch of the config's size at a time). "
                         "n > 1: n engines on n streams, every batch as ONE chain; steps are dealt round-robin and all K finish inside "
                         "the timed region (supplementary number, see DESIGN.md section 7)")
    args = ap.parse_args()
    args.batch = args.batch or (13 if args.config == "c5" else (1 if args.config == "c1" else 32))
    args.dtype = args.dtype or ("fp8" if args.config == "c5" else ("f32" if args.config == "c4" else "bf16"))
    if args.config in ("c4", "c1"):
        args.in_flight, args.no_pipelined = 1, True

    # stdout carries exactly ONE line, the JSON result: RCCL prints a banner (version / hostname / library path) on
    # file descriptor 1 when the process group comes up, so route fd 1 to stderr until the result is written
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank, local_rank, world = parallel.init_process_group("nccl")
    # --gpus N means N RCCL ranks: anything else (a forgotten torch.distributed.run, a group that did not come up) fails loudly instead
    # of printing a one-GPU number under an N-GPU label
    n_ranks = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    if world != args.gpus or n_ranks != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: WORLD_SIZE={world}, RCCL ranks={n_ranks} -- launch it as `python -m torch.distributed.run "
                         f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`")
    if world > 1 and torch.distributed.get_backend() != "nccl":
        raise SystemExit(f"bench.py: the process group runs on {torch.distributed.get_backend()}, not RCCL")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    kbps = 1.5 if args.config == "c3" else 3.0
    cc = CodecConfig(enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=kbps)
    if args.config == "c8":
        mc = CodecConfig(enc_ratios=(8,), quantization=False)
        u = UnetConfig(dim=args.diff_dims, upsampling_ratios=(5, 4, 2), unet_scale_cond=True)
    else:
        mc = CodecConfig(enc_ratios=(8, 4), quantization=False)
        u = UnetConfig(dim=args.diff_dims, upsampling_ratios=(5, 2), unet_scale_cond=True)
    T = int(args.seconds * 16000) // 640 * 640
    N = args.denoise_steps or (200 if args.config == "c3" else 50)
    if world > 1:
        log(f"rank {rank}: RCCL process group up, {torch.distributed.get_world_size()} ranks (backend {torch.distributed.get_backend()})")

    # ---- weights: rank 0 builds the synthetic checkpoints, RCCL broadcasts them (one flat buffer each) ----
    main_layout = spec.codec_keys(mc) + spec.unet_keys(u, "diff_model") + [(f"diffusion.{b}", (u.timesteps,)) for b in spec.SCHEDULE_BUFFERS]
    cond_layout = spec.codec_keys(cc)
    sd_main = sd_cond = None
    if rank == 0:
        full = synth.ladiff_state_dict(mc, u, seed=1)
        sd_main = {k: full[k] for k, _ in main_layout}      # the diffusion.model.* aliases are skipped (same tensors)
        sd_cond = synth.codec_state_dict(cc, seed=0)
    sd_main = parallel.broadcast_state_dict(sd_main, main_layout, device=dev)
    sd_cond = parallel.broadcast_state_dict(sd_cond, cond_layout, device=dev)

    log(f"rank {rank}/{world}: checkpoints ready ({len(sd_main)} + {len(sd_cond)} tensors)")
    from ladiffcodec_amd.model import Engine
    n_fl = max(1, args.in_flight)
    engines, slot_streams = [], []
    for k in range(n_fl):
        e_k = Engine(mc, u, cc, dtype=args.dtype, device=local_rank, noise_seed=4321 + rank + 1000 * k)
        if n_fl > 1:
            e_k.set_option("split", 1)     # with several batches in flight each batch is one chain
        e_k.load_state_dict(L.MODEL_MAIN, sd_main)
        e_k.load_state_dict(L.MODEL_COND, sd_cond)
        e_k.finalize(strict=True)
        engines.append(e_k)
        slot_streams.append(torch.cuda.Stream(device=dev) if n_fl > 1 else None)
    eng = engines[0]
    log("weights folded/packed/uploaded")

    B = args.batch
    wav = torch.from_numpy(synth.synthetic_wav(B, T, seed=1234 + rank)).to(dev)   # resident in HBM before timing

    if args.config == "c1":
        bench_c1(args, eng, cc, sd_cond, wav, T, rank, world, dev, result_fd)
        eng.close()
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return
    if args.config == "c4":
        bench_training(args, eng, mc, u, cc, sd_main, sd_cond, wav, T, rank, world, dev, result_fd)
        eng.close()
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return

    step_no = [0]

    failures = []      # device-side failure flags (a bounded in-launch wait that gave up: [gn_wait] / [coop_lstm]) seen by the timed calls

    def checked_decode(e_k):
        try:
            return e_k.decode(wav, N, noise=None, per_item=True)
        except RuntimeError as ex:
            if "device-side failure" not in str(ex):
                raise
            failures.append(str(ex)[:160])
            log(f"DEVICE-SIDE FAILURE in a timed call: {ex}")
            return torch.full((B, 1, T), float("nan"), device=dev)

    def step():
        k = step_no[0] % n_fl
        step_no[0] += 1
        if n_fl == 1:
            out = checked_decode(eng)
            parallel.gather_results(out, world)      # RCCL all_gather of the decoded waveforms (no-op without a process group)
            return out
        with torch.cuda.stream(slot_streams[k]):
            out = checked_decode(engines[k])
            parallel.gather_results(out, world)
        return out

    for i in range(max(args.warmup, n_fl if args.warmup else 0)):     # every engine captures its graphs before the timed region
        step()
        torch.cuda.synchronize(dev)
        log(f"warmup {i} done")
    step_no[0] = 0
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    for e_k in engines:
        e_k.host_stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(dev)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    mine_s = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(mine_s, device=dev)
    per_rank_ms = [1000.0 * float(v) / args.steps for v in parallel.gather_results(torch.tensor([mine_s], dtype=torch.float64, device=dev), world)]
    if failures:
        raise SystemExit(f"bench.py: {len(failures)} device-side failure(s) inside the timed region -- the number would be invalid: {failures[0]}")
    assert os.environ.get("LDC_CONV_DEBUG") or bool(torch.isfinite(out).all()), "non-finite output"   # (ablation runs compute garbage)
    log(f"timed region: {elapsed:.3f} s for {args.steps} step(s)")
    hs = [e_k.host_stats(reset=True) for e_k in engines]

    audio_s = world * B * (T / 16000.0) * args.steps
    result = {
        "metric": f"audio-sec decoded / wall-sec, 16kHz {kbps:g}kbps {N}-step DDPM",
        "value": audio_s / elapsed, "unit": "audio-s/wall-s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": fp8_label() if args.dtype == "fp8" else args.dtype, "data": "synthetic (seeded weights with the reference key set, synthetic 16 kHz audio)",
        "config": {"workload": f"LaDiffCodec {kbps:g} kbps, diff_dims={args.diff_dims}, enc_ratios {' '.join(map(str, mc.enc_ratios))}, "
                               f"{N}-step DDPM, batch={B}x{T / 16000.0:.1f} s utterances per GPU", "name": args.config,
                   "global_batch": world * B, "latent_len": T // mc.hop_length, "denoise_steps": N,
                   "rccl_ranks": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                   "batches_in_flight": n_fl, "ms_per_step_by_rank": per_rank_ms,
                   "parallelism": f"dp{world} (utterance-sharded, no data-path collective)"},
        # host side of the timed region (rank 0): what one process spends inside hipGraphLaunch per bench step -- the ceiling of
        # one process once the kernels get faster -- and how long it waited for its own look-ahead window (GPU-bound when > 0)
        "host": {"graph_launch_ms_per_step": sum(h[0] for h in hs) / args.steps, "lookahead_wait_ms_per_step": sum(h[1] for h in hs) / args.steps,
                 "graph_replays_per_step": sum(h[2] for h in hs) / args.steps,
                 "device_side_failures": len(failures),    # (raised flags of the bounded in-launch waits; a non-zero count aborts the run above)
                 # preconditions of the timed mode: the part streams the context chose by measured overlap -- accepted
                 # candidates, one 150 us spin alone vs the caller's stream and all accepted streams spinning together (equal = they overlap) --
                 # and the hardware-queue setting of the runtime
                 "part_streams": eng.stream_info(), "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "unset (runtime default: 4)")},
    }
    log("part streams: %s" % json.dumps(result["host"]["part_streams"]))

    if rank == 0 and not args.no_roofline:
        # dominant kernel = the conv-GEMM family: per-launch hipEvents on the launch stream over one full
        # eager decode of the same batch (profiling pass, separate from the timed region above)
        eng.profile(True)
        eng.decode(wav, N, noise=None, per_item=True)
        classes = eng.profile_read_classes()
        ms, launches, flops = eng.profile_read()
        eng.profile(False)
        log(f"profile pass: {launches} conv launches, {ms:.1f} ms")
        # the engine decodes the batch as independent chains on their own streams (ldc_api.cpp get_halves: two by default;
        # option "split" overrides): account the launches as they are issued
        nparts = max(1, min(eng.stream_info()["parts"], B))
        parts = [B * (k + 1) // nparts - B * k // nparts for k in range(nparts)]
        step_flops = step_bytes = 0.0
        for pb in parts:
            f_, b_ = eng.unet_step_cost(pb, T // mc.hop_length)
            step_flops += f_
            step_bytes += b_
        ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, profiles/)
        traffic, traffic_src = None, None
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from pmc_traffic import csrc_hash            # the figure is reported only while it was measured on THESE kernel sources
        for name in ("r06_conv_traffic.json", "r05_conv_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(tpath):
                continue
            if args.dtype == "bf16" and B == 32 and N == 50 and args.config == "c2":
                rec = json.load(open(tpath))
                if rec.get("csrc_sha256_16") == csrc_hash():
                    traffic = rec["hbm_bytes_per_launch"]
                    traffic_src = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on these kernel sources)"
                else:
                    traffic_src = f"profiles/{name} was measured on other kernel sources (hash mismatch): not reported"
                break
        convs_per_step = launches / max(1, N)
        result["roofline"] = {"bound": "mfma", "kernel": "conv_lean_kernel / conv_fast_kernel / conv_gemm_kernel (implicit-GEMM Conv1d on MFMA)",
                              "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                              "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src,
                              "algorithmic_bytes_per_launch": step_bytes / max(1.0, convs_per_step),
                              "algorithmic_flops_per_launch": flops / max(1, launches),
                              "launches": launches, "avg_launch_us": 1000.0 * ms / max(1, launches),
                              "unet_step_gflop": step_flops / 1e9, "unet_step_conv_algorithmic_gb": step_bytes / 1e9}
        # matrix-pipe occupancy of the dominant kernel from the committed SQ counter passes (profiles/r05_conv_counters.md): busy cycles of
        # the MFMA pipes / (1024 SIMDs x launch duration), over the five top shape classes at the bench's per-part batch
        # (like the traffic figure, reported only for the configuration and the kernel sources it was measured on)
        cpath = os.path.join(ROOT, "profiles", "r06_conv_counters.json")
        if os.path.exists(cpath) and args.dtype == "bf16" and B == 32 and N == 50 and args.config == "c2":
            crec = json.load(open(cpath))
            if crec.get("csrc_sha256_16") == csrc_hash():
                result["roofline"]["mfma_busy_frac"] = crec["mfma_busy_frac"]
                result["roofline"]["mfma_busy_frac_by_shape"] = {k: round(v["mfma_busy_frac_of_launch"], 4) for k, v in crec["shapes"].items()}
                result["roofline"]["non_mfma_instructions_per_mfma_by_shape"] = {k: round(v["non_mfma_per_mfma"], 2) for k, v in crec["shapes"].items() if "non_mfma_per_mfma" in v}
                result["roofline"]["mfma_busy_source"] = "profiles/r06_conv_counters.json (rocprofv3 --pmc SQ passes over tools/conv_one.py at the per-part batch, isolated launches, these kernel sources)"
            else:
                result["roofline"]["mfma_busy_source"] = "profiles/r06_conv_counters.json was measured on other kernel sources (hash mismatch): not reported"
        # the other kernel classes of the UNet step against the roof that bounds them (same profiling pass)
        other = []
        for name, cms, cn, cfl, cby in classes:
            if name in ("other", "conv_gemm") or cn == 0 or cms <= 0:
                continue
            gbs = cby / (cms * 1e-3) / 1e9
            other.append({"kernel": name, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": gbs / HBM_PEAK_GBS, "launches": cn, "avg_launch_us": 1000.0 * cms / cn,
                          "share_of_profiled_ms": cms / max(1e-9, sum(c[1] for c in classes))})
        result["roofline"]["other_kernels"] = other
    if rank == 0 and not args.no_roofline:
        # timed-mode evidence (graph replay, batch parts on their own streams): device-side stamps of every step of
        # every part during one more decode of the same batch -- rocprofv3's kernel trace serialises the streams, so it
        # cannot show the overlap the timed region runs with
        eng.timeline_enable(True)
        eng.decode(wav, N, noise=None, per_item=True)     # re-captures the step graphs with the stamp pointer
        torch.cuda.synchronize(dev)
        ck0 = eng.clock_sample()                          # (100 MHz wall ticks, shader cycles) around one decode: the shader clock it runs at
        t1 = time.perf_counter()
        eng.decode(wav, N, noise=None, per_item=True)
        torch.cuda.synchronize(dev)
        wall_ms = 1000.0 * (time.perf_counter() - t1)
        ck1 = eng.clock_sample()
        if ck1[0] > ck0[0]:
            result["host"]["shader_mhz_during_decode"] = round(100.0 * (ck1[1] - ck0[1]) / (ck1[0] - ck0[0]), 1)
        tl = eng.timeline(N, nparts)
        eng.timeline_enable(False)
        span = max(float(a[:, 1].max()) for a in tl) - min(float(a[:, 0].min()) for a in tl)
        busy = [float((a[:, 1] - a[:, 0]).sum()) for a in tl]
        both = 0.0
        if nparts >= 2:
            ev = sorted([(float(b), 1) for a in tl for b in a[:, 0]] + [(float(e), -1) for a in tl for e in a[:, 1]])
            live, last = 0, ev[0][0]
            for tm, d in ev:
                if live == nparts:
                    both += tm - last
                live += d
                last = tm
        result["roofline"]["timeline"] = {
            "source": "device stamps (100 MHz clock) at the first and last kernel of every denoise step of every batch part, "
                      "graph-replayed multi-stream decode", "parts": nparts,
            "decode_wall_ms": wall_ms, "denoise_span_ms": span / 1e3, "part_busy_ms": [b / 1e3 for b in busy],
            "all_parts_in_a_step_ms": both / 1e3, "overlap_factor": sum(busy) / max(span, 1e-9),
            "mean_step_ms_per_part": [float((a[:, 1] - a[:, 0]).mean()) / 1e3 for a in tl]}
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "timeline_steps.csv"), "w") as f:
            f.write("part,step,begin_us,end_us\n")
            for k, a in enumerate(tl):
                for j in range(a.shape[0]):
                    f.write(f"{k},{j},{a[j, 0]:.2f},{a[j, 1]:.2f}\n")
    if rank == 0 and world == 1 and n_fl == 1 and not args.no_pipelined and "split=" not in os.environ.get("LDC_OPTIONS", ""):
        # Supplementary (NOT `value`): the same K steps with TWO batches of the config's size in flight -- two more engines on
        # two streams, every batch decoded as ONE chain (kernels of 32 items instead of 16), steps dealt round-robin, all K
        # finished inside the timed region.  Per-batch latency doubles; throughput and per-launch efficiency rise.
        try:
            eng2, st2 = [], []
            for k in range(2):
                e_k = Engine(mc, u, cc, dtype=args.dtype, device=local_rank, noise_seed=8765 + k)
                e_k.set_option("split", 1)
                e_k.load_state_dict(L.MODEL_MAIN, sd_main)
                e_k.load_state_dict(L.MODEL_COND, sd_cond)
                e_k.finalize(strict=True)
                eng2.append(e_k)
                st2.append(torch.cuda.Stream(device=dev))
                engines.append(e_k)
            for k in range(2):
                with torch.cuda.stream(st2[k]):
                    eng2[k].decode(wav, N, noise=None, per_item=True)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for j in range(args.steps):
                with torch.cuda.stream(st2[j % 2]):
                    out2 = eng2[j % 2].decode(wav, N, noise=None, per_item=True)
            torch.cuda.synchronize(dev)
            el2 = time.perf_counter() - t1
            assert bool(torch.isfinite(out2).all()), "non-finite output"
            pip = {"batches_in_flight": 2, "chains_per_batch": 1, "value": B * (T / 16000.0) * args.steps / el2, "unit": "audio-s/wall-s",
                   "ms_per_step": 1000.0 * el2 / args.steps, "steps": args.steps,
                   "note": "same workload and K; two engines on two streams, each batch of the config's size decoded as one chain"}
            if not args.no_roofline:
                eng2[0].profile(True)
                eng2[0].decode(wav, N, noise=None, per_item=True)
                ms2, launches2, flops2 = eng2[0].profile_read()
                eng2[0].profile(False)
                ach2 = flops2 / (ms2 * 1e-3) / 1e12 if ms2 > 0 else 0.0
                pip["roofline"] = {"bound": "mfma", "kernel": "conv_lean_kernel / conv_fast_kernel / conv_gemm_kernel", "achieved": ach2, "peak": MFMA_PEAK_TFLOPS[args.dtype],
                                   "unit": "TFLOP/s", "frac": ach2 / MFMA_PEAK_TFLOPS[args.dtype], "launches": launches2,
                                   "avg_launch_us": 1000.0 * ms2 / max(1, launches2), "traffic": None}
            result["pipelined"] = pip
            log(f"pipelined (2 batches in flight): {pip['value']:.1f} audio-s/s, {pip['ms_per_step']:.1f} ms per batch")
        finally:
            pass
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(cc, mc, u, sd_cond, sd_main, N, args.seconds, args.cpu_batch)
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    os.close(result_fd)
    for e_k in engines:
        e_k.close()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
