"""Worker of tests/test_gpu_multi.py: one process per GPU over RCCL (`python -m torch.distributed.run --nproc-per-node N tests/dist_gpu_worker.py DIR`).

What north_star's data-parallel split needs from more than one rank, on the hardware (the reference's only
distributed code is the dead DDP scaffold of srcs/train.py:302-320,374-377):
  1. checkpoint broadcast: rank 0 builds the (synthetic) state dicts, every rank receives them as ONE flat RCCL broadcast
     (parallel.broadcast_state_dict) and loads its engine from them; checksums agree on every rank;
  2. utterance sharding + gather: every rank decodes its contiguous shard, `parallel.gather_results` brings the waveforms to every
     rank; rank 0 compares them with its own decode of the whole list;
  3. file sharding through the CLI: `sample.synthesis` over a directory of wav files writes every file exactly once across the ranks
     (configs[2]'s shape: a corpus sharded over the GPUs of a node);
  4. one data-parallel training step: every rank runs the step on its half of the batch, the flat gradient goes through the padded
     reduce_scatter + all_gather (parallel.allreduce_gradients), and the result equals the single-process gradient of the whole batch.
Runs at world size 1 as well (the collectives then move nothing), which is what a one-GPU box can check of this file."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import CASES, COND_CFG, T, cond_sd_np, load_golden, main_sd_np  # noqa: E402
from ladiffcodec_amd import lib as L, parallel, spec, synth  # noqa: E402
from ladiffcodec_amd.model import Engine  # noqa: E402


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def main():
    tmp = sys.argv[1]
    rank, local_rank, world = parallel.init_process_group("nccl")
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    mc, u, _ = CASES["r84"]

    # 1. one flat broadcast per model; only rank 0 holds the weights beforehand
    sd_main = main_sd_np("r84") if rank == 0 else None
    sd_cond = cond_sd_np() if rank == 0 else None
    lay_main = [(k, tuple(v.shape)) for k, v in main_sd_np("r84").items()]       # (the layout is config-derived: every rank can build it)
    lay_cond = [(k, tuple(v.shape)) for k, v in cond_sd_np().items()]
    got_main = parallel.broadcast_state_dict(sd_main, lay_main, device=dev)
    got_cond = parallel.broadcast_state_dict(sd_cond, lay_cond, device=dev)
    cs = torch.tensor([sum(float(np.abs(v).sum()) for v in got_main.values()) + sum(float(np.abs(v).sum()) for v in got_cond.values())],
                      dtype=torch.float64, device=dev)
    all_cs = parallel.gather_results(cs, world)
    assert all(float(c) == float(all_cs[0]) for c in all_cs), [float(c) for c in all_cs]
    eng = Engine(mc, u, COND_CFG, dtype="f32", device=local_rank)
    eng.load_state_dict(L.MODEL_MAIN, got_main)
    eng.load_state_dict(L.MODEL_COND, got_cond)
    eng.finalize(strict=True)

    # 2. contiguous utterance shards, injected noise (so that the result does not depend on who decodes what), gather
    n_utt, Tn, n_steps = 2 * max(world, 2) + 1, 5120, 3              # (an odd count: the shards are uneven)
    wav = torch.from_numpy(synth.synthetic_wav(n_utt, Tn, seed=77)) * 0.5
    noise = torch.randn(n_steps, n_utt, 128, Tn // mc.hop_length, generator=torch.Generator().manual_seed(5))
    lo, hi = parallel.shard_range(n_utt, rank, world)
    per = -(-n_utt // world)
    mine = torch.zeros(per, 1, Tn, device=dev)
    if hi > lo:
        mine[:hi - lo] = eng.decode(wav[lo:hi].to(dev), n_steps, noise[:, lo:hi].contiguous().to(dev), per_item=True)
    parts = parallel.gather_results(mine, world)
    if rank == 0:
        whole = eng.decode(wav.to(dev), n_steps, noise.to(dev), per_item=True).cpu().numpy()
        for r in range(world):
            a, b = parallel.shard_range(n_utt, r, world)
            if b > a:
                err = rel(parts[r][:b - a].cpu().numpy(), whole[a:b])
                assert err < 1e-4, ("gathered shard", r, err)

    # 3. the CLI over a directory: every file written exactly once, by the rank that owns it
    from scipy.io import wavfile
    from ladiffcodec_amd import sample as cli
    ind, outd = os.path.join(tmp, "in"), os.path.join(tmp, "out")
    if rank == 0:
        synth.save_amlt(main_sd_np("r84"), os.path.join(tmp, "ladiff.amlt"), ddp_prefix=True)
        synth.save_amlt(cond_sd_np(), os.path.join(tmp, "codec.amlt"))
        os.makedirs(os.path.join(ind, "spk"), exist_ok=True)
        for i in range(5):
            wavfile.write(os.path.join(ind, "spk" if i % 2 else "", f"u{i}.wav"), 16000, (synth.synthetic_wav(1, 2560 * (1 + i % 3), seed=10 + i)[0, 0] * 0.5).astype(np.float32))
    dist.barrier()
    args = cli.build_parser().parse_args([
        "--model_for_cond", os.path.join(tmp, "codec.amlt"), "--model_path", os.path.join(tmp, "ladiff.amlt"), "--run_diff",
        "--scaling_global", "--cond_bandwidth", "3", "--unet_scale_cond", "--enc_ratios", "8", "4", "--upsampling_ratios", "5", "2",
        "--diff_dims", "32", "--input_dir", ind + "/", "--output_dir", outd + "/", "--midway_t", "2", "--dtype", "f32"])
    written = cli.synthesis(args)
    counts = parallel.gather_results(torch.tensor([len(written)], device=dev), world)
    dist.barrier()
    if rank == 0:
        assert sum(int(c) for c in counts) == 5, [int(c) for c in counts]
        found = sorted(os.path.relpath(os.path.join(d, f), outd) for d, _, fs in os.walk(outd) for f in fs if f.endswith(".wav"))
        assert found == sorted(["u0.wav", "spk/u1.wav", "u2.wav", "spk/u3.wav", "u4.wav"]), found

    # 4. data-parallel training step: flat gradient through reduce_scatter + all_gather == the whole batch on one rank
    from ladiffcodec_amd import train as TR
    g = load_golden("train_unet")
    sd = {k[2:]: T(g[k]) for k in list(g.keys()) if k.startswith("p.")}
    gen = torch.Generator().manual_seed(31)
    nb = 2 * world
    x0 = torch.randn(nb, 8, 32, generator=gen).clamp(-1, 1)
    cond = torch.randn(nb, 8, 32, generator=gen)
    t = torch.randint(0, 1000, (nb,), generator=gen)
    nz = torch.randn(nb, 8, 32, generator=gen)
    tr = TR.DiffusionTrainer(eng, {k: v.clone() for k, v in sd.items()}, dim=16, dim_mults=(1, 2), lr=2e-3)
    s = slice(2 * rank, 2 * rank + 2)
    tr.step(x0[s], cond[s], t[s], nz[s])
    g_dp = tr.flat_g.clone()
    if rank == 0:
        keep = parallel.allreduce_gradients
        parallel.allreduce_gradients = lambda flat, **kw: flat          # the reference run is rank 0's alone: no collective
        try:
            ref = TR.DiffusionTrainer(eng, {k: v.clone() for k, v in sd.items()}, dim=16, dim_mults=(1, 2), lr=2e-3)
            ref.step(x0, cond, t, nz)
        finally:
            parallel.allreduce_gradients = keep
        # (an l1 objective: a 1e-6 difference in `pred` can flip one sign(pred - noise) / N -- the bar of tests/test_gpu_train.py)
        err = rel(g_dp.cpu().numpy(), ref.flat_g.cpu().numpy())
        assert err < 4e-3, ("data-parallel gradient", err)
    dist.barrier()
    ms = parallel.max_over_ranks(1.0 + rank, device=dev)
    assert ms == float(world)
    dist.destroy_process_group()
    if rank == 0:
        print(f"DIST_GPU_OK world={world}")


if __name__ == "__main__":
    main()
