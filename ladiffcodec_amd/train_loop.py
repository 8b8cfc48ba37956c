"""`python -m srcs.train --run_diff --freeze_ed ...`: the diffusion-training loop of the reference (srcs/train.py:227-417 around
`run_model` :110-177) on the MI355X training row -- Dataset_Libri-shaped walker -> DiffusionTrainer (frozen encoders of the next batch
prefetched on a second engine) -> validation pass with the reference's monitoring losses -> `model_best.amlt` by the last monitored
value (neg_loss) and `model_<step>.amlt` every 100 outer steps.  SURVEY.md section 8(f) row 2: "replaces the dead DDP scaffold at
train.py:298-377" -- one process per GPU, RANK / WORLD_SIZE from the environment, every rank walks its DistributedSampler share and the
flat gradient buffer is reduce-scattered / all-gathered inside `DiffusionTrainer.step` (parallel.allreduce_gradients).

Only the mode the north star's training row names is implemented and everything else is refused loudly: --run_diff with the encoder /
decoder frozen (the reference's `optim.Adam(model.diffusion.parameters())` branch, train.py:361-365), the l1 objective, no
discriminator, no EMA, the 1-D UNet.  The main model's frozen parts come from --finetune_model + '.amlt' (train.py:345-347; the
reference can also start them from random initialisation, which has no use with a frozen codec), the condition model from
--model_for_cond + '/model_best.amlt' (train.py:356).  Differences that are deliberate: --num_steps (the reference hard-codes 50 000
outer steps), --max_files, a one-line log per evaluation on stdout instead of the reference's log file, and the default of --model_type
('unet' here; the reference's parser defaults to 'transformer', a backbone this repo does not build -- it is refused, like every other flag
that would change the computation without being implemented: --unet_scale_x, a missing --cond_quantization)."""
import argparse
import time
from typing import List, Optional

import numpy as np

REFERENCE_FLAGS = [
    ("--output_dir", dict(type=str, default="saved_models")),
    ("--data_folder_path", dict(type=str, default="/data/hy17/librispeech/librispeech")),
    ("--seq_len_p_sec", dict(type=float, default=1.0)),
    ("--sample_rate", dict(type=int, default=16000)),
    ("--debug", dict(dest="debug", action="store_true")),
    ("--lr", dict(type=float, default=5e-4)),
    ("--batch_size", dict(type=int, default=5)),
    ("--exp_name", dict(type=str, default="")),
    ("--finetune_model", dict(type=str, default="")),
    ("--write_on_every", dict(type=int, default=50)),       # parsed and, as in the reference, overridden: 5, or 1 with --debug (train.py:379)
    ("--model_type", dict(type=str, default="unet")),
    ("--freeze_ed", dict(dest="freeze_ed", action="store_true")),
    ("--train_time_diff", dict(dest="train_time_diff", action="store_true")),
    ("--rep_dims", dict(type=int, default=128)),
    ("--emb_dims", dict(type=int, default=128)),
    ("--quantization", dict(dest="quantization", action="store_true")),
    ("--bandwidth", dict(type=float, default=3.0)),
    ("--n_filters", dict(type=int, default=32)),
    ("--lstm", dict(type=int, default=2)),
    ("--n_residual_layers", dict(type=int, default=1)),
    ("--enc_ratios", dict(nargs="+", type=int)),
    ("--final_activation", dict(type=str, default=None)),
    ("--diff_dims", dict(type=int, default=128)),
    ("--qtz_condition", dict(dest="qtz_condition", action="store_true")),
    ("--self_condition", dict(dest="self_condition", action="store_true")),
    ("--seq_length", dict(type=int, default=800)),
    ("--run_diff", dict(dest="run_diff", action="store_true")),
    ("--run_vae", dict(dest="run_vae", action="store_true")),
    ("--scaling_frame", dict(dest="scaling_frame", action="store_true")),
    ("--scaling_feature", dict(dest="scaling_feature", action="store_true")),
    ("--scaling_global", dict(dest="scaling_global", action="store_true")),
    ("--scaling_dim", dict(dest="scaling_dim", action="store_true")),
    ("--use_film", dict(dest="use_film", action="store_true")),
    ("--unet_scale_cond", dict(dest="unet_scale_cond", action="store_true")),
    ("--unet_scale_x", dict(dest="unet_scale_x", action="store_true")),
    ("--model_for_cond", dict(type=str, default="")),
    ("--cond_enc_ratios", dict(nargs="+", type=int)),
    ("--upsampling_ratios", dict(nargs="+", type=int)),
    ("--cond_quantization", dict(dest="cond_quantization", action="store_true")),
    ("--cond_bandwidth", dict(type=float, default=3.0)),
    ("--cond_global", dict(type=float, default=3.0)),
    ("--use_disc", dict(dest="use_disc", action="store_true")),
    ("--disc_freq", dict(type=int, default=1)),
]


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="LaDiffCodec diffusion training on MI355X (reference flags of srcs/train.py:229-289)")
    for flag, kw in REFERENCE_FLAGS:
        p.add_argument(flag, **kw)
    p.add_argument("--num_steps", type=int, default=50000, help="outer steps (epochs over the training files); the reference hard-codes 50000")
    p.add_argument("--max_files", type=int, default=10000, help="dataset_libri.py:36 keeps the first 10000 files of the glob")
    return p


def _unsupported(a) -> None:
    bad = []
    if not a.run_diff: bad.append("only --run_diff is implemented (the codec / autoencoder training modes are out of scope)")
    if not a.freeze_ed: bad.append("--freeze_ed is required: only model.diffusion's parameters are optimised (train.py:361-365)")
    if a.use_disc: bad.append("--use_disc (GAN loss) is out of scope")
    if a.train_time_diff: bad.append("--train_time_diff (DiffAudioTime) is out of scope")
    if a.model_type != "unet": bad.append(f"--model_type {a.model_type}: only the 1-D UNet of DiffAudioRep is implemented (the reference builds "
                                          "TransformerDDPM for 'transformer', its own default; here the default is 'unet')")
    if a.unet_scale_x: bad.append("--unet_scale_x is implemented on the decode path only (the training step has no per-item max scaling of cat(cond, x))")
    if not a.cond_quantization: bad.append("--cond_quantization is required: the condition codec of the training row is the quantised one "
                                           "(BASELINE configs[3]; without it the reference conditions on unquantised latents, train.py:355)")
    from .lib import FINAL_ACTIVATIONS
    if a.final_activation not in FINAL_ACTIVATIONS: bad.append(f"--final_activation {a.final_activation}: supported are {sorted(k for k in FINAL_ACTIVATIONS if k)}")
    if a.self_condition or a.qtz_condition: bad.append("--self_condition / --qtz_condition are not implemented")
    if a.run_vae or a.use_film or a.scaling_frame or a.scaling_feature or a.scaling_dim:
        bad.append("only --scaling_global is implemented among the scaling / VAE / FiLM options")
    if not a.scaling_global: bad.append("--scaling_global is required (the latents are divided by 18, model.py:165)")
    if not a.model_for_cond: bad.append("--model_for_cond is required (the condition codec of the decode path)")
    if not a.finetune_model: bad.append("--finetune_model is required: the frozen encoder / decoder of the main model come from it")
    if a.quantization: bad.append("--quantization on the main model is not the LaDiffCodec configuration")
    if bad:
        raise SystemExit("ladiffcodec_amd.train_loop: " + "; ".join(bad))


def sampler_indices(n: int, rank: int, world: int, epoch: int = 0, seed: int = 0) -> List[int]:
    """torch.utils.data.DistributedSampler(dataset, shuffle=True) after set_epoch(epoch): randperm(n) from a generator seeded
    seed + epoch, padded by wrapping to a multiple of `world`, rank takes every world-th index (train.py:327, 386).  world == 1 is the
    reference's plain DataLoader: sequential."""
    import torch
    if world <= 1:
        return list(range(n))
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    idx = torch.randperm(n, generator=g).tolist()
    total = -(-n // world) * world
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * (-(-pad // max(len(idx), 1))))[:pad]
    return idx[rank:total:world]


def run(a, log=print) -> dict:
    """-> {'best_loss', 'saved': [paths], 'history': [(step, train losses, validation losses)]}"""
    import torch
    from . import checkpoint, lib as L, parallel
    from .dataset import BatchWalker, DatasetLibri
    from .model import Engine
    from .spec import CodecConfig, UnetConfig
    from .train import DiffusionTrainer
    _unsupported(a)
    rank, local_rank, world = parallel.init_process_group("nccl")
    torch.manual_seed(rank)                                    # train.py:332 (crop positions and t / noise draws differ per rank)
    enc_ratios = tuple(a.enc_ratios) if a.enc_ratios else (8, 5, 4, 2)
    mc = CodecConfig(rep_dims=a.rep_dims, n_filters=a.n_filters, n_residual_layers=a.n_residual_layers, lstm=a.lstm, enc_ratios=enc_ratios,
                     quantization=False, final_activation=a.final_activation)                  # train.py:340-343
    cc = CodecConfig(rep_dims=a.rep_dims, n_filters=a.n_filters, n_residual_layers=a.n_residual_layers, lstm=a.lstm,
                     enc_ratios=(8, 5, 4, 2), quantization=bool(a.cond_quantization), bandwidth=a.cond_bandwidth,
                     final_activation=a.final_activation)   # train.py:355; `ratios=` is ignored by the reference
    u = UnetConfig(dim=a.diff_dims, upsampling_ratios=tuple(a.upsampling_ratios) if a.upsampling_ratios else None,
                   unet_scale_cond=a.unet_scale_cond, unet_scale_x=a.unet_scale_x)
    sd_main = checkpoint.read_amlt(a.finetune_model + ".amlt")
    sd_cond = checkpoint.read_amlt(a.model_for_cond + "/model_best.amlt")

    def make_engine(seed):
        e = Engine(mc, u, cc, dtype="f32", device=local_rank, noise_seed=seed)
        e.load_state_dict(L.MODEL_MAIN, {k: v for k, v in sd_main.items() if not k.startswith("diffusion.model.")})
        e.load_state_dict(L.MODEL_COND, sd_cond)
        e.finalize(strict=True)
        return e

    eng, front = make_engine(1000 + rank), make_engine(2000 + rank)
    sd = {k[len("diff_model."):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_main.items() if k.startswith("diff_model.")}
    trainer = DiffusionTrainer(eng, sd, dim=u.dim, dim_mults=u.dim_mults, lr=a.lr, frontend=front, upsampling_ratios=u.upsampling_ratios,
                               unet_scale_cond=u.unet_scale_cond)
    train_ds = DatasetLibri("train", a.seq_len_p_sec, a.data_folder_path, a.max_files)
    valid_ds = DatasetLibri("valid", a.seq_len_p_sec, a.data_folder_path, a.max_files)
    if len(train_ds) == 0 or len(valid_ds) == 0:
        raise SystemExit(f"no wav files under {a.data_folder_path}/train-clean-100 or /dev-clean")
    dev = torch.device("cuda", local_rank)
    write_on_every = 1 if a.debug else 5                       # train.py:379
    best, saved, history = float("inf"), [], []

    def epoch(ds, indices, update: bool):
        tot, n = {}, 0
        walker = BatchWalker(ds, a.batch_size, device=dev, indices=indices)
        for wav in walker:
            rep = trainer.step_from_wav(wav, monitor=True, next_wav=walker.peek(), update=update)
            for key in ("diff_loss", "neg_loss"):                # summed on the device: a .cpu() per step would stall the host behind every step
                v = rep[key].reshape(-1)[0].detach()
                tot[key] = v.clone() if key not in tot else tot[key] + v
            n += 1
            if a.debug:
                break                                          # train.py:170-171
        out = {k: float(v.cpu()) / max(n, 1) for k, v in tot.items()}
        if world > 1:                                          # the reference never averaged over ranks (its DDP path was dead)
            vals = torch.tensor([out["diff_loss"], out["neg_loss"]], device=dev)
            torch.distributed.all_reduce(vals)
            out = {"diff_loss": float(vals[0]) / world, "neg_loss": float(vals[1]) / world}
        return out

    for step in range(a.num_steps):
        t0 = time.time()
        tr = epoch(train_ds, sampler_indices(len(train_ds), rank, world, epoch=step), update=True)
        if step % write_on_every != 0:
            continue
        va = epoch(valid_ds, sampler_indices(len(valid_ds), rank, world, epoch=0), update=False)
        vall = va["neg_loss"]                                  # "list(val_losses.values())[-1] # negsdr", train.py:401
        history.append((step, tr, va))
        if not a.debug:
            if vall < best:
                best = vall
                if rank == 0:
                    saved.append(checkpoint.save_checkpoints(checkpoint.merged_state_dict(sd_main, trainer.state_dict()), a.output_dir, a.exp_name, "best"))
            if step % 100 == 0 and step > 0 and rank == 0:
                saved.append(checkpoint.save_checkpoints(checkpoint.merged_state_dict(sd_main, trainer.state_dict()), a.output_dir, a.exp_name, str(step)))
        if rank == 0:
            log(f"step {step}: train diff_loss {tr['diff_loss']:.5f} neg_loss {tr['neg_loss']:.3f} | valid diff_loss {va['diff_loss']:.5f} "
                f"neg_loss {va['neg_loss']:.3f} | best {best:.3f} | {time.time() - t0:.1f} s")
    return {"best_loss": best, "saved": saved, "history": history, "trainer": trainer}


def main(argv: Optional[List[str]] = None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
