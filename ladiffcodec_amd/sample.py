"""`python -m srcs.sample` -- the synthesis CLI of the reference (srcs/sample.py:50-203) on the MI355X engine.

Same flags, defaults and output naming as the reference.  What differs, by design:
  * mono files are decoded in BATCHES of equal trimmed length (the reference walks them one by one,
    sample.py:73); every utterance is normalised on its own, which is what the reference computes for
    a mono file (its "batch" is the channel axis of one file, sample.py:85,129,133-134).  A
    multi-channel file is decoded as ONE batch of its channels with the reference's joint
    (whole-tensor) normalisation, so its channels keep their relative levels exactly as in the reference;
  * `midway_t` (a literal 100 at sample.py:69) is a flag, `--midway_t`, default 100;
  * `--seed` (default 0) seeds the device noise stream (rank r uses seed + r); every decode call draws fresh noise,
    as torch.randn_like does in the reference (ddpm_loss.py:249);
  * under `torch.distributed.run` the FILE list is sharded over the ranks (one process per GPU; all channels of a
    file stay on one rank, every output file has exactly one writer).
Flags that are inert in the reference stay accepted and inert (`--sampling_timesteps`,
`--cond_enc_ratios`: quirk Q1, the cond codec is always built with ratios [8,5,4,2]).
"""
from __future__ import annotations

import argparse
import glob
import os
from typing import Dict, List, Tuple

import numpy as np

# (flag, kwargs) in the reference's order -- srcs/sample.py:141-198
_FLAGS: List[Tuple[str, dict]] = [
    ("--data_folder_path", dict(type=str, default="/data/hy17/librispeech/librispeech")),
    ("--n_spks", dict(type=int, default=500)),
    ("--seq_len_in_sec", dict(type=float, default=1.8)),
    ("--sample_rate", dict(type=int, default=16000)),
    ("--model_path", dict(type=str, default="")),
    ("--qtzer_path", dict(type=str, default="")),
    ("--note", dict(type=str, default="")),
    ("--rep_dims", dict(type=int, default=128)),
    ("--emb_dims", dict(type=int, default=128)),
    ("--quantization", dict(dest="quantization", action="store_true")),
    ("--bandwidth", dict(type=float, default=3.0)),
    ("--n_filters", dict(type=int, default=32)),
    ("--lstm", dict(type=int, default=2)),
    ("--n_residual_layers", dict(type=int, default=1)),
    ("--enc_ratios", dict(nargs="+", type=int, default=[8])),
    ("--final_activation", dict(type=str, default=None)),
    ("--run_diff", dict(dest="run_diff", action="store_true")),
    ("--run_vae", dict(dest="run_vae", action="store_true")),
    ("--train_time_diff", dict(dest="train_time_diff", action="store_true")),
    ("--diff_dims", dict(type=int, default=256)),
    ("--qtz_condition", dict(dest="qtz_condition", action="store_true")),
    ("--self_condition", dict(dest="self_condition", action="store_true")),
    ("--seq_length", dict(type=int, default=16000)),
    ("--model_type", dict(type=str, default="unet")),
    ("--scaling_frame", dict(dest="scaling_frame", action="store_true")),
    ("--scaling_feature", dict(dest="scaling_feature", action="store_true")),
    ("--scaling_global", dict(dest="scaling_global", action="store_true")),
    ("--scaling_dim", dict(dest="scaling_dim", action="store_true")),
    ("--sampling_timesteps", dict(type=int, default=1000)),
    ("--use_film", dict(dest="use_film", action="store_true")),
    ("--model_for_cond", dict(type=str, default="")),
    ("--upsampling_ratios", dict(nargs="+", type=int, default=[5, 4, 2])),
    ("--cond_enc_ratios", dict(nargs="+", type=int, default=[8, 5, 4, 2])),
    ("--cond_bandwidth", dict(type=float, default=3.0)),
    ("--cond_global", dict(type=float, default=3.0)),
    ("--unet_scale_cond", dict(dest="unet_scale_cond", action="store_true")),
    ("--unet_scale_x", dict(dest="unet_scale_x", action="store_true")),
    ("--input_dir", dict(type=str, default="")),
    ("--output_dir", dict(type=str, default="outputs/")),
]
# additions of this implementation
_EXTRA: List[Tuple[str, dict]] = [
    ("--midway_t", dict(type=int, default=100, help="reverse-diffusion steps (literal 100 in the reference)")),
    ("--dtype", dict(type=str, default="bf16", choices=["bf16", "f32", "fp8"], help="UNet compute dtype on the GPU (fp8: e4m3 conv weights, bf16 math)")),
    ("--batch_size", dict(type=int, default=32, help="utterances decoded per engine call")),
    ("--seed", dict(type=int, default=0, help="seed of the device noise stream (rank r uses seed + r)")),
    ("--in_flight", dict(type=int, default=2, help="engine calls kept in flight when a run has several batches: n engines on n "
                                                    "streams, batches dealt round-robin, every batch decoded as one chain "
                                                    "(+19 %% throughput at 32 x 2.4 s on MI355X); 1 = one batch at a time")),
    ("--chunk_sec", dict(type=float, default=0.0, help="long-form mode (BASELINE config 5): mono recordings longer than this are "
                                                        "decoded as chunks of this length batched together, the chunks' raw decoder "
                                                        "outputs are joined and normalised over the whole recording; 0 = whole files")),
]


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Encodec_baseline")
    for flag, kw in _FLAGS + _EXTRA:
        p.add_argument(flag, **kw)
    return p


def _unsupported(a) -> None:
    bad = [n for n in ("train_time_diff", "self_condition", "qtz_condition", "use_film", "run_vae") if getattr(a, n)]
    if bad:
        raise SystemExit(f"flags {bad} select paths outside the decode path this implementation covers (SURVEY.md section 8)")
    if a.model_type != "unet":
        raise SystemExit("only --model_type unet is supported")
    if not a.model_for_cond:
        raise SystemExit("--model_for_cond is required: halfway sampling starts from the quantised condition (sample.py:125-130)")
    from .lib import FINAL_ACTIVATIONS
    if a.final_activation not in FINAL_ACTIVATIONS:
        raise SystemExit(f"--final_activation {a.final_activation}: supported are {sorted(k for k in FINAL_ACTIVATIONS if k)}")


def read_wav(path: str):
    """torchaudio.load (sample.py:83): -> (float32 [channels, T] in [-1, 1), sample rate)."""
    from scipy.io import wavfile
    sr, x = wavfile.read(path)
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
    elif x.dtype.kind == "u":
        x = (x.astype(np.float32) - 128.0) / 128.0
    x = x.astype(np.float32)
    x = x[None, :] if x.ndim == 1 else x.T
    return np.ascontiguousarray(x), int(sr)


def read_wav_16k(path: str, eng=None) -> np.ndarray:
    """-> float32 [channels, T] at 16 kHz: torchaudio.load + torchaudio.functional.resample(wav, sr, 16000) of the reference
    (sample.py:83-84); the resampling runs on the GPU (ldc_resample, the same windowed-sinc filter bank)."""
    x, sr = read_wav(path)
    if sr != 16000:
        import torch
        if eng is None:
            raise RuntimeError(f"{path}: {sr} Hz input needs the engine's resampler")
        x = eng.resample(torch.from_numpy(x), sr, 16000).cpu().numpy()
    return x


def wav_header(path: str) -> Tuple[int, int, int]:
    """(channels, samples at 16 kHz, sample rate) from the RIFF header alone: the 'fmt ' and 'data' chunk headers are parsed
    directly, no sample is touched and every PCM / float container size works (scipy's memory-mapped read refuses the 3-byte
    samples of 24-bit files: one such file used to abort the whole run on every rank).  The length after resampling is what
    the resampler itself will produce (ldc_resample_out_len = torchaudio's ceil(16000 * T / sr)), so the batch plan and the
    data agree."""
    import struct
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] not in (b"RIFF", b"RF64") or head[8:12] != b"WAVE":
            raise ValueError(f"{path}: not a RIFF/WAVE file")
        ch = sr = block = None
        n_bytes = None
        while True:
            hdr = f.read(8)
            if len(hdr) < 8:
                break
            cid, size = hdr[:4], struct.unpack("<I", hdr[4:])[0]
            if cid == b"fmt ":
                fmt = f.read(size + (size & 1))
                if len(fmt) < 16:
                    raise ValueError(f"{path}: 'fmt ' chunk of {len(fmt)} bytes (16 needed)")
                tag, ch, sr, _, block, _ = struct.unpack("<HHIIHH", fmt[:16])
                if tag == 0xFFFE and len(fmt) >= 26:            # WAVE_FORMAT_EXTENSIBLE: the real tag opens the sub-format GUID
                    tag = struct.unpack("<H", fmt[24:26])[0]
                if tag not in (1, 3):                            # PCM / IEEE float: the only containers whose block count is a sample count
                    raise ValueError(f"{path}: WAVE format tag {tag:#x} is not PCM or IEEE float")
            elif cid == b"data":
                here = f.tell()
                f.seek(0, 2)
                n_bytes = min(size, f.tell() - here) if size not in (0, 0xFFFFFFFF) else f.tell() - here   # streamed files leave the size open
                break
            else:
                f.seek(size + (size & 1), 1)
    if not ch or not sr or not block or n_bytes is None:
        raise ValueError(f"{path}: no 'fmt ' / 'data' chunk")
    n = n_bytes // block
    return int(ch), resample_out_len(n, int(sr), 16000), int(sr)


def resample_out_len(n: int, sr_in: int, sr_out: int) -> int:
    """Samples torchaudio.functional.resample returns for n input samples: ceil(new * n / orig) on the gcd-reduced rates
    (the arithmetic of ldc_resample_out_len, restated in Python so that header planning needs no built library;
    tests/test_cli_and_parallel_cpu.py holds the two to each other)."""
    import math
    g = math.gcd(int(sr_in), int(sr_out))
    o, nw = int(sr_in) // g, int(sr_out) // g
    return int(-(-nw * int(n) // o))


class LazyWavs:
    """The run's recordings at 16 kHz, loaded (and resampled on the GPU) on first use: a rank touches only the files of its own
    shard (through round 2 every rank read and resampled the whole corpus before sharding it: world-times redundant I/O and
    O(corpus) host memory per rank).  `shapes[i]` = (channels, samples) from the header."""

    def __init__(self, files: List[str], eng):
        self.files, self.eng = files, eng
        self.shapes = [wav_header(f)[:2] for f in files]
        self._cache: Dict[int, np.ndarray] = {}

    def subset(self, idx: List[int]) -> "LazyWavs":
        out = LazyWavs.__new__(LazyWavs)
        out.files, out.eng = [self.files[i] for i in idx], self.eng
        out.shapes = [self.shapes[i] for i in idx]
        out._cache = {}
        return out

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i: int) -> np.ndarray:
        if i not in self._cache:
            w = read_wav_16k(self.files[i], self.eng if hasattr(self.eng, "resample") else None)
            if w.shape[-1] != self.shapes[i][1]:               # a resampler that rounds differently: trust the data
                self.shapes[i] = (w.shape[0], w.shape[-1])
            self._cache[i] = w
        return self._cache[i]

    def drop(self, i: int) -> None:
        self._cache.pop(i, None)


def output_path(wav_file: str, input_dir: str, output_dir: str) -> str:
    """sample.py:75-81,136: save_path = output_dir + wav_file[len(input_dir):][:-4]; file = save_path + '.wav'."""
    local_path = wav_file[len(input_dir):][:-4]
    save_path = output_dir + local_path
    return os.path.join(output_dir, f"{save_path}.wav")


def synthesis(inp_args) -> List[str]:
    import torch
    from scipy.io import wavfile

    from . import checkpoint, lib as L, parallel
    from .model import Engine
    from .spec import CodecConfig, UnetConfig

    _unsupported(inp_args)
    rank, local_rank, world = parallel.init_process_group("nccl")
    main_codec = CodecConfig(rep_dims=inp_args.rep_dims, n_filters=inp_args.n_filters,
                             n_residual_layers=inp_args.n_residual_layers, lstm=inp_args.lstm,
                             enc_ratios=tuple(inp_args.enc_ratios), quantization=False,
                             final_activation=inp_args.final_activation)
    cond_codec = CodecConfig(rep_dims=inp_args.rep_dims, n_filters=inp_args.n_filters,
                             n_residual_layers=inp_args.n_residual_layers, lstm=inp_args.lstm,
                             enc_ratios=(8, 5, 4, 2), quantization=True, bandwidth=inp_args.cond_bandwidth,
                             final_activation=inp_args.final_activation)   # quirk Q1: ratios are always [8,5,4,2]
    unet = UnetConfig(dim=inp_args.diff_dims, inp_channels=inp_args.rep_dims, upsampling_ratios=tuple(inp_args.upsampling_ratios),
                      unet_scale_cond=inp_args.unet_scale_cond, unet_scale_x=inp_args.unet_scale_x)
    files = sorted(glob.glob(os.path.join(inp_args.input_dir, "**/*.wav"), recursive=True))
    n_eng = max(1, int(getattr(inp_args, "in_flight", 1)))
    if len(files) <= inp_args.batch_size * world:        # a single batch per rank: nothing to pipeline
        n_eng = 1
    sd_main, sd_cond = checkpoint.read_amlt(inp_args.model_path), checkpoint.read_amlt(inp_args.model_for_cond)
    engines = []
    for k in range(n_eng):
        eng = Engine(main_codec, unet, cond_codec, dtype=inp_args.dtype, device=local_rank, noise_seed=inp_args.seed + rank + 7919 * k)
        if n_eng > 1:
            eng.set_option("split", 1)                   # with batches in flight each batch is one chain (a per-context option,
        eng.load_state_dict(L.MODEL_MAIN, sd_main)       # not an environment variable: other engines of the process keep theirs)
        eng.load_state_dict(L.MODEL_COND, sd_cond)       # load_model(model, path, strict=True)
        eng.finalize(strict=True)
        engines.append(eng)
    written = decode_files(engines if n_eng > 1 else engines[0], files, inp_args, rank, world, local_rank)
    for eng in engines:
        eng.close()
    return written


def apply_device_fallback(eng, err) -> bool:
    """A kernel whose bounded in-launch wait gave up poisons its output with NaN and raises the context's device-side failure
    flag; the call that sees the flag fails with LDC_E_HIP and a tagged message (ldc_api.cpp: check_dev_flag).  Switch the engine
    to the form that needs no co-residency: "[coop_lstm]" -> the streamed LSTM kernel, "[gn_wait]" -> separate conv + gn_apply
    launches.  Returns False for any other error."""
    from . import lib as L
    if getattr(err, "code", None) != L.E_HIP or "device-side failure" not in str(err):
        return False
    if "[gn_wait]" in str(err):
        eng.set_option("fuse_gn_epi", 0)
    else:
        eng.set_option("lstm_stream", 1)
    return True


def decode_with_retry(eng, batch, n_steps: int, noise, per_item: bool):
    """One engine call; a batch hit by a device-side failure (or by the report of an earlier call's) is decoded again after
    the matching fallback (apply_device_fallback), twice at most: the LSTM and the GroupNorm exchange can each give up once."""
    from . import lib as L
    for attempt in range(3):
        try:
            return eng.decode(batch, n_steps, noise=noise, per_item=per_item)
        except L.LdcError as e:
            if attempt == 2 or not apply_device_fallback(eng, e):
                raise


def plan_batches(lengths: List[int], channels: List[int], rank: int, world: int, batch_size: int) -> List[Tuple[List[int], bool]]:
    """Work list of one rank: [(file indices, joint)].  Files are dealt to ranks whole (parallel.shard_utterances over
    FILES), so a file has one writer.  Mono files of equal trimmed length share batches (`joint` False: per-utterance
    normalisation = the reference's result for a mono file); a multi-channel file is its own batch, normalised jointly
    over its channels as sample.py:129,133-134 do."""
    from . import parallel
    mine = parallel.shard_utterances(lengths, rank, world)
    work: List[Tuple[List[int], bool]] = []
    by_len: Dict[int, List[int]] = {}
    for i in mine:
        if channels[i] > 1:
            work.append(([i], True))
        else:
            by_len.setdefault(lengths[i] // 640 * 640, []).append(i)
    for _, idxs in sorted(by_len.items(), reverse=True):
        for s in range(0, len(idxs), batch_size):
            work.append((idxs[s:s + batch_size], False))
    return work


_CHUNK_QUANTUM = 2560     # default quantum (enc_ratios 8 4): see chunk_quantum


def chunk_quantum(enc_ratios, unet_levels: int = 5) -> int:
    """Samples a chunk must be a multiple of: whole condition frames (320 samples, cond codec hop 8*5*4*2) AND a latent length that
    survives the UNet's unet_levels - 1 halvings (unet.py:336-349): lcm(320, hop * 2^(levels-1)) -- 2560 for enc_ratios 8 4 (hop
    32), 640 for enc_ratios 8.  (Through round 2 this was a fixed 1280, which let a 1.2 s tail -- the tail of a 30 s recording cut
    into 2.4 s chunks -- reach the UNet with L = 600.)"""
    import math
    hop = int(np.prod(list(enc_ratios)))
    return math.lcm(320, hop * (1 << (unet_levels - 1)))


def plan_chunks(n_samples: int, chunk: int, quantum: int = _CHUNK_QUANTUM) -> List[Tuple[int, int]]:
    """[(start, length)] of a recording cut into `chunk`-sample pieces; the tail keeps whole quanta (a shorter last chunk),
    what is left of it (less than one quantum) is dropped as the reference drops the sub-frame tail (sample.py:87-88)."""
    out, pos = [], 0
    while n_samples - pos >= chunk:
        out.append((pos, chunk)); pos += chunk
    tail = (n_samples - pos) // quantum * quantum
    if tail > 0:
        out.append((pos, tail))
    return out


def decode_long_files(eng, files: List[str], wavs, inp_args, rank: int, world: int, local_rank: int) -> List[str]:
    """Long-form mode: every chunk of every recording of this rank is one batch item (equal-length chunks share engine calls
    across recordings); the chunks' latents go through the decoder, the raw waveforms are joined per recording and the
    reference's output normalisation (sample.py:133-134) runs once over the whole recording."""
    import torch
    from scipy.io import wavfile
    from . import lib as L, parallel
    quantum = chunk_quantum(getattr(inp_args, "enc_ratios", [8, 4]))
    chunk = max(quantum, int(round(inp_args.chunk_sec * 16000)) // quantum * quantum)
    mine = parallel.shard_utterances([sh[1] for sh in wavs.shapes], rank, world)
    pieces: Dict[int, List[Tuple[int, int, int]]] = {}            # chunk length -> [(file, order, start)]
    nchunks = {}
    for i in mine:
        plan = plan_chunks(wavs.shapes[i][1], chunk, quantum)
        nchunks[i] = len(plan)
        for k, (st, ln) in enumerate(plan):
            pieces.setdefault(ln, []).append((i, k, st))
    dev = torch.device("cuda", local_rank)
    raw: Dict[int, List] = {i: [None] * nchunks[i] for i in mine}
    for ln, items in sorted(pieces.items(), reverse=True):
        for s in range(0, len(items), inp_args.batch_size):
            part = items[s:s + inp_args.batch_size]
            batch = torch.from_numpy(np.stack([wavs[i][0, st:st + ln] for i, _, st in part])[:, None, :])
            # test seam (see decode_files): keys are (file index, chunk number)
            provider = getattr(inp_args, "noise_provider", None)
            hop = int(np.prod(getattr(inp_args, "enc_ratios", [8])))
            noise = provider([(i, k) for i, k, _ in part], inp_args.midway_t, ln // hop).to(dev) if provider is not None else None
            stages = eng.decode(batch.to(dev), inp_args.midway_t, noise=noise, per_item=True, want_stages=True)
            wav_raw = eng.decode_latents(L.MODEL_MAIN, stages["latents"])          # un-normalised decoder output
            for j, (i, k, _) in enumerate(part):
                raw[i][k] = wav_raw[j:j + 1]
    written = []
    for i in mine:
        if not raw[i]:
            continue
        whole = eng.output_normalise(torch.cat(raw[i], dim=-1), per_item=False)
        if not bool(torch.isfinite(whole).all()):
            raise RuntimeError(f"non-finite audio decoded for {files[i]}")
        path = output_path(files[i], inp_args.input_dir, inp_args.output_dir)
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        wavfile.write(path, 16000, np.ascontiguousarray(whole.cpu().numpy()[0, 0]))
        written.append(path)
    return written


def decode_files(eng, files: List[str], inp_args, rank: int, world: int, local_rank: int) -> List[str]:
    """`eng`: one engine, or a list of engines (one batch in flight per engine, each on its own stream)."""
    import torch
    from scipy.io import wavfile
    engines = list(eng) if isinstance(eng, (list, tuple)) else [eng]
    eng = engines[0]
    wavs = LazyWavs(files, eng)                     # headers only: (channels, samples at 16 kHz); data is loaded per shard
    keep = [i for i, sh in enumerate(wavs.shapes) if sh[1] // 640 * 640 > 0]                  # sample.py:87-88
    files, wavs = [files[i] for i in keep], wavs.subset(keep)
    chunk_sec = float(getattr(inp_args, "chunk_sec", 0.0) or 0.0)
    if chunk_sec > 0:
        # recordings longer than a chunk (mono) take the long-form path, everything else the reference's whole-file path
        is_long = [sh[0] == 1 and sh[1] > int(round(chunk_sec * 16000)) for sh in wavs.shapes]
        long_i = [i for i, m in enumerate(is_long) if m]
        short_i = [i for i, m in enumerate(is_long) if not m]
        long_f, long_w = [files[i] for i in long_i], wavs.subset(long_i)
        files, wavs = [files[i] for i in short_i], wavs.subset(short_i)
        written_long = decode_long_files(eng, long_f, long_w, inp_args, rank, world, local_rank) if long_f else []
    else:
        written_long = []
    lengths = [sh[1] for sh in wavs.shapes]
    channels = [sh[0] for sh in wavs.shapes]
    written = []
    dev = torch.device("cuda", local_rank)
    streams = [torch.cuda.Stream(device=dev) for _ in engines] if len(engines) > 1 else [None]
    pending: List[tuple] = []                 # (output tensor on the device, file indices, joint, stream, redo), oldest first
    hop = int(np.prod(getattr(inp_args, "enc_ratios", [8])))   # samples per latent frame of the main codec (noise seam only)

    def retire(item):
        out, idxs, joint, stream, redo = item
        if stream is not None:
            stream.synchronize()               # the producer stream, not the current one: waits for that engine's batch only
        out = out.cpu()
        if not bool(torch.isfinite(out).all()):
            # the device-side failure of this batch (cooperative LSTM timed out: its output is poisoned with NaN) is reported by
            # the engine's NEXT call; decode the batch again on the streamed LSTM instead of losing the run
            # (that report -- LDC_E_HIP with the failure's tag -- may well arrive in the redo itself: decode_with_retry applies the
            # matching fallback and decodes again; without a report the cooperative LSTM is the one that poisons silently)
            eng_k, batch_k, per_item_k, noise_k = redo
            out = decode_with_retry(eng_k, batch_k.to(dev), inp_args.midway_t, noise_k, per_item_k).cpu()
            if not bool(torch.isfinite(out).all()):
                eng_k.set_option("lstm_stream", 1)
                eng_k.set_option("fuse_gn_epi", 0)
                out = decode_with_retry(eng_k, batch_k.to(dev), inp_args.midway_t, noise_k, per_item_k).cpu()
            if not bool(torch.isfinite(out).all()):
                raise RuntimeError(f"non-finite audio decoded for {[files[i] for i in idxs]}")
        out = out.numpy()
        for k, i in enumerate(idxs):
            path = output_path(files[i], inp_args.input_dir, inp_args.output_dir)
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
            data = out[:, 0, :].T if joint else out[k, 0]                                     # [T, channels] | [T]
            wavfile.write(path, 16000, np.ascontiguousarray(data))
            written.append(path)

    for j, (idxs, joint) in enumerate(plan_batches(lengths, channels, rank, world, inp_args.batch_size)):
        n = lengths[idxs[0]] // 640 * 640
        if joint:
            batch = torch.from_numpy(np.ascontiguousarray(wavs[idxs[0]][:, None, :n]))      # [channels, 1, T], as sample.py:85
        else:
            batch = torch.from_numpy(np.stack([wavs[i][0, :n] for i in idxs])[:, None, :])
        slot = j % len(engines)
        if len(pending) >= len(engines):
            retire(pending.pop(0))             # the batch this engine decoded last: its output is read before the slot is reused
        # test seam: a callable (file indices, steps, latent length) -> noise [steps, B, 128, L] replaces the device-drawn noise
        provider = getattr(inp_args, "noise_provider", None)
        noise = provider(idxs, inp_args.midway_t, n // hop).to(dev) if provider is not None else None
        if streams[slot] is not None:
            with torch.cuda.stream(streams[slot]):
                out = decode_with_retry(engines[slot], batch.to(dev, non_blocking=True), inp_args.midway_t, noise, not joint)
        else:
            out = decode_with_retry(engines[slot], batch.to(dev), inp_args.midway_t, noise, not joint)
        pending.append((out, idxs, joint, streams[slot], (engines[slot], batch, not joint, noise)))
    while pending:
        retire(pending.pop(0))
    return written_long + written


def main(argv=None):
    synthesis(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
