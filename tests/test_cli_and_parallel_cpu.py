"""Host logic without a GPU: CLI contract (flags / defaults / output naming of srcs/sample.py) and the
data-parallel helpers over a 2-process gloo group."""
import os
import subprocess
import sys

import numpy as np
import pytest

from ladiffcodec_amd import parallel, sample, spec
from helpers import COND_CFG

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# flag -> default, transcribed from SURVEY.md section 8b (reference srcs/sample.py:141-201)
REFERENCE_DEFAULTS = {
    "data_folder_path": "/data/hy17/librispeech/librispeech", "n_spks": 500, "seq_len_in_sec": 1.8, "sample_rate": 16000,
    "model_path": "", "qtzer_path": "", "note": "", "rep_dims": 128, "emb_dims": 128, "quantization": False,
    "bandwidth": 3.0, "n_filters": 32, "lstm": 2, "n_residual_layers": 1, "enc_ratios": [8], "final_activation": None,
    "run_diff": False, "run_vae": False, "train_time_diff": False, "diff_dims": 256, "qtz_condition": False,
    "self_condition": False, "seq_length": 16000, "model_type": "unet", "scaling_frame": False, "scaling_feature": False,
    "scaling_global": False, "scaling_dim": False, "sampling_timesteps": 1000, "use_film": False, "model_for_cond": "",
    "upsampling_ratios": [5, 4, 2], "cond_enc_ratios": [8, 5, 4, 2], "cond_bandwidth": 3.0, "cond_global": 3.0,
    "unet_scale_cond": False, "unet_scale_x": False, "input_dir": "", "output_dir": "outputs/",
}


def test_cli_flags_and_defaults_match_reference():
    ns = vars(sample.build_parser().parse_args([]))
    for k, v in REFERENCE_DEFAULTS.items():
        assert k in ns, k
        assert ns[k] == v, (k, ns[k], v)
    assert ns["midway_t"] == 100                      # the reference's literal (sample.py:69)
    assert set(ns) - set(REFERENCE_DEFAULTS) == {"midway_t", "dtype", "batch_size", "seed", "chunk_sec", "in_flight"}


def test_cli_readme_invocation_parses():
    a = sample.build_parser().parse_args(
        "--model_for_cond EnCodec_libri_3kb/model_best.amlt --model_path Ladiff_3kb_8/model_best.amlt --run_diff "
        "--scaling_global --cond_bandwidth 3 --unet_scale_cond --input_dir /in/ --output_dir /out/".split())
    assert a.unet_scale_cond and a.run_diff and a.scaling_global and a.cond_bandwidth == 3.0
    assert a.enc_ratios == [8] and a.upsampling_ratios == [5, 4, 2]


def test_output_path_rule():
    # sample.py:75-76,136 with an absolute --output_dir (quirk Q10)
    assert sample.output_path("/in/spk/a/utt1.wav", "/in/", "/out/") == "/out/spk/a/utt1.wav"


def test_module_entry_point_exists():
    r = subprocess.run([sys.executable, "-m", "srcs.sample", "--help"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and "--model_for_cond" in r.stdout and "--midway_t" in r.stdout


def test_shard_helpers():
    assert [parallel.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    lengths = [6400, 640, 38400, 38400, 12800, 700, 38401]
    shards = [parallel.shard_utterances(lengths, r, 3) for r in range(3)]
    assert sorted(sum(shards, [])) == list(range(len(lengths)))
    assert abs(len(shards[0]) - len(shards[2])) <= 1


_WORKER = r'''
import json, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from ladiffcodec_amd import parallel, sample, spec, synth
from helpers import COND_CFG
rank, local_rank, world = parallel.init_process_group("gloo")
layout = spec.codec_keys(COND_CFG)
sd = synth.codec_state_dict(COND_CFG, seed=3) if rank == 0 else None
got = parallel.broadcast_state_dict(sd, layout)
ref = synth.codec_state_dict(COND_CFG, seed=3)
assert list(got) == [k for k, _ in layout]
assert all(np.array_equal(got[k], ref[k]) for k in ref), "broadcast mismatch"
lo, hi = parallel.shard_range(7, rank, world)
local = torch.full((4, 1, 8), float(rank))
outs = parallel.gather_results(local, world)
assert len(outs) == world and all(float(o.mean()) == float(r) for r, o in enumerate(outs))
t = parallel.max_over_ranks(1.0 + rank)
assert t == float(world)
grads = torch.arange(10, dtype=torch.float32) * (rank + 1)          # training-step gradient reduction (flat buffer)
parallel.allreduce_gradients(grads)
assert torch.allclose(grads, torch.arange(10, dtype=torch.float32) * (1 + 2) / 2)

# synthesis()'s sharding / bucketing with a stub engine: every file decoded exactly once by exactly one rank,
# multi-channel files whole and jointly normalised, mono files batched by equal trimmed length
from scipy.io import wavfile
ind, outd = sys.argv[2], sys.argv[3]
class Stub:
    calls = []
    def decode(self, batch, n_steps, noise=None, per_item=False):
        Stub.calls.append((tuple(batch.shape), bool(per_item)))
        return batch * 0.5
files = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(ind) for f in fs if f.endswith(".wav"))
class A: pass
a = A(); a.batch_size = 2; a.midway_t = 3; a.input_dir = ind + "/"; a.output_dir = outd + "/"
torch.Tensor.to = (lambda orig: (lambda self, *x, **k: self))(torch.Tensor.to)     # no GPU here: .to(device) is a no-op
written = sample.decode_files(Stub(), files, a, rank, world, 0)
with open(os.path.join(outd, f"rank{rank}.json"), "w") as f:
    json.dump({"rank": rank, "shard": [lo, hi], "written": written, "calls": Stub.calls}, f)
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_gloo_broadcast_gather_and_cli_sharding(tmp_path):
    # Each rank writes its own result file: nothing is asserted on interleaved stdout.
    import json
    import socket
    from scipy.io import wavfile
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    ind, outd = tmp_path / "in", tmp_path / "out"
    (ind / "a").mkdir(parents=True); outd.mkdir()
    rng = np.random.default_rng(0)
    spec_files = {"a/m1.wav": (1, 1280), "a/m2.wav": (1, 1280 + 5), "a/m3.wav": (1, 1280), "m4.wav": (1, 640), "st.wav": (2, 1920),
                  "a/short.wav": (1, 100), "m5.wav": (1, 1280)}
    for name, (ch, n) in spec_files.items():
        x = (rng.standard_normal((n, ch)) * 0.1).astype(np.float32)
        wavfile.write(str(ind / name), 16000, x[:, 0] if ch == 1 else x)
    with socket.socket() as sk:          # a free port: back-to-back runs must not collide on a fixed one
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", port, str(script), ROOT, str(ind), str(outd)], env=env, capture_output=True,
                       text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = [json.load(open(outd / f"rank{k}.json")) for k in range(2)]
    assert res[0]["shard"] == [0, 4] and res[1]["shard"] == [4, 7]
    all_written = res[0]["written"] + res[1]["written"]
    expect = sorted(str(outd / n) for n in spec_files if n != "a/short.wav")              # < 640 samples: skipped (sample.py:87)
    assert sorted(all_written) == expect and len(set(all_written)) == len(all_written)     # one writer per file
    assert abs(len(res[0]["written"]) - len(res[1]["written"])) <= 1
    calls = [tuple(c) for k in range(2) for c in res[k]["calls"]]
    assert ((2, 1, 1920), False) in [(tuple(c[0]), c[1]) for c in calls]                   # stereo: one joint batch of its channels
    for shape, per_item in calls:
        assert shape[0] <= 2 and (per_item or shape[0] == 2)
    sr, st = wavfile.read(str(outd / "st.wav"))
    assert sr == 16000 and st.shape == (1920, 2)
    sr, m2 = wavfile.read(str(outd / "a/m2.wav"))
    assert m2.shape == (1280,)


def test_plan_batches_properties():
    lengths = [1280, 1285, 1280, 640, 1920, 1280, 6400, 6400, 6400]
    channels = [1, 1, 1, 1, 2, 1, 1, 1, 3]
    seen = []
    for rank in range(3):
        for idxs, joint in sample.plan_batches(lengths, channels, rank, 3, batch_size=2):
            assert len(idxs) <= 2
            assert len({lengths[i] // 640 * 640 for i in idxs}) == 1
            assert joint == (channels[idxs[0]] > 1) and (not joint or len(idxs) == 1)
            seen += idxs
    assert sorted(seen) == list(range(len(lengths)))


def test_long_form_chunk_plan_and_reassembly(tmp_path, monkeypatch):
    """--chunk_sec (BASELINE config 5): a 30 s recording becomes 12 chunks of 2.4 s + a shorter tail, chunks of equal length
    batch together across recordings, the raw decoder outputs are joined in order and normalised ONCE per recording; short
    and multi-channel files keep the whole-file path."""
    import torch
    from scipy.io import wavfile
    from ladiffcodec_amd import sample
    # enc_ratios 8 4: a chunk is whole condition frames (320 samples) AND a latent length that survives the UNet's four halvings
    # (hop 32 x 16): quanta of 2560 samples -- the 1.2 s tail of a 30 s recording keeps 1.12 s
    assert sample.chunk_quantum([8, 4]) == 2560 and sample.chunk_quantum([8]) == 640
    assert sample.plan_chunks(480000, 38400) == [(k * 38400, 38400) for k in range(12)] + [(460800, 17920)]
    assert sample.plan_chunks(38400 + 2559, 38400) == [(0, 38400)]                       # less than a quantum of tail: dropped
    assert sample.plan_chunks(2560 + 1280, 2560, 640) == [(0, 2560), (2560, 1280)]
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir(); outd.mkdir()
    rng = np.random.default_rng(1)
    spec_files = {"long1.wav": (1, 2560 * 3 + 1280 + 7), "long2.wav": (1, 2560 * 2), "short.wav": (1, 1920), "st.wav": (2, 2560 * 4)}
    for name, (ch, n) in spec_files.items():
        x = (rng.standard_normal((n, ch)) * 0.1).astype(np.float32)
        wavfile.write(str(ind / name), 16000, x[:, 0] if ch == 1 else x)

    class Stub:
        def __init__(self):
            self.calls, self.norm_calls = [], []
        def decode(self, batch, n_steps, noise=None, per_item=False, want_stages=False):
            self.calls.append((tuple(batch.shape), bool(per_item), bool(want_stages)))
            return {"latents": batch * 2.0} if want_stages else batch * 0.5
        def decode_latents(self, which, z):
            return z + 1.0                                                                  # "raw decoder output"
        def output_normalise(self, wav, per_item=False):
            self.norm_calls.append((tuple(wav.shape), bool(per_item)))
            return wav * 0.25

    class A:
        pass
    a = A(); a.batch_size = 4; a.midway_t = 3; a.input_dir = str(ind) + "/"; a.output_dir = str(outd) + "/"; a.chunk_sec = 2560 / 16000.0
    a.enc_ratios = [8]                                                                      # quantum 640: long1's 1280-sample tail survives
    monkeypatch.setattr(torch.Tensor, "to", lambda self, *x, **k: self)                     # no GPU here
    eng = Stub()
    files = sorted(str(ind / n) for n in spec_files)
    written = sample.decode_files(eng, files, a, 0, 1, 0)
    assert sorted(written) == sorted(str(outd / n) for n in spec_files)
    # long1: 3 chunks of 2560 + tail 1280; long2: exactly 2 chunks is NOT longer than... (5120 > 2560 -> long): 2 chunks
    long_calls = [c for c in eng.calls if c[2]]
    assert sorted(c[0] for c in long_calls) == [(1, 1, 1280), (4, 1, 2560), (1, 1, 2560)] or \
        sorted(c[0] for c in long_calls) == sorted([(4, 1, 2560), (1, 1, 2560), (1, 1, 1280)])
    assert all(c[1] for c in long_calls)
    assert sorted(eng.norm_calls) == [((1, 1, 2560 * 2), False), ((1, 1, 2560 * 3 + 1280), False)]   # once per recording
    sr, y = wavfile.read(str(outd / "long1.wav"))
    src = wavfile.read(str(ind / "long1.wav"))[1]
    assert y.shape == (2560 * 3 + 1280,)
    np.testing.assert_allclose(y, (src[:y.shape[0]] * 2.0 + 1.0) * 0.25, rtol=1e-6, atol=1e-7)   # chunks joined in order
    whole = [c for c in eng.calls if not c[2]]
    assert ((2, 1, 2560 * 4), False, False) in whole and ((1, 1, 1920), True, False) in whole       # stereo / short: whole-file path


def test_cli_batches_in_flight_round_robin(tmp_path, monkeypatch):
    """--in_flight: several engines, batches dealt round-robin, an engine's previous output is read before its slot is reused,
    every file written once."""
    import contextlib
    import torch
    from scipy.io import wavfile
    from ladiffcodec_amd import sample
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir(); outd.mkdir()
    rng = np.random.default_rng(2)
    names = [f"f{k}.wav" for k in range(7)]
    for k, name in enumerate(names):
        wavfile.write(str(ind / name), 16000, (rng.standard_normal(1280) * 0.1).astype(np.float32))
    log = []

    class Stub:
        def __init__(self, tag):
            self.tag = tag
        def decode(self, batch, n_steps, noise=None, per_item=False):
            log.append(("decode", self.tag, batch.shape[0]))
            return batch * (1.0 + self.tag)

    class A:
        pass
    a = A(); a.batch_size = 2; a.midway_t = 3; a.input_dir = str(ind) + "/"; a.output_dir = str(outd) + "/"; a.chunk_sec = 0.0
    monkeypatch.setattr(torch.Tensor, "to", lambda self, *x, **k: self)
    class FakeStream:
        def synchronize(self):
            log.append(("sync",))
    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: FakeStream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    files = sorted(str(ind / n) for n in names)
    written = sample.decode_files([Stub(0), Stub(1)], files, a, 0, 1, 0)
    assert sorted(written) == sorted(str(outd / n) for n in names) and len(set(written)) == 7
    dec = [e for e in log if e[0] == "decode"]
    assert [e[1] for e in dec] == [0, 1, 0, 1]                      # 7 files / batch 2 = 4 batches, alternating engines
    assert sum(e[2] for e in dec) == 7
    assert sum(1 for e in log if e[0] == "sync") == 4               # every batch's producer stream is waited for before its read
    assert [e[0] for e in log[:3]] == ["decode", "decode", "sync"]  # two batches in flight before the first output is read
    for n in names:                                                  # each output carries its engine's factor
        y = wavfile.read(str(outd / n))[1]; x = wavfile.read(str(ind / n))[1]
        r = float(np.abs(y).max() / np.abs(x).max())
        assert abs(r - 1.0) < 1e-5 or abs(r - 2.0) < 1e-5


def test_wav_header_reads_every_container_without_touching_samples(tmp_path):
    """sample.wav_header parses the RIFF chunks itself (scipy's mmap read refuses 24-bit PCM and one such file aborted
    the run on every rank): 16-bit, 24-bit, 32-bit float, stereo, an odd-sized extra chunk in front of 'data', and a 44.1 kHz
    file whose 16 kHz length must be the resampler's own (ldc_resample_out_len = ceil(16000 T / sr))."""
    import struct
    from scipy.io import wavfile
    from ladiffcodec_amd import sample as cli
    rng = np.random.default_rng(0)
    wavfile.write(tmp_path / "a16.wav", 16000, (rng.standard_normal(5000) * 3000).astype(np.int16))
    assert cli.wav_header(str(tmp_path / "a16.wav")) == (1, 5000, 16000)
    wavfile.write(tmp_path / "f32.wav", 16000, rng.standard_normal((4000, 2)).astype(np.float32))
    assert cli.wav_header(str(tmp_path / "f32.wav")) == (2, 4000, 16000)
    wavfile.write(tmp_path / "a44.wav", 44100, (rng.standard_normal(44100) * 3000).astype(np.int16))
    assert cli.wav_header(str(tmp_path / "a44.wav")) == (1, 16000, 44100)
    wavfile.write(tmp_path / "b44.wav", 44100, (rng.standard_normal(44101) * 3000).astype(np.int16))
    assert cli.wav_header(str(tmp_path / "b44.wav")) == (1, -(-16000 * 44101 // 44100), 44100)
    # 24-bit PCM, hand-written, with a 'LIST' chunk of odd size before 'data'
    n, ch = 3001, 2
    data = rng.integers(0, 256, n * ch * 3, dtype=np.uint8).tobytes()
    extra = b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\0"
    fmt = b"fmt " + struct.pack("<IHHIIHH", 16, 1, ch, 16000, 16000 * ch * 3, ch * 3, 24)
    body = b"WAVE" + fmt + extra + b"data" + struct.pack("<I", len(data)) + data
    (tmp_path / "p24.wav").write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
    assert cli.wav_header(str(tmp_path / "p24.wav")) == (ch, n, 16000)
    x, sr = cli.read_wav(str(tmp_path / "p24.wav"))          # the full read decodes the same file
    assert x.shape == (ch, n) and sr == 16000
    import pytest
    (tmp_path / "bad.wav").write_bytes(b"not a wave file at all")
    with pytest.raises(ValueError):
        cli.wav_header(str(tmp_path / "bad.wav"))
