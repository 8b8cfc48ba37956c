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
    assert set(ns) - set(REFERENCE_DEFAULTS) == {"midway_t", "dtype", "batch_size"}


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
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from ladiffcodec_amd import parallel, spec, synth
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
dist.barrier(); dist.destroy_process_group()
print("ok", rank, lo, hi)
'''


def test_two_rank_gloo_broadcast_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    import socket
    with socket.socket() as sk:          # a free port: back-to-back runs must not collide on a fixed one
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", port, str(script), ROOT], env=env, capture_output=True, text=True,
                       timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "ok 0 0 4" in r.stdout and "ok 1 4 7" in r.stdout
