"""More than one RCCL rank on the hardware: tests/dist_gpu_worker.py under torch.distributed.run.  The
two-rank test runs the moment a box exposes two devices; a one-GPU box runs the same worker at world size 1."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_worker(n, tmp_path, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0 and f"DIST_GPU_OK world={n}" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]


def test_distributed_worker_at_world_size_one(tmp_path):
    """Broadcast, sharded decode + gather, the CLI over a directory and the flat-gradient reduction under torch.distributed.run with one rank."""
    run_worker(1, tmp_path, 29641)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's multi-GPU node)")
def test_two_ranks_over_rccl(tmp_path):
    """The same with two ranks: uneven utterance shards, every file of the directory written once, the data-parallel gradient of a
    four-item batch equal to the single-process one."""
    run_worker(2, tmp_path, 29642)
