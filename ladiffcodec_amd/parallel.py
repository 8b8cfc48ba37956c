"""Data-parallel decode over the GPUs of one node: one process per GPU, utterances sharded, no
data-path collective.  torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU
tests) is used for exactly two things, as SURVEY.md section 8e lays out:

  * `broadcast_state_dict`: rank 0 reads / builds the checkpoint once and broadcasts it as ONE flat
    float32 buffer (a single large collective suits point-to-point xGMI better than one message per
    tensor, which is what the reference's `broadcast_tensors`, srcs/quantization/distrib.py:55-68, does);
  * `gather_results`: decoded waveforms come back to rank 0 with one all_gather.

The reference has no inference-time parallelism at all (synthesis() walks files one by one,
srcs/sample.py:73); utterances are independent, so sharding them is exact.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def dist_env() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend: str):
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    # under torch.distributed.run ("RANK" is set) the group is created even for a single rank, so that the same
    # broadcast / gather code runs at N = 1 and at N = 8
    if (world > 1 or "RANK" in os.environ) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n_items for `rank`; the first (n_items % world) ranks get one more."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_utterances(lengths: Sequence[int], rank: int, world: int, multiple: int = 640) -> List[int]:
    """Length-bucketed round-robin: utterance indices sorted by trimmed length (sample.py:87 trims to a
    multiple of 640) are dealt to ranks in turn, so every rank sees the same mix of lengths."""
    trimmed = [(int(n) // multiple * multiple, i) for i, n in enumerate(lengths)]
    order = [i for _, i in sorted(trimmed, key=lambda p: (-p[0], p[1]))]
    return order[rank::world]


def broadcast_state_dict(sd: Optional["OrderedDict[str, np.ndarray]"], layout: List[Tuple[str, Tuple[int, ...]]],
                         device=None, src: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Every rank passes the same `layout` (key, shape) list (spec.py derives it from the config);
    only `src` needs `sd`.  One flat fp32 broadcast."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        assert sd is not None
        return sd
    total = sum(int(np.prod(s)) if len(s) else 1 for _, s in layout)
    dev = device if device is not None else torch.device("cpu")
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    if dist.get_rank() == src:
        assert sd is not None
        host = np.concatenate([np.asarray(sd[k], np.float32).reshape(-1) for k, _ in layout])
        flat.copy_(torch.from_numpy(host))
    dist.broadcast(flat, src=src)
    host = flat.cpu().numpy()
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    off = 0
    for k, s in layout:
        n = int(np.prod(s)) if len(s) else 1
        out[k] = host[off:off + n].reshape(s)
        off += n
    return out


def gather_results(local, world: int):
    """all_gather of equally-shaped per-rank result tensors -> list indexed by rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return [local]
    outs = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(outs, local)
    return outs


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_gradients(flat, average: bool = True, force_collective: bool = False):
    """Data-parallel gradient reduction of the training step: ONE flat fp32 buffer per step
    (542 MB for the diff_dims = 256 model) instead of the reference's per-parameter all-reduce (srcs/encodec/distrib.py:
    sync_grad).  On RCCL the sum is a reduce-scatter followed by an all-gather: xGMI is point-to-point, and the two halves keep
    all seven links of every GPU busy with 1/world-sized shards (each rank can also run its optimiser step on its own shard
    between the two); gloo (CPU tests) has no reduce-scatter and takes the plain all-reduce.  In place; returns `flat`.
    The buffer is padded so that every rank's shard is a whole number of 16-byte pieces.  `force_collective` runs the collectives
    in a one-rank group too (the GPU suite brings up a world-size-1 RCCL group so that this exact code path has executed on
    the hardware before an 8-GPU node sees it)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        return flat
    world = dist.get_world_size()
    if dist.get_backend() == "nccl":
        n = flat.numel()
        pad = (-n) % (world * 4)
        buf = torch.cat([flat, flat.new_zeros(pad)]) if pad else flat
        shard = torch.empty(buf.numel() // world, dtype=buf.dtype, device=buf.device)
        dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM)
        if average:
            shard /= world
        dist.all_gather_into_tensor(buf, shard)
        if pad:
            flat.copy_(buf[:n])
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= world
    return flat
