"""Multi-GPU plumbing: one process per GPU, batch sharding, one weight broadcast.

The path shards over independent samples (no op mixes samples - SURVEY.md 8e), so there is
no collective in the step loop.  The only exchange is the start-up broadcast of the packed
weight blob (rank 0 packs; the others receive it over RCCL/xGMI - backend "nccl" on ROCm - or
gloo in the CPU tests) and, optionally, a final gather of the results.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of ``total`` samples for ``rank`` (first ``total % world`` ranks get one more)."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def init_from_env():
    """Initialise torch.distributed from the torchrun environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def broadcast_blob(blob: torch.Tensor, src: int = 0) -> torch.Tensor:
    """In-place broadcast of a packed weight blob (a no-op in single-process runs)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src=src)
    return blob


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float) -> float:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([value], dtype=torch.float64, device="cuda" if torch.cuda.is_available() else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return value
