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
    # PF_LOCAL_DEVICE / PF_DIST_BACKEND exist for ONE purpose: running the N-rank code path on a box with a single GPU (every rank
    # on device 0, gloo instead of RCCL, which refuses two ranks on one device) - tests/test_gpu_multirank.py.  Production: unset.
    if "PF_LOCAL_DEVICE" in os.environ:
        local = int(os.environ["PF_LOCAL_DEVICE"])
    backend = os.environ.get("PF_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend)
        else:
            dist.init_process_group("gloo")
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


_direct_comm = None


def _direct(rank: int, world: int):
    """pf_comm over librccl directly (include/pfhip.h): the 128-byte unique id travels through the process group the
    launcher's rendezvous already set up (or, without one, a TCPStore on PF_COMM_PORT / MASTER_PORT + 1); the broadcast of the
    weight blobs itself involves no torch.distributed collective."""
    global _direct_comm
    if _direct_comm is None:
        import ctypes as C
        from datetime import timedelta
        from . import _lib
        lib = _lib.load()
        uid = (C.c_char * 128)()
        if rank == 0:
            _lib.check(lib.pf_comm_unique_id(uid), "pf_comm_unique_id")
        store = None
        if dist.is_available() and dist.is_initialized():
            # the launcher's own rendezvous carries the id: no second listening port that could be taken on the node
            box = [bytes(uid.raw) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            if rank != 0:
                uid.raw = box[0]
        else:
            # no process group: a TCPStore of our own.  PF_COMM_PORT names its port; default MASTER_PORT + 1 (rank 0 cannot tell the
            # others a probed port without a channel, so the number has to be agreed on beforehand)
            host = os.environ.get("MASTER_ADDR", "127.0.0.1")
            port = int(os.environ.get("PF_COMM_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
            try:
                store = dist.TCPStore(host, port, world, is_master=(rank == 0), timeout=timedelta(seconds=120))
            except (RuntimeError, OSError) as e:
                raise RuntimeError(f"PF_COMM_DIRECT: cannot open the id-exchange store on {host}:{port} ({e}); set PF_COMM_PORT to a free port "
                                   "(the same on every rank)") from e
            if rank == 0:
                store.set("pf_comm_uid", bytes(uid.raw))
            else:
                uid.raw = store.get("pf_comm_uid")
        h = C.c_void_p()
        _lib.check(lib.pf_comm_init(uid, rank, world, C.byref(h)), "pf_comm_init")
        _direct_comm = (lib, h, store)
    return _direct_comm


def broadcast_blob(blob: torch.Tensor, src: int = 0) -> torch.Tensor:
    """In-place broadcast of a packed weight blob (a no-op in single-process runs).  Default transport: torch.distributed
    (backend nccl = RCCL over xGMI; gloo in the CPU tests).  PF_COMM_DIRECT=1 uses the library's own pf_comm_bcast (librccl
    opened by libpfhip.so itself) - same wire protocol, no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("PF_COMM_DIRECT") == "1" and blob.is_cuda:
        from . import _lib
        lib, h, _ = _direct(int(os.environ.get("RANK", "0")), world)
        _lib.check(lib.pf_comm_bcast(h, blob.data_ptr(), blob.numel() * blob.element_size(), src, _lib.current_stream()), "pf_comm_bcast")
        return blob
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


def broadcast_int(value: int, src: int = 0) -> int:
    """Rank ``src``'s integer on every rank (e.g. a randomly drawn seed)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([int(value)], dtype=torch.int64, device="cuda" if torch.cuda.is_available() else "cpu")
        dist.broadcast(t, src=src)
        return int(t.item())
    return int(value)


def gather_floats(value: float) -> list:
    """Every rank's value, in rank order, on every rank (a one-element list in single-process runs).  One all_reduce of a vector
    that is zero except at the caller's own rank: works on every backend the other helpers here work on."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.zeros(dist.get_world_size(), dtype=torch.float64, device="cuda" if torch.cuda.is_available() else "cpu")
        t[dist.get_rank()] = float(value)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]
    return [float(value)]


def gather_rows(local: torch.Tensor, total: int, rank: int, world: int) -> torch.Tensor:
    """Concatenate the ranks' row blocks (rank r holds rows ``shard_range(total, r, world)``) on every rank.

    The end-of-run exchange of SURVEY.md 8e: ``all_gather`` needs equal shapes, so every rank pads its block to the
    largest shard (``ceil(total / world)`` rows) and the padding is dropped after the collective."""
    lo, hi = shard_range(total, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"gather_rows: rank {rank} holds {local.shape[0]} rows, expected {hi - lo}")
    if world == 1 or not (dist.is_available() and dist.is_initialized()):
        return local
    cap = -(-total // world)
    padded = local.new_zeros((cap, *local.shape[1:]))
    padded[: hi - lo] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded.contiguous())
    out = []
    for r, part in enumerate(parts):
        a, b = shard_range(total, r, world)
        out.append(part[: b - a])
    return torch.cat(out, dim=0)


def ranks_seen():
    """(world size the process group reports, the device index every rank runs on) - bench.py prints it so a run that
    silently fell back to fewer ranks is visible in the JSON line."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        me = torch.tensor([torch.cuda.current_device() if torch.cuda.is_available() else -1], dtype=torch.int64,
                          device="cuda" if torch.cuda.is_available() else "cpu")
        parts = [torch.empty_like(me) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, me)
        return dist.get_world_size(), [int(p.item()) for p in parts]
    return 1, [torch.cuda.current_device() if torch.cuda.is_available() else -1]
