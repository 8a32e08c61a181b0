"""Multi-GPU path: contiguous path-range sharding + one all-reduce of the fp64 triple.

New relative to the reference (single device, no NCCL/MPI: SURVEY 2.2 "Collectives: none").
One process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm).  Paths are
i.i.d., so rank g of G simulates the contiguous global range
``[floor(g P / G), floor((g+1) P / G))`` (``mc_shard_range``) with the SAME seed -- a path's
normals depend only on (seed, global path index), so the union over ranks is exactly the
single-GPU stream -- and the only exchange is ``all_reduce(SUM)`` of ``{sum, sum2, n}``
(3 doubles = 24 bytes, latency-bound; link bandwidth is irrelevant).  The closing formulas run
on every rank from the reduced triple.

torch is used for the process group, the device tensor that receives the triple, and streams:
plumbing only.  ``compute`` is injected, so the same code is exercised on CPU under gloo with
the oracle as the per-shard engine (tests/test_distributed_cpu.py).
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from .engine import Estimate, closing, shard_range


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).

    Returns (rank, world, local_rank).  A group is created whenever a launcher set RANK (also for
    a world of 1, so the collective plumbing is exercised on a one-GPU box); a plain
    `python bench.py` creates none."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or "RANK" in os.environ) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def my_shard(total_paths: int, rank: Optional[int] = None, world: Optional[int] = None) -> Tuple[int, int]:
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    return shard_range(total_paths, rank, world)


def reduce_triple(triple: torch.Tensor, async_op: bool = False):
    """In-place SUM all-reduce of the {sum, sum2, n} tensor (float64[3]) across ranks."""
    if dist.is_initialized():
        return dist.all_reduce(triple, op=dist.ReduceOp.SUM, async_op=async_op)
    return None


def estimate_from_triple(triple, discount: float) -> Estimate:
    s, s2, n = (float(x) for x in triple)
    e, c = closing(s, s2, int(n), discount)
    return Estimate(e, c, s, s2, int(n))


def sharded_estimate(compute: Callable[[int, int, torch.Tensor], None], total_paths: int, discount: float,
                     device: Optional[torch.device] = None) -> Estimate:
    """compute(first_path, n_paths, out) must leave this rank's {sum, sum2, n} in ``out``
    (float64[3] on ``device``), enqueued on the current stream if ``device`` is a GPU."""
    first, count = my_shard(total_paths)
    out = torch.zeros(3, dtype=torch.float64, device=device)
    if count:
        compute(first, count, out)
    reduce_triple(out)
    return estimate_from_triple(out.cpu(), discount)
