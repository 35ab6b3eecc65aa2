"""Multi GPU plumbing: one process per GPU, reads sharded with no data path collective, one all-reduce
of the integer counter buffer at the end of the run.

The reference merges its per thread decoder copies with field wise `+=`
(Transcode::finalize transcode.cpp:1816-1818 -> TranscodingDecoder::collect transcode.cpp:162-179 ->
Classifier::collect classifier.h:87-93 -> AccumulatingOption::collect selector.cpp:68-77).  Every word of
the counter buffer is such a sum, so the merge over GPUs is a single SUM all-reduce (RCCL on GPUs,
gloo in the CPU tests); being integer it is exact and independent of rank order.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def environment() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when not launched by it."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def initialise(backend: str) -> Tuple[int, int, int]:
    rank, local_rank, world = environment()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_read: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous read range of one rank: output order is reproducible by concatenation."""
    return (rank * n_read) // world, ((rank + 1) * n_read) // world


def allreduce_counters(words: torch.Tensor) -> torch.Tensor:
    """In place SUM over ranks of an int64 view of the counter buffer.  uint64 sums wrap like int64 sums."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(words, op=dist.ReduceOp.SUM)
    return words
