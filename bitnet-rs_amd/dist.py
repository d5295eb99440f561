"""Multi-GPU plumbing for bench.py: one process per GPU, torch.distributed
(backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).  No data-path collective:
batch-1 decode does not shard, so ranks are independent replicas and the
only communication is the barrier around the timed region and the max-over-ranks of the
elapsed time.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass


@dataclass
class Rank:
    world: int
    rank: int
    local_rank: int
    backend: str | None


def init(backend: str | None = None) -> Rank:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return Rank(1, 0, local_rank, None)
    import torch
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's driver only supports dmabuf IPC (RCCL peer buffers)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    backend = os.environ.get("BITNET_DIST_BACKEND", backend)  # rehearsals: gloo on a box with fewer GPUs than ranks
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif torch.cuda.is_available():
        local_rank %= max(1, torch.cuda.device_count())
        dist.init_process_group(backend)
    else:
        dist.init_process_group(backend)
    return Rank(world, rank, local_rank, backend)


def barrier(r: Rank) -> None:
    import torch

    if r.world > 1:
        import torch.distributed as dist

        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(r: Rank, value: float) -> float:
    if r.world == 1:
        return value
    import torch
    import torch.distributed as dist

    dev = "cuda" if r.backend == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed_region(r: Rank, fn) -> float:
    """barrier + synchronize on both sides of fn(); returns the MAX elapsed seconds over ranks."""
    barrier(r)
    t0 = time.perf_counter()
    fn()
    import torch

    if torch.cuda.is_available():
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier(r)
    return max_over_ranks(r, elapsed)


def aggregate_throughput(r: Rank, units_per_rank: int, elapsed_max: float) -> float:
    """Whole-job units/s for independent replicas (weak scaling)."""
    return r.world * units_per_rank / elapsed_max


def shard_rows(n_rows: int, world: int, rank: int, multiple: int = 1):
    """Contiguous [lo, hi) slice of n_rows for this rank, boundaries on `multiple`
    (token-parallel / row-sharded prefill: SURVEY.md 8e; QK256 K-shards need 256-multiples)."""
    units = -(-n_rows // multiple)
    lo_u = (units * rank) // world
    hi_u = (units * (rank + 1)) // world
    return min(n_rows, lo_u * multiple), min(n_rows, hi_u * multiple)


def finalize(r: Rank) -> None:
    if r.world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()
