"""Multi-GPU: one process per GPU, ``torch.distributed`` (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards by RAY: parameters are replicated, every rank
renders / trains its own rays.

* render: no data-path collective at all; ``gather_rows`` (optional) collects
  the per-rank outputs when one process must own the whole image.
* train: each rank back-propagates its shard; ``allreduce_sum_`` adds the four
  flat parameter gradients (hash grid 52 MB + three small MLPs coalesced into
  one 57 kB buffer) once per step; every rank then applies the same Adam step.
  Over fully connected xGMI (7 links/GPU) RCCL's direct reduce-scatter +
  all-gather moves 52/8 MB per link per phase; nothing here forces a ring.
* the reference normalises its losses by N (or by the number of valid depth
  pixels); ``global_mean_scale`` turns local sums into terms of the global
  mean so that SUM-reduced gradients equal the single-process gradient.
* metrics: the 40x40 confusion matrix is all-reduced instead of the
  reference's all_gather of label maps (joint_train_lightning_net.py:666-667).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world() -> Tuple[int, int]:
    if is_dist():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """torchrun-style env (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_*)."""
    w = int(os.environ.get("WORLD_SIZE", "1"))
    r = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if w > 1 and not is_dist():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(lr)
            kw["device_id"] = torch.device("cuda", lr)
        dist.init_process_group(backend, **kw)
    return r, lr, w


def shard_range(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of n items for `rank`."""
    base, rem = divmod(n, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_round_robin(n: int, rank: int, world_size: int) -> List[int]:
    return list(range(rank, n, world_size))


def allreduce_sum_(tensors: Sequence[torch.Tensor], small_bytes: int = 1 << 20):
    """In-place SUM all-reduce.  Tensors below `small_bytes` are coalesced into
    one flat buffer (one collective for the three MLP gradients); large ones
    (the hash grid) go on their own, un-copied."""
    if not is_dist() or dist.get_world_size() == 1:
        return
    small = [t for t in tensors if t.numel() * t.element_size() < small_bytes]
    large = [t for t in tensors if t.numel() * t.element_size() >= small_bytes]
    handles = [dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
               for t in large]
    if small:
        flat = torch.cat([t.reshape(-1) for t in small])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        off = 0
        for t in small:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
    for h in handles:
        h.wait()


def allreduce_grads_(params: Iterable[torch.nn.Parameter]):
    grads = [p.grad for p in params if p.grad is not None]
    allreduce_sum_(grads)


def global_count(local_count: torch.Tensor) -> torch.Tensor:
    """Sum of a (scalar) count over ranks; float64 to stay exact."""
    c = local_count.detach().to(torch.float64).clone()
    if is_dist() and dist.get_world_size() > 1:
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return c


def gather_rows(local: torch.Tensor, sizes: Sequence[int], dst: int = 0):
    """Gather per-rank row blocks (ragged along dim 0) on `dst`; returns the
    concatenation there, None elsewhere."""
    rank, w = world()
    if w == 1:
        return local
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype,
                      device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(w)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)


def allreduce_confusion_(cm: torch.Tensor) -> torch.Tensor:
    if is_dist() and dist.get_world_size() > 1:
        dist.all_reduce(cm, op=dist.ReduceOp.SUM)
    return cm
