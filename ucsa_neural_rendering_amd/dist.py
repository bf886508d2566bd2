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
* loss normalisation, two documented modes:
  - DDP semantics (what the reference's Lightning DDP does and what the
    module / ``bench.py --mode train`` use): every rank draws its OWN rays,
    normalises its loss terms by its local counts, gradients are SUMMED and
    divided by the world size (``average_grads_``) -- the mean of per-rank
    means.
  - one shared batch split over the ranks (``shard_range``):
    ``global_mean_scale`` returns the factors (n_local/n_global for the
    colour and semantics terms, valid_local/valid_global for the depth term)
    that turn the locally normalised terms into the terms of the GLOBAL mean,
    so that SUM-reduced gradients equal the single-process gradient exactly.
* sharded optimizer (``nerf.optim.ShardedHipAdam``):
  ``reduce_scatter_sum_`` of the hash-grid gradient, Adam on each rank's 1/N
  slice, ``all_gather_into_`` of the updated slice.
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


def forced() -> bool:
    """``UCSA_FORCE_DIST=1``: run the distributed code path even with ONE
    rank -- ``init_from_env`` / ``bench.py`` create a world-size-1 process
    group (``nccl`` = RCCL on a GPU) and every collective below is issued
    instead of short-circuited.  A 1-GPU box can execute the RCCL branches
    (reduce_scatter_tensor / all_gather_into_tensor on device tensors,
    ``device_id=`` init, all_gather_object) this way; results equal the
    non-distributed run bit for bit (tests/test_gpu_dist.py)."""
    return os.environ.get("UCSA_FORCE_DIST", "") not in ("", "0")


def active() -> bool:
    """Collectives are issued: a process group exists and it has more than
    one rank, or ``forced()``."""
    return is_dist() and (dist.get_world_size() > 1 or forced())


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """torchrun-style env (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_*)."""
    w = int(os.environ.get("WORLD_SIZE", "1"))
    r = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if (w > 1 or forced()) and not is_dist():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if w == 1 and "MASTER_PORT" not in os.environ:
            import socket
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            s.close()
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(lr)
            kw["device_id"] = torch.device("cuda", lr)
        dist.init_process_group(backend, rank=r, world_size=w, **kw)
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
    if not active():
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


def broadcast_parameters_(module: torch.nn.Module, src: int = 0):
    """Every rank takes rank `src`'s parameters and buffers (what DDP does at
    construction).  Needed whenever the replicas were produced by a
    non-deterministic computation -- e.g. a pre-training whose hash-grid
    gradient uses float atomics -- before they are trained data-parallel."""
    if not active():
        return
    gloo_cuda = dist.get_backend() == "gloo"
    for t in list(module.parameters()) + list(module.buffers()):
        if gloo_cuda and t.is_cuda:
            h = t.detach().cpu()
            dist.broadcast(h, src=src)
            t.data.copy_(h)
        else:
            dist.broadcast(t.data, src=src)
        if isinstance(t, torch.nn.Parameter):
            torch.autograd.graph.increment_version(t)


def broadcast_buffers_(module: torch.nn.Module, src: int = 0):
    """Rank `src`'s buffers (DeepLab's BatchNorm running statistics) on every
    rank.  Lightning DDP re-broadcasts buffers from rank 0 at every forward
    (``broadcast_buffers=True``); here the ranks' statistics are allowed to
    drift during a training epoch and are re-aligned before every
    evaluation / predict pass, so that sharded metrics and the written
    pseudo-labels all come from the model rank 0 saves."""
    if not active():
        return
    gloo_cuda = dist.get_backend() == "gloo"
    for t in module.buffers():
        if gloo_cuda and t.is_cuda:
            h = t.detach().cpu()
            dist.broadcast(h, src=src)
            t.data.copy_(h)
        else:
            dist.broadcast(t.data, src=src)


def average_grads_(params: Iterable[torch.nn.Parameter]):
    """DDP semantics: SUM all-reduce of the gradients, then / world."""
    params = [p for p in params if p.grad is not None]
    _, w = world()
    if not active():
        return
    allreduce_sum_([p.grad for p in params])
    for p in params:
        p.grad.div_(w)


def reduce_scatter_sum_(out: torch.Tensor, inp: torch.Tensor):
    """out[per] = sum over ranks of inp[rank*per:(rank+1)*per] (1-D,
    inp.numel() == world * out.numel())."""
    if dist.get_backend() == "gloo" and inp.is_cuda:
        # gloo (the 1-GPU test hook) has no reduce-scatter for device tensors:
        # all-reduce a copy and keep this rank's slice.  RCCL takes the direct
        # path below.
        tmp = inp.clone()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM)
        r = dist.get_rank()
        out.copy_(tmp[r * out.numel():(r + 1) * out.numel()])
        return
    dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM)


def all_gather_into_(out: torch.Tensor, inp: torch.Tensor):
    """out[rank*per:(rank+1)*per] = inp of that rank (1-D)."""
    if dist.get_backend() == "gloo" and inp.is_cuda:
        parts = [torch.empty_like(inp) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, inp)
        out.copy_(torch.cat(parts))
        return
    dist.all_gather_into_tensor(out, inp)


def allreduce_sum_tensor(t: torch.Tensor) -> torch.Tensor:
    if active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


class RankShardSampler(torch.utils.data.Sampler):
    """Evaluation / predict loaders under torch.distributed: rank r takes
    items r, r+W, r+2W, ... in order -- no padding, no duplicates (a
    DistributedSampler would repeat frames to equalise the ranks, which would
    count them twice in the confusion matrix)."""

    def __init__(self, n: int, rank: int, world_size: int):
        self.idx = list(range(rank, n, world_size))

    def __iter__(self):
        return iter(self.idx)

    def __len__(self):
        return len(self.idx)


def allreduce_max_(t: torch.Tensor) -> torch.Tensor:
    if active():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def all_gather_ints(value: int, device) -> List[int]:
    """Every rank's integer, in rank order (one 8-byte-per-rank all_gather and
    a host read-back: used once per joint training step to agree on the
    number of NeRF updates, see ``training_step_joint``)."""
    if not active():
        return [int(value)]
    mine = torch.tensor([int(value)], dtype=torch.int64, device=device)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [int(t.item()) for t in parts]


def global_mean_scale(n_local: int, valid_local: torch.Tensor):
    """One shared batch split over the ranks: factors that turn locally
    normalised loss terms into terms of the global mean.  ``valid_local`` is
    this rank's number of valid depth pixels (a tensor, stays on its device:
    one 2-element all-reduce, no host read-back).  Returns (scale_mean,
    scale_depth) as 0-d tensors: multiply the colour / semantics terms by the
    first and the depth term by the second, then SUM-reduce the gradients
    (no division by the world size)."""
    v = torch.stack([torch.as_tensor(float(n_local), dtype=torch.float64,
                                     device=valid_local.device),
                     valid_local.detach().to(torch.float64).reshape(())])
    tot = v.clone()
    if active():
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    sc = (v / tot.clamp_min(1.0)).to(torch.float32)
    return sc[0], sc[1]


def global_count(local_count: torch.Tensor) -> torch.Tensor:
    """Sum of a (scalar) count over ranks; float64 to stay exact."""
    c = local_count.detach().to(torch.float64).clone()
    if active():
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return c


def gather_rows(local: torch.Tensor, sizes: Sequence[int], dst: int = 0):
    """Gather per-rank row blocks (ragged along dim 0) on `dst`; returns the
    concatenation there, None elsewhere."""
    rank, w = world()
    if not active():
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
    if active():
        dist.all_reduce(cm, op=dist.ReduceOp.SUM)
    return cm
