"""Adam for the NeRF parameter groups, backed by the HIP kernel
``ucsa_adam_step``.

Same constructor contract as ``torch.optim.Adam`` as the reference calls it
(nr4seg/lightning/joint_train_lightning_net.py:897-919): named param groups,
``lr``, ``betas``, ``eps``, per-group ``weight_decay`` (L2 folded into the
gradient, not AdamW).  Under ``torch.amp.GradScaler`` (the reference steps
its NeRF optimizer through one, :46,:509-513) it declares
``_step_supports_amp_scaling``: the scaler then hands over its scale and its
found-inf flag as device tensors instead of reading the flag back, and the
kernel (``ucsa_adam_step_scaled``) unscales / skips on the device -- no host
synchronisation per training step.

``ShardedHipAdam`` is the multi-GPU form: the hash-grid
gradient is reduce-scattered, every rank runs Adam on its 1/N slice with 1/N
of the moment buffers, and the updated slices are all-gathered.
``CollectiveGradScaler`` makes the scaler's found-inf flag collective (one
4-byte all-reduce), which the sharded step needs and the replicated one
tolerates."""
from __future__ import annotations

import torch

from .. import dist as udist
from .. import ops


class CollectiveGradScaler(torch.amp.GradScaler):
    """``torch.amp.GradScaler`` whose found-inf flag is the MAX over the
    ranks (what torch's own ShardedGradScaler does for FSDP): every rank then
    skips the same steps and keeps the same scale, also when each rank has
    only looked at its own, not yet reduced, gradients.  With one process it
    is exactly ``GradScaler``."""

    def _unscale_grads_(self, optimizer, inv_scale, found_inf, allow_fp16):
        out = super()._unscale_grads_(optimizer, inv_scale, found_inf,
                                      allow_fp16)
        if udist.active():
            for t in out.values():
                udist.allreduce_max_(t)
            optimizer._found_inf_is_collective = True
        return out


class HipAdam(torch.optim.Optimizer):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._step_supports_amp_scaling = True
        self._skipped = {}  # device -> int32[1]: steps skipped by the scaler

    @staticmethod
    def _require_gpu(p):
        if not p.is_cuda:
            raise RuntimeError("HipAdam updates GPU parameters only "
                               "(no CPU fallback)")

    def _scaler_state(self):
        """(grad_scale, found_inf, scaled): set by GradScaler.step() around
        the call (device tensors); grad_scale None = already unscaled."""
        grad_scale = getattr(self, "grad_scale", None)
        found_inf = getattr(self, "found_inf", None)
        return grad_scale, found_inf, found_inf is not None

    def _apply(self, p, g, m, v, step, group, grad_scale, found_inf, scaled,
               devices, extra_div=1.0):
        """One tensor (or slice): p, g, m, v are flat fp32 views of equal
        length.  ``extra_div``: the gradient is additionally divided by it
        (world size, when g is a SUM over ranks)."""
        b1, b2 = group["betas"]
        dev = p.device
        skipped = self._skipped.get(dev)
        if scaled or skipped is not None:
            if skipped is None:
                skipped = torch.zeros(1, dtype=torch.int32, device=dev)
                self._skipped[dev] = skipped
            if scaled:
                gs = (torch.ones(1, device=dev) if grad_scale is None
                      else grad_scale.to(dev, torch.float32).reshape(1))
                fi = found_inf.to(dev, torch.float32).reshape(1)
            else:  # plain step after scaled ones: keep the step count
                gs = torch.ones(1, device=dev)
                fi = torch.zeros(1, device=dev)
            if extra_div != 1.0:
                gs = gs * float(extra_div)
            ops.adam_step_scaled(p, g, m, v, step, group["lr"], b1, b2,
                                 group["eps"], group["weight_decay"], gs, fi,
                                 skipped)
            if scaled:
                devices.add((dev, fi))
        else:
            ops.adam_step(p, g, m, v, step, group["lr"], b1, b2, group["eps"],
                          group["weight_decay"],
                          inv_grad_scale=1.0 / float(extra_div))

    def _finish(self, devices):
        for dev, fi in {d: f for d, f in devices}.items():
            ops.adam_count_skipped(fi, self._skipped[dev])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grad_scale, found_inf, scaled = self._scaler_state()
        devices = set()
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                self._require_gpu(p)
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                self._apply(p.data, p.grad.contiguous(), st["exp_avg"],
                            st["exp_avg_sq"], st["step"], group, grad_scale,
                            found_inf, scaled, devices)
                # the kernel wrote through the raw pointer: tell autograd (and the
                # packed-weight cache keyed on ._version) that p changed
                torch.autograd.graph.increment_version(p)
        self._finish(devices)
        return loss


class ShardedHipAdam(HipAdam):
    """the NeRF optimizer of
    ``joint_train_lightning_net.py:897-919`` with its state and work sharded
    over the ranks of ``torch.distributed`` (RCCL over xGMI).

    ``step()`` contains the gradient collectives (do NOT all-reduce the
    gradients before it):

    * a parameter of >= ``shard_min_numel`` elements (the 13 M-element hash
      grid): ``reduce_scatter`` of its flat gradient (SUM), Adam on this
      rank's contiguous 1/N slice -- the two moment buffers exist for that
      slice only -- then ``all_gather`` of the updated slices into the
      replicated parameter.  The slice length is a multiple of 4 (16-byte
      aligned slices); the < 4N elements past the last slice are handled
      like a small parameter.
    * small parameters (the three MLPs, 14 k elements): one coalesced SUM
      all-reduce, the same Adam step on every rank.
    * ``average=True`` (DDP semantics) divides the summed gradients by N
      inside the Adam kernel (the unscale factor), not in a pass of its own.
    * ``comm_dtype`` (None | torch.float16 | torch.bfloat16): the gradient
      payload of the reduce-scatter is cast to it (halves the bytes on the
      links).  fp16 payloads are range-normalised by the collective max|g|
      first (``_reduce_scatter_low_precision``) so that neither a
      GradScaler-scaled value nor a sum over the ranks can overflow after
      the inf check.  Moments, parameters and the all-gather stay fp32.
    * under ``CollectiveGradScaler`` the found-inf flag is already the MAX
      over ranks when it arrives here; with a plain GradScaler it is
      MAX-reduced here (4 bytes) so that every rank skips the same step --
      the scaler's own copy then still differs, hence use the collective one.

    Per step and rank this moves (N-1)/N x 52 MB twice, like the ring
    all-reduce it replaces, but Adam reads/writes 28 B x 13 M / N instead of
    28 B x 13 M, and the moments take 2 x 52 MB / N.  With one process it
    is ``HipAdam``."""

    handles_collectives = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, shard_min_numel=1 << 20, average=True,
                 comm_dtype=None):
        super().__init__(params, lr=lr, betas=betas, eps=eps,
                         weight_decay=weight_decay)
        self.shard_min_numel = int(shard_min_numel)
        self.average = bool(average)
        self.comm_dtype = comm_dtype
        self.last_comm_bytes = 0

    def _reduce_scatter_low_precision(self, g, per, world):
        """SUM reduce-scatter of the fp32 gradient ``g`` with a 16-bit payload.

        fp16 cannot hold what the fp32 gradient may: under the GradScaler the
        values are scaled by up to 2^16 (one above 65504, or a sum over the
        ranks that passes it, would turn into inf AFTER the scaler's inf check
        ran on the fp32 values), and without a scaler small values flush to
        zero.  So the payload is range-normalised first: the MAX over the
        ranks of max|g| (one 4-byte all-reduce) is mapped to 2^14 / world --
        no element and no partial sum can overflow, and the largest
        gradients keep all of fp16's mantissa -- and the reduced slice is
        scaled back in fp32.  A non-finite max (an overflow the scaler has
        already flagged, on any rank) makes the factor 0: the payload is
        finite zeros and the step is skipped by found_inf as before.  bf16
        has fp32's range and is cast directly."""
        if self.comm_dtype == torch.float16:
            amax = torch.linalg.vector_norm(g, ord=float("inf")).reshape(1)
            amax = udist.allreduce_max_(amax)
            ok = torch.isfinite(amax) & (amax > 0)
            amax = torch.where(ok, amax, torch.ones_like(amax))
            # a power of two: the scaling itself is exact
            k = torch.exp2(torch.floor(torch.log2((16384.0 / world) / amax)))
            k = torch.where(ok, k, torch.zeros_like(k))
            gh = torch.nan_to_num(g * k, nan=0.0, posinf=0.0, neginf=0.0).to(torch.float16)
            inv = torch.where(ok, 1.0 / torch.where(ok, k, torch.ones_like(k)),
                              torch.zeros_like(k))
        else:
            gh, inv = g.to(self.comm_dtype), None
        oh = torch.empty(per, dtype=self.comm_dtype, device=g.device)
        udist.reduce_scatter_sum_(oh, gh)
        out = oh.float()
        return out if inv is None else out * inv

    @torch.no_grad()
    def step(self, closure=None):
        rank, world = udist.world()
        if not udist.active():      # one process, no (forced) group: plain HipAdam
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grad_scale, found_inf, scaled = self._scaler_state()
        if scaled and not getattr(self, "_found_inf_is_collective", False):
            found_inf = udist.allreduce_max_(found_inf.clone())
        # the flag describes ONE step (set by CollectiveGradScaler around it):
        # a later step driven by a plain GradScaler must reduce again
        self._found_inf_is_collective = False
        div = float(world) if self.average else 1.0
        devices = set()
        small = []  # (group, p_view, g_view, exp_avg, exp_avg_sq, step)
        large = []  # (group, p, pf, g_shard, state, per, body)
        comm = 0
        # what this rank's Adam owns of every sharded parameter (read by the
        # bench / the width-8 test): (numel, first, last + 1, first of the tail)
        self.last_shards = []
        # phase 1: every gradient collective of the large parameters
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                self._require_gpu(p)
                st = self.state[p]
                g = p.grad.contiguous().view(-1)
                pf = p.data.view(-1)
                n = g.numel()
                if not st:
                    st["step"] = 0
                st["step"] += 1
                if n < self.shard_min_numel:
                    if "exp_avg" not in st:
                        st["exp_avg"] = torch.zeros_like(pf)
                        st["exp_avg_sq"] = torch.zeros_like(pf)
                    small.append((group, pf, g, st["exp_avg"], st["exp_avg_sq"],
                                  st["step"]))
                    torch.autograd.graph.increment_version(p)
                    continue
                per = (n // world) // 4 * 4   # slices stay 16-byte aligned
                body = per * world
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros(per, device=p.device)
                    st["exp_avg_sq"] = torch.zeros(per, device=p.device)
                    st["tail_exp_avg"] = torch.zeros(n - body, device=p.device)
                    st["tail_exp_avg_sq"] = torch.zeros(n - body, device=p.device)
                if self.comm_dtype is not None:
                    g_shard = self._reduce_scatter_low_precision(g[:body], per, world)
                    comm += body * torch.finfo(self.comm_dtype).bits // 8
                else:
                    g_shard = torch.empty(per, device=p.device)
                    udist.reduce_scatter_sum_(g_shard, g[:body])
                    comm += body * 4
                large.append((group, p, pf, g_shard, st, per, body))
                self.last_shards.append((n, rank * per, (rank + 1) * per, body))
                if n > body:
                    small.append((group, pf[body:], g[body:], st["tail_exp_avg"],
                                  st["tail_exp_avg_sq"], st["step"]))
        # phase 2: Adam on this rank's slices, all-gather of the result
        for group, p, pf, g_shard, st, per, body in large:
            p_shard = pf[rank * per:(rank + 1) * per]
            self._apply(p_shard, g_shard, st["exp_avg"], st["exp_avg_sq"],
                        st["step"], group, grad_scale, found_inf, scaled,
                        devices, extra_div=div)
            udist.all_gather_into_(pf[:body], p_shard.clone())
            comm += body * 4
            torch.autograd.graph.increment_version(p)
        if small:
            flat = torch.cat([g for (_, _, g, _, _, _) in small])
            udist.allreduce_sum_([flat], small_bytes=0)
            comm += flat.numel() * 4
            off = 0
            for group, pv, g, m, v, step in small:
                k = g.numel()
                # copies: the kernels need 16-byte-aligned-agnostic flat fp32
                # views; slices of `flat` are fine as they are contiguous
                self._apply(pv, flat[off:off + k], m, v, step, group, grad_scale,
                            found_inf, scaled, devices, extra_div=div)
                off += k
        self.last_comm_bytes = comm
        self._finish(devices)
        return loss
