"""Mirror of reference ``nr4seg/lightning/joint_train_lightning_net.py``.

The four hot-path methods keep their names, arguments and return values:
``get_rays_train`` (:108-157), ``forward_seg`` (:159-165),
``forward_nerf_train`` (:167-223), ``forward_nerf_test`` (:225-257); so do
``training_step_nerf`` (:473-513), ``training_step_joint`` (:363-471) and
``configure_optimizers`` (:876-921).  What they call is the HIP path:

* rays            -> ``ucsa_get_rays`` (pixel indices drawn with torch.randint
                     exactly like :141)
* NeRF render     -> ``SemanticNeRFNetwork.render`` (HIP pipeline + backward)
* NeRF losses     -> ``ucsa_nerf_loss`` (fused forward+gradient)
* softmax/argmax, CE-on-softmax -> ``ucsa_seg_tail``
* post-processing -> ``ucsa_semantic_postproc``
* mIoU            -> ``ucsa_confusion_matrix`` + reference formula
* NeRF optimizer  -> ``HipAdam`` (``ucsa_adam_step``), same two param groups.

Out of scope and therefore reduced: the colour-coded ``*_vis``
PNGs of ``predict_step``, the Visualizer and WandB logging (``self.log`` goes
to the JSONL logger of the thin Trainer).  PyTorch-Lightning itself is not
required: the class is a plain ``nn.Module`` with the few LightningModule
members the code uses.
"""
from __future__ import annotations

import os
import warnings
import random

import contextlib

import torch
import torch.nn as nn

from .. import dist as udist
from .. import losses as ulosses
from .. import ops
from ..nerf.network_tcnn_semantics import SemanticNeRFNetwork
from ..nerf.optim import CollectiveGradScaler, HipAdam, ShardedHipAdam
from ..network import DeepLabV3
from ..utils.metrics import SemanticsMeter


class JointTrainLightningNet(nn.Module):

    def __init__(self, exp, env):
        super().__init__()
        self.num_classes = exp["model"]["num_classes"]
        self.seg_model = DeepLabV3(exp["model"])
        nerf_cfg = exp.get("nerf", {})  # optional block, defaults = reference
        # The reference hard-codes cuda_ray=False (:29-35).  `nerf: {cuda_ray:
        # true}` trains and renders through the occupancy-grid marcher
        # instead: the density grid is refreshed every 16
        # NeRF steps and before every evaluation epoch.
        self.cuda_ray = bool(nerf_cfg.get("cuda_ray", False))
        self.dt_gamma = float(nerf_cfg.get("dt_gamma", 1.0 / 256))
        # far_closure: end every marched ray like run() does (needed to render
        # a field that was trained through run(); a field trained through the
        # marcher carries its own opacity)
        self.far_closure = bool(nerf_cfg.get("far_closure", False))
        self.nerf_model = SemanticNeRFNetwork(
            encoding="hashgrid", bound=4, cuda_ray=self.cuda_ray,
            density_scale=1, num_semantic_classes=self.num_classes,
            seed=exp.get("nerf_seed"))
        self.nerf_model.march_training = self.cuda_ray
        # `nerf: {precision: ...}`: arithmetic of the three MLPs in the no-grad
        # renders.  "f16x2" (default since round 4): fp32-grade on the f16 MFMA
        # pipe (two f16 terms per operand, the second scaled by 2^11, three
        # partial products, fp32 accumulation: the f32-input MFMA chain's error
        # against fp64; the RANGE of the reference's own fp16 nets, 65504);
        # "bf16x3": fp32-grade with fp32's range (three bf16 terms, six partial
        # products, ~11 % more time per view); "fp32": the f32-input MFMA (an
        # exact fmaf chain); "fp16": like tiny-cuda-nn (fp16 weights / layer
        # inputs, fp32 accumulate).  Training: see train_precision.
        self.nerf_model.precision = str(nerf_cfg.get("precision", "f16x2"))
        # `nerf: {fp16_table: true}` (with precision: fp16): the renders read the
        # hash grid from an fp16 copy of the table, as tiny-cuda-nn stores it
        self.nerf_model.fp16_table = bool(nerf_cfg.get("fp16_table", False))
        # `nerf: {train_precision: fp16}`: colour / semantics nets of the
        # training pass on f16 MFMA too; the GradScaler's scale (reference :46)
        # already protects the fp16 gradient operands, so no extra one
        # Default "bf16x3": the training forward's colour / semantics stage on
        # the split pair with the bf16x3 nets (fp32-grade, 1e-7 from the
        # f32-input MFMA chain that "fp32" selects); the backward is the same.
        self.nerf_model.train_precision = str(nerf_cfg.get("train_precision", "bf16x3"))
        # `nerf: {bwd_precision: fp32}`: keep the f32-input MFMA kernels for the
        # backward of the bf16x3 mode (default bf16x2: renderer_semantics.py)
        self.nerf_model.bwd_precision = str(nerf_cfg.get("bwd_precision", "bf16x2"))
        # round-4 switches of the default training mode (all default on; the
        # UCSA_* environment variables of renderer_semantics.py set the defaults):
        #   train_fwd_f16x2      forward nets as f16x2 instead of bf16x3
        #   grid_records_packed  8-byte packed bin records in the grid backward
        #   grid_bwd_merged      both density passes' grid backward in one call
        #   fused_train_calls    ucsa_render_fused_fwd / _bwd instead of staged calls
        for key in ("train_fwd_f16x2", "grid_records_packed", "grid_bwd_merged",
                    "fused_train_calls"):
            if key in nerf_cfg:
                setattr(self.nerf_model, key, bool(nerf_cfg[key]))
        self.nerf_model.f16_bwd_scale = float(nerf_cfg.get("f16_bwd_scale", 1.0))
        # `nerf: {h2_guard: off | weights | full}`: range guard of the f16x2 nets
        # (default weights: max|W| checked at every refreshed pack; full also the
        # activations of a sample of every no-grad render) -- out-of-range values
        # raise instead of turning into zeros (network_tcnn_semantics.py)
        if "h2_guard" in nerf_cfg:
            self.nerf_model.h2_guard = str(nerf_cfg["h2_guard"])
        # `model: {amp: bf16}` (optional; the reference trains DeepLab in fp32)
        # runs the segmentation network under bf16 autocast in channels_last
        # (MIOpen's fast path on MI355X: 49 -> 34 ms per 8-image train step)
        self.seg_amp = str(exp["model"].get("amp", "")).lower() == "bf16"
        # `model: {channels_last: true}`: fp32, NHWC layout (1x1 convolutions as
        # one GEMM, MIOpen's NHWC kernels without layout transposes)
        # Default ON: the fused BatchNorm (+ add) (+ ReLU) kernels
        # (network/fused_bn.py) work on channels-last activations; NCHW
        # (`channels_last: false`) runs the same modules through F.batch_norm.
        self.seg_channels_last = bool(exp["model"].get("channels_last", True))
        if self.seg_amp or self.seg_channels_last:
            self.seg_model = self.seg_model.to(memory_format=torch.channels_last)
        # NeRF-only steps replay the frozen segmentation forward as a HIP graph
        self.seg_graph = bool(exp["model"].get("seg_graph", True))
        self._seg_graphs = {}
        self._nerf_steps = 0
        self._grid_stale = True  # refresh the density grid before evaluating
        self.n_rays_train = int(nerf_cfg.get("n_rays", 4096))
        # "tile": the drawn pixels are handed to the renderer tile by tile
        # (ops.tile_order; default on the live path, where it makes the grid
        # backward 2x faster); "random": in the order drawn (default with
        # cuda_ray, where it measured no gain)
        self.ray_order = str(nerf_cfg.get(
            "ray_order", "random" if nerf_cfg.get("cuda_ray", False) else "tile"))
        self.num_steps = int(nerf_cfg.get("num_steps", 256))
        self.upsample_steps = int(nerf_cfg.get("upsample_steps", 256))

        self.weight_depth = ulosses.WEIGHT_DEPTH
        self.weight_semantics = ulosses.WEIGHT_SEMANTICS
        # found-inf flag MAX-reduced over the ranks (== GradScaler with one)
        self.nerf_scaler = CollectiveGradScaler("cuda", enabled=True)
        # multi-GPU: reduce-scatter + per-rank Adam slice + all-gather
        #  instead of all-reduce + replicated Adam;
        # `nerf: {sharded_optimizer: false}` keeps the replicated step,
        # `nerf: {grad_comm_dtype: fp16|bf16}` halves the gradient payload
        self.sharded_optimizer = bool(nerf_cfg.get("sharded_optimizer", True))
        self.grad_comm_dtype = {None: None, "": None, "fp32": None,
                                "fp16": torch.float16, "bf16": torch.bfloat16}[
            nerf_cfg.get("grad_comm_dtype")]
        self.automatic_optimization = False
        self.joint_train = False
        self.fix_nerf = exp.get("fix_nerf", False)
        self.root_new_scene = os.path.join(
            env.get("scannet", "."), str(exp["scenes"][-1]),
            str(exp.get("exp_name", "debug")))
        self.prev_scene_name = None
        names = ["train_nerf", "train_seg", "train_nerf_seg", "train_seg_nerf",
                 "val_seg", "train_val_seg", "test_nerf", "test_25k"]
        self._meter = {n: SemanticsMeter(number_classes=self.num_classes)
                       for n in names}
        self._exp, self._env = exp, env
        self._mode = "train"
        self._output_size = (240, 320)
        self._flip_p = 0.5
        self._degrees = 10
        self._jitter = dict(brightness=0.3, contrast=0.3, saturation=0.3,
                            hue=0.05)
        self._default_H = None
        self._default_W = None
        # thin-Trainer plumbing
        self.trainer = None
        self._optimizers = None
        self._logged_raw = {}

    # ---- LightningModule members used by the code --------------------------
    @property
    def current_epoch(self):
        return self.trainer.current_epoch if self.trainer else 0

    def log(self, name, value, **kw):
        # tensors stay on the device (no read-back per training step);
        # `logged` converts on access, the logger in batches
        self._logged_raw[name] = value.detach() if torch.is_tensor(value) else value
        if self.trainer is not None and udist.world()[0] == 0:
            self.trainer.logger.log(name, value, self.trainer.global_step)

    @property
    def logged(self):
        """Last logged value of every metric, as floats."""
        return {k: float(v) for k, v in self._logged_raw.items()}

    def optimizers(self, use_pl_optimizer=False):
        if self._optimizers is None:
            self._optimizers = self.configure_optimizers()
        return self._optimizers

    def manual_backward(self, loss):
        loss.backward()

    def on_train_epoch_start(self):
        self._mode = "train"

    # ---- a1 ------------------------------------------------------------------
    @torch.no_grad()
    def get_rays_train(self, batch, bs, N=None):
        """reference :108-157 -> rays_o, rays_d [1,N,3], direction_norms
        [1,N,1], inds [1,N]."""
        N = self.n_rays_train if N is None else N
        # (a slice, not `[[bs], ...]`: indexing a device tensor with a Python list
        # uploads the index with a blocking copy -- one device synchronisation per
        # training step, 15 ms of host time each in the continual loop's profile)
        poses = batch["pose"][bs:bs + 1]
        device = poses.device
        fx, fy, cx, cy = [float(v) for v in batch["intrinsics"][bs]]
        H, W = int(batch["H"][bs]), int(batch["W"][bs])
        N = min(N, H * W)
        inds = torch.randint(0, H * W, size=[N], device=device)  # may duplicate
        if self.ray_order == "tile":  # same pixels, neighbours adjacent
            inds = ops.tile_order(inds, W, H=H)
        o, d, n = ops.get_rays(poses, (fx, fy, cx, cy), H, W, inds=inds)
        return o, d, n, inds.expand([1, N])

    def _seg_logits(self, image):
        if not self.seg_amp:
            if self.seg_channels_last:
                image = image.contiguous(memory_format=torch.channels_last)
            return self.seg_model(image)["out"]
        image = image.contiguous(memory_format=torch.channels_last)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = self.seg_model(image)["out"]
        return out.float()

    # ---- a14 / a15 -------------------------------------------------------------
    def forward_seg(self, batch, image=None):
        """reference :159-165 -> {"seg_semantics" argmax, "seg_semantics_raw"
        softmax probabilities, "seg_logits" (extra: lets seg_loss fuse the
        double softmax with its backward)}."""
        if image is None:
            image = batch["img"]
        logits = self._seg_logits(image)
        with torch.no_grad():
            tail = ops.seg_tail(logits.detach().contiguous(), None,
                                want_prob=True)
        return {"seg_semantics": tail["argmax"],
                "seg_semantics_raw": tail["prob"], "seg_logits": logits}

    @torch.no_grad()
    def forward_seg_frozen(self, batch):
        """``forward_seg`` of the NeRF-only step (reference :478-481: eval mode,
        no gradient) replayed as a HIP graph.  One DeepLabV3-R101 forward is
        ~350 launches of a few microseconds each -- on one image the host, not
        the GPU, sets its pace (measured 5.6 ms of Python/launch time, plus
        2.4 ms for walking the module tree in ``.eval()`` / ``.train()``) -- so
        it is captured once per input shape (``torch.cuda.CUDAGraph``, i.e.
        hipGraph) and replayed; parameters are read from their own storage, so
        optimizer steps in between are seen.  ``model: {seg_graph: false}`` or
        any capture failure falls back to the eager call."""
        image = batch["img"]
        if not self.seg_graph or not image.is_cuda:
            return self._forward_seg_eval_eager(batch)
        first = next(self.seg_model.parameters())
        key = (tuple(image.shape), image.dtype, self.seg_amp, first.data_ptr())
        entry = self._seg_graphs.get(key)
        if entry is None:
            entry = self._capture_seg_graph(image)
            self._seg_graphs[key] = entry
        if entry is False:
            return self._forward_seg_eval_eager(batch)
        graph, static_in, out = entry
        static_in.copy_(image)
        graph.replay()
        return out  # static buffers: consumed before the next replay

    def _forward_seg_eval_eager(self, batch):
        self.seg_model.eval()
        try:
            return self.forward_seg(batch)
        finally:
            self.seg_model.train()

    def _capture_seg_graph(self, image):
        was_training = self.seg_model.training
        self.seg_model.eval()
        try:
            static_in = image.clone()
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):  # MIOpen picks its kernels here
                for _ in range(3):
                    self.forward_seg(None, static_in)
            cur.wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.forward_seg(None, static_in)
            out = {k: v for k, v in out.items() if k != "seg_logits"}
            return graph, static_in, out
        except Exception as e:  # noqa: BLE001 -- any capture problem: eager
            warnings.warn(f"segmentation forward not captured as a HIP graph "
                          f"({type(e).__name__}: {e}); running it eagerly")
            torch.cuda.synchronize()
            return False
        finally:
            self.seg_model.train(was_training)

    # ---- a3-a12 ----------------------------------------------------------------
    def forward_nerf_train(self, batch, output_seg, bs):
        """reference :167-223 -> (loss_color, loss_semantics, loss_depth).
        Where the reference returns ``loss_semantics = None`` (every ray with
        invalid semantics, :212-213) this returns a zero with zero gradient:
        same total and gradients, without a host read-back per step
        (``losses.nerf_losses``)."""
        rays_o, rays_d, direction_norms, inds = self.get_rays_train(batch, bs)
        images = batch["img_fp16"][bs:bs + 1]
        label_nerf = output_seg["seg_semantics"][bs:bs + 1]
        depths = batch["depth"][bs:bs + 1]
        uom = batch["one_m_to_scene_uom"][bs]
        uom = float(uom)
        B, C, H, W = images.shape
        self._default_H, self._default_W = H, W
        # the reference gathers from the fp16 image / depth (:180-189)
        gt_rgb = torch.gather(images.reshape(B, C, -1).permute(0, 2, 1), 1,
                              torch.stack(C * [inds], -1))
        labels = torch.gather(label_nerf.reshape(B, -1), 1, inds)
        gt_depth = torch.gather(depths.reshape(B, -1), 1, inds)
        if self.cuda_ray:
            if self.nerf_model.refresh_due(self._nerf_steps):
                self.nerf_model.update_extra_state()
            self._nerf_steps += 1
            self._grid_stale = True
        outputs = self.nerf_model.render(
            rays_o, rays_d, direction_norms=direction_norms, staged=False,
            bg_color=None, perturb=True, epoch=self.current_epoch,
            num_steps=self.num_steps, upsample_steps=self.upsample_steps,
            **({"dt_gamma": self.dt_gamma} if self.cuda_ray else {}))
        return ulosses.nerf_losses(outputs["image"], outputs["semantics"],
                                   outputs["depth"], gt_rgb.float(), labels,
                                   gt_depth.float(), uom)

    @torch.no_grad()
    def forward_nerf_test(self, batch):
        """reference :225-257."""
        rays_o, rays_d = batch["rays_o"], batch["rays_d"]
        direction_norms = batch["direction_norms"]
        if batch["viewpoint_is_novel"][0]:
            B = len(batch["viewpoint_is_novel"])
            H, W = self._default_H, self._default_W
        else:
            B, C, H, W = batch["img"].shape
        if self.cuda_ray and self._grid_stale:
            self.nerf_model.update_extra_state()
            self._grid_stale = False
        outputs = self.nerf_model.render(
            rays_o, rays_d, direction_norms=direction_norms, staged=True,
            bg_color=1, perturb=False, num_steps=self.num_steps,
            upsample_steps=self.upsample_steps, image_width=W,
            **({"dt_gamma": self.dt_gamma, "far_closure": self.far_closure}
               if self.cuda_ray else {}))
        pred_rgb = outputs["image"].reshape(B, H, W, 3)
        sem = outputs["semantics"].reshape(B, H, W, self.num_classes)
        sem_norm, pred_sem = ops.semantic_postproc(sem)
        return {"nerf_rgb": pred_rgb.permute(0, 3, 1, 2),
                "nerf_semantics": pred_sem, "nerf_semantics_raw": sem_norm}

    # ---- rendered-image augmentation (reference :259-302; 8f rank 2) --------
    @staticmethod
    @torch.no_grad()
    def data_aug_static(img, label, degrees=10, flip_p=0.5,
                        jitter=(0.3, 0.3, 0.3, 0.05), output_size=(240, 320),
                        record=None):
        """img [3,H,W] in [0,1], label [H,W] -> ColorJitter (drawn order and
        factors, torchvision 0.12.0 ``ColorJitter.get_params``), rotate
        (``random.uniform`` angle, reference :266), RandomCrop / CenterCrop to
        ``output_size`` (identities at 240x320), flip (``torch.rand``, :292).
        One fused HIP kernel pair (``ucsa_augment``) instead of ~40 torch
        kernels."""
        order = torch.randperm(4).tolist()
        b, c, s, h = jitter
        draw = lambda lo, hi: float(torch.empty(1).uniform_(lo, hi))
        params = dict(order=order,
                      brightness=draw(max(0.0, 1 - b), 1 + b),
                      contrast=draw(max(0.0, 1 - c), 1 + c),
                      saturation=draw(max(0.0, 1 - s), 1 + s),
                      hue=draw(-h, h),
                      angle_deg=random.uniform(-degrees, degrees))
        H, W = img.shape[-2:]
        th, tw = output_size
        th, tw = min(th, H), min(tw, W)
        params["crop_i"] = 0 if H == th else int(torch.randint(0, H - th + 1, (1,)))
        params["crop_j"] = 0 if W == tw else int(torch.randint(0, W - tw + 1, (1,)))
        params["flip"] = bool(torch.rand(1) < flip_p)
        if record is not None:
            record.update(params)
        out, out_l = ops.augment(img[None], label[None], [params], (th, tw))
        return out[0], out_l[0]

    def data_aug(self, img, label):
        j = self._jitter
        return self.data_aug_static(
            img, label, self._degrees, self._flip_p,
            (j["brightness"], j["contrast"], j["saturation"], j["hue"]),
            self._output_size)

    # ---- training ------------------------------------------------------------
    def training_step(self, batch, batch_idx):
        if self.joint_train:
            self.training_step_joint(batch)
        else:
            self.training_step_nerf(batch)

    def _nerf_update(self, optimizer_nerf, loss_color, loss_semantics,
                     loss_depth, contributors=None):
        """reference :497-513.  Under torch.distributed every rank has drawn
        its own rays on its own frames (DDP semantics, reference
        scripts/train_joint.py:137-142): the gradients are averaged over the
        ranks -- inside ``ShardedHipAdam.step`` (reduce-scatter / all-gather),
        or by one all-reduce here for the replicated ``HipAdam``.

        ``contributors`` (joint training, N > 1): the number of ranks that
        hold a new-scene frame for this update.  The collectives divide the
        summed gradient by the world size, so the loss is pre-scaled by
        world / contributors: the update is the mean over the frames that
        exist.  A rank WITHOUT a frame calls ``_nerf_update_idle`` instead and
        enters the very same collectives with a zero gradient -- the number of
        collectives per step must not depend on what a rank's sampler drew
        (replayed old-scene frames, ``from_old_scene``)."""
        for nm, v in (("loss_nerf_rgb", loss_color), ("loss_depth", loss_depth),
                      ("loss_nerf_semantics", loss_semantics)):
            if v is not None:
                self.log(f"{self._mode}/{nm}", v.detach())
        total = ulosses.nerf_total_loss(loss_color, loss_semantics, loss_depth)
        self._nerf_backward_and_step(optimizer_nerf, total, contributors)

    def _nerf_update_idle(self, optimizer_nerf, contributors):
        """This rank has no new-scene frame for the update the others make: a
        zero loss that depends on every NeRF parameter gives dense zero
        gradients, and the step runs through the same scaler / optimizer /
        collective sequence as on the contributing ranks."""
        total = None
        for p in self.nerf_model.parameters():
            if p.requires_grad:
                z = p.reshape(-1)[:1].sum() * 0.0
                total = z if total is None else total + z
        self._nerf_backward_and_step(optimizer_nerf, total, contributors,
                                     dense_zero=True)

    def _nerf_backward_and_step(self, optimizer_nerf, total, contributors,
                                dense_zero=False):
        w = udist.world()[1]
        if contributors is not None and w > 1 and contributors != w:
            total = total * (float(w) / float(contributors))
        optimizer_nerf.zero_grad()
        total = self.nerf_scaler.scale(total)
        self.manual_backward(total)
        if dense_zero:
            for p in self.nerf_model.parameters():
                if p.requires_grad:
                    p.grad = torch.zeros_like(p) if p.grad is None else p.grad.zero_()
        if not getattr(optimizer_nerf, "handles_collectives", False):
            udist.average_grads_(self.nerf_model.parameters())
        self.nerf_scaler.step(optimizer_nerf)
        self.nerf_scaler.update()

    def _new_frame_counts(self, n_local):
        """How many new-scene frames every rank holds in this step (one small
        all_gather; [n_local] without torch.distributed)."""
        if not udist.active():
            return [int(n_local)]
        return udist.all_gather_ints(int(n_local), self._reduce_device())

    def training_step_nerf(self, batch):
        """reference :473-513."""
        optimizer_seg, optimizer_nerf = self.optimizers()
        output_seg = self.forward_seg_frozen(batch)
        for bs in range(batch["img"].shape[0]):
            lc, ls, ld = self.forward_nerf_train(batch, output_seg, bs)
            self._nerf_update(optimizer_nerf, lc, ls, ld)

    def training_step_joint(self, batch):
        """reference :363-471.

        One stream: the renders of the new-scene frames, the pseudo-label forward,
        the NeRF updates, then DeepLab's own step, in the reference's order.  (A
        two-stream schedule -- renders || pseudo-label forward, NeRF updates ||
        DeepLab step -- was built and measured in round 5: 177 -> 187 ms on the
        R-101 step, every kernel of either half fills the chip; removed in round
        6, docs/DESIGN_NOTEBOOK.md.)"""
        optimizer_seg, optimizer_nerf = self.optimizers()
        batch_old, batch_new, batch_cl = batch
        if batch_new is not None:
            with torch.no_grad():
                self.nerf_model.eval()
                output_nerf = self.forward_nerf_test(batch_new)
                self.nerf_model.train()
        # NeRF updates: one per new-scene frame (reference :381-393).  Under
        # torch.distributed the ranks' samplers mix replayed old-scene frames
        # in, so a rank may hold 0..B new frames: every rank runs
        # max-over-ranks updates, idle ones contribute zero gradients.
        n_new = 0 if (self.fix_nerf or batch_new is None) else int(batch_new["img"].shape[0])
        counts = [0] if self.fix_nerf else self._new_frame_counts(n_new)
        if n_new > 0:
            self.seg_model.eval()
            if batch_new["img"].shape[0] > 1:  # BN trains only when B > 1
                for m in self.seg_model.modules():
                    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                        m.train()
            with torch.no_grad():
                output_seg = self.forward_seg(batch_new)
            self.seg_model.train()
        for bs in range(max(counts)):
            contributors = sum(1 for c in counts if c > bs)
            if bs < n_new:
                lc, ls, ld = self.forward_nerf_train(batch_new, output_seg, bs)
                self._nerf_update(optimizer_nerf, lc, ls, ld, contributors)
            else:
                self._nerf_update_idle(optimizer_nerf, contributors)
        with torch.no_grad():
            rgb_seg = label_seg = None
            if batch_new is not None:
                aug = [self.data_aug(output_nerf["nerf_rgb"][bs],
                                     output_nerf["nerf_semantics"][bs])
                       for bs in range(batch_new["img"].shape[0])]
                rgb_seg = torch.stack([a[0] for a in aug], dim=0)
                label_seg = torch.stack([a[1] for a in aug], dim=0)
            if batch_old is not None:
                o_rgb, o_lab = batch_old["img"], batch_old["nerf_label"]
                rgb_seg = o_rgb if rgb_seg is None else torch.cat([rgb_seg, o_rgb], 0)
                label_seg = o_lab if label_seg is None else torch.cat([label_seg, o_lab], 0)
            if batch_cl is not None:
                rimg = batch_cl["replay_img"]
                _, _, C, H, W = rimg.shape
                rgb_seg = torch.cat([rgb_seg, rimg.reshape(-1, C, H, W)], 0)
                label_seg = torch.cat(
                    [label_seg, batch_cl["replay_label"].reshape(-1, H, W)], 0)
        # (round 6: this forward + loss + backward was also captured as ONE HIP graph per
        # batch shape and measured on the continual loop -- 778 replays, 234.7 s against
        # 233.8 s eager: the joint step at batch 2 is not paced by DeepLab's launches;
        # removed again, profiles/r06_cfg5.json)
        logits = self._seg_logits(rgb_seg)
        loss = ulosses.seg_loss(logits, label_seg)  # CE on softmax (:456-458)
        optimizer_seg.zero_grad()
        self.manual_backward(loss)
        udist.average_grads_(self.seg_model.parameters())
        optimizer_seg.step()
        self.log(f"{self._mode}/loss_seg", loss.detach())

    def on_train_epoch_end(self):
        for net_name in ["seg", "nerf", "nerf_seg", "seg_nerf"]:
            m = self._meter[f"train_{net_name}"]
            if m.conf_mat is not None:
                m_iou, total_acc, m_acc = m.measure()
                self.log(f"train/{net_name}_total_accuracy", total_acc)
                self.log(f"train/{net_name}_mean_accuracy", m_acc)
                self.log(f"train/{net_name}_mean_IoU", m_iou)
                m.clear()

    # ---- validation (:541-646) ------------------------------------------------
    def on_validation_epoch_start(self):
        udist.broadcast_buffers_(self.seg_model)   # rank 0's BN statistics
        self._mode = "val"
        self._meter["val_seg"].clear()
        self._meter["train_val_seg"].clear()

    def validation_step(self, batch, batch_idx, dataloader_idx=0):
        output_seg = self.forward_seg(batch)
        mode = "val" if dataloader_idx == 0 else "train_val"
        self.prev_scene_name = batch["current_scene_name"][0]
        self._meter[f"{mode}_seg"].update(output_seg["seg_semantics"],
                                          batch["label"])
        loss = ops.seg_tail(output_seg["seg_logits"].contiguous(),
                            batch["label"], want_prob=False)["loss"]
        self.log(f"{self._mode}/loss", loss)
        return loss

    def on_validation_epoch_end(self):
        out = {}
        for mode in ("val", "train_val"):
            m = self._meter[f"{mode}_seg"]
            self._reduce_meter(m)   # every rank enters, frames or not
            if m.conf_mat is None:
                continue
            m_iou, total_acc, m_acc = m.measure()
            tag = self.prev_scene_name
            self.log(f"{mode}/seg_total_accuracy_{tag}", total_acc)
            self.log(f"{mode}/seg_mean_accuracy_{tag}", m_acc)
            self.log(f"{mode}/seg_mean_IoU_{tag}", m_iou)
            out[f"{mode}_mIoU"] = m_iou
            m.clear()
        self.prev_scene_name = None
        return out

    # ---- test (:648-693) --------------------------------------------------------
    def on_test_epoch_start(self):
        udist.broadcast_buffers_(self.seg_model)   # rank 0's BN statistics
        self._mode = "test"
        self._meter["test_nerf"].clear()
        self._meter["test_25k"].clear()
        self._psnr = []

    def test_step(self, batch, batch_idx, dataloader_idx=0):
        if dataloader_idx == 0:
            out = self.forward_nerf_test(batch)
            self._meter["test_nerf"].update(out["nerf_semantics"], batch["label"])
            mse = torch.mean((out["nerf_rgb"] - batch["img"]) ** 2)
            # (kept on the device: a float() here is one synchronisation per frame)
            self._psnr.append((-10.0 * torch.log10(mse)).detach().double().reshape(1))
        else:
            tail = ops.seg_tail(self.seg_model(batch["img"])["out"].contiguous(),
                                None, want_prob=False)
            self._meter["test_25k"].update(tail["argmax"], batch["label"])

    def on_test_epoch_end(self):
        out = {}
        for net_name in ["nerf", "25k"]:
            m = self._meter[f"test_{net_name}"]
            self._reduce_meter(m)   # every rank enters, frames or not
            if m.conf_mat is not None:
                m_iou, total_acc, m_acc = m.measure()
                self.log(f"test/{net_name}_total_accuracy", total_acc)
                self.log(f"test/{net_name}_mean_accuracy", m_acc)
                self.log(f"test/{net_name}_mean_IoU", m_iou)
                out[f"test_{net_name}_mIoU"] = m_iou
                m.clear()
        psnr_sum = float(torch.cat(self._psnr).sum()) if self._psnr else 0.0
        tot = torch.tensor([psnr_sum, float(len(self._psnr))], dtype=torch.float64)
        if udist.active():  # the ranks evaluated disjoint frames
            tot = udist.allreduce_sum_tensor(tot.to(self._reduce_device()))
        if float(tot[1]) > 0:
            out["test_nerf_PSNR"] = float(tot[0] / tot[1])
            self.log("test/nerf_PSNR", out["test_nerf_PSNR"])
        return out

    def _reduce_device(self):
        import torch.distributed as tdist
        return (next(self.parameters()).device
                if tdist.get_backend() == "nccl" else torch.device("cpu"))

    def _reduce_meter(self, m):
        """Sum the 40x40 confusion matrix over the ranks (each evaluated its
        own frames) instead of the reference's all_gather of label maps
        (:666-667)."""
        if udist.active():
            import numpy as np
            local = (m.conf_mat if m.conf_mat is not None else
                     np.zeros((m.number_classes, m.number_classes), dtype=np.int64))
            cm = torch.from_numpy(np.ascontiguousarray(local)).to(self._reduce_device())
            cm = udist.allreduce_confusion_(cm).cpu().numpy()
            # a meter nobody fed stays empty (measure() is skipped for it)
            m.conf_mat = cm if (m.conf_mat is not None or cm.any()) else None

    # ---- predict (:695-782) ---------------------------------------------------
    # Returns the tensors; with ``predict_to_disk`` (set by train_joint when the
    # data come from a ScanNet-layout root) also writes the PNGs the next
    # stage's replay reads (``ScanNetNGPJoint``): nerf_image (RGB), nerf_label
    # and seg_label (class id + 1, 0 = unknown).  The colour-coded *_vis copies
    # of the reference are visualisation only and are not written.
    predict_to_disk = False

    def on_predict_epoch_start(self):
        udist.broadcast_buffers_(self.seg_model)   # rank 0's BN statistics
        self._mode = "predict"
        if self.predict_to_disk:
            for sub in ("", "novel_viewpoints"):
                for name in ("nerf_image", "nerf_label", "seg_label"):
                    os.makedirs(os.path.join(self.root_new_scene, sub, name),
                                exist_ok=True)

    def predict_step(self, batch, batch_idx, dataloader_idx=0):
        out = self.forward_nerf_test(batch)
        novel = bool(batch["viewpoint_is_novel"][0])
        if novel:
            seg = self.forward_seg(batch, out["nerf_rgb"].contiguous())
        else:
            seg = self.forward_seg(batch)
        res = {"nerf_image": out["nerf_rgb"],
               "nerf_label": out["nerf_semantics"] + 1,  # +1 when saved (:763)
               "seg_label": seg["seg_semantics"] + 1,
               "index": batch["current_index"]}
        if self.predict_to_disk:
            from PIL import Image
            import numpy as np
            sub = "novel_viewpoints" if novel else ""
            # one read-back per batch and array (quantised on the device), not three
            # per frame
            rgb8 = (res["nerf_image"].permute(0, 2, 3, 1).detach() * 255).to(torch.uint8).cpu().numpy()
            lab8 = {name: res[name].detach().to(torch.uint8).cpu().numpy()
                    for name in ("nerf_label", "seg_label")}
            for i, idx in enumerate(batch["current_index"]):
                Image.fromarray(rgb8[i]).save(os.path.join(
                    self.root_new_scene, sub, "nerf_image", idx + ".png"))
                for name in ("nerf_label", "seg_label"):
                    Image.fromarray(lab8[name][i]).save(os.path.join(
                        self.root_new_scene, sub, name, idx + ".png"))
        return res

    def on_predict_epoch_end(self):
        if self.predict_to_disk:
            # the PNGs above may replace files the decode cache has seen (same size,
            # same mtime tick): ADVICE r5
            from ..dataset.scannet_ngp_joint import decode_cache
            decode_cache().invalidate()
        return None

    # ---- optimizers (:876-921) ---------------------------------------------------
    def configure_optimizers(self):
        name = self._exp["optimizer"]["name"]
        lr_seg = self._exp["optimizer"]["lr_seg"]
        params = list(self.seg_model.parameters())
        if name == "Adam":
            # torch.optim.Adam as the reference configures it (:876-896); on the
            # GPU its fused implementation (one multi-tensor kernel chain for
            # the 58.6 M parameters instead of ~10 passes: 1.7 -> 0.6 ms per
            # step), same update rule
            fused = bool(params) and params[0].is_cuda
            optimizer_seg = torch.optim.Adam(params, lr=lr_seg, fused=fused)
        elif name == "SGD":
            cfg = self._exp["optimizer"]["sgd_cfg"]
            optimizer_seg = torch.optim.SGD(params, lr=lr_seg,
                                            momentum=cfg["momentum"],
                                            weight_decay=cfg["weight_decay"])
        elif name == "Adadelta":
            optimizer_seg = torch.optim.Adadelta(params, lr=lr_seg)
        elif name == "RMSprop":
            optimizer_seg = torch.optim.RMSprop(params, momentum=0.9, lr=lr_seg)
        else:
            raise ValueError(name)
        lr_nerf = self._exp["optimizer"]["lr_nerf"]
        sharded = self.sharded_optimizer and udist.active()
        ctor = ShardedHipAdam if sharded else HipAdam
        extra = {"comm_dtype": self.grad_comm_dtype} if sharded else {}
        optimizer_nerf = ctor(
            [{"name": "encoding",
              "params": list(self.nerf_model.encoder.parameters())},
             {"name": "net",
              "params": list(self.nerf_model.sigma_net.parameters()) +
              list(self.nerf_model.color_net.parameters()) +
              list(self.nerf_model.semantics_net.parameters()),
              "weight_decay": 1e-6}],
            lr=lr_nerf, betas=(0.9, 0.99), eps=1e-15, **extra)
        return optimizer_seg, optimizer_nerf
