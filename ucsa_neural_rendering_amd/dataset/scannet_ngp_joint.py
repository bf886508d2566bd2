"""Mirror of reference ``nr4seg/dataset/scannet_ngp_joint.py`` (SURVEY 8f
rank 3): the per-scene dataset that feeds the hot path -- same constructor,
same on-disk layout (``<root>/<scene>/transforms_train.json`` with ``h, w,
fl_x, fl_y, cx, cy, one_m_to_scene_uom, frames[file_path, label_path,
transform_matrix]``, ``depth/<stem>.png`` in millimetres, generated
``<exp_name>/[novel_viewpoints/]nerf_{image,label}/<stem>.png``), same 80/20
split, replay selection, Slerp novel viewpoints and item dictionary
(:320-458), same ``collate`` (:460-495).

What differs, deliberately:
  * rays are generated on the GPU by ``ucsa_get_rays`` from the pose
    (reference :417-419 / :380-382 call the per-item CPU ``get_rays``), and
    every tensor of an item is placed on ``device``;
  * the replay augmentation (reference ``helper.AugmentationList.apply``,
    helper.py:157-232) runs in ``ucsa_augment`` with the same draws;
  * images are decoded with PIL instead of cv2 (not installed).  Files whose
    size already is ``output_size`` -- what the reference's preprocessing
    writes -- decode to identical arrays; other sizes are resampled with
    PIL's BOX (images, cv2.INTER_AREA's equivalent for integer factors) and
    NEAREST (labels, depth) filters.
"""
from __future__ import annotations

import json
import os
import random
import re
import threading
from collections import OrderedDict, defaultdict

import numpy as np
import torch
from torch.utils.data import Dataset

from .. import dist as udist
from .. import ops
from .ngp_utils import get_rays, nerf_matrix_to_ngp

__all__ = ["ScanNetNGPJoint", "DecodeCache", "decode_cache"]

_TEN_SCENES = [f"scene{i:04d}_00" for i in range(10)]


def _pil():
    from PIL import Image
    return Image


class DecodeCache:
    """Decoded frames, kept on the host between epochs AND between the stages of
    the continual loop (round 5).

    The reference decodes every PNG again each time a frame is drawn (its
    DataLoader workers hide that); here the trainer's loop is synchronous and a
    640x480 frame costs ~4 + 3.5 + 6.5 ms of PIL decode (image, label, depth)
    -- 11 of the 63 s of a profiled three-stage run (profiles/r05_cfg5_hostprofile.txt),
    for frames that never change: a stage trains 10 + 2 epochs on the same ~100
    frames, and the next stages replay them.  One process-wide LRU keyed by
    (kind, path, output size, st_mtime_ns, st_size): a file that is rewritten --
    the pseudo-labels of the predict pass -- misses and is decoded again; files
    rewritten inside the filesystem's mtime granularity with the same byte size
    would not, so whoever rewrites label files calls ``invalidate()`` (the predict
    pass does: a generation counter is part of the key).  The entries are the
    cache's own tensors: callers must not modify them in place (they copy to the
    device or clone).  ``UCSA_DECODE_CACHE_MB`` (default 2048, DIVIDED by the number
    of ranks on this node; 0 switches it off) bounds it; 100 frames are ~0.8 GB.
    Page-locking the entries is opt-in (``UCSA_DECODE_CACHE_PIN=1``): 8 ranks x 2 GB
    of pinned host memory is not something to take silently."""

    def __init__(self, budget_mb=None):
        if budget_mb is None:
            budget_mb = float(os.environ.get("UCSA_DECODE_CACHE_MB", "2048"))
            local = int(os.environ.get("LOCAL_WORLD_SIZE", "0") or 0)
            if local > 1:
                budget_mb /= local
        self.pin = os.environ.get("UCSA_DECODE_CACHE_PIN", "0") == "1"
        self.budget = int(budget_mb * (1 << 20))
        self.generation = 0
        self.used = 0
        self.hits = self.misses = 0
        self._d = OrderedDict()
        self._lock = threading.Lock()      # (the trainer's prefetch thread)

    def lookup(self, kind, path, size, decode):
        """The cached host tensor of ``decode(path)`` -- read-only for the caller."""
        if self.budget <= 0:
            return decode(path)
        st = os.stat(path)
        key = (kind, os.path.abspath(path), tuple(size), st.st_mtime_ns, st.st_size,
               self.generation if self._rewritable(path) else 0)
        with self._lock:
            t = self._d.get(key)
            if t is not None:
                self._d.move_to_end(key)
                self.hits += 1
                return t
        t = decode(path)
        if self.pin and torch.cuda.is_available():
            try:
                t = t.pin_memory()
            except RuntimeError:
                pass
        n = t.numel() * t.element_size()
        with self._lock:
            self.misses += 1
            if n <= self.budget and key not in self._d:
                self._d[key] = t
                self.used += n
                while self.used > self.budget:
                    _, old = self._d.popitem(last=False)
                    self.used -= old.numel() * old.element_size()
        return t

    @staticmethod
    def _rewritable(path):
        # the predict pass's outputs (joint_train_lightning_net.predict_step); the
        # dataset's own frames never change
        return any(d in path for d in ("nerf_image", "nerf_label", "seg_label"))

    def invalidate(self):
        """The predict pass has (re)written its PNGs: later look-ups of those files
        decode again whatever their mtime says."""
        with self._lock:
            self.generation += 1

    def clear(self):
        with self._lock:
            self._d.clear()
            self.used = 0


_DECODE_CACHE = None


def decode_cache() -> DecodeCache:
    global _DECODE_CACHE
    if _DECODE_CACHE is None:
        _DECODE_CACHE = DecodeCache()
    return _DECODE_CACHE


class ScanNetNGPJoint(Dataset):

    def __init__(self, root, scene_list, mode="train", output_size=(240, 320),
                 degrees=10, flip_p=0.5, jitter_bcsh=[0.3, 0.3, 0.3, 0.05],
                 data_augmentation=True, exp_name="debug",
                 use_novel_viewpoints=False, only_new_scene=True,
                 fix_nerf=False, replay_buffer_size=None, device="cuda",
                 val_scene_list=None):
        super().__init__()
        self._mode = mode
        self.H, self.W = output_size
        self.num_rays = 4096
        self.root = root
        self.exp_name = exp_name
        self.fix_nerf = fix_nerf
        self.device = torch.device(device)
        if only_new_scene:
            scene_list = [scene_list[-1]]
        self.replay_buffer_size = replay_buffer_size
        self.replay_per_scene = None
        if replay_buffer_size is not None:
            n_old = len(scene_list) - 1
            if n_old > 0:
                self.replay_per_scene = replay_buffer_size // n_old
        if mode in ("val", "train_val"):  # hard-coded in the reference (:66-93)
            # (val_scene_list: extra, for roots that do not hold all ten scenes)
            scene_list = list(val_scene_list or _TEN_SCENES)
        if mode == "predict":
            self._use_novel_viewpoints = use_novel_viewpoints
        elif mode == "train":
            self._use_novel_viewpoints = (use_novel_viewpoints and
                                          self.replay_per_scene is not None)
        else:
            assert not use_novel_viewpoints
            self._use_novel_viewpoints = False
        self.get_ngp_info(scene_list)
        self.length = (len(self.nerf_image_pths) if self._use_novel_viewpoints
                       else len(self.image_pths))
        self._output_size = tuple(output_size)
        self._degrees, self._flip_p = degrees, flip_p
        self._jitter_bcsh = list(jitter_bcsh)
        self._data_augmentation = data_augmentation

    # ------------------------------------------------------------------ index
    def get_ngp_info(self, scene_list):
        """reference :113-291."""
        self.poses, self.image_pths, self.label_pths = [], [], []
        self.nerf_label_pths, self.nerf_image_pths, self.depth_pths = [], [], []
        self.from_old_scene, self.viewpoint_is_novel = [], []
        for i, scene_name in enumerate(scene_list):
            last = i == len(scene_list) - 1
            scene_root = os.path.join(self.root, scene_name)
            with open(os.path.join(scene_root, "transforms_train.json")) as f:
                info = json.load(f)
            if last:
                self.ngp_H, self.ngp_W = int(info["h"]), int(info["w"])
                self.ngp_fl_x, self.ngp_fl_y = info["fl_x"], info["fl_y"]
                self.ngp_cx, self.ngp_cy = info["cx"], info["cy"]
                self.one_m_to_scene_uom = info["one_m_to_scene_uom"]
                self.ngp_intrinsics = np.array([self.ngp_fl_x, self.ngp_fl_y,
                                                self.ngp_cx, self.ngp_cy])
            frames = info["frames"]
            if self._mode != "predict":
                n_val = int(0.2 * len(frames))
                frames = frames[-n_val:] if self._mode == "val" else frames[:-n_val]
            gen_json = os.path.join(scene_root, self.exp_name,
                                    "novel_viewpoints", "interpolated_data.json")
            replayed = (self._mode == "train" and
                        self.replay_per_scene is not None and not last)
            if replayed:
                if self._use_novel_viewpoints:
                    with open(gen_json) as f:
                        frames = json.load(f)["frames"]
                random.Random(0).shuffle(frames)
                frames = frames[:self.replay_per_scene]
            novel_replay = replayed and self._use_novel_viewpoints
            novel = self._use_novel_viewpoints and (novel_replay or
                                                    self._mode == "predict")
            sub = "novel_viewpoints" if self._use_novel_viewpoints else ""
            current_poses, gen_images, gen_labels = [], [], []
            for fr in frames:
                if novel_replay:
                    nerf_image_path = fr["nerf_image"]
                    nerf_label_path = fr["nerf_label"]
                    pose = np.array(fr["pose"], dtype=np.float32)
                else:
                    image_path = os.path.join(scene_root, fr["file_path"])
                    label_path = os.path.join(scene_root, fr["label_path"])
                    stem = os.path.basename(image_path).split(".")[0]
                    depth_path = os.path.join(scene_root, "depth", stem + ".png")
                    nerf_label_path = os.path.join(scene_root, self.exp_name,
                                                   sub, "nerf_label",
                                                   stem + ".png")
                    nerf_image_path = os.path.join(scene_root, self.exp_name,
                                                   sub, "nerf_image",
                                                   stem + ".png")
                    gen_labels.append(nerf_label_path)
                    gen_images.append(nerf_image_path)
                    pose = np.array(fr["transform_matrix"], dtype=np.float32)
                current_poses.append(pose)
                self.viewpoint_is_novel.append(bool(novel))
                if novel:
                    self.image_pths.append(None)
                    self.label_pths.append(None)
                    self.depth_pths.append(None)
                else:
                    self.image_pths.append(image_path)
                    self.label_pths.append(label_path)
                    self.depth_pths.append(depth_path)
                self.nerf_label_pths.append(nerf_label_path)
                self.nerf_image_pths.append(nerf_image_path)
                if self._mode in ("val", "train_val"):
                    self.from_old_scene.append(False)
                else:
                    self.from_old_scene.append(bool(not last or self.fix_nerf))
            if self._use_novel_viewpoints and self._mode == "predict":
                current_poses = self._interpolate_poses(current_poses)
                assert len(gen_images) == len(gen_labels) == len(current_poses)
                os.makedirs(os.path.dirname(gen_json), exist_ok=True)
                if udist.world()[0] == 0:  # one writer under torch.distributed
                    frames_out = [
                        {"nerf_image": a, "nerf_label": b, "pose": p.tolist()}
                        for a, b, p in zip(gen_images, gen_labels,
                                           current_poses)]
                    with open(gen_json, "w") as f:
                        json.dump({"frames": frames_out}, f, indent=5)
            self.poses += [nerf_matrix_to_ngp(p) for p in current_poses]
        self.poses = torch.from_numpy(np.stack(self.poses, axis=0))

    @staticmethod
    def _interpolate_poses(poses):
        """reference :232-262: Slerp half way between consecutive views (and
        between the last and the first), mean of the translations."""
        from scipy.spatial.transform import Rotation, Slerp
        poses = list(poses) + [poses[0]]
        times = list(range(len(poses)))
        slerp = Slerp(times=times, rotations=Rotation.from_matrix(
            [p[:3, :3] for p in poses]))
        rots = slerp(times=[0.5 + k for k in range(len(poses) - 1)]).as_matrix()
        out = []
        for k in range(len(poses) - 1):
            m = np.eye(4)
            m[:3, :3] = rots[k]
            m[:3, 3] = (poses[k][:3, 3] + poses[k + 1][:3, 3]) / 2.0
            out.append(m)
        return out

    # ----------------------------------------------------------------- decode
    def _cached(self, kind, path, decode):
        """A private copy of the decoded frame: the cache's tensor is cloned on
        the CPU; ``__getitem__``'s ``.to(device)`` copies on a GPU anyway."""
        t = decode_cache().lookup(kind, path, (self.H, self.W), decode)
        return t.clone() if self.device.type == "cpu" else t

    def preprocess_image(self, image_path):
        """reference :293-300 -> [3,H,W] fp32 in [0,1]."""
        return self._cached("image", image_path, self._decode_image)

    def preprocess_label(self, label_path):
        """reference :302-308 -> [H,W] int64, -1 unknown, 0..39 NYU40."""
        return self._cached("label", label_path, self._decode_label)

    def preprocess_depth(self, depth_path):
        """reference :310-319 -> [H,W] fp32 metres from uint16 millimetres."""
        return self._cached("depth", depth_path, self._decode_depth)

    def _decode_image(self, image_path):
        Image = _pil()
        im = Image.open(image_path).convert("RGB")
        if im.size != (self.W, self.H):
            im = im.resize((self.W, self.H), Image.BOX)
        a = np.asarray(im, dtype=np.float32) / 255.0
        return torch.from_numpy(a).permute(2, 0, 1).contiguous()

    def _decode_label(self, label_path):
        Image = _pil()
        im = Image.open(label_path)
        if im.size != (self.W, self.H):
            im = im.resize((self.W, self.H), Image.NEAREST)
        return torch.from_numpy(np.asarray(im).astype(np.int64)) - 1

    def _decode_depth(self, depth_path):
        Image = _pil()
        im = Image.open(depth_path)
        if im.size != (self.W, self.H):
            im = im.resize((self.W, self.H), Image.NEAREST)
        a = np.asarray(im)
        assert a.ndim == 2 and a.dtype in (np.uint16, np.int32), a.dtype
        return torch.from_numpy(a.astype(np.float32) / 1000.0)

    # ------------------------------------------------------------ augmentation
    def _augment(self, img, labels, only_crop=False):
        """helper.AugmentationList.apply (reference helper.py:157-232) on the
        device: optional rescale, then jitter / rotate / crop / flip with one
        set of draws for the image and all its label maps."""
        oh, ow = self._output_size
        H, W = img.shape[1:]
        sf = None
        if H >= 2 * oh:
            sf = max(oh / H, ow / W) * 1.2
        elif H < oh or W < ow:
            sf = max(oh / H, ow / W) * 1.2
        if sf is not None:
            F = torch.nn.functional
            img = F.interpolate(img[None], scale_factor=(sf, sf),
                                mode="bilinear", recompute_scale_factor=False,
                                align_corners=False)[0]
            labels = [F.interpolate(l[None], scale_factor=(sf, sf),
                                    mode="nearest",
                                    recompute_scale_factor=False)[0]
                      for l in labels]
            H, W = img.shape[1:]
        b, c, s, h = self._jitter_bcsh
        if only_crop:
            p = dict(order=[0, 1, 2, 3], brightness=1.0, contrast=1.0,
                     saturation=1.0, hue=0.0, angle_deg=0.0, flip=False)
            p["crop_i"] = int(round((H - oh) / 2.0))  # CenterCrop
            p["crop_j"] = int(round((W - ow) / 2.0))
        else:
            draw = lambda lo, hi: float(torch.empty(1).uniform_(lo, hi))
            p = dict(order=torch.randperm(4).tolist(),
                     brightness=draw(max(0.0, 1 - b), 1 + b),
                     contrast=draw(max(0.0, 1 - c), 1 + c),
                     saturation=draw(max(0.0, 1 - s), 1 + s), hue=draw(-h, h),
                     angle_deg=random.uniform(-self._degrees, self._degrees))
            p["crop_i"] = 0 if H == oh else int(torch.randint(0, H - oh + 1, (1,)))
            p["crop_j"] = 0 if W == ow else int(torch.randint(0, W - ow + 1, (1,)))
            p["flip"] = bool(torch.rand(1) < self._flip_p)
        img = img.to(self.device)
        out_img = None
        out_labels = []
        for l in labels:  # labels arrive as float (label + 1), [1,H,W]
            li = l[0].to(self.device).long() - 1
            out_img, ol = ops.augment(img[None], li[None], [p], (oh, ow))
            out_labels.append((ol[0] + 1)[None].float())
        return out_img[0], out_labels

    # ------------------------------------------------------------------ items
    @torch.no_grad()
    def __getitem__(self, index):
        """reference :320-458."""
        dev = self.device
        novel = self.viewpoint_is_novel[index]
        if self.from_old_scene[index]:
            nerf_label = self.preprocess_label(self.nerf_label_pths[index])
            nerf_image = self.preprocess_image(self.nerf_image_pths[index])
            if novel:
                img, img_fp16, label, depth = nerf_image, None, nerf_label, None
            else:
                img = self.preprocess_image(self.image_pths[index])
                img_fp16 = img.half().to(dev)
                label = self.preprocess_label(self.label_pths[index])
                depth = self.preprocess_depth(self.depth_pths[index]).half().to(dev)
            train_aug = self._mode == "train" and self._data_augmentation
            img, labels = self._augment(
                nerf_image if train_aug else img,
                [(label[None] + 1).float(), (nerf_label[None] + 1).float()],
                only_crop=not train_aug)
            label = None if novel else (labels[0][0] - 1).long()
            nerf_label = (labels[1][0] - 1).long()
            pose = self.poses[-1].unsqueeze(0)
            from_old = True
        else:
            if novel:
                img, img_fp16, label, depth = [], [], [], []
            else:
                img = self.preprocess_image(self.image_pths[index]).to(dev)
                img_fp16 = img.half()
                label = self.preprocess_label(self.label_pths[index]).to(dev)
                depth = self.preprocess_depth(self.depth_pths[index]).half().to(dev)
            nerf_label = label
            pose = self.poses[index].unsqueeze(0)
            from_old = False
        rays = get_rays(pose.to(dev), self.ngp_intrinsics, self.ngp_H,
                        self.ngp_W)
        item = {
            "img": img, "label": label, "depth": depth, "img_fp16": img_fp16,
            "nerf_label": nerf_label, "pose": pose[0].to(dev),
            "from_old_scene": from_old, "viewpoint_is_novel": novel,
            "H": self.ngp_H, "W": self.ngp_W, "intrinsics": self.ngp_intrinsics,
            "one_m_to_scene_uom": self.one_m_to_scene_uom,
            "rays_o": rays["rays_o"][0], "rays_d": rays["rays_d"][0],
            "direction_norms": rays["direction_norms"][0],
        }
        if novel:
            names = re.findall(r"scene\d\d\d\d_\d\d", self.nerf_image_pths[index])
            assert len(names) == 1
            item["current_scene_name"] = names[0]
            item["current_index"] = str(
                os.path.basename(self.nerf_image_pths[index])[:-4])
        else:
            item["current_scene_name"] = os.path.normpath(
                self.image_pths[index]).split(os.path.sep)[-3]
            item["current_index"] = str(
                os.path.basename(self.image_pths[index])[:-4])
        return item

    @staticmethod
    def collate(batch):
        """reference :460-495 -> (batch_old, batch_new, batch_cl)."""
        groups = [defaultdict(list), defaultdict(list), defaultdict(list)]
        for key in batch[0]:
            for b in batch:
                if key in ("replay_img", "replay_label"):
                    groups[2][key].append(b[key])
                elif b["from_old_scene"]:
                    groups[0][key].append(b[key])
                else:
                    groups[1][key].append(b[key])
        out = []
        for grp, probe in zip(groups, ("img", "img", "replay_img")):
            if probe not in grp:
                out.append(None)
                continue
            for key in grp:
                if type(grp[key][0]) == torch.Tensor:
                    grp[key] = torch.stack(grp[key], dim=0)
            out.append(grp)
        return tuple(out)

    def __len__(self):
        return self.length

    def __str__(self):
        return ("=" * 90 + "\nScannet Dataset: \n" +
                f"    Total Samples: {len(self)}  »  Mode: {self._mode} \n" +
                f"  »  DataAug: {self._data_augmentation}" + "=" * 90)
