"""Mirror of reference ``nr4seg/lightning/joint_train_data_module.py``.  Same
loader method names and batch sizes: nerf loaders batch_size 1, joint loader
``cfg.batch_size`` with the (old, new, cl) collate (:119-202).

Data source: when ``env["scannet"]/<scene>/transforms_train.json`` exists the
datasets are ``ScanNetNGPJoint`` instances built like the reference's
(:27-85; the ScanNet-25k test / continual-learning sets, C9, stay out of
scope); otherwise the synthetic box-room scene."""
from __future__ import annotations

import os

import torch
from torch.utils.data import DataLoader

from .. import dist as udist
from ..dataset.scannet_ngp_joint import ScanNetNGPJoint, _TEN_SCENES

from ..dataset.synthetic_scene import (SyntheticSceneDataset,
                                       default_collate_dict)


class JointTrainDataModule:

    def __init__(self, exp, env=None):
        self.exp = exp
        self.env = env
        self.cfg_loader = exp["data_module"]
        self._setup = False

    def _setup_scannet(self, root):
        """reference :27-85 (new-scene and joint datasets, val, predict)."""
        scenes = [str(s) for s in self.exp["scenes"]]
        name = self.exp.get("exp_name", "debug")
        cl = self.exp.get("cl", {})
        if cl.get("active"):
            raise NotImplementedError(
                "cl.active mixes in ScanNet-25k frames (reference :87-104); "
                "that dataset is out of scope (SURVEY C9)")
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        have = [s for s in _TEN_SCENES
                if os.path.exists(os.path.join(root, s, "transforms_train.json"))]
        val_scenes = None if len(have) == len(_TEN_SCENES) else scenes
        novel = bool(cl.get("use_novel_viewpoints", False))
        # frame size of the files; the reference's preprocessing writes 240x320
        size = tuple(self.cfg_loader.get("output_size", (240, 320)))
        kw = dict(root=root, scene_list=scenes, exp_name=name, device=dev,
                  output_size=size)
        self.val_set = ScanNetNGPJoint(mode="val", only_new_scene=False,
                                       val_scene_list=val_scenes, **kw)
        self.train_val_set = ScanNetNGPJoint(mode="train_val",
                                             only_new_scene=False,
                                             val_scene_list=val_scenes, **kw)
        self.predict_set = ScanNetNGPJoint(mode="predict",
                                           use_novel_viewpoints=novel,
                                           only_new_scene=True, **kw)
        self.train_set = ScanNetNGPJoint(mode="train", only_new_scene=True,
                                         **kw)
        self.joint_set = ScanNetNGPJoint(
            mode="train", only_new_scene=False, use_novel_viewpoints=novel,
            fix_nerf=False,
            replay_buffer_size=cl.get("replay_buffer_size"), **kw)
        self.H, self.W = self.train_set.ngp_H, self.train_set.ngp_W
        self._scannet = True
        self._setup = True

    def setup(self, stage=None):
        self._scannet = False
        root = (self.env or {}).get("scannet")
        if root and os.path.exists(os.path.join(
                root, str(self.exp["scenes"][-1]), "transforms_train.json")):
            return self._setup_scannet(root)
        syn = self.exp.get("synthetic", {})
        scene = self.exp["scenes"][-1]
        seed = int("".join(ch for ch in str(scene) if ch.isdigit())[:4] or 0)
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        n_views = int(syn.get("n_views", 20))
        self.H = int(syn.get("H", 240))
        self.W = int(syn.get("W", 320))
        full = SyntheticSceneDataset(seed, n_views, self.H, self.W,
                                     self.exp["model"]["num_classes"], dev,
                                     scene_name=str(scene))
        n_train = max(1, int(round(n_views * 0.8)))  # reference 80/20 split
        idx = list(range(n_views))
        self.train_set = torch.utils.data.Subset(full, idx[:n_train])
        self.val_set = torch.utils.data.Subset(full, idx[n_train:] or idx[-1:])
        self._setup = True

    def _dl(self, ds, bs, shuffle, collate, drop_last=False):
        """One process per GPU: training loaders get a DistributedSampler
        (every rank its own frames, reshuffled per epoch by the Trainer's
        ``set_epoch``; what Lightning DDP does in the reference), evaluation
        and predict loaders a strided shard without duplicates."""
        rank, world = udist.world()
        if udist.active():
            if shuffle:
                sampler = torch.utils.data.distributed.DistributedSampler(
                    ds, num_replicas=world, rank=rank, shuffle=True,
                    seed=int(self.exp.get("seed", 0)), drop_last=False)
            else:
                sampler = udist.RankShardSampler(len(ds), rank, world)
            return DataLoader(ds, batch_size=bs, sampler=sampler, num_workers=0,
                              drop_last=drop_last, collate_fn=collate)
        return DataLoader(ds, batch_size=bs, shuffle=shuffle, num_workers=0,
                          drop_last=drop_last, collate_fn=collate)

    def train_dataloader_nerf(self):
        return self._dl(self.train_set, 1, True, default_collate_dict)

    def train_dataloader_joint(self):
        if self._scannet:
            return self._dl(self.joint_set, self.cfg_loader["batch_size"], True,
                            ScanNetNGPJoint.collate, drop_last=True)
        return self._dl(self.train_set, self.cfg_loader["batch_size"], True,
                        SyntheticSceneDataset.collate, drop_last=True)

    def val_dataloader(self):
        if self._scannet:
            return [self._dl(self.val_set, 1, False, default_collate_dict),
                    self._dl(self.train_val_set, 1, False, default_collate_dict)]
        return [self._dl(self.val_set, 1, False, default_collate_dict),
                self._dl(self.train_set, 1, False, default_collate_dict)]

    def test_dataloader_nerf(self):
        return self._dl(self.train_set, 1, False, default_collate_dict)

    def test_dataloader(self):
        return [self.test_dataloader_nerf(),
                self._dl(self.val_set, 4, False, default_collate_dict)]

    def predict_dataloader(self):
        if self._scannet:
            return self._dl(self.predict_set, 1, False, default_collate_dict)
        return self._dl(self.train_set, 1, False, default_collate_dict)
