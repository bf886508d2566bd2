"""Mirror of reference ``nr4seg/lightning/joint_train_data_module.py`` over the
synthetic scenes (ScanNet IO is out of scope).  Same loader
method names and batch sizes: nerf loaders batch_size 1, joint loader
``cfg.batch_size`` with the (old, new, cl) collate (:119-202)."""
from __future__ import annotations

import torch
from torch.utils.data import DataLoader

from ..dataset.synthetic_scene import (SyntheticSceneDataset,
                                       default_collate_dict)


class JointTrainDataModule:

    def __init__(self, exp, env=None):
        self.exp = exp
        self.env = env
        self.cfg_loader = exp["data_module"]
        self._setup = False

    def setup(self, stage=None):
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
        return DataLoader(ds, batch_size=bs, shuffle=shuffle, num_workers=0,
                          drop_last=drop_last, collate_fn=collate)

    def train_dataloader_nerf(self):
        return self._dl(self.train_set, 1, True, default_collate_dict)

    def train_dataloader_joint(self):
        return self._dl(self.train_set, self.cfg_loader["batch_size"], True,
                        SyntheticSceneDataset.collate, drop_last=True)

    def val_dataloader(self):
        return [self._dl(self.val_set, 1, False, default_collate_dict),
                self._dl(self.train_set, 1, False, default_collate_dict)]

    def test_dataloader_nerf(self):
        return self._dl(self.train_set, 1, False, default_collate_dict)

    def test_dataloader(self):
        return [self.test_dataloader_nerf(),
                self._dl(self.val_set, 4, False, default_collate_dict)]

    def predict_dataloader(self):
        return self._dl(self.train_set, 1, False, default_collate_dict)
