"""Synthetic stand-in for the ScanNet scene datasets.

ScanNet is licensed and absent; BASELINE configs are defined on a synthetic
analytic scene: a box room inside [-3,3]^3 (scene units, NeRF bound 4) with a
few axis-aligned boxes, every surface carrying an NYU40-style class id and a
smooth procedural colour.  Ground-truth rgb / z-depth / label come from exact
ray casting.  ``SyntheticSceneDataset.__getitem__`` emits the same dict schema
as the reference's ``ScanNetNGPJoint.__getitem__``
(nr4seg/dataset/scannet_ngp_joint.py:420-458) and ``collate`` the same
``(batch_old, batch_new, batch_cl)`` triple (:460-495).

This is data generation (excluded from timing) and uses plain torch ops.
"""
from __future__ import annotations

import math
from collections import defaultdict

import torch

from .. import ops


def _slerp_loop_poses(n, radius=1.8, height=0.2, seed=123):
    """Cameras on a loop inside the room looking slightly outwards/around,
    +z forward (ray convention dir = (x, y, 1)), NGP frame."""
    g = torch.Generator().manual_seed(seed)
    poses = []
    for k in range(n):
        a = 2 * math.pi * k / max(n, 1)
        eye = torch.tensor([radius * math.cos(a), radius * math.sin(a),
                            height + 0.2 * math.sin(3 * a)])
        look = torch.tensor([-0.6 * math.cos(a + 0.7), -0.6 * math.sin(a + 0.7),
                             0.1 * float(torch.randn(1, generator=g))])
        f = look - eye
        f = f / f.norm()
        up = torch.tensor([0.0, 0.0, 1.0])
        r = torch.linalg.cross(f, up)
        r = r / r.norm()
        dn = torch.linalg.cross(f, r)
        m = torch.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = r, dn, f, eye
        poses.append(m)
    return torch.stack(poses)


class SyntheticRoom:
    """Room = inside of the box [-3,3]^3; objects = solid axis-aligned boxes."""

    def __init__(self, seed=0, n_boxes=4, n_classes=40, palette_seed=None):
        """``palette_seed``: None = every room draws its own class colours (the
        benchmarks' rooms); an int = ONE class -> colour table shared by all
        rooms built with it, so that appearance predicts the class ACROSS
        rooms -- what a segmentation network pre-trained on other rooms needs
        (the continual loop, cfg5: `synthetic: {palette_seed: ...}`)."""
        g = torch.Generator().manual_seed(1000 + seed)
        self.n_classes = n_classes
        self.room = torch.tensor([[-3.0, -3.0, -3.0], [3.0, 3.0, 3.0]])
        # classes of the six room faces (-x,+x,-y,+y,-z,+z): walls, floor, ceiling
        self.room_cls = torch.tensor([0, 0, 0, 0, 1, 21])
        boxes, cls = [], []
        for _ in range(n_boxes):
            c = (torch.rand(3, generator=g) * 2 - 1) * torch.tensor([2.2, 2.2, 0.0])
            half = 0.25 + 0.45 * torch.rand(3, generator=g)
            c[2] = -3.0 + half[2]  # standing on the floor
            # keep the camera loop (radius ~1.8) free
            if c[:2].norm() > 1.2 and c[:2].norm() < 2.4:
                c[:2] = c[:2] / c[:2].norm() * 2.6
            boxes.append(torch.stack([c - half, c + half]))
            cls.append(int(torch.randint(2, n_classes, (1,), generator=g)))
        self.boxes = torch.stack(boxes) if boxes else torch.zeros(0, 2, 3)
        self.box_cls = torch.tensor(cls, dtype=torch.int64)
        self.palette = torch.rand(n_classes, 3, generator=g) * 0.7 + 0.2
        if palette_seed is not None:
            gp = torch.Generator().manual_seed(77000 + int(palette_seed))
            self.palette = torch.rand(n_classes, 3, generator=gp) * 0.7 + 0.2

    def to(self, device):
        for k in ("room", "room_cls", "boxes", "box_cls", "palette"):
            setattr(self, k, getattr(self, k).to(device))
        return self

    @torch.no_grad()
    def cast(self, rays_o, rays_d):
        """rays [N,3] (unit d) -> t_hit [N], rgb [N,3], label [N]."""
        o, d = rays_o, rays_d
        inv = 1.0 / torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
        # room: exit point of the ray from the box (camera is inside)
        t0 = (self.room[0] - o) * inv
        t1 = (self.room[1] - o) * inv
        tfar = torch.maximum(t0, t1)
        t_room, ax = tfar.min(dim=-1)
        side = (torch.gather(d, 1, ax[:, None])[:, 0] > 0).long()
        label = self.room_cls[ax * 2 + side]
        t_hit = t_room.clone()
        for b in range(self.boxes.shape[0]):
            a0 = (self.boxes[b, 0] - o) * inv
            a1 = (self.boxes[b, 1] - o) * inv
            tn = torch.minimum(a0, a1).max(dim=-1)[0]
            tf = torch.maximum(a0, a1).min(dim=-1)[0]
            hit = (tn < tf) & (tn > 1e-4) & (tn < t_hit)
            t_hit = torch.where(hit, tn, t_hit)
            label = torch.where(hit, self.box_cls[b].expand_as(label), label)
        p = o + d * t_hit[:, None]
        shade = 0.75 + 0.25 * torch.sin(p[:, 0] * 2.1 + 0.3) * torch.cos(
            p[:, 1] * 1.7 - 0.2) * torch.sin(p[:, 2] * 1.3 + 0.9)
        rgb = (self.palette[label] * shade[:, None]).clamp(0, 1)
        return t_hit, rgb, label


class SyntheticSceneDataset(torch.utils.data.Dataset):
    """One synthetic scene: n_views posed images with rays, rgb, depth, label."""

    def __init__(self, scene_seed=0, n_views=16, H=240, W=320, n_classes=40,
                 device="cuda", scene_name=None, label_noise=0.0, palette_seed=None):
        self.H, self.W = H, W
        self.device = torch.device(device)
        self.room = SyntheticRoom(scene_seed, n_classes=n_classes,
                                  palette_seed=palette_seed).to(self.device)
        self.poses = _slerp_loop_poses(n_views, seed=123 + scene_seed).to(self.device)
        # ScanNet-like pinhole scaled to W x H
        self.intrinsics = torch.tensor([0.89 * W, 0.89 * W, W / 2.0, H / 2.0])
        self.one_m_to_scene_uom = 1.0
        self.scene_name = scene_name or f"synthetic{scene_seed:04d}_00"
        self.n_classes = n_classes
        self._cache = {}

    def __len__(self):
        return self.poses.shape[0]

    @torch.no_grad()
    def __getitem__(self, index):
        if index in self._cache:
            return self._cache[index]
        H, W = self.H, self.W
        pose = self.poses[index:index + 1]
        o, d, n = ops.get_rays(pose, self.intrinsics.tolist(), H, W)
        t_hit, rgb, label = self.room.cast(o[0], d[0])
        depth = (t_hit / n[0, :, 0]).view(H, W)  # z-depth in metres (uom = 1)
        img = rgb.view(H, W, 3).permute(2, 0, 1).contiguous()
        item = {
            "img": img,
            "label": label.view(H, W),
            "depth": depth.half(),
            "img_fp16": img.half(),
            "nerf_label": label.view(H, W),
            "pose": pose[0],
            "from_old_scene": False,
            "viewpoint_is_novel": False,
            "H": H,
            "W": W,
            "intrinsics": self.intrinsics,
            "one_m_to_scene_uom": self.one_m_to_scene_uom,
            "rays_o": o[0],
            "rays_d": d[0],
            "direction_norms": n[0],
            "current_scene_name": self.scene_name,
            "current_index": f"{index:06d}",
        }
        self._cache[index] = item
        return item

    @staticmethod
    def collate(batch):
        """reference scannet_ngp_joint.py:460-495."""
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
                if isinstance(grp[key][0], torch.Tensor):
                    grp[key] = torch.stack(grp[key], dim=0)
            out.append(dict(grp))
        return tuple(out)


def default_collate_dict(batch):
    """torch DataLoader's default collation for the dict items (batch_size 1
    loaders of the reference): tensors stacked, scalars -> lists/tensors."""
    out = {}
    for key in batch[0]:
        v = [b[key] for b in batch]
        if isinstance(v[0], torch.Tensor):
            out[key] = torch.stack(v, dim=0)
        elif type(v[0]).__module__ == "numpy" and hasattr(v[0], "shape"):
            import numpy as np
            out[key] = torch.from_numpy(np.stack(v))
        elif isinstance(v[0], (int, float)) and not isinstance(v[0], bool):
            out[key] = torch.tensor(v)
        else:
            out[key] = v
    return out
