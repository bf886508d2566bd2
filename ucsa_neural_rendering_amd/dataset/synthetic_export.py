"""Write a synthetic scene in the on-disk layout the reference's
``ScanNetNGPJoint`` reads (``<root>/<scene>/transforms_train.json``,
``color/*.png``, ``label_40/*.png`` (NYU40 id, 0 = unknown), ``depth/*.png``
uint16 millimetres), so that the dataset mirror and the entry points can be
exercised without ScanNet (``scripts/cl_deeplab.py`` does this for the ten
synthetic rooms of BASELINE cfg5; CLI: ``tools/export_synthetic_scannet.py``).
Data generation, excluded from any timing."""
import json
import os

import numpy as np


def ngp_to_nerf_matrix(n):
    """inverse of dataset.ngp_utils.nerf_matrix_to_ngp."""
    n = np.asarray(n, dtype=np.float64)
    p = np.eye(4)
    p[1] = [n[0, 0], -n[0, 1], -n[0, 2], n[0, 3]]
    p[2] = [n[1, 0], -n[1, 1], -n[1, 2], n[1, 3]]
    p[0] = [n[2, 0], -n[2, 1], -n[2, 2], n[2, 3]]
    return p


def export(root, scene_seed=0, n_views=10, H=240, W=320, device="cuda",
           scene_name=None, palette_seed=None):
    from PIL import Image
    from .synthetic_scene import SyntheticSceneDataset
    ds = SyntheticSceneDataset(scene_seed, n_views, H, W, device=device,
                               palette_seed=palette_seed)
    name = scene_name or f"scene{scene_seed:04d}_00"
    sroot = os.path.join(root, name)
    for sub in ("color", "label_40", "depth"):
        os.makedirs(os.path.join(sroot, sub), exist_ok=True)
    frames = []
    for i in range(n_views):
        it = ds[i]
        stem = f"{i:06d}"
        img = (it["img"].permute(1, 2, 0).cpu().numpy() * 255.0).round().astype(np.uint8)
        Image.fromarray(img).save(os.path.join(sroot, "color", stem + ".png"))
        lab = (it["label"].cpu().numpy() + 1).astype(np.uint8)
        Image.fromarray(lab).save(os.path.join(sroot, "label_40", stem + ".png"))
        dep = (it["depth"].float().cpu().numpy() * 1000.0).round().astype(np.uint16)
        Image.fromarray(dep).save(os.path.join(sroot, "depth", stem + ".png"))
        frames.append({
            "file_path": f"color/{stem}.png", "label_path": f"label_40/{stem}.png",
            "transform_matrix": ngp_to_nerf_matrix(it["pose"].cpu().numpy()).tolist()})
    fx, fy, cx, cy = [float(v) for v in ds.intrinsics]
    with open(os.path.join(sroot, "transforms_train.json"), "w") as f:
        json.dump({"h": H, "w": W, "fl_x": fx, "fl_y": fy, "cx": cx, "cy": cy,
                   "one_m_to_scene_uom": 1.0, "frames": frames}, f)
    return ds, sroot
