"""Index building of the ScanNet-layout dataset mirror (no GPU needed): split,
paths, pose convention, replay selection, novel-viewpoint bookkeeping --
reference nr4seg/dataset/scannet_ngp_joint.py:113-291."""
import json
import os
import random

import numpy as np

from ucsa_neural_rendering_amd.dataset.ngp_utils import nerf_matrix_to_ngp
from ucsa_neural_rendering_amd.dataset.scannet_ngp_joint import ScanNetNGPJoint


def _scene(root, name, n):
    rs = np.random.RandomState(hash(name) % 1000)
    frames = []
    for i in range(n):
        m = np.eye(4)
        q, _ = np.linalg.qr(rs.randn(3, 3))
        m[:3, :3] = q * np.sign(np.linalg.det(q))
        m[:3, 3] = rs.randn(3)
        frames.append({"file_path": f"color/{i}.jpg", "label_path": f"label_40/{i}.png",
                       "transform_matrix": m.tolist()})
    os.makedirs(os.path.join(root, name), exist_ok=True)
    with open(os.path.join(root, name, "transforms_train.json"), "w") as f:
        json.dump({"h": 240, "w": 320, "fl_x": 290.0, "fl_y": 291.0, "cx": 160.0,
                   "cy": 120.0, "one_m_to_scene_uom": 0.37, "frames": frames}, f)
    return frames


def test_split_paths_and_poses(tmp_path):
    root = str(tmp_path)
    fr = _scene(root, "scene0003_00", 23)
    tr = ScanNetNGPJoint(root, ["scene0001_00", "scene0003_00"], mode="train",
                         exp_name="x", device="cpu")
    assert len(tr) == 23 - int(0.2 * 23) == 19        # only_new_scene default
    assert tr.image_pths[2] == os.path.join(root, "scene0003_00", "color/2.jpg")
    assert tr.depth_pths[2] == os.path.join(root, "scene0003_00", "depth", "2.png")
    assert tr.nerf_label_pths[2] == os.path.join(root, "scene0003_00", "x", "",
                                                 "nerf_label", "2.png")
    assert tr.one_m_to_scene_uom == 0.37 and tr.ngp_intrinsics.tolist() == [290.0, 291.0, 160.0, 120.0]
    want = nerf_matrix_to_ngp(np.array(fr[5]["transform_matrix"], np.float32))
    assert np.allclose(tr.poses[5].numpy(), want)
    pr = ScanNetNGPJoint(root, ["scene0003_00"], mode="predict", device="cpu")
    assert len(pr) == 23 and not any(pr.viewpoint_is_novel)
    va = ScanNetNGPJoint(root, ["scene0003_00"], mode="val", device="cpu",
                         val_scene_list=["scene0003_00"])
    assert len(va) == 4 and va.image_pths[0].endswith("color/19.jpg")
    assert not any(va.from_old_scene)
    fx = ScanNetNGPJoint(root, ["scene0003_00"], mode="train", fix_nerf=True,
                         device="cpu")
    assert all(fx.from_old_scene)


def test_replay_selection_and_novel_viewpoints(tmp_path):
    root = str(tmp_path)
    _scene(root, "scene0000_00", 20)
    _scene(root, "scene0001_00", 15)
    fr2 = _scene(root, "scene0002_00", 10)
    j = ScanNetNGPJoint(root, ["scene0000_00", "scene0001_00", "scene0002_00"],
                        mode="train", only_new_scene=False, replay_buffer_size=10,
                        device="cpu")
    assert j.replay_per_scene == 5 and len(j) == 5 + 5 + 8
    assert j.from_old_scene == [True] * 10 + [False] * 8
    order = list(range(16))                      # scene 0: 16 training frames
    random.Random(0).shuffle(order)
    assert [os.path.basename(p) for p in j.image_pths[:5]] == [f"{k}.jpg" for k in order[:5]]
    assert np.allclose(j.poses[-1].numpy(), nerf_matrix_to_ngp(
        np.array(fr2[7]["transform_matrix"], np.float32)))
    # novel viewpoints: predict writes the interpolated poses ...
    p = ScanNetNGPJoint(root, ["scene0000_00"], mode="predict", exp_name="e",
                        use_novel_viewpoints=True, device="cpu")
    js = os.path.join(root, "scene0000_00", "e", "novel_viewpoints",
                      "interpolated_data.json")
    frames = json.load(open(js))["frames"]
    assert len(p) == 20 == len(frames) and all(p.viewpoint_is_novel)
    assert frames[0]["nerf_image"].endswith("e/novel_viewpoints/nerf_image/0.png")
    # ... which the next stage's replay reads back
    _scene(root, "scene0001_00", 15)
    n = ScanNetNGPJoint(root, ["scene0000_00", "scene0001_00"], mode="train",
                        exp_name="e", only_new_scene=False, replay_buffer_size=6,
                        use_novel_viewpoints=True, device="cpu")
    assert len(n) == 6 + 12 and n.viewpoint_is_novel[:6] == [True] * 6
    assert n.image_pths[0] is None and "novel_viewpoints" in n.nerf_image_pths[0]
    R = n.poses[0][:3, :3].double().numpy()
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-5)


def test_tile_order_keeps_the_multiset_and_groups_tiles():
    """ops.tile_order: same pixel indices (duplicates kept), tile by tile."""
    import torch
    from ucsa_neural_rendering_amd.ops import tile_order
    g = torch.Generator().manual_seed(3)
    W, H = 50, 37   # not multiples of the tile
    inds = torch.randint(0, W * H, (700,), generator=g)
    out = tile_order(inds, W, tile=16)
    assert out.shape == inds.shape
    assert torch.equal(out.sort().values, inds.sort().values)
    ty, tx = (out // W) // 16, (out % W) // 16
    tile_id = ty * 4 + tx
    assert bool((tile_id[1:] >= tile_id[:-1]).all())      # tiles in order
    same = tile_id[1:] == tile_id[:-1]
    inner = (out // W % 16) * 16 + out % W % 16
    assert bool((inner[1:][same] >= inner[:-1][same]).all())  # row-major inside
    assert tile_order(inds.view(1, -1), W).shape == (1, 700)


def test_jsonl_logger_buffers_and_flushes(tmp_path):
    """Metrics are kept as given (tensors included) and converted / written
    in batches: nothing on disk before a flush, everything after, in order."""
    import json
    import torch
    from ucsa_neural_rendering_amd.lightning.trainer import JsonlLogger
    lg = JsonlLogger(str(tmp_path), flush_every=4)
    lg.log("a", torch.tensor(1.5), 0)
    lg.log("b", 2, 1)
    lg.log("c", torch.tensor([3.25]), 2)
    path = tmp_path / "metrics.jsonl"
    assert not path.exists()
    lg.log("d", 4.0, 3)                      # 4th record: automatic flush
    rows = [json.loads(l) for l in path.read_text().splitlines()]
    assert [(r["name"], r["value"], r["step"]) for r in rows] == \
        [("a", 1.5, 0), ("b", 2.0, 1), ("c", 3.25, 2), ("d", 4.0, 3)]
    lg.log("e", torch.tensor(5.0), 4)
    assert len(path.read_text().splitlines()) == 4
    assert [r["name"] for r in lg.history] == ["a", "b", "c", "d", "e"]   # history flushes
    assert len(path.read_text().splitlines()) == 5


def test_trainer_keeps_per_item_scalars_on_the_host():
    import torch
    from ucsa_neural_rendering_amd.lightning.trainer import Trainer
    tr = Trainer(device="cpu")
    batch = {"img": torch.zeros(2, 3, 4, 4), "intrinsics": torch.ones(2, 4),
             "H": torch.tensor([4, 4]), "W": torch.tensor([4, 4]),
             "one_m_to_scene_uom": torch.tensor([1.0, 1.0]),
             "current_index": ["000001", "000002"], "nested": [{"H": torch.tensor([1])}]}
    out = tr._to_device(batch)
    assert set(out) == set(batch)
    for k in ("intrinsics", "H", "W", "one_m_to_scene_uom"):
        assert out[k].device.type == "cpu" and torch.equal(out[k], batch[k])
    assert out["current_index"] == batch["current_index"]
    assert out["nested"][0]["H"].device.type == "cpu"
