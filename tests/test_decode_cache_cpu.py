"""dataset.scannet_ngp_joint.DecodeCache: decoded frames survive epochs and stages,
a rewritten file (the pseudo-labels of the predict pass) is decoded again, the
budget bounds it, and nobody gets the cached tensor itself on the CPU."""
import json
import os

import numpy as np
import pytest
import torch

from ucsa_neural_rendering_amd.dataset import scannet_ngp_joint as sj

PIL = pytest.importorskip("PIL.Image")


def _png(path, arr):
    PIL.fromarray(arr).save(path)


def _bump(path):
    st = os.stat(path)
    os.utime(path, ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000))


def test_hit_miss_invalidation_and_budget(tmp_path):
    calls = []

    def decode(p):
        calls.append(p)
        return torch.from_numpy(np.asarray(PIL.open(p)).astype(np.int64))

    a, b = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    _png(a, np.full((64, 64), 3, np.uint8))
    _png(b, np.full((64, 64), 5, np.uint8))
    c = sj.DecodeCache(budget_mb=64 * 64 * 8 * 1.5 / (1 << 20))     # room for ONE 64x64 int64 frame
    t1 = c.lookup("label", a, (64, 64), decode)
    t2 = c.lookup("label", a, (64, 64), decode)
    assert t1 is t2 and len(calls) == 1 and (c.hits, c.misses) == (1, 1)
    # another output size or kind is another entry
    c.lookup("label", a, (32, 32), decode)
    assert len(calls) == 2
    # the file is rewritten: decoded again, new content
    _png(a, np.full((64, 64), 9, np.uint8))
    _bump(a)
    t3 = c.lookup("label", a, (64, 64), decode)
    assert len(calls) == 3 and int(t3[0, 0]) == 9
    # budget: b evicts a
    c.lookup("label", b, (64, 64), decode)
    assert c.used <= c.budget
    n = len(calls)
    c.lookup("label", a, (64, 64), decode)
    assert len(calls) == n + 1
    # switched off: straight through
    off = sj.DecodeCache(budget_mb=0)
    off.lookup("label", b, (64, 64), decode)
    off.lookup("label", b, (64, 64), decode)
    assert len(calls) == n + 3 and off.used == 0


def test_the_dataset_hands_out_private_copies(tmp_path, monkeypatch):
    root = tmp_path
    scene = root / "scene0000_00"
    for d in ("color", "label_40", "depth"):
        (scene / d).mkdir(parents=True)
    frames = []
    rng = np.random.default_rng(0)
    for i in range(5):
        _png(str(scene / "color" / f"{i}.png"), rng.integers(0, 255, (48, 64, 3), dtype=np.uint8))
        _png(str(scene / "label_40" / f"{i}.png"), rng.integers(0, 41, (48, 64)).astype(np.uint8))
        _png(str(scene / "depth" / f"{i}.png"), rng.integers(500, 5000, (48, 64)).astype(np.uint16))
        frames.append({"file_path": f"color/{i}.png", "label_path": f"label_40/{i}.png",
                       "transform_matrix": np.eye(4).tolist()})
    json.dump({"h": 48, "w": 64, "fl_x": 50.0, "fl_y": 50.0, "cx": 32.0, "cy": 24.0,
               "one_m_to_scene_uom": 1.0, "frames": frames}, open(scene / "transforms_train.json", "w"))
    monkeypatch.setattr(sj, "_DECODE_CACHE", sj.DecodeCache(budget_mb=64))
    ds = sj.ScanNetNGPJoint(str(root), ["scene0000_00"], mode="train", output_size=(48, 64), device="cpu")
    p = ds.image_pths[0]
    x = ds.preprocess_image(p)
    ref = x.clone()
    x.mul_(0.0)                                  # a consumer scribbles on its copy
    y = ds.preprocess_image(p)
    assert torch.equal(y, ref) and sj.decode_cache().hits == 1
    lab, dep = ds.preprocess_label(ds.label_pths[0]), ds.preprocess_depth(ds.depth_pths[0])
    assert lab.dtype == torch.int64 and dep.dtype == torch.float32
    assert torch.equal(lab, ds._decode_label(ds.label_pths[0]))
    assert torch.equal(dep, ds._decode_depth(ds.depth_pths[0]))
    # a second dataset object (the next stage) finds the frames decoded
    ds2 = sj.ScanNetNGPJoint(str(root), ["scene0000_00"], mode="train", output_size=(48, 64), device="cpu")
    before = sj.decode_cache().misses
    assert torch.equal(ds2.preprocess_image(p), ref)
    assert sj.decode_cache().misses == before


def test_a_predict_output_rewritten_inside_one_mtime_tick_is_decoded_again(tmp_path):
    """a pseudo-label PNG rewritten with the same byte size inside the
    filesystem's mtime granularity looks unchanged to (mtime, size); the predict pass
    therefore calls ``invalidate()`` -- a generation counter in the key of its output
    directories -- while the dataset's own frames stay cached."""
    calls = []

    def decode(p):
        calls.append(p)
        return torch.from_numpy(np.asarray(PIL.open(p)).astype(np.int64))

    (tmp_path / "nerf_label").mkdir()
    (tmp_path / "label_40").mkdir()
    out, own = str(tmp_path / "nerf_label" / "0.png"), str(tmp_path / "label_40" / "0.png")
    _png(out, np.full((32, 32), 3, np.uint8))
    _png(own, np.full((32, 32), 4, np.uint8))
    c = sj.DecodeCache(budget_mb=4)
    assert int(c.lookup("label", out, (32, 32), decode)[0, 0]) == 3
    c.lookup("label", own, (32, 32), decode)
    st = os.stat(out)
    _png(out, np.full((32, 32), 7, np.uint8))           # same size ...
    os.utime(out, ns=(st.st_atime_ns, st.st_mtime_ns))   # ... and the same mtime
    if os.stat(out).st_size == st.st_size:
        assert int(c.lookup("label", out, (32, 32), decode)[0, 0]) == 3      # the stale hit
    c.invalidate()
    n = len(calls)
    assert int(c.lookup("label", out, (32, 32), decode)[0, 0]) == 7
    c.lookup("label", own, (32, 32), decode)                                 # still cached
    assert len(calls) == n + 1
    assert not c.pin                                                         # pinning is opt-in
