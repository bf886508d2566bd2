"""The ScanNet-layout dataset mirror against the REFERENCE's own index logic
(SURVEY 8f rank 3): ``tests/golden/g8_dataset_index.json`` was produced by
importing reference ``nr4seg/dataset/scannet_ngp_joint.py`` (cv2 /
torchvision stubbed; ``tests/golden/make_golden.py --dataset-only``) on ten
seeded ``transforms_train.json`` files and records, for ten constructor
configurations, what it built: the 80/20 split (:141-147), the path lists
(:160-193), the replay selection ``random.Random(0).shuffle`` + per-scene
quota (:155-163), the old/new and novel flags (:197-216), the NGP poses, and
the Slerp novel viewpoints with their ``interpolated_data.json`` hand-over
(:218-283).  The mirror must reproduce every list and pose.  No GPU needed
(only the constructor runs)."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from tests.golden.make_golden import DATASET_CASES, dataset_layout, dataset_state
from ucsa_neural_rendering_amd.dataset.scannet_ngp_joint import ScanNetNGPJoint

FIX = os.path.join(os.path.dirname(__file__), "golden", "g8_dataset_index.json")


@pytest.fixture(scope="module")
def layout(tmp_path_factory):
    with open(FIX) as f:
        fix = json.load(f)
    root = str(tmp_path_factory.mktemp("scans"))
    dataset_layout(root, fix["n_frames"])
    return root, fix


def test_mirror_reproduces_the_reference_index(layout):
    root, fix = layout
    for tag, kw in DATASET_CASES:       # order matters: predict_novel_* write the json
        with contextlib.redirect_stdout(io.StringIO()):
            ds = ScanNetNGPJoint(root, exp_name="e", device="cpu", **kw)
        got, want = dataset_state(ds, root), fix["cases"][tag]
        for k in ("length", "image_pths", "label_pths", "depth_pths",
                  "nerf_image_pths", "nerf_label_pths", "from_old_scene",
                  "viewpoint_is_novel", "ngp_intrinsics", "one_m_to_scene_uom",
                  "ngp_HW"):
            assert got[k] == want[k], (tag, k)
        assert np.allclose(np.array(got["poses"]), np.array(want["poses"]),
                           rtol=0, atol=1e-6), tag
    # the hand-over file the predict pass leaves for the next stage's replay
    for scene, frames in fix["interpolated_data"].items():
        pth = os.path.join(root, scene, "e", "novel_viewpoints",
                           "interpolated_data.json")
        with open(pth) as f:
            mine = json.load(f)["frames"]
        assert len(mine) == len(frames)
        for a, b in zip(mine, frames):
            assert os.path.relpath(a["nerf_image"], root) == b["nerf_image"]
            assert os.path.relpath(a["nerf_label"], root) == b["nerf_label"]
            assert np.allclose(np.array(a["pose"]), np.array(b["pose"]), atol=1e-9)


def test_fixture_covers_the_interesting_branches(layout):
    _, fix = layout
    c = fix["cases"]
    assert c["train_new_only"]["length"] == 12           # 15 - int(0.2 * 15)
    assert c["val"]["length"] == sum(int(0.2 * n) for n in fix["n_frames"])
    assert all(c["train_fix_nerf"]["from_old_scene"])
    j = c["joint_replay"]                                 # 7 // 2 = 3 per old scene
    assert j["from_old_scene"].count(True) == 6 and j["length"] == 6 + 12
    jn = c["joint_replay_novel"]
    assert jn["viewpoint_is_novel"].count(True) == 6
    assert jn["image_pths"][:6] == [None] * 6
    assert all(c["predict_novel_1"]["viewpoint_is_novel"])
