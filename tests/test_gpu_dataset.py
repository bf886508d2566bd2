"""The ScanNet-layout dataset mirror on a synthetic scene
exported to that layout: items equal the in-memory synthetic dataset's up to
the 8-bit / millimetre quantisation of the files, rays bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def exported(tmp_path_factory):
    from tools.export_synthetic_scannet import export
    root = str(tmp_path_factory.mktemp("scannet"))
    ds0, _ = export(root, 0, 10, 48, 64)
    ds1, _ = export(root, 1, 10, 48, 64)
    return root, ds0, ds1


def test_new_scene_items_match_the_synthetic_dataset(exported):
    from ucsa_neural_rendering_amd.dataset.scannet_ngp_joint import \
        ScanNetNGPJoint
    root, ds0, _ = exported
    d = ScanNetNGPJoint(root, ["scene0000_00"], mode="train",
                        output_size=(48, 64))
    assert len(d) == 8 and not any(d.from_old_scene)   # 80 % of 10 frames
    v = ScanNetNGPJoint(root, ["scene0000_00"], mode="predict",
                        output_size=(48, 64))
    assert len(v) == 10
    for k in (0, 5, 7):
        it, ref = d[k], ds0[k]
        assert float((it["img"] - ref["img"]).abs().max()) <= 0.5 / 255 + 1e-6
        assert torch.equal(it["label"], ref["label"])
        assert float((it["depth"].float() - ref["depth"].float()).abs().max()) <= 2e-3
        assert it["img_fp16"].dtype == torch.float16 and it["depth"].dtype == torch.float16
        assert float((it["pose"] - ref["pose"]).abs().max()) <= 1e-6
        for key in ("rays_o", "rays_d", "direction_norms"):
            assert float((it[key] - ref[key]).abs().max()) <= 1e-5, key
        assert it["current_scene_name"] == "scene0000_00"
        assert it["current_index"] == f"{k:06d}"
        assert it["H"] == 48 and it["W"] == 64 and it["one_m_to_scene_uom"] == 1.0
        assert it["from_old_scene"] is False and it["viewpoint_is_novel"] is False
        assert it["nerf_label"] is it["label"]
    old, new, cl = ScanNetNGPJoint.collate([d[0], d[1]])
    assert old is None and cl is None and new["img"].shape == (2, 3, 48, 64)
    assert new["rays_o"].shape == (2, 48 * 64, 3)


def test_novel_viewpoints_and_replay(exported):
    from PIL import Image
    from ucsa_neural_rendering_amd.dataset.scannet_ngp_joint import \
        ScanNetNGPJoint
    root, ds0, ds1 = exported
    # predict with novel viewpoints: poses half way, json written
    p = ScanNetNGPJoint(root, ["scene0000_00"], mode="predict",
                        output_size=(48, 64), exp_name="e",
                        use_novel_viewpoints=True)
    js = os.path.join(root, "scene0000_00", "e", "novel_viewpoints",
                      "interpolated_data.json")
    frames = json.load(open(js))["frames"]
    assert len(p) == 10 and len(frames) == 10 and all(p.viewpoint_is_novel)
    it = p[3]
    assert it["img"] == [] and it["viewpoint_is_novel"] and it["current_index"] == "000003"
    mid = 0.5 * (ds0[3]["pose"][:3, 3] + ds0[4]["pose"][:3, 3])
    assert float((it["pose"][:3, 3] - mid).abs().max()) <= 1e-5
    R = it["pose"][:3, :3].double().cpu()
    assert float((R @ R.T - torch.eye(3, dtype=torch.float64)).abs().max()) <= 1e-5
    assert it["rays_o"].shape == (48 * 64, 3)
    # write the "rendered" frames the replay reads, then the joint dataset
    for fr in frames:
        for key, arr in (("nerf_image", np.full((48, 64, 3), 120, np.uint8)),
                         ("nerf_label", np.full((48, 64), 7, np.uint8))):
            os.makedirs(os.path.dirname(fr[key]), exist_ok=True)
            Image.fromarray(arr).save(fr[key])
    torch.manual_seed(0)
    j = ScanNetNGPJoint(root, ["scene0000_00", "scene0001_00"], mode="train",
                        output_size=(48, 64), exp_name="e",
                        use_novel_viewpoints=True, only_new_scene=False,
                        replay_buffer_size=4)
    assert len(j) == 4 + 8 and j.from_old_scene[:4] == [True] * 4
    assert j.viewpoint_is_novel[:4] == [True] * 4 and not any(j.viewpoint_is_novel[4:])
    old = j[1]
    assert old["from_old_scene"] and old["label"] is None and old["depth"] is None
    assert old["img"].shape == (3, 48, 64) and old["nerf_label"].shape == (48, 64)
    inside = old["nerf_label"] >= 0                  # rotated-in border is -1
    assert inside.float().mean() > 0.7 and bool((old["nerf_label"][inside] == 6).all())
    assert float(old["img"].max()) <= 1.0
    new = j[6]
    assert not new["from_old_scene"] and torch.equal(new["label"], ds1[2]["label"])
    b_old, b_new, b_cl = ScanNetNGPJoint.collate([j[0], j[5], j[6]])
    assert b_old["img"].shape[0] == 1 and b_new["img"].shape[0] == 2 and b_cl is None


def test_train_joint_on_the_scannet_layout(exported, tmp_path):
    """scripts/train_joint.train on the exported scene: the data module picks
    ScanNetNGPJoint, NeRF training raises the render PSNR, the predict pass
    writes the nerf_image / nerf_label PNGs where the next stage's replay
    (ScanNetNGPJoint(..., replay_buffer_size=...)) looks for them."""
    import argparse
    from PIL import Image
    from scripts import train_joint as tj
    from ucsa_neural_rendering_amd.dataset.scannet_ngp_joint import \
        ScanNetNGPJoint
    root = exported[0]
    env = {"results": str(tmp_path / "experiments"), "scannet": root}
    cfgp = tmp_path / "exp.yml"
    cfgp.write_text("x: 1\n")

    def exp():
        return {
            "general": {"name": "joint_train/layout", "checkpoint_load": "",
                        "clean_up_folder_if_exists": True},
            "model": {"pretrained": False, "pretrained_backbone": False,
                      "num_classes": 40, "backbone": "resnet50"},
            "optimizer": {"lr_seg": 1e-5, "lr_nerf": 1e-2, "name": "Adam"},
            "trainer": {"load_from_checkpoint": False},
            "data_module": {"batch_size": 2, "output_size": (48, 64)},
            "scenes": ["scene0000_00"], "exp_name": "t",
            "cl": {"active": False, "use_novel_viewpoints": False,
                   "replay_buffer_size": None},
            "nerf": {"n_rays": 1024, "num_steps": 32, "upsample_steps": 32},
            "nerf_seed": 1,
        }
    args = argparse.Namespace(exp_name="t", fix_nerf=False, seed=123,
                              nerf_train_epoch=0, joint_train_epoch=0,
                              limit_batches=None)
    r0 = tj.train(exp(), env, str(cfgp), str(cfgp), args)
    args.nerf_train_epoch, args.joint_train_epoch = 12, 1
    r1 = tj.train(exp(), env, str(cfgp), str(cfgp), args)
    assert r1["test_after_nerf"]["test_nerf_PSNR"] > r0["test_after_nerf"]["test_nerf_PSNR"] + 3.0
    out = os.path.join(root, "scene0000_00", "t")
    names = sorted(os.listdir(os.path.join(out, "nerf_image")))
    assert len(names) == 10 and names[0] == "000000.png"
    lab = np.asarray(Image.open(os.path.join(out, "nerf_label", "000003.png")))
    assert lab.shape == (48, 64) and lab.dtype == np.uint8 and 1 <= lab.min() and lab.max() <= 40
    assert np.asarray(Image.open(os.path.join(out, "nerf_image", "000003.png"))).shape == (48, 64, 3)
    # the next stage (scene 1 new, scene 0 replayed from those files)
    j = ScanNetNGPJoint(root, ["scene0000_00", "scene0001_00"], mode="train",
                        output_size=(48, 64), exp_name="t", only_new_scene=False,
                        replay_buffer_size=4)
    old = j[0]
    assert old["from_old_scene"] and old["label"] is not None
    assert old["nerf_label"].shape == (48, 64) and old["img"].shape == (3, 48, 64)
