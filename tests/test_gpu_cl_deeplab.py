"""BASELINE cfg5's driver, ``scripts/cl_deeplab.py`` (reference
``scripts/cl_deeplab.py:11-22,53-91`` + ``run_scripts/multi_step.sh``), on two
tiny synthetic scenes: stage naming, checkpoint chaining (stage 1 starts from
stage 0's ``deeplab.ckpt``; only stage 0 applies the pre-training key
rewrite), the growing scene list and the replay of the previous scene from
the PNGs its predict pass wrote.  ``-m gpu`` (the stages train on the HIP
path).  The pure planning logic is also checked on the CPU
(tests/test_cl_plan_cpu.py)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _exp():
    return {
        "general": {"name": "x", "clean_up_folder_if_exists": True, "checkpoint_load": ""},
        "model": {"pretrained": False, "pretrained_backbone": False, "num_classes": 40,
                  "backbone": "resnet50"},
        "optimizer": {"lr_seg": 1e-5, "lr_nerf": 1e-2, "name": "Adam"},
        "trainer": {"load_from_checkpoint": True, "resume_from_checkpoint": False,
                    "cudnn_benchmark": False},
        "data_module": {"batch_size": 2, "output_size": (48, 64)},
        "scenes": ["scene0000_00"],
        "cl": {"active": False, "use_novel_viewpoints": False, "replay_buffer_size": 4},
        "synthetic": {"n_views": 6, "H": 48, "W": 64},
        "nerf": {"n_rays": 512, "num_steps": 32, "upsample_steps": 32},
        "nerf_seed": 1,
    }


def test_two_stage_continual_loop(tmp_path, monkeypatch):
    from scripts import cl_deeplab
    from ucsa_neural_rendering_amd.network import DeepLabV3
    env = {"results": str(tmp_path / "experiments"), "scannet": str(tmp_path / "scans")}
    # a "pre-training" checkpoint in the reference's Lightning layout: keys
    # prefixed by the LightningModule attribute, plus an aux head to drop
    torch.manual_seed(0)
    pre = DeepLabV3(_exp()["model"])
    with torch.no_grad():
        pre._model.classifier[-1].bias.fill_(0.321)
    sd = {"_model." + k: v for k, v in pre.state_dict().items()}
    sd["_model._model.aux_classifier.0.weight"] = torch.zeros(3)
    ck = tmp_path / "pretrain.ckpt"
    torch.save({"state_dict": sd}, ck)
    exp = _exp()
    exp["general"]["checkpoint_load"] = str(ck)

    loads = []
    orig_load = torch.load

    def spy(path, *a, **k):
        loads.append(str(path))
        return orig_load(path, *a, **k)

    monkeypatch.setattr(torch, "load", spy)
    seen = {}
    from ucsa_neural_rendering_amd.lightning import joint_train_lightning_net as jl
    orig_fit_hook = jl.JointTrainLightningNet.on_train_epoch_start

    def hook(self):
        # the bias value the stage STARTS from (before any optimizer step of it)
        seen.setdefault(self._exp["general"]["name"],
                        float(self.seg_model._model.classifier[-1].bias[0]))
        return orig_fit_hook(self)

    monkeypatch.setattr(jl.JointTrainLightningNet, "on_train_epoch_start", hook)
    res = cl_deeplab.main(["--exp_name", "cl", "--scenes", "2", "--nerf_train_epoch", "1",
                           "--joint_train_epoch", "1", "--limit_batches", "2"],
                          exp=exp, env=env)
    assert len(res) == 2
    s0 = os.path.join(env["results"], "cl", "stage_0")
    s1 = os.path.join(env["results"], "cl", "stage_1")
    assert os.path.exists(os.path.join(s0, "deeplab.ckpt"))
    assert os.path.exists(os.path.join(s1, "deeplab.ckpt"))
    # stage 0 loaded the pre-training checkpoint (key rewrite), stage 1 stage 0's
    assert loads[0] == str(ck)
    assert os.path.join("cl", "stage_0", "deeplab.ckpt") in loads[1]
    assert abs(seen[s0] - 0.321) < 1e-6            # rewrite + strict load worked
    end0 = orig_load(os.path.join(s0, "deeplab.ckpt"))["state_dict"]
    assert abs(seen[s1] - float(end0["_model.classifier.4.bias"][0])) < 1e-7
    # both synthetic rooms were written in the ScanNet layout, and stage 0's
    # predict pass left the PNGs stage 1 replays
    for sc in ("scene0000_00", "scene0001_00"):
        assert os.path.exists(os.path.join(env["scannet"], sc, "transforms_train.json"))
    lab_dir = os.path.join(env["scannet"], "scene0000_00", "cl", "nerf_label")
    assert len(os.listdir(lab_dir)) == 6
    assert "test_nerf_PSNR" in res[1]["test_after_nerf"]


@pytest.mark.timeout(1800)
def test_ten_stage_continual_loop_replay_composition_and_checkpoint_chain(tmp_path, monkeypatch):
    """All TEN stages of BASELINE cfg5's loop (reference
    ``scripts/cl_deeplab.py:53-91``) on tiny synthetic rooms, one process.
    Against the reference loop's own arithmetic, per stage i:

    * ``exp["scenes"]`` is the first i+1 scenes of SCENE_ORDER, the run is
      named ``<exp_name>/stage_<i>`` (:66-70);
    * DeepLab starts from ``stage_<i-1>/deeplab.ckpt`` (:77-82) -- the bias
      value the stage starts from equals the one the previous stage saved;
      only stage 0 reads the pre-training checkpoint (:74-76);
    * the joint loader's dataset holds every training frame of the NEW scene
      (``from_old_scene`` False) plus, of each of the i old scenes,
      ``min(replay_buffer_size // i, n_train)`` frames flagged
      ``from_old_scene`` (``scannet_ngp_joint.py:56-63,150-156``: the integer
      division makes the replay vanish once i exceeds the buffer size);
    * the replayed frames read the ``nerf_label`` PNGs the earlier stages'
      predict passes wrote."""
    from scripts import cl_deeplab
    from ucsa_neural_rendering_amd.lightning import joint_train_data_module as dmod
    from ucsa_neural_rendering_amd.lightning import joint_train_lightning_net as jl
    env = {"results": str(tmp_path / "experiments"), "scannet": str(tmp_path / "scans")}
    exp = _exp()
    R, n_views = 8, 5
    n_train = n_views - int(0.2 * n_views)
    exp["cl"]["replay_buffer_size"] = R
    exp["synthetic"] = {"n_views": n_views, "H": 24, "W": 32}
    exp["data_module"] = {"batch_size": 2, "output_size": (24, 32)}
    exp["nerf"] = {"n_rays": 128, "num_steps": 8, "upsample_steps": 8}

    comp = []
    orig_setup = dmod.JointTrainDataModule.setup

    def setup_spy(self, stage=None):
        r = orig_setup(self, stage)
        js = self.joint_set
        per_scene = {}
        for pth, old, lab in zip(js.image_pths, js.from_old_scene, js.nerf_label_pths):
            sc = [p for p in pth.split(os.sep) if p.startswith("scene")][0]
            per_scene.setdefault(sc, []).append((bool(old), lab))
        comp.append(per_scene)
        return r

    monkeypatch.setattr(dmod.JointTrainDataModule, "setup", setup_spy)
    loads, start_bias = [], {}
    orig_load = torch.load
    monkeypatch.setattr(torch, "load", lambda path, *a, **k: (loads.append(str(path)),
                                                              orig_load(path, *a, **k))[1])
    orig_hook = jl.JointTrainLightningNet.on_train_epoch_start

    def hook(self):
        start_bias.setdefault(self._exp["general"]["name"],
                              float(self.seg_model._model.classifier[-1].bias[0]))
        return orig_hook(self)

    monkeypatch.setattr(jl.JointTrainLightningNet, "on_train_epoch_start", hook)
    res = cl_deeplab.main(["--exp_name", "cl", "--nerf_train_epoch", "1",
                           "--joint_train_epoch", "1", "--limit_batches", "1"],
                          exp=exp, env=env)
    assert len(res) == 10 and len(comp) == 10
    scenes = cl_deeplab.SCENE_ORDER
    for i in range(10):
        sd = os.path.join(env["results"], "cl", f"stage_{i}")
        assert os.path.exists(os.path.join(sd, "deeplab.ckpt")), i
        got = comp[i]
        want_old = 0 if i == 0 else min(R // i, n_train)
        assert [f for f in got[scenes[i]] if f[0]] == [], i          # new frames are new
        assert len(got[scenes[i]]) == n_train, i
        for k in range(i):
            frames = got.get(scenes[k], [])
            assert len(frames) == want_old, (i, k, len(frames), want_old)
            assert all(old for old, _ in frames)
            for _, lab in frames:                                    # written by stage k
                assert os.path.exists(lab), lab
        assert set(got) == set(scenes[:i + 1] if want_old or i == 0 else [scenes[i]]), (i, set(got))
        if i > 0:
            prev = os.path.join(env["results"], "cl", f"stage_{i - 1}", "deeplab.ckpt")
            assert prev in loads, i
            end_prev = orig_load(prev)["state_dict"]["_model.classifier.4.bias"][0]
            assert abs(start_bias[sd] - float(end_prev)) < 1e-7, i
        assert "test_nerf_PSNR" in res[i]["test_after_nerf"], i
    # no checkpoint but the chain was read: stage 0 had none (random init, warned)
    assert all("deeplab.ckpt" in l for l in loads) and len(loads) >= 9
