"""GPU parity of the loss / post-processing / metric kernels (rows a12, a15,
a16, M) against the oracle and the torch modules the reference configures, and
an end-to-end run of the drop-in LightningModule + entry point on a tiny
synthetic scene.  ``-m gpu``."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import losses as olosses
from oracle import metrics as ometrics
from tests.util import load_golden, maxabs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from ucsa_neural_rendering_amd import ops as _ops
    return _ops


def _loss_inputs(N=1000, C=40, seed=0):
    g = torch.Generator().manual_seed(seed)
    rgb, gt = torch.rand(1, N, 3, generator=g), torch.rand(1, N, 3, generator=g)
    sem = torch.rand(1, N, C, generator=g) * 0.05
    sem[0, :7] = 0
    labels = torch.randint(-1, C, (1, N), generator=g)
    depth = torch.rand(1, N, generator=g) * 3
    gtd = torch.rand(1, N, generator=g) * 3
    gtd[0, ::9] = 0
    return rgb, sem, depth, gt, labels, gtd


def test_nerf_loss_kernel_values_and_gradients(ops):
    from ucsa_neural_rendering_amd import losses as ul
    rgb, sem, depth, gt, labels, gtd = _loss_inputs()
    a = [t.clone().requires_grad_() for t in (rgb, sem, depth)]
    lc, ls, ld = olosses.nerf_losses(a[0], a[1], a[2], gt, labels, gtd, 0.7)
    olosses.nerf_total_loss(lc, ls, ld).backward()
    b = [t.clone().cuda().requires_grad_() for t in (rgb, sem, depth)]
    hc, hs, hd = ul.nerf_losses(b[0], b[1], b[2], gt.cuda(), labels.cuda(),
                                gtd.cuda(), 0.7)
    (ul.nerf_total_loss(hc, hs, hd) * 128.0).backward()  # GradScaler-like scale
    assert abs(float(hc) - float(lc)) <= 1e-6
    assert abs(float(hs) - float(ls)) <= 2e-6 * abs(float(ls))
    assert abs(float(hd) - float(ld)) <= 1e-6
    for x, y in zip(b, a):
        assert maxabs(x.grad / 128.0, y.grad) <= 1e-6 * max(1.0, float(y.grad.abs().max()))
    # all-invalid branch -> None, no semantic gradient
    z = torch.zeros_like(sem).cuda().requires_grad_()
    _, hs0, _ = ul.nerf_losses(b[0].detach(), z, b[2].detach(), gt.cuda(),
                               labels.cuda(), gtd.cuda(), 0.7, none_if_invalid=True)
    assert hs0 is None
    # default (no host read-back): a zero term with zero gradient, i.e. the
    # same total loss and gradients as skipping the term
    z2 = torch.zeros_like(sem).cuda().requires_grad_()
    hc1, hs1, hd1 = ul.nerf_losses(b[0].detach(), z2, b[2].detach(), gt.cuda(),
                                   labels.cuda(), gtd.cuda(), 0.7)
    assert float(hs1) == 0.0
    ul.nerf_total_loss(hc1, hs1, hd1).backward()
    assert z2.grad is None or float(z2.grad.abs().max()) == 0.0


def test_nerf_loss_second_backward_through_a_retained_graph(ops):
    """the loss node's backward used to scale its saved gradients
    IN PLACE (unseen by autograd's version counter); per-term gradient norms
    with ``torch.autograd.grad(..., retain_graph=True)`` followed by
    ``.backward()`` then returned gradients scaled twice.  Every backward
    through the retained node must give the gradient of what was asked."""
    from ucsa_neural_rendering_amd import losses as ul
    rgb, sem, depth, gt, labels, gtd = _loss_inputs(300)
    a = [t.clone().requires_grad_() for t in (rgb, sem, depth)]
    lc, ls, ld = olosses.nerf_losses(a[0], a[1], a[2], gt, labels, gtd, 0.7)
    want_c = torch.autograd.grad(lc, a[0], retain_graph=True)[0]
    want_s = torch.autograd.grad(ls, a[1], retain_graph=True)[0]
    olosses.nerf_total_loss(lc, ls, ld).backward()
    b = [t.clone().cuda().requires_grad_() for t in (rgb, sem, depth)]
    hc, hs, hd = ul.nerf_losses(b[0], b[1], b[2], gt.cuda(), labels.cuda(), gtd.cuda(), 0.7)
    total = ul.nerf_total_loss(hc, hs, hd)
    got_c = torch.autograd.grad(hc * 3.0, b[0], retain_graph=True)[0]
    got_s = torch.autograd.grad(hs, b[1], retain_graph=True)[0]
    total.backward(retain_graph=True)
    first = [x.grad.clone() for x in b]
    total.backward()
    assert maxabs(got_c / 3.0, want_c) <= 1e-6 * float(want_c.abs().max())
    assert maxabs(got_s, want_s) <= 2e-6 * float(want_s.abs().max())
    for x, f, y in zip(b, first, a):
        tol = 1e-6 * max(1.0, float(y.grad.abs().max()))
        assert maxabs(f, y.grad) <= tol
        assert maxabs(x.grad, 2.0 * y.grad) <= 2 * tol   # accumulated, not rescaled


def test_semantic_postproc(ops):
    g = torch.Generator().manual_seed(2)
    sem = torch.rand(3, 50, 40, generator=g)
    sem[0, :5] = 0
    ref_n, ref_a = olosses.semantic_postproc(sem)
    n, a = ops.semantic_postproc(sem.cuda())
    assert maxabs(n, ref_n) <= 1e-7
    assert torch.equal(a.cpu(), ref_a)


def test_seg_tail_matches_the_reference_modules(ops):
    from ucsa_neural_rendering_amd import losses as ul
    g = torch.Generator().manual_seed(3)
    logits = (torch.randn(2, 40, 24, 32, generator=g) * 3)
    labels = torch.randint(-1, 40, (2, 24, 32), generator=g)
    x = logits.clone().requires_grad_()
    pred = F.softmax(x, dim=1)
    loss = torch.nn.CrossEntropyLoss(ignore_index=-1, reduction="none")(pred, labels).mean()
    loss.backward()
    xh = logits.clone().cuda().requires_grad_()
    lh = ul.seg_loss(xh, labels.cuda())
    lh.backward()
    assert abs(float(lh) - float(loss)) <= 2e-6
    assert maxabs(xh.grad, x.grad) <= 1e-9 + 1e-5 * float(x.grad.abs().max())
    t = ops.seg_tail(logits.cuda(), None)
    assert maxabs(t["prob"], pred) <= 1e-6
    assert torch.equal(t["argmax"].cpu(), torch.argmax(pred, dim=1))


def test_confusion_matrix_and_meter(ops):
    from ucsa_neural_rendering_amd.utils.metrics import SemanticsMeter
    g = load_golden("g7_meter.npz")
    C = g["C"]
    m = SemanticsMeter(C)
    m.update(g["preds"][:2].cuda(), g["truths"][:2].cuda())
    m.update(g["preds"][2:].cuda(), g["truths"][2:].cuda())
    assert np.array_equal(m.conf_mat, g["conf_mat"].numpy())
    miou, acc, cacc = m.measure()
    assert abs(miou - g["miou"]) < 1e-12 and abs(acc - g["total_acc"]) < 1e-12
    assert abs(cacc - g["class_avg_acc"]) < 1e-12
    m2 = SemanticsMeter(C)  # CPU inputs take the numpy path, same result
    m2.update(g["preds"], g["truths"])
    assert np.array_equal(m2.conf_mat, m.conf_mat)


def _tiny_exp():
    return {
        "general": {"name": "joint_train/test_tiny", "clean_up_folder_if_exists": True,
                    "checkpoint_load": ""},
        "model": {"pretrained": False, "pretrained_backbone": False,
                  "num_classes": 40, "backbone": "resnet50"},
        "optimizer": {"lr_seg": 1e-5, "lr_nerf": 1e-2, "name": "Adam"},
        "trainer": {"load_from_checkpoint": False, "cudnn_benchmark": False},
        "data_module": {"batch_size": 2},
        "scenes": ["scene0000_00"],
        "synthetic": {"n_views": 5, "H": 48, "W": 64},
        "nerf": {"n_rays": 1024, "num_steps": 32, "upsample_steps": 32},
        "nerf_seed": 1,
    }


def test_dataset_schema_matches_reference_dict():
    from ucsa_neural_rendering_amd.dataset import SyntheticSceneDataset
    ds = SyntheticSceneDataset(0, n_views=3, H=48, W=64)
    it = ds[1]
    for k in ["img", "img_fp16", "label", "depth", "pose", "H", "W", "intrinsics",
              "one_m_to_scene_uom", "rays_o", "rays_d", "direction_norms",
              "from_old_scene", "viewpoint_is_novel", "current_scene_name",
              "current_index", "nerf_label"]:
        assert k in it, k
    assert it["img"].shape == (3, 48, 64) and it["img"].dtype == torch.float32
    assert it["img_fp16"].dtype == torch.float16 and it["depth"].dtype == torch.float16
    assert it["label"].shape == (48, 64) and it["label"].dtype == torch.int64
    assert it["rays_o"].shape == (48 * 64, 3) and it["direction_norms"].shape == (48 * 64, 1)
    assert float(it["depth"].float().min()) > 0 and int(it["label"].max()) < 40


def test_train_joint_entrypoint_tiny(tmp_path):
    """Reference call order on a 64x48 synthetic scene: a few NeRF steps must
    raise the render PSNR, joint step runs, checkpoint is written."""
    import argparse
    from scripts import train_joint as tj
    exp = _tiny_exp()
    env = {"results": str(tmp_path / "experiments"), "scannet": str(tmp_path)}
    cfgp = tmp_path / "exp.yml"
    cfgp.write_text("x: 1\n")
    args = argparse.Namespace(exp_name="t", fix_nerf=False, seed=123,
                              nerf_train_epoch=0, joint_train_epoch=0,
                              limit_batches=None)
    r0 = tj.train(exp, env, str(cfgp), str(cfgp), args)
    exp = _tiny_exp()
    args.nerf_train_epoch, args.joint_train_epoch = 25, 1
    r1 = tj.train(exp, env, str(cfgp), str(cfgp), args)
    p0 = r0["test_after_nerf"]["test_nerf_PSNR"]
    p1 = r1["test_after_nerf"]["test_nerf_PSNR"]
    assert p1 > p0 + 3.0, (p0, p1)
    assert (tmp_path / "experiments/joint_train/test_tiny/deeplab.ckpt").exists()
    assert "test_nerf_mIoU" in r1["test_after_joint"]


def test_train_joint_entrypoint_tiny_cuda_ray(tmp_path):
    """The same entry point with `nerf: {cuda_ray: true}`: NeRF training and
    evaluation go through the occupancy-grid marcher."""
    import argparse
    from scripts import train_joint as tj
    env = {"results": str(tmp_path / "experiments"), "scannet": str(tmp_path)}
    cfgp = tmp_path / "exp.yml"
    cfgp.write_text("x: 1\n")
    args = argparse.Namespace(exp_name="t", fix_nerf=False, seed=123,
                              nerf_train_epoch=0, joint_train_epoch=0,
                              limit_batches=None)
    exp = _tiny_exp()
    exp["nerf"].update(cuda_ray=True, dt_gamma=1.0 / 128)
    r0 = tj.train(exp, env, str(cfgp), str(cfgp), args)
    exp = _tiny_exp()
    exp["nerf"].update(cuda_ray=True, dt_gamma=1.0 / 128)
    args.nerf_train_epoch, args.joint_train_epoch = 25, 1
    r1 = tj.train(exp, env, str(cfgp), str(cfgp), args)
    p0 = r0["test_after_nerf"]["test_nerf_PSNR"]
    p1 = r1["test_after_nerf"]["test_nerf_PSNR"]
    assert p1 > p0 + 3.0, (p0, p1)
    assert "test_nerf_mIoU" in r1["test_after_joint"]


def test_frozen_seg_forward_graph_replay_equals_eager(tmp_path):
    """NeRF-only steps replay the eval-mode DeepLab forward as a HIP graph:
    same outputs as the eager eval forward, parameter updates made between
    replays are seen, a new input shape gets its own capture, and the
    model's train/eval flags are left as they were."""
    from ucsa_neural_rendering_amd.lightning import JointTrainLightningNet
    exp = _tiny_exp()
    model = JointTrainLightningNet(exp, {"results": str(tmp_path), "scannet": str(tmp_path)}).cuda()
    model.train()
    g = torch.Generator().manual_seed(5)

    def eager(img):
        model.seg_model.eval()
        with torch.no_grad():
            o = model.forward_seg({"img": img})
        model.seg_model.train()
        return o

    img = torch.rand(2, 3, 48, 64, generator=g).cuda()
    want = eager(img)
    got = model.forward_seg_frozen({"img": img})
    assert model.seg_model.training
    assert len(model._seg_graphs) == 1 and next(iter(model._seg_graphs.values())) is not False, \
        "the forward was not captured"
    assert float((got["seg_semantics_raw"] - want["seg_semantics_raw"]).abs().max()) <= 1e-5
    assert (got["seg_semantics"] != want["seg_semantics"]).float().mean() < 1e-3
    # another image through the same graph
    img2 = torch.rand(2, 3, 48, 64, generator=g).cuda()
    want2 = eager(img2)
    got2 = model.forward_seg_frozen({"img": img2})
    assert len(model._seg_graphs) == 1
    assert float((got2["seg_semantics_raw"] - want2["seg_semantics_raw"]).abs().max()) <= 1e-5
    # in-place parameter update (an optimizer step) is seen by the replay
    with torch.no_grad():
        model.seg_model._model.classifier[-1].bias[0] += 3.0
    want3 = eager(img2)
    got3 = model.forward_seg_frozen({"img": img2})
    assert float((want3["seg_semantics_raw"] - want2["seg_semantics_raw"]).abs().max()) > 1e-4
    assert float((got3["seg_semantics_raw"] - want3["seg_semantics_raw"]).abs().max()) <= 1e-5
    # a different shape: second capture
    img4 = torch.rand(1, 3, 48, 64, generator=g).cuda()
    want4 = eager(img4)
    got4 = model.forward_seg_frozen({"img": img4})
    assert len(model._seg_graphs) == 2
    assert float((got4["seg_semantics_raw"] - want4["seg_semantics_raw"]).abs().max()) <= 1e-5
    # switch: eager path
    model.seg_graph = False
    got5 = model.forward_seg_frozen({"img": img4})
    assert float((got5["seg_semantics_raw"] - want4["seg_semantics_raw"]).abs().max()) <= 1e-6


@pytest.mark.parametrize("opts", [{"channels_last": True}, {"amp": "bf16"}, {"seg_graph": False}])
def test_module_layout_and_precision_options_train(tmp_path, opts):
    """`model: {channels_last | amp: bf16 | seg_graph: false}`: a NeRF-only
    epoch and a joint epoch run and log finite losses."""
    from ucsa_neural_rendering_amd.lightning import (JointTrainDataModule,
                                                     JointTrainLightningNet, Trainer)
    exp = _tiny_exp()
    exp["model"].update(opts)
    env = {"results": str(tmp_path / "r"), "scannet": str(tmp_path)}
    model = JointTrainLightningNet(exp, env)
    dm = JointTrainDataModule(exp)
    dm.setup()
    tr = Trainer(max_epochs=1, default_root_dir=str(tmp_path), limit_batches=2)
    model.joint_train = False
    tr.fit(model, train_dataloaders=dm.train_dataloader_nerf())
    model.joint_train = True
    tr.fit(model, train_dataloaders=dm.train_dataloader_joint())
    logged = model.logged
    for k in ("train/loss_nerf_rgb", "train/loss_depth", "train/loss_seg"):
        assert k in logged and logged[k] == logged[k] and abs(logged[k]) < 1e6, (k, logged)
    rows = tr.logger.history
    assert any(r["name"] == "train/loss_seg" for r in rows)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_nerf_loss_kernel_matches_reference_fixture(ops, tag):
    """ucsa_nerf_loss against G6: values and gradients produced by the
    reference's own forward_nerf_train + weighting (:167-223, :503-507)."""
    from tests.test_oracle_golden import _g6_case
    from ucsa_neural_rendering_amd import losses as ul
    g = load_golden("g6_nerf_losses.npz")
    gt_rgb, labels, gt_depth = _g6_case(g, tag)
    image = g[f"{tag}_image"].clone().cuda().requires_grad_()
    depth = g[f"{tag}_depth"].clone().cuda().requires_grad_()
    sem = g[f"{tag}_sem"].clone().cuda().requires_grad_()
    lc, ls, ld = ul.nerf_losses(image, sem, depth, gt_rgb.cuda(), labels.cuda(),
                                gt_depth.cuda(), float(g[f"{tag}_uom"]),
                                none_if_invalid=True)
    assert (ls is None) == bool(g[f"{tag}_sem_is_none"])
    total = ul.nerf_total_loss(lc, ls, ld)
    total.backward()
    assert abs(float(lc) - float(g[f"{tag}_loss_color"])) <= 1e-6
    assert abs(float(ld) - float(g[f"{tag}_loss_depth"])) <= 1e-6
    if ls is not None:
        assert abs(float(ls) - float(g[f"{tag}_loss_sem"])) <= 2e-6 * abs(float(g[f"{tag}_loss_sem"]))
    assert abs(float(total) - float(g[f"{tag}_total"])) <= 2e-6
    assert maxabs(image.grad, g[f"{tag}_g_image"]) <= 1e-8
    assert maxabs(depth.grad, g[f"{tag}_g_depth"]) <= 1e-8
    gs = torch.zeros_like(sem) if sem.grad is None else sem.grad
    assert maxabs(gs, g[f"{tag}_g_sem"]) <= 1e-6 * max(1.0, float(g[f"{tag}_g_sem"].abs().max()))


def test_no_valid_depth_pixel_behaves_like_the_reference(ops):
    """what happens when no drawn pixel has depth.  The reference
    (joint_train_lightning_net.py:218-221): loss_depth = mean of an empty
    selection = NaN, logged; its gradient is the scatter of an empty tensor =
    zeros, the other gradients stay finite, GradScaler does not skip the step.
    The kernel does the same."""
    from ucsa_neural_rendering_amd import losses as ul
    rgb, sem, depth, gt, labels, _ = _loss_inputs(200)
    gtd = torch.zeros(1, 200)
    # reference arithmetic, inline
    p = depth.clone().requires_grad_()
    c = rgb.clone().requires_grad_()
    ld = torch.nn.L1Loss(reduction="none")(p[gtd != 0] / 0.7, gtd[gtd != 0]).mean(-1)
    lc = torch.nn.MSELoss(reduction="none")(c, gt).mean()
    ((lc + 0.1 * ld) * 1024.0).backward()
    assert torch.isnan(ld) and float(p.grad.abs().max()) == 0.0 and torch.isfinite(c.grad).all()
    a = [t.clone().cuda().requires_grad_() for t in (rgb, sem, depth)]
    hc, hs, hd = ul.nerf_losses(a[0], a[1], a[2], gt.cuda(), labels.cuda(), gtd.cuda(), 0.7)
    (ul.nerf_total_loss(hc, hs, hd) * 1024.0).backward()
    assert torch.isnan(hd) and float(a[2].grad.abs().max()) == 0.0
    assert torch.isfinite(a[0].grad).all() and torch.isfinite(a[1].grad).all()
    assert maxabs(a[0].grad, c.grad) <= 1e-6 * float(c.grad.abs().max())
