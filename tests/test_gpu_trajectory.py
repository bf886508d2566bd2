"""Trained-quality parity (north_star: "matched mIoU/PSNR +-0.5"): the CPU
oracle (its runs: committed fixtures, tests/golden/make_trajectory_golden.py) and
the HIP path TRAINED on the same schedule -- same initialisation (tcnn
style, seed 123), same frames, same ray indices, same stratified-sampling
(``rng_t``) and inverse-CDF (``rng_u``) tensors at every step, the reference's
loss weights and Adam settings (reference
``nr4seg/lightning/joint_train_lightning_net.py:473-513`` training_step_nerf,
``:897-919`` optimizer) -- for 150 steps of 2048 rays x (16+16) samples on the
synthetic room.  At 9 checkpoints (every 10 steps from step 70) all 12 views
(8 trained on, 4 held out) are rendered by each side with ITS OWN parameters
and compared with the analytic ground truth; the MEANS over the checkpoints
must agree:

    |PSNR_hip - PSNR_oracle| <= 0.5 dB,  |mIoU_hip - mIoU_oracle| <= 1.0 pt

on the training views (what the reference's final test pass renders; held-out
views: +-1.5 dB, see ``_compare``), for ``train_precision`` fp32, bf16x3
(against the fp32 oracle) and tcnn (against the oracle emulating tiny-cuda-nn's
fp16 roundings: fp16 table and features, fp16 weights / layer inputs).

Why means over checkpoints, and 2048 rays: with the reference's Adam (lr 1e-2,
eps 1e-15) the loss of a small-batch run oscillates by tens of per cent from
step to step, and two fp32 runs decorrelate -- measured at 512 rays x (32+32):
HIP vs HIP (float atomics in the grid backward) +-0.5 dB on a single parameter
state after 150 steps, +-0.9 dB on held-out views.  At 2048 rays the same
comparison gives 0.007 - 0.23 dB; single checkpoints of HIP and oracle differ
by 0.05 - 0.25 dB up to step 150 and by up to 2 dB / 3 pt beyond (hence 150
steps), their means by 0.05 - 0.3 dB.  The oracle itself is not reproducible
to better than that: its trajectory depends on the host's thread count
(33.1 / 33.4 / 33.6 dB over three boxes).  mIoU: after 150 steps the semantic
head (loss weight 0.04) is still in its chance-to-learning transition
(24.8 -> 40 -> 31 -> 38 pt from checkpoint to checkpoint, a dozen classes
present: one class flipping is 8 pt on a checkpoint), so its mean is held to
+-1.0 pt, not +-0.5 (observed |d| 0.02 - 0.7).  Round-4 measurements, DESIGN 2.  Per-step gradient
parity is tests/test_gpu_backward.py / test_gpu_configs.py; this is the
trajectory.  ``-m gpu``."""
import copy

import numpy as np
import pytest
import torch

from oracle import field as ofield
from oracle import losses as olosses
from oracle import metrics as ometrics
from oracle import renderer as oren
from tests.util import AABB4, hip_network_from_oracle

pytestmark = pytest.mark.gpu

import collections

C = 40
H, W, VIEWS, HELD = 48, 64, 12, 4   # frames 0..7 train, frames 8..11 held out
LR, WD = 1e-2, 1e-6
# quality = mean over the parameter states after `checkpoints` (0-based steps)
Run = collections.namedtuple("Run", "steps n T t checkpoints seed")
# round 4: by step 70 the loss has fallen from 0.545 to ~0.009; the semantic head is
# still in its chance-to-learning transition at step 150
SHORT = Run(150, 2048, 16, 16, (69, 79, 89, 99, 109, 119, 129, 139, 149), 2024)
# round 6: a horizon on which the semantic head HAS converged --
# 600 steps of 4096 rays (the reference's batch, joint_train_lightning_net.py:141),
# quality at 100 checkpoints (every 5 steps from step 100), the oracle's side a
# committed fixture (tests/golden/g9_trajectory_long.npz) -- held to north_star's
# +-0.5 dB / +-0.5 pt
LONG = Run(600, 4096, 16, 16, tuple(range(99, 600, 5)), 2025)
STEPS, N, T, t = SHORT.steps, SHORT.n, SHORT.T, SHORT.t      # (the short run's, for the tools)
CHECKPOINTS = SHORT.checkpoints


@pytest.fixture(scope="module")
def scene():
    """Frames of the synthetic room (GT by analytic ray casting on the GPU),
    and the per-step random tensors, all as CPU tensors."""
    return (_frames(),) + _draws(SHORT)


def _draws(run):
    """The per-step random tensors of a run (frame, ray indices, rng_t, rng_u) and the
    uniforms of the evaluation renders, from the run's seed."""
    g = torch.Generator().manual_seed(run.seed)
    draws = [dict(frame=int(torch.randint(0, VIEWS - HELD, (1,), generator=g)),
                  inds=torch.randint(0, H * W, (run.n,), generator=g),
                  rt=torch.rand(run.n, run.T, generator=g), ru=torch.rand(run.n, run.t, generator=g))
             for _ in range(run.steps)]
    u_eval = torch.rand(VIEWS * H * W, run.t, generator=g)
    return draws, u_eval


def _batch(frames, dr):
    f, i = frames[dr["frame"]], dr["inds"]
    return (f["o"][i][None], f["d"][i][None], f["nrm"][i][None], f["rgb"][i][None],
            f["label"][i][None], f["depth"][i][None])


def _all(frames, key):
    return torch.cat([f[key] for f in frames], 0)


def _quality(img, sem, frames):
    """Per view set -- "train" = every pixel of the 8 training frames (what the
    reference's final test pass renders: scripts/train_joint.py tests on the
    NeRF TRAIN loader after the joint phase), "held" = the 4 held-out frames
    (its test_after_nerf split) -- PSNR as the mean over the views of the
    per-view PSNR (the module's test loop) and mIoU from one
    confusion matrix."""
    img, gt = img.cpu().view(VIEWS, H * W, 3), _all(frames, "rgb").view(VIEWS, H * W, 3)
    per_view = -10 * torch.log10(((img - gt) ** 2).mean((1, 2)))
    _, lab = olosses.semantic_postproc(sem.cpu())
    lab, gt_lab = lab.view(VIEWS, -1).numpy(), _all(frames, "label").view(VIEWS, -1).numpy()
    k = VIEWS - HELD
    miou = lambda a, b: 100.0 * ometrics.measure(ometrics.confusion(a, b, C))[0]
    return {"train": (float(per_view[:k].mean()), miou(lab[:k], gt_lab[:k])),
            "held": (float(per_view[k:].mean()), miou(lab[k:], gt_lab[k:]))}


def _mean_quality(quals):
    """Mean over the checkpoints (module docstring)."""
    out = {k: (float(np.mean([q[k][0] for q in quals])), float(np.mean([q[k][1] for q in quals])))
           for k in ("train", "held")}
    out["per_checkpoint_train_psnr"] = [round(q["train"][0], 2) for q in quals]
    out["per_checkpoint_train_miou"] = [round(q["train"][1], 1) for q in quals]
    return out


def _train_oracle(frames, draws, u_eval, emulate_tcnn, checkpoints=SHORT.checkpoints,
                  raw_quals=False, progress=None):
    fld = ofield.OracleField(bound=4.0, num_semantic_classes=C, seed=123)
    if emulate_tcnn:
        fld.emulate_fp16 = True
        fld.fp16_table = True
    fld.requires_grad_(True)
    st = [dict(m=torch.zeros_like(p), v=torch.zeros_like(p)) for p in fld.parameters()]
    losses, quals = [], []

    def evaluate():
        with torch.no_grad():
            out = oren.render(fld, _all(frames, "o")[None], _all(frames, "d")[None],
                              _all(frames, "nrm")[None], AABB4, staged=True,
                              max_ray_batch=6144, num_steps=T, upsample_steps=t,
                              u=u_eval[None])
        quals.append(_quality(out["image"][0], out["semantics"][0], frames))

    for k, dr in enumerate(draws):
        o, d, nrm, rgb, lab, dep = _batch(frames, dr)
        out = oren.run(fld, o, d, nrm, AABB4, num_steps=T, upsample_steps=t,
                       t_rand=dr["rt"], u=dr["ru"])
        lc, ls, ld = olosses.nerf_losses(out["image"], out["semantics"], out["depth"],
                                         rgb, lab, dep, 1.0)
        loss = olosses.nerf_total_loss(lc, ls, ld)
        for p in fld.parameters():
            p.grad = None
        loss.backward()
        losses.append(float(loss.detach()))
        with torch.no_grad():
            for i, (p, s) in enumerate(zip(fld.parameters(), st)):
                pn, s["m"], s["v"] = olosses.adam_step(
                    p, p.grad, s["m"], s["v"], k + 1, LR, weight_decay=0.0 if i == 0 else WD)
                p.copy_(pn)
        if k in checkpoints:
            evaluate()
            if progress is not None:
                progress(k, quals[-1], losses)
    return (quals if raw_quals else _mean_quality(quals)), losses, {}


def _train_hip(frames, draws, u_eval, precision, deterministic=False, steps=None,
               checkpoints=SHORT.checkpoints, raw_quals=False):
    from ucsa_neural_rendering_amd import losses as ul
    from ucsa_neural_rendering_amd.nerf.optim import HipAdam
    net = hip_network_from_oracle(ofield.OracleField(bound=4.0, num_semantic_classes=C,
                                                     seed=123)).train()
    net.train_precision = precision
    net.deterministic = deterministic
    if steps is not None:
        draws = draws[:steps]
    opt = HipAdam([{"name": "encoding", "params": list(net.encoder.parameters())},
                   {"name": "net", "params": list(net.sigma_net.parameters()) +
                    list(net.color_net.parameters()) + list(net.semantics_net.parameters()),
                    "weight_decay": WD}], lr=LR, betas=(0.9, 0.99), eps=1e-15)
    # the reference steps the NeRF optimizer through a GradScaler (:46, :509-513);
    # the tcnn arithmetic (fp16 gradients between layers) needs its scale
    scaler = torch.amp.GradScaler("cuda", enabled=True)
    losses, quals = [], []
    ev_rays = [_all(frames, k)[None].cuda() for k in ("o", "d", "nrm")]
    infer = {"fp32": "fp32", "bf16x3": "bf16x3", "tcnn": "fp16"}[precision]

    def evaluate():
        # inference arithmetic of the same family as the training one
        net.eval()
        net.precision, net.fp16_table = infer, precision == "tcnn"
        with torch.no_grad():
            out = net.render(*ev_rays, staged=True, num_steps=T, upsample_steps=t,
                             rng_u=u_eval.cuda())
        net.train()
        quals.append(_quality(out["image"][0], out["semantics"][0], frames))

    for k, dr in enumerate(draws):
        o, d, nrm, rgb, lab, dep = [x.cuda() for x in _batch(frames, dr)]
        out = net.render(o, d, nrm, perturb=True, num_steps=T, upsample_steps=t,
                         rng_t=dr["rt"].cuda(), rng_u=dr["ru"].cuda())
        lc, ls, ld = ul.nerf_losses(out["image"], out["semantics"], out["depth"], rgb, lab,
                                    dep, 1.0)
        loss = ul.nerf_total_loss(lc, ls, ld)
        opt.zero_grad()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(loss.detach())
        if k in checkpoints:
            evaluate()
    skipped = sum(int(v) for v in opt._skipped.values()) if opt._skipped else 0
    info = {"skipped_steps": skipped, "final_scale": float(scaler.get_scale())}
    return (quals if raw_quals else _mean_quality(quals)), [float(x) for x in torch.stack(losses).cpu()], info


def _hip_mean(scene, precision, runs=2):
    """Mean quality of `runs` HIP trajectories (each run sees a different
    round-off: float atomics in the grid backward), losses / info of the first;
    halves the HIP side's share of the run-to-run spread for ~6 s per run."""
    outs = [_train_hip(*scene, precision) for _ in range(runs)]
    q = {k: (float(np.mean([o[0][k][0] for o in outs])), float(np.mean([o[0][k][1] for o in outs])))
         for k in ("train", "held")}
    for k in ("per_checkpoint_train_psnr", "per_checkpoint_train_miou"):
        q[k] = [round(float(x), 2) for x in np.mean([o[0][k] for o in outs], 0)]
    spread = max(abs(outs[0][0]["train"][0] - o[0]["train"][0]) for o in outs)
    return q, outs[0][1], dict(outs[0][2], hip_runs=runs, hip_run_spread_db=round(spread, 3))


def short_oracle(kind):
    """(mean quality, losses, {}) of the oracle's SHORT run `kind` ("fp32" | "tcnn") from
    the committed fixture tests/golden/g9_trajectory_short.npz -- what _train_oracle
    returns for it (tests/golden/make_trajectory_golden.py short;
    tests/test_trajectory_golden_cpu.py re-runs its first steps)."""
    from tests.util import load_golden
    g = load_golden("g9_trajectory_short.npz")
    assert tuple(int(x) for x in g["checkpoints"]) == SHORT.checkpoints and int(g["steps"]) == SHORT.steps
    assert int(g["rays"]) == SHORT.n and int(g["seed"]) == SHORT.seed
    col = lambda k: [float(x) for x in g[f"{kind}_{k}"]]
    quals = [{"train": (p, m), "held": (hp, hm)} for p, m, hp, hm in
             zip(col("psnr"), col("miou"), col("held_psnr"), col("held_miou"))]
    return _mean_quality(quals), col("losses"), {}


@pytest.fixture(scope="module")
def oracles():
    """Both short oracle trajectories (fp32, fp16-emulating).  Until round 6 they were
    trained here, in CPU worker processes beside the HIP runs (147 s of the suite's
    500); they are a committed fixture now, like the long horizon's."""
    return {k: short_oracle(k) for k in ("fp32", "tcnn")}


def _frames():
    """Frames of the synthetic room as CPU tensors, built WITHOUT the GPU: rays from
    oracle.rays.pixel_rays (bit-equal to ucsa_get_rays, G1), ground truth by the
    room's analytic ray casting (plain torch) -- the same frames for the HIP side, the
    oracle trainer and tests/golden/make_trajectory_golden.py."""
    from oracle.rays import pixel_rays
    from ucsa_neural_rendering_amd.dataset.synthetic_scene import SyntheticRoom, _slerp_loop_poses
    room = SyntheticRoom(3, n_classes=C)
    poses = _slerp_loop_poses(VIEWS, seed=123 + 3)
    intr = (0.89 * W, 0.89 * W, W / 2.0, H / 2.0)
    frames = []
    with torch.no_grad():
        for i in range(VIEWS):
            o, d, n = pixel_rays(poses[i:i + 1], intr, H, W)
            t_hit, rgb, label = room.cast(o[0], d[0])
            depth = (t_hit / n[0, :, 0]).half().float()          # the dataset hands out fp16 depth
            frames.append(dict(o=o[0].contiguous(), d=d[0].contiguous(), nrm=n[0].contiguous(),
                               rgb=rgb.contiguous(), label=label.reshape(-1), depth=depth.reshape(-1)))
    return frames


HELD_OUT_DB = 1.5


def _compare(tag, hip, ora, tol_db=0.5, tol_pt=1.0):
    """+-0.5 dB / +-1.0 pt on the training views (the reference's final test
    set).  Held-out views are reported and held to +-1.5 dB only: with the
    reference's Adam (eps 1e-15) a grid entry whose gradient is round-off
    noise still moves by a full lr step in the noise's direction, so runs that
    agree on the training loss to 2 % differ more where only such entries
    decide -- in cells no training ray constrained (observed here: <= 0.3 dB)."""
    (qh, lh, ih), (qo, lo, _) = hip, ora
    lh, lo = np.array(lh), np.array(lo)
    for k in ("train", "held"):
        (ph, mh), (po, mo) = qh[k], qo[k]
        print(f"{tag} [{k} views]: PSNR hip {ph:.3f} dB / oracle {po:.3f} dB (d {ph - po:+.3f}); "
              f"mIoU hip {mh:.2f} / oracle {mo:.2f} pt (d {mh - mo:+.2f})")
    print(f"{tag} train-view PSNR per checkpoint hip {qh['per_checkpoint_train_psnr']} / oracle "
          f"{qo['per_checkpoint_train_psnr']}")
    print(f"{tag} train-view mIoU per checkpoint hip {qh['per_checkpoint_train_miou']} / oracle "
          f"{qo['per_checkpoint_train_miou']}")
    print(f"{tag}: loss step 1 {lh[0]:.5f} / {lo[0]:.5f}, mean of last 20 {lh[-20:].mean():.5f} / "
          f"{lo[-20:].mean():.5f}; max |loss difference| first 20 steps "
          f"{np.abs(lh[:20] - lo[:20]).max():.2e}; {ih}")
    print(f"{tag} loss every 10 steps hip   : " + " ".join(f"{x:.4f}" for x in lh[::10]))
    print(f"{tag} loss every 10 steps oracle: " + " ".join(f"{x:.4f}" for x in lo[::10]))
    assert lo[-10:].mean() < 0.6 * lo[0] and lh[-10:].mean() < 0.6 * lh[0]   # both learned
    assert qo["train"][0] > 14.0                          # the run means something
    assert abs(qh["train"][0] - qo["train"][0]) <= tol_db, (tag, qh, qo)
    assert abs(qh["train"][1] - qo["train"][1]) <= tol_pt, (tag, qh, qo)
    assert abs(qh["held"][0] - qo["held"][0]) <= HELD_OUT_DB, (tag, qh, qo)
    return lh, lo


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_trajectory_quality_matches_the_fp32_oracle(scene, oracles, precision):
    hip = _train_hip(*scene, precision)
    oracle_fp32 = oracles["fp32"]
    if precision == "fp32":
        # run-to-run spread of the HIP path itself (float atomics in the grid
        # backward: a different round-off every run), for the record
        again = _train_hip(*scene, precision)[0]
        print("fp32 HIP run 2 vs run 1: train-view PSNR %+.3f dB, held-out %+.3f dB"
              % (again["train"][0] - hip[0]["train"][0], again["held"][0] - hip[0]["held"][0]))
    lh, lo = _compare(precision, hip, oracle_fp32)
    # before round-off has had time to grow the two runs are the same run
    assert np.abs(lh[:5] - lo[:5]).max() <= 2e-5 * max(1.0, lo[0])


def long_run_stats(psnr, miou, late):
    """The two statistics of ONE run that are comparable between runs: the mean PSNR
    over all checkpoints, the MEDIAN mIoU over the converged window (a run now and
    then leaves the converged state for a few dozen steps -- the oracle itself does,
    76 -> 36 pt and 75 -> 83 pt in two of its six runs -- the median ignores that)."""
    psnr, miou = np.asarray(psnr), np.asarray(miou)
    return float(psnr.mean()), float(np.median(miou[late]))


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
def test_long_horizon_quality_within_half_a_db_and_half_a_point(precision):
    """north_star: ">= 10x ... at matched mIoU / PSNR (+-0.5)" -- on a horizon where the
    semantic head HAS converged: 600 steps of 4096 rays (the
    reference's batch) from the same initial state on the same draws, quality at 101
    checkpoints (every 5 steps from step 100).  The oracle's side is a committed
    fixture, tests/golden/g9_trajectory_long.npz: SIX runs of the CPU oracle trainer
    with different BLAS thread counts (tests/golden/make_trajectory_golden.py, ~25 min
    of CPU each -- too long for the suite, and the stored series is the same evidence).

    What is comparable.  With the reference's Adam (lr 1e-2, eps 1e-15, no schedule)
    two runs of ONE implementation decorrelate after ~150 steps: single checkpoints
    differ by +-2 ... 4 dB, the mean PSNR of a run over its 101 checkpoints scatters
    with sigma ~0.25 dB (six oracle runs, four HIP runs), the late-window mean mIoU by
    several points when a run has an excursion.  So the test compares run-AVERAGED
    statistics (long_run_stats: mean PSNR over all checkpoints; median mIoU from step
    425 on, the head converges between steps 350 and 400 in every run: ~37 -> ~74 pt):

        |mean over HIP runs - mean over the six oracle runs| <= 0.5 dB and <= 0.5 pt

    for the default training arithmetic (bf16x3 forward, bf16x2 backward, x-pair grid
    records) and the exact fp32 one.  Two HIP runs first (~25 s each); if the
    difference of the means is outside the bound, two more are added (at most six) and
    the means re-evaluated: more runs only shrink the sampling error of the HIP mean,
    the criterion stays +-0.5.  The oracle runs' own scatter is printed as the
    yardstick."""
    from tests.util import load_golden
    gold = load_golden("g9_trajectory_long.npz")
    ck = tuple(int(x) for x in gold["checkpoints"])
    assert ck == LONG.checkpoints and int(gold["steps"]) == LONG.steps and int(gold["rays"]) == LONG.n
    late = np.array(ck) >= 424
    psnr_o, miou_o = gold["psnr"].numpy(), gold["miou"].numpy()          # [runs, 101]
    ora = np.array([long_run_stats(p, m, late) for p, m in zip(psnr_o, miou_o)])
    for k, (p, m) in enumerate(ora):
        print(f"long[{precision}] oracle run {k} ({int(gold['threads'][k])} threads): PSNR mean {p:.3f} dB, "
              f"late-window mIoU median {m:.2f} pt (mean {miou_o[k][late].mean():.2f})")
    assert ora[:, 1].mean() > 60.0 and ora[:, 0].mean() > 30.0          # the run means something
    frames = _frames()
    draws, u_eval = _draws(LONG)
    hip = []
    while True:
        for _ in range(2):
            q, losses, info = _train_hip(frames, draws, u_eval, precision, checkpoints=LONG.checkpoints,
                                         raw_quals=True)
            p, m = [x["train"][0] for x in q], [x["train"][1] for x in q]
            hip.append(long_run_stats(p, m, late))
            print(f"long[{precision}] hip run {len(hip)}: PSNR mean {hip[-1][0]:.3f} dB, late-window mIoU median "
                  f"{hip[-1][1]:.2f} pt (mean {np.asarray(m)[late].mean():.2f}); loss last 20 "
                  f"{np.mean(losses[-20:]):.5f}; {info}")
        h = np.array(hip)
        d_psnr, d_miou = float(h[:, 0].mean() - ora[:, 0].mean()), float(h[:, 1].mean() - ora[:, 1].mean())
        print(f"long[{precision}] {len(hip)} hip runs - {len(ora)} oracle runs: PSNR {d_psnr:+.3f} dB, mIoU "
              f"{d_miou:+.3f} pt; oracle run-to-run sigma {ora[:, 0].std(ddof=1):.3f} dB / "
              f"{ora[:, 1].std(ddof=1):.3f} pt, hip {h[:, 0].std(ddof=1):.3f} dB / {h[:, 1].std(ddof=1):.3f} pt")
        if (abs(d_psnr) <= 0.5 and abs(d_miou) <= 0.5) or len(hip) >= 6:
            break
    assert abs(d_psnr) <= 0.5, (d_psnr, hip, ora.tolist())
    assert abs(d_miou) <= 0.5, (d_miou, hip, ora.tolist())


def test_trajectory_quality_tcnn_numerics_matches_the_fp16_emulating_oracle(scene, oracles):
    """+-1.0 dB / +-1.0 pt here: the HIP path also rounds the gradients between
    the layers to fp16 under a loss scale (as tiny-cuda-nn does) and its grid
    gradient travels as half2 records; the oracle emulates the FORWARD roundings
    only (gradients pass in fp32).  The two therefore decorrelate from the first
    steps on, not after ~100 like two fp32-grade runs (measured: -0.41 / +0.05
    / +0.12 dB over three runs)."""
    hip = _hip_mean(scene, "tcnn")
    _compare("tcnn", hip, oracles["tcnn"], tol_db=1.0, tol_pt=1.0)


def test_deterministic_mode_makes_the_trajectory_reproducible(scene):
    """`UCSA_DETERMINISTIC=1` (net.deterministic): the grid gradient through the
    order-independent fixed-point reduction.  Two trainings from the same state
    on the same draws then follow the SAME trajectory bit for bit -- every loss of
    80 steps and the rendered quality at the checkpoints inside them -- where
    two default runs decorrelate (module docstring: +-0.007 ... 0.23 dB at this
    batch size).  The default path's trajectory stays within its own run-to-run
    spread of it."""
    a = _train_hip(*scene, "bf16x3", deterministic=True, steps=80)
    b = _train_hip(*scene, "bf16x3", deterministic=True, steps=80)
    assert a[1] == b[1]                                   # all 80 losses, exactly
    assert a[0]["per_checkpoint_train_psnr"] == b[0]["per_checkpoint_train_psnr"]
    assert a[0]["train"] == b[0]["train"] and a[0]["held"] == b[0]["held"]
    c = _train_hip(*scene, "bf16x3", deterministic=False, steps=80)
    print("deterministic vs default run, 80 steps: train-view PSNR %+.3f dB, first differing loss at step %s"
          % (c[0]["train"][0] - a[0]["train"][0],
             next((i for i, (x, y) in enumerate(zip(a[1], c[1])) if x != y), None)))
    assert abs(c[0]["train"][0] - a[0]["train"][0]) <= 0.5
