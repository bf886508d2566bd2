#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the
REFERENCE's own Python for the parts of the hot path that are importable here
(SURVEY.md 8c): ``nr4seg/nerf/renderer_semantics.py``,
``nr4seg/nerf/activation.py``, ``nr4seg/dataset/ngp_utils.py`` and
``nr4seg/utils/metrics.py``.

Run in the authoring container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference source never travels: only the input/output arrays written here
are committed.  Stubs are supplied for modules the reference imports but the
path does not use (``trimesh`` -- only in plot_pointcloud) or that are CUDA
only (``nr4seg.nerf.raymarching`` -- replaced by the oracle's CPU slab test,
which is itself KAT-pinned in tests/test_oracle_rays.py).

Random draws inside the reference (``torch.rand`` in run() and sample_pdf())
are replayed from recorded tensors by temporarily replacing ``torch.rand``.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("UCSA_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from oracle import field as ofield  # noqa: E402
from oracle import rays as orays  # noqa: E402


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name,
                                                  os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    if not hasattr(np, "float"):
        np.float = float  # reference metrics.py:51 uses the removed alias
    sys.modules.setdefault("trimesh", types.ModuleType("trimesh"))
    for pkg in ("nr4seg", "nr4seg.nerf", "nr4seg.dataset", "nr4seg.utils"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    rm_pkg = types.ModuleType("nr4seg.nerf.raymarching")
    rm = types.SimpleNamespace(near_far_from_aabb=lambda o, d, aabb, min_near=0.2: orays.near_far_from_aabb(o, d, aabb, min_near))
    rm_pkg.raymarching = rm
    sys.modules["nr4seg.nerf.raymarching"] = rm_pkg
    act = _load("nr4seg.nerf.activation", "nr4seg/nerf/activation.py")
    ren = _load("nr4seg.nerf.renderer_semantics",
                "nr4seg/nerf/renderer_semantics.py")
    ngp = _load("nr4seg.dataset.ngp_utils", "nr4seg/dataset/ngp_utils.py")
    met = _load("nr4seg.utils.metrics", "nr4seg/utils/metrics.py")
    return act, ren, ngp, met


class RandReplay:
    """Replace torch.rand by a queue of recorded tensors."""

    def __init__(self, tensors):
        self.q = list(tensors)

    def __enter__(self):
        self._orig = torch.rand

        def fake(*size, **kw):
            t = self.q.pop(0)
            shape = tuple(size[0]) if len(size) == 1 and not isinstance(
                size[0], int) else tuple(size)
            assert tuple(t.shape) == shape, (t.shape, shape)
            return t.clone()

        torch.rand = fake
        return self

    def __exit__(self, *a):
        torch.rand = self._orig
        assert not self.q, "unused recorded random tensors"


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} kB)")


# ---------------------------------------------------------------------------
def make_rays(n, gen, inside=True):
    """Camera inside the bound-4 box, looking around."""
    o = (torch.rand(n, 3, generator=gen) * 2 - 1) * (2.5 if inside else 6.0)
    d = torch.randn(n, 3, generator=gen)
    d = d / d.norm(dim=-1, keepdim=True)
    norms = 1.0 + torch.rand(n, 1, generator=gen) * 0.3
    return o, d, norms


class TableField:
    """Field whose per-sample outputs are explicit leaf tensors: pins the
    sampling / merge / compositing arithmetic independently of tcnn.
    geo_feat column 0 carries the sample id so colour/semantics can look up
    their tables after the renderer's sort+gather."""

    def __init__(self, sig_c, sig_f, rgb_tab, prob_tab, C):
        self.sig = [sig_c, sig_f]
        self.rgb_tab, self.prob_tab, self.C = rgb_tab, prob_tab, C
        self.calls = 0
        self.base = [0, sig_c.numel()]

    def density(self, x):
        k = self.calls
        self.calls += 1
        s = self.sig[k].reshape(-1)
        ids = torch.arange(s.numel(), dtype=torch.float32) + self.base[k]
        geo = torch.zeros(s.numel(), 15)
        geo[:, 0] = ids
        return {"sigma": s, "geo_feat": geo}

    def color(self, x, d, mask=None, geo_feat=None, **kw):
        ids = geo_feat[:, 0].long()
        out = torch.zeros(mask.shape[0], 3)
        if not mask.any():
            return out
        out[mask] = self.rgb_tab[ids[mask]]
        return out

    def semantics(self, x, d, mask=None, geo_feat=None, **kw):
        ids = geo_feat[:, 0].long()
        out = torch.zeros(mask.shape[0], self.C)
        if not mask.any():
            return out
        out[mask] = self.prob_tab[ids[mask]]
        return out


def build_ref_renderer(ren, field_obj, bound, C):

    class R(ren.SemanticNeRFRenderer):

        def density(self, x):
            return field_obj.density(x)

        def color(self, x, d, mask=None, geo_feat=None, **kw):
            return field_obj.color(x, d, mask=mask, geo_feat=geo_feat)

        def semantics(self, x, d, mask=None, geo_feat=None, **kw):
            return field_obj.semantics(x, d, mask=mask, geo_feat=geo_feat)

    r = R(bound=bound, cuda_ray=False, density_scale=1,
          num_semantic_classes=C)
    return r


def main():
    act, ren, ngp, met = load_reference()
    torch.manual_seed(0)

    # ---- G0: trunc_exp -----------------------------------------------------
    x = torch.linspace(-20, 20, 41, requires_grad=True)
    y = act.trunc_exp(x)
    gy = torch.linspace(0.5, 1.5, 41)
    y.backward(gy)
    save("g0_trunc_exp.npz", x=x, y=y, gy=gy, gx=x.grad)

    # ---- G1: get_rays / nerf_matrix_to_ngp ---------------------------------
    g = torch.Generator().manual_seed(11)
    c2w = []
    for _ in range(3):
        q, _r = torch.linalg.qr(torch.randn(3, 3, generator=g))
        m = torch.eye(4)
        m[:3, :3] = q
        m[:3, 3] = torch.randn(3, generator=g)
        c2w.append(m.numpy())
    ngp_poses = np.stack([ngp.nerf_matrix_to_ngp(m) for m in c2w])
    poses = torch.from_numpy(ngp_poses)
    small_intr = np.array([7.1, 6.9, 4.2, 2.8], dtype=np.float32)
    r_small = ngp.get_rays(poses, small_intr, 6, 8)
    big_intr = np.array([0.89 * 640, 0.89 * 640, 320.0, 240.0],
                        dtype=np.float32)
    r_big = ngp.get_rays(poses[:1], big_intr, 480, 640)
    pick = torch.tensor([0, 1, 639, 640, 153600 + 320, 307199])
    save("g1_rays.npz", c2w=np.stack(c2w), ngp_poses=ngp_poses,
         small_intr=small_intr, small_o=r_small["rays_o"],
         small_d=r_small["rays_d"], small_n=r_small["direction_norms"],
         big_intr=big_intr, big_pick=pick,
         big_o=r_big["rays_o"][:, pick], big_d=r_big["rays_d"][:, pick],
         big_n=r_big["direction_norms"][:, pick])

    # ---- G3: sample_pdf -----------------------------------------------------
    g = torch.Generator().manual_seed(3)
    Nr, T, t = 8, 12, 9
    bins = torch.sort(torch.rand(Nr, T - 1, generator=g) * 5 + 0.2, -1)[0]
    w = torch.rand(Nr, T - 2, generator=g)
    w[1] = 0.0  # degenerate: all-zero weights
    w[2, :5] = 0.0  # flat cdf segments
    w[3] = 1e-7
    u = torch.rand(Nr, t, generator=g)
    u[4, 0] = 0.0
    u[4, 1] = 0.99999994
    with RandReplay([u]):
        z_new = ren.sample_pdf(bins, w, t, det=False)
    save("g3_sample_pdf.npz", bins=bins, weights=w, u=u, samples=z_new)

    # ---- G4: run() with the table field (fwd + autograd grads) -------------
    for tag, (N, T, t, perturb, staged) in {
            "a": (24, 16, 16, True, False),
            "b": (24, 24, 8, False, False),
            "c": (40, 16, 16, False, True),  # chunked: max_ray_batch 16
            "d": (16, 32, 0, True, False),  # upsample_steps = 0 path
    }.items():
        g = torch.Generator().manual_seed(40 + ord(tag))
        C = 5
        o, d, norms = make_rays(N, g)
        sig_c = (torch.rand(N, T, generator=g) * 6).pow(2).requires_grad_()
        sig_f = (torch.rand(N, max(t, 1), generator=g) *
                 6).pow(2)[:, :t].clone().requires_grad_()
        rgb_tab = torch.rand(N * (T + t), 3, generator=g).requires_grad_()
        prob_tab = torch.softmax(
            torch.randn(N * (T + t), C, generator=g) * 2,
            -1).clone().requires_grad_()
        t_rand = torch.rand(N, T, generator=g)
        u = torch.rand(N, max(t, 1), generator=g)[:, :t]
        chunk = 16 if staged else 4096

        class ChunkTable(TableField):
            pass

        if staged:
            # one independent table field per chunk (run() calls density twice
            # per chunk); rebuild the field for each chunk via a dispatcher
            outs = {"depth": [], "image": [], "semantics": []}
            for head in range(0, N, chunk):
                tail = min(head + chunk, N)
                n = tail - head
                # the rgb/prob tables are indexed by ids local to the chunk:
                # [coarse n*T | fine n*t]
                idc = torch.arange(head * T, tail * T)
                idf = N * T + torch.arange(head * t, tail * t)
                ids = torch.cat([idc, idf])
                fld = TableField(sig_c[head:tail], sig_f[head:tail],
                                 rgb_tab[ids], prob_tab[ids], C)
                rr = build_ref_renderer(ren, fld, 4, C)
                rr.eval()
                rec = []
                if perturb:
                    rec.append(t_rand[head:tail])
                if t > 0:
                    rec.append(u[head:tail])
                with RandReplay(rec):
                    res = rr.render(o[None, head:tail], d[None, head:tail],
                                    norms[None, head:tail], staged=False,
                                    perturb=perturb, num_steps=T,
                                    upsample_steps=t)
                for k in outs:
                    outs[k].append(res[k])
            res = {k: torch.cat(v, dim=1) for k, v in outs.items()}
        else:
            fld = TableField(sig_c, sig_f, rgb_tab, prob_tab, C)
            rr = build_ref_renderer(ren, fld, 4, C)
            rr.train()
            rec = []
            if perturb:
                rec.append(t_rand)
            if t > 0:
                rec.append(u)
            with RandReplay(rec):
                res = rr.render(o[None], d[None], norms[None], staged=False,
                                perturb=perturb, num_steps=T,
                                upsample_steps=t)
        ci = torch.rand(1, N, 3, generator=g)
        cd = torch.rand(1, N, generator=g)
        cs = torch.rand(1, N, C, generator=g)
        loss = (res["image"] * ci).sum() + (res["depth"] * cd).sum() + (
            res["semantics"] * cs).sum()
        loss.backward()
        save(f"g4{tag}_run_table.npz", N=N, T=T, t=t, perturb=perturb, C=C,
             staged=staged, chunk=chunk, rays_o=o, rays_d=d, norms=norms,
             sig_c=sig_c, sig_f=sig_f, rgb_tab=rgb_tab, prob_tab=prob_tab,
             t_rand=t_rand, u=u, image=res["image"], depth=res["depth"],
             semantics=res["semantics"], ci=ci, cd=cd, cs=cs,
             g_sig_c=sig_c.grad, g_sig_f=sig_f.grad if t > 0 else sig_f * 0,
             g_rgb=rgb_tab.grad, g_prob=prob_tab.grad)

    # ---- G5: run() with the restated hash/SH/MLP field ---------------------
    # (renderer arithmetic = reference; field arithmetic = oracle restatement)
    fld = ofield.OracleField(bound=4.0, num_semantic_classes=40, seed=123)
    gs = torch.Generator().manual_seed(77)
    # livelier grid than the 1e-4 init so sigma / masks are non-trivial
    fld.grid_params = (torch.rand(fld.grid.n_params, generator=gs) * 2 -
                       1) * 3.0
    chk = dict(grid_sum=fld.grid_params.double().sum(),
               grid_head=fld.grid_params[:8],
               sigma_head=fld.sigma_params[:8],
               color_head=fld.color_params[:8], sem_head=fld.sem_params[:8])
    for tag, (N, T, t, perturb, staged) in {
            "a": (64, 16, 16, True, False),
            "b": (64, 16, 16, False, True),
            "c": (96, 96, 96, False, True),
            "d": (32, 96, 96, True, False),
    }.items():
        g = torch.Generator().manual_seed(500 + ord(tag))
        o, d, norms = make_rays(N, g)
        t_rand = torch.rand(N, T, generator=g)
        u = torch.rand(N, t, generator=g)
        rr = build_ref_renderer(ren, fld, 4, 40)
        chunk = 32
        if staged:
            rr.eval()
            rec = []
            for head in range(0, N, chunk):
                if perturb:
                    rec.append(t_rand[head:head + chunk])
                rec.append(u[head:head + chunk])
            with RandReplay(rec), torch.no_grad():
                res = rr.render(o[None], d[None], norms[None], staged=True,
                                max_ray_batch=chunk, perturb=perturb,
                                num_steps=T, upsample_steps=t)
        else:
            rr.train()
            rec = ([t_rand] if perturb else []) + [u]
            with RandReplay(rec), torch.no_grad():
                res = rr.render(o[None], d[None], norms[None], staged=False,
                                perturb=perturb, num_steps=T,
                                upsample_steps=t)
        save(f"g5{tag}_run_field.npz", N=N, T=T, t=t, perturb=perturb,
             staged=staged, chunk=chunk, rays_o=o, rays_d=d, norms=norms,
             t_rand=t_rand, u=u, image=res["image"], depth=res["depth"],
             semantics=res["semantics"], grid_seed=77, grid_amp=3.0, **chk)

    # ---- G7: SemanticsMeter -------------------------------------------------
    g = np.random.RandomState(5)
    C = 40
    truths = g.randint(-1, C, size=(3, 24, 32))
    truths[truths == 7] = 3  # class 7 absent from GT
    truths[truths == 21] = -1
    preds = np.where(g.rand(3, 24, 32) < 0.6, np.maximum(truths, 0),
                     g.randint(0, C, size=(3, 24, 32)))
    m = met.SemanticsMeter(C)
    m.update(torch.from_numpy(preds[:2]), torch.from_numpy(truths[:2]))
    m.update(torch.from_numpy(preds[2:]), torch.from_numpy(truths[2:]))
    miou, acc, cacc = m.measure()
    save("g7_meter.npz", preds=preds, truths=truths, conf_mat=m.conf_mat,
         miou=miou, total_acc=acc, class_avg_acc=float(cacc), C=C)

    make_dataset_fixture()
    make_loss_fixture()


# ---------------------------------------------------------------------------
# G6: the NeRF losses of forward_nerf_train (reference
# nr4seg/lightning/joint_train_lightning_net.py:167-223: gather of the ground
# truth at the drawn pixels from the fp16 image / depth, the invalid-semantics
# rule, MSE / NLL(log(p+1e-15)) / L1 on valid depth) and the weighting of
# :503-507.  The module imports PyTorch-Lightning, torchvision, cv2 and the
# CUDA-only field at import time: all stubbed (none is used by this method);
# the method then runs on a stand-in `self` whose get_rays_train / render
# return recorded tensors.
# ---------------------------------------------------------------------------
def load_reference_lightning():
    for name in ("cv2", "torchvision", "torchvision.transforms",
                 "torchvision.transforms.functional", "pytorch_lightning",
                 "nr4seg.nerf.network_tcnn_semantics", "nr4seg.network",
                 "nr4seg.visualizer"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = \
        sys.modules["torchvision.transforms.functional"]
    sys.modules["pytorch_lightning"].LightningModule = torch.nn.Module
    sys.modules["nr4seg.nerf.network_tcnn_semantics"].SemanticNeRFNetwork = object
    sys.modules["nr4seg.network"].DeepLabV3 = object
    sys.modules["nr4seg.visualizer"].Visualizer = object
    if "nr4seg.dataset.ngp_utils" not in sys.modules:
        _load("nr4seg.dataset.ngp_utils", "nr4seg/dataset/ngp_utils.py")
    if "nr4seg.utils.metrics" not in sys.modules:
        _load("nr4seg.utils.metrics", "nr4seg/utils/metrics.py")
    return _load("nr4seg.lightning.joint_train_lightning_net",
                 "nr4seg/lightning/joint_train_lightning_net.py")


def make_loss_fixture():
    import warnings
    warnings.simplefilter("ignore")
    mod = load_reference_lightning()
    cls = mod.JointTrainLightningNet
    g = torch.Generator().manual_seed(31)
    H, W, C, N = 12, 16, 40, 150
    out = {}
    for tag in ("a", "b"):
        img16 = torch.rand(2, 3, H, W, generator=g).half()
        depth16 = (torch.rand(2, H, W, generator=g) * 3).half()
        depth16[:, ::3, ::2] = 0                       # invalid depth pixels
        seg = torch.randint(0, C, (2, H, W), generator=g)
        inds = torch.randint(0, H * W, (1, N), generator=g)   # duplicates allowed
        uom = 0.73
        image = torch.rand(1, N, 3, generator=g).requires_grad_()
        depth = (torch.rand(1, N, generator=g) * 3).requires_grad_()
        sem0 = torch.rand(1, N, C, generator=g) * 0.05
        if tag == "a":
            sem0[0, :9] = 0                            # some invalid rows
        else:
            sem0[:] = 0                                # every row invalid -> None
        sem = sem0.clone().requires_grad_()

        class _Nerf:
            def render(self, *a, **k):
                # the method writes into outputs["semantics"] in place
                return {"image": image * 1.0, "semantics": sem * 1.0,
                        "depth": depth * 1.0}

        me = types.SimpleNamespace(
            get_rays_train=lambda batch, bs: (None, None, None, inds),
            nerf_model=_Nerf(), current_epoch=0,
            criterion_nerf_rgb=torch.nn.MSELoss(reduction="none"),
            criterion_nerf_semantics=torch.nn.NLLLoss(ignore_index=-1, reduction="none"),
            criterion_nerf_depth=torch.nn.L1Loss(reduction="none"))
        # fp16-ROUNDED values held in fp32 tensors: on the GPU the method runs
        # under CUDA autocast, which casts the operands of mse_loss / l1_loss /
        # nll_loss to fp32; that context is inert here (no CUDA), and CPU
        # autograd refuses mixed Half/Float operands, so the cast is made up
        # front -- same values, same fp32 arithmetic
        batch = {"img_fp16": img16.float(), "depth": depth16.float(),
                 "one_m_to_scene_uom": torch.tensor([uom, uom])}
        bs = 1
        lc, ls, ld = cls.forward_nerf_train(me, batch, {"seg_semantics": seg}, bs)
        total = lc
        if ls is not None:
            total = total + ls * 0.04                  # weight_semantics (:45)
        total = total + ld * 0.1                       # weight_depth (:44)
        total.backward()
        out.update({
            f"{tag}_img_fp16": img16.float(), f"{tag}_depth_fp16": depth16.float(),
            f"{tag}_seg": seg, f"{tag}_inds": inds, f"{tag}_uom": uom, f"{tag}_bs": bs,
            f"{tag}_image": image, f"{tag}_depth": depth, f"{tag}_sem": sem0,
            f"{tag}_loss_color": lc, f"{tag}_loss_depth": ld,
            f"{tag}_loss_sem": float("nan") if ls is None else ls,
            f"{tag}_sem_is_none": ls is None, f"{tag}_total": total,
            f"{tag}_g_image": image.grad, f"{tag}_g_depth": depth.grad,
            f"{tag}_g_sem": torch.zeros_like(sem0) if sem.grad is None else sem.grad})
    save("g6_nerf_losses.npz", **out)


# ---------------------------------------------------------------------------
# G8: the per-scene dataset's INDEX logic: 80/20 split,
# path construction, replay selection (random.Random(0).shuffle + per-scene
# quota), old/new flags, Slerp novel viewpoints and the interpolated_data.json
# hand-over -- reference nr4seg/dataset/scannet_ngp_joint.py:113-291, imported
# here with stubs for cv2 (only used by __getitem__'s decoders) and for
# helper.AugmentationList (torchvision).  Only transforms_train.json files are
# needed for this part of the class; they are generated below from a seed and
# stored in the fixture together with what the reference made of them.
# ---------------------------------------------------------------------------
def dataset_layout(root, n_frames):
    """Ten scenes with seeded poses; returns {scene: frames}."""
    import json
    out = {}
    for k in range(10):
        name = f"scene{k:04d}_00"
        rs = np.random.RandomState(100 + k)
        frames = []
        for i in range(n_frames[k]):
            m = np.eye(4)
            q, _ = np.linalg.qr(rs.randn(3, 3))
            m[:3, :3] = q * np.sign(np.linalg.det(q))
            m[:3, 3] = rs.randn(3)
            frames.append({"file_path": f"color/{i * 10}.jpg",
                           "label_path": f"label_40/{i * 10}.png",
                           "transform_matrix": m.tolist()})
        os.makedirs(os.path.join(root, name), exist_ok=True)
        info = {"h": 240, "w": 320, "fl_x": 290.0 + k, "fl_y": 291.0 + k,
                "cx": 160.0, "cy": 120.0, "one_m_to_scene_uom": 0.3 + 0.01 * k,
                "frames": frames}
        with open(os.path.join(root, name, "transforms_train.json"), "w") as f:
            json.dump(info, f)
        out[name] = info
    return out


DATASET_CASES = [
    # (tag, kwargs) -- scene_list is always given in full; the class trims it
    ("train_new_only", dict(scene_list=["scene0000_00", "scene0001_00", "scene0002_00"],
                            mode="train")),
    ("train_fix_nerf", dict(scene_list=["scene0002_00"], mode="train", fix_nerf=True)),
    ("val", dict(scene_list=["scene0002_00"], mode="val", only_new_scene=False)),
    ("train_val", dict(scene_list=["scene0002_00"], mode="train_val", only_new_scene=False)),
    ("predict", dict(scene_list=["scene0000_00", "scene0001_00"], mode="predict")),
    # writes <scene>/e/novel_viewpoints/interpolated_data.json for scenes 0, 1
    ("predict_novel_0", dict(scene_list=["scene0000_00"], mode="predict",
                             use_novel_viewpoints=True)),
    ("predict_novel_1", dict(scene_list=["scene0000_00", "scene0001_00"], mode="predict",
                             use_novel_viewpoints=True)),
    ("joint_replay", dict(scene_list=["scene0000_00", "scene0001_00", "scene0002_00"],
                          mode="train", only_new_scene=False, replay_buffer_size=7)),
    ("joint_replay_novel", dict(scene_list=["scene0000_00", "scene0001_00", "scene0002_00"],
                                mode="train", only_new_scene=False, replay_buffer_size=7,
                                use_novel_viewpoints=True)),
    ("joint_no_replay", dict(scene_list=["scene0000_00", "scene0002_00"], mode="train",
                             only_new_scene=False)),
]


def dataset_state(ds, root):
    """What the index holds, with paths made relative to the root."""
    rel = lambda p: None if p is None else os.path.relpath(p, root)
    return {
        "length": len(ds),
        "image_pths": [rel(p) for p in ds.image_pths],
        "label_pths": [rel(p) for p in ds.label_pths],
        "depth_pths": [rel(p) for p in ds.depth_pths],
        "nerf_image_pths": [rel(p) for p in ds.nerf_image_pths],
        "nerf_label_pths": [rel(p) for p in ds.nerf_label_pths],
        "from_old_scene": [bool(v) for v in ds.from_old_scene],
        "viewpoint_is_novel": [bool(v) for v in ds.viewpoint_is_novel],
        "poses": np.asarray(ds.poses, dtype=np.float64).tolist(),
        "ngp_intrinsics": [float(v) for v in ds.ngp_intrinsics],
        "one_m_to_scene_uom": float(ds.one_m_to_scene_uom),
        "ngp_HW": [int(ds.ngp_H), int(ds.ngp_W)],
    }


def make_dataset_fixture():
    import contextlib
    import io
    import json
    import tempfile
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    helper = types.ModuleType("nr4seg.dataset.helper")
    helper.AugmentationList = lambda *a, **k: None
    sys.modules["nr4seg.dataset.helper"] = helper
    if "nr4seg.dataset.ngp_utils" not in sys.modules:
        _load("nr4seg.dataset.ngp_utils", "nr4seg/dataset/ngp_utils.py")
    mod = _load("nr4seg.dataset.scannet_ngp_joint",
                "nr4seg/dataset/scannet_ngp_joint.py")
    n_frames = [10, 12, 15, 5, 5, 5, 6, 5, 5, 7]
    with tempfile.TemporaryDirectory() as root:
        dataset_layout(root, n_frames)
        cases = {}
        for tag, kw in DATASET_CASES:
            with contextlib.redirect_stdout(io.StringIO()):  # the class prints
                ds = mod.ScanNetNGPJoint(root, exp_name="e", **kw)
            cases[tag] = dataset_state(ds, root)
        gen = {}
        for k in (0, 1):
            pth = os.path.join(root, f"scene{k:04d}_00", "e", "novel_viewpoints",
                               "interpolated_data.json")
            with open(pth) as f:
                fr = json.load(f)["frames"]
            gen[f"scene{k:04d}_00"] = [
                {"nerf_image": os.path.relpath(x["nerf_image"], root),
                 "nerf_label": os.path.relpath(x["nerf_label"], root),
                 "pose": x["pose"]} for x in fr]
    path = os.path.join(HERE, "g8_dataset_index.json")
    with open(path, "w") as f:
        json.dump({"n_frames": n_frames, "cases": cases,
                   "interpolated_data": gen}, f)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} kB)")


if __name__ == "__main__":
    if "--dataset-only" in sys.argv:
        load_reference()
        make_dataset_fixture()
    elif "--losses-only" in sys.argv:
        load_reference()
        make_loss_fixture()
    else:
        main()
