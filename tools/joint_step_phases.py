"""Phases of the cfg3 joint step (8 frames 320x240, 256+256 samples, 4096-ray NeRF updates, DeepLabV3 BACKBONE=resnet101):
each method of JointTrainLightningNet.training_step_joint wrapped with a device synchronisation and a host clock --
the sum exceeds the asynchronous step (also printed) by the waits it adds."""
import os, sys, time, tempfile, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucsa_neural_rendering_amd.lightning import JointTrainDataModule, JointTrainLightningNet, Trainer
from ucsa_neural_rendering_amd._miopen_db import default_cudnn_benchmark
B = 8
exp = {"general": {"name": "phases", "clean_up_folder_if_exists": True, "checkpoint_load": ""},
       "model": {"pretrained": False, "pretrained_backbone": False, "num_classes": 40,
                 "backbone": os.environ.get("BACKBONE", "resnet101"), "amp": ""},
       "optimizer": {"lr_seg": 1e-5, "lr_nerf": 1e-2, "name": "Adam"}, "trainer": {},
       "data_module": {"batch_size": B}, "scenes": ["scene0000_00"],
       "synthetic": {"n_views": 2 * B, "H": 240, "W": 320},
       "nerf": {"n_rays": 4096, "num_steps": 256, "upsample_steps": 256, "precision": "f16x2"},
       "nerf_seed": 123, "seed": 123}
tmp = tempfile.mkdtemp()
torch.manual_seed(123)
torch.backends.cudnn.benchmark = default_cudnn_benchmark()
model = JointTrainLightningNet(exp, {"results": tmp, "scannet": tmp})
dm = JointTrainDataModule(exp); dm.setup()
tr = Trainer(max_epochs=1, device="cuda:0"); tr._attach(model)
model.train(); model.joint_train = True
batches = [tr._to_device(b) for b in dm.train_dataloader_joint()]
for i in range(3):
    model.training_step(batches[i % len(batches)], 0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10):
    model.training_step(batches[i % len(batches)], 0)
torch.cuda.synchronize(); whole = (time.perf_counter() - t0) / 10 * 1e3
acc = collections.OrderedDict()
def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    def timed(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[label or name] = acc.get(label or name, 0.0) + (time.perf_counter() - t) * 1e3
        return r
    setattr(obj, name, timed)
for nm in ("forward_nerf_test", "forward_seg", "forward_nerf_train", "_nerf_update", "data_aug", "_seg_logits", "manual_backward"):
    wrap(model, nm)
opt_seg = model.optimizers()[0]
wrap(opt_seg, "step", "optimizer_seg.step")
n = 10
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(n):
    model.training_step(batches[i % len(batches)], 0)
torch.cuda.synchronize(); synced = (time.perf_counter() - t0) / n * 1e3
print(f"joint step asynchronous {whole:.1f} ms; with a sync around every phase {synced:.1f} ms; phases (ms per step): "
      + ", ".join(f"{k} {v / n:.1f}" for k, v in acc.items()) + f"; accounted {sum(acc.values()) / n:.1f}")
