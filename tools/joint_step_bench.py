"""Time the reference's joint training step (training_step_joint,
joint_train_lightning_net.py:363-471) at its native sizes on the synthetic
scene: batch 4 new-scene frames of 320x240, 4096 rays x (256+256) samples per
NeRF step, full no-grad render of every frame, DeepLabV3-R101 fwd/bwd."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucsa_neural_rendering_amd.lightning import JointTrainDataModule, JointTrainLightningNet, Trainer
exp = {
    "general": {"name": "bench_joint", "clean_up_folder_if_exists": True, "checkpoint_load": ""},
    "model": {"pretrained": False, "pretrained_backbone": False, "num_classes": 40,
              "amp": os.environ.get("AMP", ""),
              "channels_last": bool(int(os.environ.get("CL", "0")))},
    "optimizer": {"lr_seg": 1e-5, "lr_nerf": 1e-2, "name": "Adam"},
    "trainer": {}, "data_module": {"batch_size": int(os.environ.get("BS", "4"))},
    "scenes": ["scene0000_00"], "synthetic": {"n_views": 12, "H": 240, "W": 320},
    "nerf": {"n_rays": 4096, "num_steps": 256, "upsample_steps": 256,
             "cuda_ray": bool(int(os.environ.get("CUDA_RAY", "0")))}, "nerf_seed": 1,
}
model = JointTrainLightningNet(exp, {"results": "/tmp/exp", "scannet": "/tmp"})
dm = JointTrainDataModule(exp); dm.setup()
tr = Trainer(max_epochs=1)
tr._attach(model)
model.train(); model.joint_train = True
loader = dm.train_dataloader_joint()
batches = [tr._to_device(b) for b in loader]
def step(b):
    model.training_step(b, 0)
for _ in range(int(os.environ.get("WARM", "1"))):
    for b in batches:
        step(b)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
for b in batches:
    step(b); n += 1
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
B = exp["data_module"]["batch_size"]
rays = B * (240 * 320 + 4096)
print(f"joint step (B={B}): {dt*1e3:.1f} ms  -> {rays/dt/1e6:.2f} M NeRF rays/s (render+train) + {B/dt:.1f} seg img/s; losses {model.logged}")
model.joint_train = False
nb = [tr._to_device(b) for b in dm.train_dataloader_nerf()]
model.training_step(nb[0], 0); torch.cuda.synchronize()
t0 = time.perf_counter()
for b in nb[:8]:
    model.training_step(b, 0)
torch.cuda.synchronize()
print(f"nerf-only step (cfg1-style, 4096 rays x 512): {(time.perf_counter()-t0)/8*1e3:.1f} ms")
