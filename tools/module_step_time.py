"""Wall time of JointTrainLightningNet NeRF-only training steps (B=1 batches),
with / without the HIP-graph replay of the frozen segmentation forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucsa_neural_rendering_amd.lightning import JointTrainDataModule, JointTrainLightningNet, Trainer
exp = {
    "general": {"name": "bench_joint", "clean_up_folder_if_exists": True, "checkpoint_load": ""},
    "model": {"pretrained": False, "pretrained_backbone": False, "num_classes": 40, "amp": ""},
    "optimizer": {"lr_seg": 1e-5, "lr_nerf": 1e-2, "name": "Adam"},
    "trainer": {}, "data_module": {"batch_size": 4},
    "scenes": ["scene0000_00"], "synthetic": {"n_views": 12, "H": 240, "W": 320},
    "nerf": {"n_rays": 4096, "num_steps": 256, "upsample_steps": 256,
             "cuda_ray": bool(int(os.environ.get("CUDA_RAY", "0")))}, "nerf_seed": 1,
}
model = JointTrainLightningNet(exp, {"results": "/tmp/exp", "scannet": "/tmp"})
dm = JointTrainDataModule(exp); dm.setup()
tr = Trainer(max_epochs=1)
tr._attach(model)
model.train(); model.joint_train = False
nb = [tr._to_device(b) for b in dm.train_dataloader_nerf()]
for k in range(int(os.environ.get("WARM", "200"))):
    model.training_step(nb[k % len(nb)], 0)
for graph in (True, False, True, False):
    model.seg_graph = graph
    for k in range(10):
        model.training_step(nb[k % len(nb)], 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(60):
        model.training_step(nb[k % len(nb)], 0)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"seg_graph={graph}: host {th/60*1e3:.2f} ms, wall {(time.perf_counter()-t0)/60*1e3:.2f} ms per NeRF-only step")
