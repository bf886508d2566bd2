#!/bin/bash
# on the GPU box: rebuild composite.hip with different (waves, column blocks)
# and time the composite stage of the bench chunk
cd "$GRAFT_REPO_ROOT"
for v in "12 2" "8 2" "8 3" "8 4" "10 2" "16 1"; do
  set -- $v
  sed -i "s/^#define CMP_MAX_WAVES .*/#define CMP_MAX_WAVES $1/; s/^#define CMP_CBS .*/#define CMP_CBS $2    \/\/ column blocks/" ucsa_neural_rendering_amd/csrc/composite.hip
  make -C ucsa_neural_rendering_amd/csrc -j8 > /dev/null 2>&1 || { echo "build failed $v"; continue; }
  echo -n "waves=$1 cbs=$2: "
  timeout 200 python - <<'PY' 2>&1 | tail -1
import torch, bench
from ucsa_neural_rendering_amd import ops
dev = torch.device("cuda:0")
net, ds = bench.build_field(dev, train_steps=200)
W, H = 640, 480
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
pose = _slerp_loop_poses(4, seed=999)[1:2].to(dev)
o, d, n = ops.get_rays(pose, (0.89 * W, 0.89 * W, W / 2, H / 2), H, W)
N = 61440
u = torch.rand(N, 96, device=dev)
st, rho = bench.stage_times(net, o[0, :N].contiguous(), d[0, :N].contiguous(), n[0, :N, 0].contiguous(), u, image_width=W)
print(f"composite {st['composite']:.3f} ms, rho {rho:.2f}")
PY
done
