#!/bin/bash
# on the GPU box: time the composite stage with parts of the kernel removed
# (results are then WRONG -- timing only) to see what the time is made of
cd "$GRAFT_REPO_ROOT"
F=ucsa_neural_rendering_amd/csrc/composite.hip
cp $F /tmp/composite.orig
run() {
  make -C ucsa_neural_rendering_amd/csrc -j8 > /dev/null 2>&1 || { echo "build failed"; return; }
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
print(f"composite {st['composite']:.3f} ms")
PY
}
edit() { python3 - "$1" "$2" <<'PY'
import sys
p="ucsa_neural_rendering_amd/csrc/composite.hip"; s=open(p).read()
a,b=sys.argv[1],sys.argv[2]
assert a in s, a
open(p,"w").write(s.replace(a,b))
PY
}
echo -n "baseline: "; run
edit "      } else if (e < T) {
        uint32_t lo = 0, hi = t;  // #fine strictly below ze" "      } else if (true) { rank = e; } else if (e < T) {
        uint32_t lo = 0, hi = t;  // #fine strictly below ze"
echo -n "no rank searches (wrong): "; run
cp /tmp/composite.orig $F
edit "      const f32x4 hv = *reinterpret_cast<const f32x4*>(hp);" "      const f32x4 hv = f32x4{0.1f, 0.2f, 0.3f, (float)row * 1e-9f}; (void)hp;"
echo -n "no h-row gathers (wrong): "; run
cp /tmp/composite.orig $F
edit "    while (cnt - head >= G) {" "    while (false && cnt - head >= G) {"
echo -n "no shading at all, phase A only (wrong): "; run
cp /tmp/composite.orig $F
make -C ucsa_neural_rendering_amd/csrc -j8 > /dev/null 2>&1
