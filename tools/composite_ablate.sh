#!/bin/bash
# on the GPU box: time the composite stage with parts of the shading removed
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
echo -n "baseline: "; run
# 1. no colour L3 MFMAs
python3 - <<'PY'
import re
p="ucsa_neural_rendering_amd/csrc/composite.hip"; s=open(p).read()
s=s.replace("for (int cb = 0; cb < CBS; ++cb) o3[cb] = mfma16(wa, hid[cb][ks], o3[cb]);","for (int cb = 0; cb < CBS; ++cb) o3[cb][ks & 3] += wa * hid[cb][ks];")
open(p,"w").write(s)
PY
echo -n "colour L3 on VALU-ish (wrong): "; run
cp /tmp/composite.orig $F
# 2. no softmax exp (cheap exp)
sed -i 's/const float ex = ok ? expf(lg\[cb\]\[rb\]\[r\] - mx) : 0.0f;/const float ex = ok ? __expf(lg[cb][rb][r] - mx) : 0.0f;/; s/rgb\[cb\]\[c\] = 1.0f \/ (1.0f + expf(-o3\[cb\]\[c\]));/rgb[cb][c] = 1.0f \/ (1.0f + __expf(-o3[cb][c]));/' $F
echo -n "fast exp: "; run
cp /tmp/composite.orig $F
# 3. skip the per-ray sequential sums (phase D)
python3 - <<'PY'
p="ucsa_neural_rendering_amd/csrc/composite.hip"; s=open(p).read()
s=s.replace("        if ((uint32_t)e < nb) {\n          const uint32_t ray","        if ((uint32_t)e < nb && e == 0) {\n          const uint32_t ray")
open(p,"w").write(s)
PY
echo -n "phase D only first row (wrong): "; run
cp /tmp/composite.orig $F
# 4. skip semantics net entirely: NRB loops -> keep MFMA count but no softmax/contrib for sem
python3 - <<'PY'
p="ucsa_neural_rendering_amd/csrc/composite.hip"; s=open(p).read()
s=s.replace("          if (cls < C) crow[3 + cls] = wgt * (lg[cb][rb][r] * inv_sum);","          if (cls < C && rb == 0) crow[3 + cls] = wgt * (lg[cb][rb][r] * inv_sum);")
open(p,"w").write(s)
PY
echo -n "contrib writes rb0 only (wrong): "; run
cp /tmp/composite.orig $F
make -C ucsa_neural_rendering_amd/csrc -j8 > /dev/null 2>&1
