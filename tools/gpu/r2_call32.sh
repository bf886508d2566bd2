#!/bin/bash
mkdir -p gpurun_out
for cfg in "resnet50:" "resnet101:" "resnet50:--seg-amp bf16 --nerf-precision fp16" "resnet50:--nerf-precision fp32"; do
  bb=${cfg%%:*}; extra=${cfg#*:}
  tag=$(echo "$bb $extra" | tr -c 'a-z0-9\n' '_')
  timeout 900 python bench.py --mode cfg3 --backbone $bb $extra --steps 5 --warmup 2 > gpurun_out/r2_cfg3_$tag.json 2> gpurun_out/r2_cfg3_$tag.err
  python - "$tag" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r2_cfg3_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["value"], {k: d["config"].get(k) for k in ("seg_precision", "nerf_render_nets", "backbone")})
PY
done
