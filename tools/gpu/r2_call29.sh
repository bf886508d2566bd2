#!/bin/bash
mkdir -p gpurun_out
S=$(date +%s)
timeout 1500 python bench.py > gpurun_out/r2_bench29.json 2> gpurun_out/r2_bench29.err
E=$(date +%s)
echo "default bench.py wall: $((E-S)) s"
grep "^\[bench" gpurun_out/r2_bench29.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_bench29.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "value_f32_mfma_nets", "value_fp16_nets")})
print("cpu", d.get("cpu_baseline")); print("seg", d.get("seg"))
PY
