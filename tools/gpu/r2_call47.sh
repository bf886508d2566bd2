#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py --no-cpu-baseline --no-train-bench > gpurun_out/r2_bench47.json 2> gpurun_out/r2_bench47.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_bench47.json").read().strip().splitlines()[-1])
print(d["value"], json.dumps(d["mlp_error_vs_fp64"], indent=1))
PY
tail -3 gpurun_out/r2_bench47.err
