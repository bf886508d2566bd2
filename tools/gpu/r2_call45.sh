#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_bench_final.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "value_f32_mfma_nets", "value_fp16_nets", "value_fp16_nets_fp16_table")})
print("train", d["train"]["ms_per_step"], d["train_f16_nets"]["ms_per_step"], "cpu", d["cpu_baseline"]["value"], d["speedup_vs_cpu"])
PY
