#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/r2_bench41.json 2> gpurun_out/r2_bench41.err
grep "^\[bench" gpurun_out/r2_bench41.err | tail -3
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_bench41.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "value_f32_mfma_nets", "value_fp16_nets", "value_fp16_nets_fp16_table")})
print("stage", {k: round(v, 3) for k, v in d["stage_ms_per_chunk"].items()})
print("roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), d["roofline"].get("hbm_utilisation"))
print("cmp", round(d["roofline_composite"]["frac"], 4), d["roofline_composite"].get("mfma_pipe_busy_frac"), d["roofline_composite"].get("valu_issue_frac"))
print("train", d["train"]["ms_per_step"], d["train_f16_nets"]["ms_per_step"], "cpu", d["cpu_baseline"]["value"], d["speedup_vs_cpu"])
PY
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
