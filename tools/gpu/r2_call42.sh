#!/bin/bash
# final validation of round 2: full GPU suite, smoke, default bench, profile refresh
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_bench_final.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "value_f32_mfma_nets", "value_fp16_nets", "value_fp16_nets_fp16_table")})
print("stage", {k: round(v, 3) for k, v in d["stage_ms_per_chunk"].items()})
print("f16 stage", {k: round(v, 3) for k, v in d["f16_mlp_option"]["stage_ms_per_chunk"].items()})
print("train", d["train"]["ms_per_step"], d["train_f16_nets"]["ms_per_step"], "cpu", d["cpu_baseline"]["value"], d["speedup_vs_cpu"], d["quality"])
PY
bash tools/refresh_profiles.sh r02 > gpurun_out/r2_refresh.log 2>&1
python3 tools/pmc_traffic.py gpurun_out r02 > gpurun_out/r02_pmc_traffic.json 2>gpurun_out/r2_pmc_traffic.err
TRAIN_PRECISION=fp32 PRE=200 STEPS=40 TAIL_FRAC=0.12 timeout 300 python tools/profile_train.py 2>&1 | tail -1
