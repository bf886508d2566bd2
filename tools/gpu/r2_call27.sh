#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x -s 2>&1 | grep -v "^$" > gpurun_out/r2_tests27.log
grep -E "passed|failed|error|bf16x3|scale|sigma MLP" gpurun_out/r2_tests27.log | tail -30
timeout 900 python bench.py --no-cpu-baseline --no-train-bench > gpurun_out/r2_bench27.json 2> gpurun_out/r2_bench27.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_bench27.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "value_f32_mfma_nets", "value_fp16_nets")})
print("stage", d["stage_ms_per_chunk"])
print("f32", d["f32_mfma_option"]["stage_ms_per_chunk"], d["f32_mfma_option"]["max_abs_image_diff_vs_value_mode"])
print("f16", d["f16_mlp_option"]["stage_ms_per_chunk"], d["f16_mlp_option"]["max_abs_image_diff_vs_value_mode"])
print(d["roofline_composite"]); print(d["quality"])
PY
tail -3 gpurun_out/r2_bench27.err
