mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_backward.py tests/test_gpu_parity.py -q -s --tb=short -k "f16 or misaligned or gradients_match or split or fp16" > gpurun_out/r2_tests11.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_tests11.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_bench_d.json 2> gpurun_out/r2_bench_d.err
grep -E "f16-train|passed|failed|FAILED|^E " gpurun_out/r2_tests11.log | cut -c1-250 | head -40
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r2_bench_d.json").read().strip().splitlines()[-1])
print("value", r["value"], "fp16", r.get("value_fp16_nets"))
print("train", r["train"]["ms_per_step"], "train_f16", r["train_f16_nets"]["ms_per_step"])
print("seg", r["seg"])
PY
tail -3 gpurun_out/r2_bench_d.err | cut -c1-300
