mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -q --tb=short -k "split or fused_encode" > gpurun_out/r2_tests7.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_tests7.log
for fe in 0 2 1; do
  UCSA_FUSED_ENCODE=$fe timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-bench > gpurun_out/r2_bench_fe$fe.json 2> gpurun_out/r2_bench_fe$fe.err
done
tail -2 gpurun_out/r2_tests7.log
python - <<'PY'
import json
for fe in (0, 2, 1):
    try:
        r = json.loads(open(f"gpurun_out/r2_bench_fe{fe}.json").read().strip().splitlines()[-1])
        f16 = r["f16_mlp_option"]
        print("fused_encode", fe, "fp32 value", r["value"], "ms", r["ms_per_step"], "| f16", f16["rays_per_s"], f16["ms_per_view"], f16["stage_ms_per_chunk"])
    except Exception as e:
        print("fe", fe, "failed", repr(e)); print(open(f"gpurun_out/r2_bench_fe{fe}.err").read()[-1500:])
PY
