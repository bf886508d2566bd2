mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -q --tb=short -k "split or fused_encode" > gpurun_out/r2_tests6.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_tests6.log
UCSA_ES_S=8 python -m pytest tests/test_gpu_parity.py -q --tb=short -k "fused_encode" >> gpurun_out/r2_tests6.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_tests6.log
timeout 900 python tools/composite_split_bench.py > gpurun_out/r2_split_bench3.log 2>&1
for es in 4 8; do
  rm -rf /tmp/pe_$es
  UCSA_ES_S=$es timeout 600 rocprofv3 --kernel-trace -d /tmp/pe_$es -o p -- python3 tools/profile_composite_split.py fp32 > gpurun_out/r2_prof_es$es.log 2>&1
  BY_GRID=1 python3 tools/rocpd_summary.py $(find /tmp/pe_$es -name "*.db" | head -1) 2>/dev/null | grep -E "k_encode_sigma|k_hashgrid_encode_tiled|\] k_sigma_mlp\(|^#" | head -20 > gpurun_out/r2_prof_es$es.txt
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-bench > gpurun_out/r2_bench_c.json 2> gpurun_out/r2_bench_c.err
tail -2 gpurun_out/r2_tests6.log; grep -v amdgpu gpurun_out/r2_split_bench3.log; cat gpurun_out/r2_prof_es4.txt gpurun_out/r2_prof_es8.txt
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r2_bench_c.json").read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "f16", r["f16_mlp_option"], r["stage_ms_per_chunk"])
PY
