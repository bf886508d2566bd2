mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_deeplab_parity.py -q --tb=short -k "split or fused_encode or section" > gpurun_out/r2_tests5.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_tests5.log
timeout 900 python tools/composite_split_bench.py > gpurun_out/r2_split_bench2.log 2>&1
for prec in fp32 fp16; do
  rm -rf /tmp/pc_$prec
  timeout 600 rocprofv3 --kernel-trace -d /tmp/pc_$prec -o p -- python3 tools/profile_composite_split.py $prec > gpurun_out/r2_prof_split_$prec.log 2>&1
  BY_GRID=1 python3 tools/rocpd_summary.py $(find /tmp/pc_$prec -name "*.db" | head -1) 2>/dev/null | grep -E "k_composite|k_weights|k_shade|k_encode_sigma|k_hashgrid_encode_tiled|k_sigma_mlp|^#" | head -20 > gpurun_out/r2_prof_split_$prec.txt
done
for fe in 1 0; do
  UCSA_FUSED_ENCODE=$fe timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-bench > gpurun_out/r2_bench_fe$fe.json 2> gpurun_out/r2_bench_fe$fe.err
done
tail -3 gpurun_out/r2_tests5.log; grep -v amdgpu gpurun_out/r2_split_bench2.log; cat gpurun_out/r2_prof_split_fp32.txt gpurun_out/r2_prof_split_fp16.txt
python - <<'PY'
import json
for fe in (1, 0):
    try:
        r = json.loads(open(f"gpurun_out/r2_bench_fe{fe}.json").read().strip().splitlines()[-1])
        print("fused_encode", fe, "value", r["value"], "ms", r["ms_per_step"], "f16", r["f16_mlp_option"]["rays_per_s"], r["stage_ms_per_chunk"])
    except Exception as e:
        print("fe", fe, "failed", e)
PY
