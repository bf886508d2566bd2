mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_backward.py -q -s --tb=short -k "half_precision or f16 or binned" > gpurun_out/r2_tests21.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_tests21.log
grep -E "half records|f16-train|passed|failed|^E " gpurun_out/r2_tests21.log | cut -c1-200
export PRE=200 STEPS=40 TAIL_FRAC=0.12
rm -rf /tmp/pt_fp16
TRAIN_PRECISION=fp16 rocprofv3 --kernel-trace -d /tmp/pt_fp16 -o t -- python3 tools/profile_train.py > gpurun_out/r2_prof_train_fp16.log 2>&1
python3 tools/rocpd_summary.py $(find /tmp/pt_fp16 -name "*.db" | head -1) > gpurun_out/r2_train_trace_fp16.txt
grep -v "^W2026\|^E2026\|amdgpu" gpurun_out/r2_prof_train_fp16.log | tail -1 | cut -c150-300
head -14 gpurun_out/r2_train_trace_fp16.txt | cut -c1-130
