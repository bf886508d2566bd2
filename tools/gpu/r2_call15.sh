mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_backward.py tests/test_gpu_raymarch.py -q --tb=short -k "binned or gradient or train or f16" > gpurun_out/r2_tests15.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_tests15.log
tail -3 gpurun_out/r2_tests15.log
export PRE=200 STEPS=40 TAIL_FRAC=0.12
for prec in fp32 fp16; do
  rm -rf /tmp/pt_$prec
  TRAIN_PRECISION=$prec rocprofv3 --kernel-trace -d /tmp/pt_$prec -o t -- python3 tools/profile_train.py > gpurun_out/r2_prof_train_$prec.log 2>&1
  python3 tools/rocpd_summary.py $(find /tmp/pt_$prec -name "*.db" | head -1) > gpurun_out/r2_train_trace_$prec.txt
  echo "== $prec"; grep -v "^W2026\|^E2026\|amdgpu" gpurun_out/r2_prof_train_$prec.log | tail -1 | cut -c150-300
  head -14 gpurun_out/r2_train_trace_$prec.txt | cut -c1-130
done
