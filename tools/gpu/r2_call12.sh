mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PRE=200 STEPS=40 TAIL_FRAC=0.12
for prec in fp32 fp16; do
  rm -rf /tmp/pt_$prec
  TRAIN_PRECISION=$prec rocprofv3 --kernel-trace -d /tmp/pt_$prec -o t -- python3 tools/profile_train.py > gpurun_out/r2_prof_train_$prec.log 2>&1
  python3 tools/rocpd_summary.py $(find /tmp/pt_$prec -name "*.db" | head -1) > gpurun_out/r2_train_trace_$prec.txt
  echo "== $prec"; grep -v "^W2026\|^E2026\|amdgpu" gpurun_out/r2_prof_train_$prec.log | tail -1 | cut -c1-300
  head -24 gpurun_out/r2_train_trace_$prec.txt
done
