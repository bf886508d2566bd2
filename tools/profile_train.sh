# on the GPU box: kernel trace of the live (run()) training step, steady state
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PRE=200 STEPS=40
rocprofv3 --kernel-trace -d /tmp/pt -o t -- python3 tools/profile_train.py > gpurun_out/prof_train.log 2>&1
grep -v "^W2026\|^E2026" gpurun_out/prof_train.log | tail -2
export TAIL_FRAC=${TAIL_FRAC:-0.12}
python3 tools/rocpd_summary.py $(find /tmp/pt -name "*.db" | head -1) > gpurun_out/train_trace.txt
head -40 gpurun_out/train_trace.txt
