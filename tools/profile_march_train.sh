# on the GPU box: kernel trace of marched training, steady state (last 20 %)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export STEPS=${STEPS:-800}
rm -rf /tmp/pm
rocprofv3 --kernel-trace -d /tmp/pm -o m -- python3 tools/profile_march_train.py > gpurun_out/prof_march_train.log 2>&1
grep -v "^W2026\|^E2026" gpurun_out/prof_march_train.log | tail -2
export TAIL_FRAC=0.2
python3 tools/rocpd_summary.py $(find /tmp/pm -name "*.db" | head -1) > gpurun_out/march_train_trace.txt
head -34 gpurun_out/march_train_trace.txt
