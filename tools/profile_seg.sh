cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export STEPS=40 WARM=3
rocprofv3 --kernel-trace -d /tmp/ps -o s -- python3 tools/profile_seg.py > gpurun_out/prof_seg.log 2>&1
grep -v "^W2026\|^E2026" gpurun_out/prof_seg.log | tail -2
export TAIL_FRAC=0.07
python3 tools/rocpd_summary.py $(find /tmp/ps -name "*.db" | head -1) > gpurun_out/seg_trace.txt
head -40 gpurun_out/seg_trace.txt
