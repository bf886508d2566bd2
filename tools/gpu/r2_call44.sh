#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for P in fp16 fp32; do
rm -rf /tmp/pt_$P
TRAIN_PRECISION=$P PRE=200 STEPS=40 timeout 600 rocprofv3 --kernel-trace -d /tmp/pt_$P -o t -- python3 tools/profile_train.py > gpurun_out/r2_train_$P.log 2>&1
TAIL_FRAC=0.12 python3 tools/rocpd_summary.py $(find /tmp/pt_$P -name "*.db" | head -1) > gpurun_out/r2_train_trace_$P.txt 2>&1
tail -1 gpurun_out/r2_train_$P.log | cut -c1-300
head -16 gpurun_out/r2_train_trace_$P.txt | cut -c1-140
done
