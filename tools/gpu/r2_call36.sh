#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -rf /tmp/c3
timeout 900 rocprofv3 --kernel-trace -d /tmp/c3 -o p -- python3 bench.py --mode cfg3 --backbone resnet50 --steps 6 --warmup 2 --no-seg-find > gpurun_out/r2_cfg3_trace.log 2>&1
TAIL_FRAC=0.5 python3 tools/rocpd_summary.py $(find /tmp/c3 -name "*.db" | head -1) > gpurun_out/r2_cfg3_trace.txt 2>&1
head -40 gpurun_out/r2_cfg3_trace.txt | cut -c1-150
grep "^{" gpurun_out/r2_cfg3_trace.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
