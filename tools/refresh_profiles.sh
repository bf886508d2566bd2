#!/bin/bash
# Runs ON THE GPU BOX (gpurun): rocprofv3 kernel trace of the default bench
# command and three separate PMC passes (never combined with other trace
# domains), summarised into gpurun_out/<tag>_*.txt.   usage: refresh_profiles.sh r01
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
mkdir -p $OUT
BENCH="bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace -d /tmp/p0 -o p -- python3 $BENCH > $OUT/${TAG}_bench.log 2>&1
BY_GRID=1 python3 tools/rocpd_summary.py /tmp/p0/*/p_results.db > $OUT/${TAG}_bench_kernel_trace.txt 2>/dev/null || \
BY_GRID=1 python3 tools/rocpd_summary.py $(find /tmp/p0 -name "*.db" | head -1) > $OUT/${TAG}_bench_kernel_trace.txt
# PMC passes on the SAME parameter state as the default bench run (200 pretrain
# steps): the w > 1e-4 mask fraction, and with it the composite kernel's work
# and traffic, depends on the field
SMALL="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-bench"
i=1
for PMC in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "TCC_REQ_sum TCC_READ_sum"; do
  rm -rf /tmp/p$i
  timeout 900 rocprofv3 --kernel-trace --pmc $PMC -d /tmp/p$i -o p -- python3 $SMALL > $OUT/${TAG}_pmc$i.log 2>&1
  python3 tools/rocpd_summary.py $(find /tmp/p$i -name "*.db" | head -1) > $OUT/${TAG}_pmc$i.txt
  i=$((i+1))
done
tail -1 $OUT/${TAG}_bench.log | head -c 600
