#!/bin/bash
# Runs ON THE GPU BOX: PMC passes (each its own run, kernel trace only beside
# them) of the steady-state DeepLabV3-R101 training step (tools/profile_seg.py,
# B = 8, 240x320); counters summed over the last 40 % of the dispatches, per
# step (marker: k_seg_tail, one launch per step).   usage: seg_pmc.sh <tag> [fp32_cl|bf16_cl]
set -u
TAG=${1:-r03}
export MODE=${2:-fp32_cl}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
mkdir -p $OUT
export B=8 WARM=4 STEPS=12 FIND=0 TAIL_FRAC=0.4 MARKER=k_seg_tail
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/sg$i
  timeout 900 rocprofv3 --kernel-trace --pmc $PMC -d /tmp/sg$i -o p -- python3 tools/profile_seg.py > $OUT/${TAG}_seg_pmc_${MODE}_$i.log 2>&1
  (echo "# pmc: $PMC   MODE=$MODE"; python3 tools/pmc_window.py $(find /tmp/sg$i -name "*.db" | head -1)) > $OUT/${TAG}_seg_pmc_${MODE}_$i.txt
  i=$((i+1))
done
head -12 $OUT/${TAG}_seg_pmc_${MODE}_0.txt
