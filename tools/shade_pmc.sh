#!/bin/bash
# Runs ON THE GPU BOX: kernel trace + SQ PMC passes (each its own run) of the
# inference composite pair alone.   usage: shade_pmc.sh <tag>
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
mkdir -p $OUT
i=0
for PMC in "" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
  "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/sp$i
  if [ -z "$PMC" ]; then
    timeout 600 rocprofv3 --kernel-trace -d /tmp/sp$i -o p -- python3 tools/shade_only.py > $OUT/${TAG}_shade_pmc$i.log 2>&1
  else
    timeout 600 rocprofv3 --kernel-trace --pmc $PMC -d /tmp/sp$i -o p -- python3 tools/shade_only.py > $OUT/${TAG}_shade_pmc$i.log 2>&1
  fi
  (echo "# pmc: $PMC"; python3 tools/rocpd_summary.py $(find /tmp/sp$i -name "*.db" | head -1) 2>/dev/null | grep -E "^#|k_shade16|k_weights_compact") > $OUT/${TAG}_shade_pmc$i.txt
  i=$((i+1))
done
cat $OUT/${TAG}_shade_pmc*.txt
