#!/bin/bash
# Runs ON THE GPU BOX: kernel trace + PMC passes (each its own run) of the
# encoder alone (tools/encode_only.py: round 5's four encoder kernels + the sort).   usage: encode_pmc.sh <tag>
# (the TA_* counter set of round 3 is gone: in round 4 it hung rocprofv3 until
# the timeout on this pool)
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
mkdir -p $OUT
i=0
for PASS in coarse fine; do
export PASS
for PMC in "" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
  "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" \
  "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum"; do
  rm -rf /tmp/ep$i
  if [ -z "$PMC" ]; then
    timeout 600 rocprofv3 --kernel-trace -d /tmp/ep$i -o p -- python3 tools/encode_only.py > $OUT/${TAG}_enc_pmc$i.log 2>&1
  else
    timeout 600 rocprofv3 --kernel-trace --pmc $PMC -d /tmp/ep$i -o p -- python3 tools/encode_only.py > $OUT/${TAG}_enc_pmc$i.log 2>&1
  fi
  (echo "# PASS=$PASS  pmc: $PMC"; python3 tools/rocpd_summary.py $(find /tmp/ep$i -name "*.db" | head -1) 2>/dev/null | grep -E "^#|hashgrid_encode_tiled|hashgrid_encode_sorted|tile_depth_order") > $OUT/${TAG}_enc_pmc$i.txt
  i=$((i+1))
done
done
cat $OUT/${TAG}_enc_pmc*.txt
