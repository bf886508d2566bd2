#!/bin/bash
# Runs ON THE GPU BOX: PMC passes (separate, kernel-trace only) of the marcher
# render on a marcher-trained field.   usage: refresh_march_pmc.sh r01
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
# counters serialise the dispatches: keep the training short (PRE) -- 600 steps took 5 min per pass
export MARCH_TRAIN=1 PRE=250 ITERS=6
i=1
# (FETCH_SIZE and WRITE_SIZE in ONE pass aborted rocprofv3 on this pool: keep them apart)
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/q$i
  timeout 900 rocprofv3 --kernel-trace --pmc $PMC -d /tmp/q$i -o p -- python3 tools/profile_march.py > $OUT/${TAG}_march_pmc$i.log 2>&1
  TAIL_FRAC=0.05 python3 tools/rocpd_summary.py $(find /tmp/q$i -name "*.db" | head -1) | grep -E "^#|k_composite<3, 2, false, true>|k_hashgrid_encode<false>|k_seg_|k_sigma_mlp\(" > $OUT/${TAG}_march_pmc$i.txt
  i=$((i+1))
done
grep points $OUT/${TAG}_march_pmc1.log
