#!/bin/bash
# Runs ON THE GPU BOX: PMC passes (each its own run) of the steady-state NeRF
# training step (tools/profile_train.py: 200 pre-training steps, then 40 steps
# of 4096 rays x (256+256)); counters summed over the last 10 % of the
# dispatches, per step.   usage: train_pmc.sh <tag> [fp32|fp16|bf16x3]
set -u
TAG=${1:-r03}
export TRAIN_PRECISION=${2:-fp32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
mkdir -p $OUT
export PRE=200 STEPS=40 TAIL_FRAC=0.1 MARKER=k_nerf_loss_grad
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/tp$i
  timeout 900 rocprofv3 --kernel-trace --pmc $PMC -d /tmp/tp$i -o p -- python3 tools/profile_train.py > $OUT/${TAG}_train_pmc_${TRAIN_PRECISION}_$i.log 2>&1
  (echo "# pmc: $PMC   TRAIN_PRECISION=$TRAIN_PRECISION"; python3 tools/pmc_window.py $(find /tmp/tp$i -name "*.db" | head -1)) > $OUT/${TAG}_train_pmc_${TRAIN_PRECISION}_$i.txt
  i=$((i+1))
done
cat $OUT/${TAG}_train_pmc_${TRAIN_PRECISION}_*.txt | head -150
