#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
i=1
for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/q$i
  timeout 600 rocprofv3 --kernel-trace --pmc $PMC -d /tmp/q$i -o p -- python3 tools/x3_bench.py > $OUT/r2_x3_pmc$i.log 2>&1
  python3 tools/rocpd_summary.py $(find /tmp/q$i -name "*.db" | head -1) 2>&1 | grep -E "k_shade_dense|k_composite<|k_weights_compact|^#" > $OUT/r2_x3_pmc$i.txt
  i=$((i+1))
done
grep -E "k_shade_dense<3, 1, (1|2), 16>|k_shade_dense<3, 2, 2, 8>" $OUT/r2_x3_pmc1.txt $OUT/r2_x3_pmc2.txt | awk '{print $2,$3,$4,$5,$6,$7, $(NF-3), $NF}'
