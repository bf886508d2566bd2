#!/bin/bash
# Runs ON THE GPU BOX: (re)generates ucsa_neural_rendering_amd/gemm_tuning/
# tunableop_gfx950.csv -- PyTorch TunableOp over rocBLAS / hipBLASLt for the
# GEMMs of DeepLabV3-R101's 1x1 convolutions at the benchmark's batch (8 x
# 240x320), fp32 and bf16 (about 2 minutes).  Merged back through
# gpurun_out/tunable/; `python tools/gemm_tune_merge.py` writes the package file.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tunable
export B=8 WARM=3 STEPS=5 FIND=0 PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_VERBOSE=0
for M in bf16_cl fp32_cl; do
  PYTORCH_TUNABLEOP_FILENAME=$GRAFT_REPO_ROOT/gpurun_out/tunable/tunableop_$M.csv MODE=$M \
    python3 tools/profile_seg.py 2>&1 | grep -v amdgpu.ids | tail -1
done
