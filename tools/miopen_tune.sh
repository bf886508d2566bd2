#!/bin/bash
# Runs ON THE GPU BOX: (re)generates ucsa_neural_rendering_amd/miopen_db/ --
# MIOpen's user find-db / perf-db for every convolution configuration of the
# DeepLab legs (bench.py's seg modes at B = 8, cfg3's joint step with both
# backbones), with the exhaustive search scripts/train_joint.py asks for
# (torch.backends.cudnn.benchmark = True).  The files are merged back through
# gpurun_out/miopen_db/; copy them into the package and commit.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# a scratch copy MIOpen may append to (the package's own files are never written)
export MIOPEN_USER_DB_PATH=/tmp/ucsa_miopen_tune_db
mkdir -p gpurun_out/miopen_db "$MIOPEN_USER_DB_PATH"
cp ucsa_neural_rendering_amd/miopen_db/*.txt "$MIOPEN_USER_DB_PATH"/ 2>/dev/null
export B=8 WARM=3 STEPS=5 FIND=1
for M in fp32_cl fp32_nchw bf16_cl; do
  MODE=$M python3 tools/profile_seg.py 2>&1 | grep -v amdgpu.ids | tail -1
done
for BB in resnet101 resnet50; do
  BACKBONE=$BB MODE=fp32_cl B=4 python3 tools/profile_seg.py 2>&1 | grep -v amdgpu.ids | tail -1
done
python3 bench.py --mode cfg3 --steps 3 --warmup 2 2>/dev/null | tail -1 | cut -c1-300
python3 bench.py --mode cfg3 --backbone resnet101 --steps 3 --warmup 2 2>/dev/null | tail -1 | cut -c1-300
python3 bench.py --mode cfg3 --seg-amp bf16 --steps 3 --warmup 2 2>/dev/null | tail -1 | cut -c1-300
cp "$MIOPEN_USER_DB_PATH"/*.txt gpurun_out/miopen_db/
# the GPU tests' (small) DeepLab configurations, so that the suite starts warm too
python3 -m pytest -q -m gpu -x ${DL_TESTS:-tests/test_gpu_cl_deeplab.py tests/test_gpu_deeplab_parity.py tests/test_gpu_fused_bn.py} 2>&1 | tail -2
cp "$MIOPEN_USER_DB_PATH"/*.txt gpurun_out/miopen_db/
wc -l gpurun_out/miopen_db/*.txt
