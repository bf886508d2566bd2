#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -q -m gpu -x 2>&1 | tail -4
timeout 600 python tools/encode_per_level.py > gpurun_out/r2_encode_levels.log 2>&1
tail -19 gpurun_out/r2_encode_levels.log
