#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/x3_bench.py > gpurun_out/r2_x3_bench.log 2>&1
tail -8 gpurun_out/r2_x3_bench.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -5
