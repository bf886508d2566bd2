#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_configs.py tests/test_gpu_losses_and_module.py -q -m gpu -x 2>&1 | tail -5
bash tools/refresh_profiles.sh r02 > gpurun_out/r2_refresh.log 2>&1
python3 tools/pmc_traffic.py gpurun_out r02 > gpurun_out/r02_pmc_traffic.json 2>gpurun_out/r2_pmc_traffic.err
grep -c "" gpurun_out/r02_bench_kernel_trace.txt; head -c 300 gpurun_out/r02_pmc_traffic.json
TRAIN_PRECISION=fp32 PRE=200 STEPS=40 timeout 300 python tools/profile_train.py 2>&1 | tail -2
