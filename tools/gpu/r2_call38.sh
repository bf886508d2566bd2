#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x -k "fp16 or cfg2 or split" 2>&1 | tail -4
timeout 600 python tools/fp16_table_bench.py > gpurun_out/r2_fp16_table.log 2>&1
head -4 gpurun_out/r2_fp16_table.log | tail -3
