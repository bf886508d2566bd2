#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fp16" 2>&1 | tail -4
timeout 600 python tools/fp16_table_bench.py > gpurun_out/r2_fp16_table.log 2>&1
tail -8 gpurun_out/r2_fp16_table.log
