#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/x3_bench.py > gpurun_out/r2_x3_bench.log 2>&1
tail -12 gpurun_out/r2_x3_bench.log
