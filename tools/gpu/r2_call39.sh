#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/encode_per_level.py > gpurun_out/r2_encode_levels.log 2>&1
tail -20 gpurun_out/r2_encode_levels.log
