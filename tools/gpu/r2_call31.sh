#!/bin/bash
mkdir -p gpurun_out
CHUNKS=65536,32768 timeout 600 python tools/streams_bench.py > gpurun_out/r2_streams.log 2>&1
cat gpurun_out/r2_streams.log | tail -20
