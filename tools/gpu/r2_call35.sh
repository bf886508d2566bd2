#!/bin/bash
mkdir -p gpurun_out
CHUNKS=65536,77824,104448,155648 timeout 600 python tools/streams_bench.py > gpurun_out/r2_streams.log 2>&1
grep "streams 1" gpurun_out/r2_streams.log
