#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "hashgrid or fp16_table or image_ordered or fused_encode or density or tile" 2>&1 | tail -3
timeout 600 python tools/fp16_table_bench.py 2>&1 | grep -E "encode|per view"
