#!/bin/bash
mkdir -p gpurun_out
timeout 300 tools/ubench/mfma_valu > gpurun_out/r2_mfma_valu.log 2>&1
cat gpurun_out/r2_mfma_valu.log
