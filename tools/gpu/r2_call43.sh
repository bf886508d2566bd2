#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_backward.py -q -m gpu -x -s -k "f16 or half" 2>&1 | grep -E "passed|failed|f16-train|Error|assert" | tail -16
TRAIN_PRECISION=fp16 PRE=200 STEPS=40 timeout 300 python tools/profile_train.py 2>&1 | tail -1
