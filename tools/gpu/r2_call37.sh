#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "class_count or bf16x3 or split" 2>&1 | tail -8
