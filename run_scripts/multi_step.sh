#!/bin/bash
# reference run_scripts/multi_step.sh: the continual loop over the ten scenes.
# One process per GPU: `torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/cl_deeplab.py "$@"`
python scripts/cl_deeplab.py "$@"
