#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x --durations=12 2>&1 | tail -40 > gpurun_out/r2_tests30.log
tail -30 gpurun_out/r2_tests30.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
