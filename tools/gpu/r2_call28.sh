#!/bin/bash
# final round-2 profiles: bench kernel trace + PMC passes, shade-kernel PMC, MFMA/VALU overlap microbenchmark
bash tools/refresh_profiles.sh r02 > gpurun_out/r2_refresh.log 2>&1
python3 tools/pmc_traffic.py gpurun_out r02 > gpurun_out/r02_pmc_traffic.json 2>gpurun_out/r2_pmc_traffic.err
bash tools/gpu/r2_call26.sh > gpurun_out/r2_call26.log 2>&1
tools/ubench/mfma_valu > gpurun_out/r2_mfma_valu.log 2>&1
tail -3 gpurun_out/r2_refresh.log | cut -c1-400; head -c 600 gpurun_out/r02_pmc_traffic.json; ls gpurun_out | grep r02_
