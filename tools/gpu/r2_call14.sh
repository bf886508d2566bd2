mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r2_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r2_smoke.log
grep -v amdgpu gpurun_out/r2_smoke.log | tail -8
timeout 600 python bench.py --mode cfg4 --views 16 --warmup 1 > gpurun_out/r2_cfg4_1gpu.json 2> gpurun_out/r2_cfg4_1gpu.err; tail -c 600 gpurun_out/r2_cfg4_1gpu.json
