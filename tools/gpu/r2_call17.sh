mkdir -p gpurun_out
python -m pytest tests/test_gpu_raymarch.py tests/test_gpu_bench_modes.py tests/test_gpu_cl_deeplab.py tests/test_gpu_deeplab_parity.py -q --tb=short --durations=8 -k "quality_gate or bench or cl_deeplab or deeplab" > gpurun_out/r2_tests17.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_tests17.log
tail -22 gpurun_out/r2_tests17.log
