mkdir -p gpurun_out
python -m pytest tests -q -m gpu --tb=short --durations=12 > gpurun_out/r2_gpu_suite2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_gpu_suite2.log
tail -25 gpurun_out/r2_gpu_suite2.log
