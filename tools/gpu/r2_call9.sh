mkdir -p gpurun_out
python -m pytest tests -q -m gpu --tb=short > gpurun_out/r2_gpu_suite.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_gpu_suite.log
tail -15 gpurun_out/r2_gpu_suite.log
