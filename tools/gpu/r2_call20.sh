mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bench_modes.py tests/test_gpu_deeplab_parity.py -q --tb=short --durations=8 -k "bench or channels_last-resnet50" > gpurun_out/r2_tests20.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_tests20.log
tail -16 gpurun_out/r2_tests20.log
