mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_backward.py tests/test_gpu_raymarch.py tests/test_gpu_configs.py -q --tb=short -k "gradient or f16 or train or cfg3 or march_train or backward" > gpurun_out/r2_tests13.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_tests13.log
tail -4 gpurun_out/r2_tests13.log
bash tools/gpu/r2_call12.sh 2>&1 | grep -E "^==|ms_per_step|k_shade_bwd|k_composite<|GPU busy" | cut -c1-200
