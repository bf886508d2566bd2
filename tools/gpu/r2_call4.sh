mkdir -p gpurun_out
python -m pytest tests/test_gpu_configs.py tests/test_gpu_deeplab_parity.py tests/test_gpu_parity.py tests/test_gpu_losses_and_module.py -q -s --tb=short -k "cfg or section or split or fixture or render_matches or chunking or image_ordered or fp16_inference or nerf_loss_kernel" > gpurun_out/r2_tests4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_tests4.log
timeout 900 python tools/composite_split_bench.py > gpurun_out/r2_split_bench.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-bench > gpurun_out/r2_bench_b.json 2> gpurun_out/r2_bench_b.err
export UCSA_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --mode train > gpurun_out/r2_bench_train2.json 2> gpurun_out/r2_bench_train2.err
tail -3 gpurun_out/r2_tests4.log; cat gpurun_out/r2_split_bench.log | grep -v amdgpu
