mkdir -p gpurun_out
set -u
python -m pytest tests/test_gpu_configs.py tests/test_gpu_deeplab_parity.py -q -s --tb=short > gpurun_out/r2_newtests2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_newtests2.log
timeout 600 python tests/scripts/aspp_probe.py --whole-only > gpurun_out/r2_aspp_fixed.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r2_bench_a.json 2> gpurun_out/r2_bench_a.err
export UCSA_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-train-bench > gpurun_out/r2_bench_n2.json 2> gpurun_out/r2_bench_n2.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --mode train > gpurun_out/r2_bench_train2.json 2> gpurun_out/r2_bench_train2.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --warmup 1 --mode cfg4 --views 8 --gather > gpurun_out/r2_bench_cfg4.json 2> gpurun_out/r2_bench_cfg4.err
unset UCSA_BENCH_BACKEND
timeout 300 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/r2_bench_train1.json 2> gpurun_out/r2_bench_train1.err
rocprofv3 --list-avail 2>/dev/null | grep -E "TCP_TCC|TCC_REQ|TCP_TA|TCC_EA0_RDREQ|TCC_READ" | head -40 > gpurun_out/r2_counters.txt
tail -3 gpurun_out/r2_newtests2.log
