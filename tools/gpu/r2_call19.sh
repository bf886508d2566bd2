mkdir -p gpurun_out
export UCSA_BENCH_BACKEND=gloo UCSA_BENCH_WATCHDOG=200
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2954$((RANDOM % 10)) bench.py --gpus 2 --steps 2 --warmup 1 --pretrain-steps 30 > gpurun_out/r2_dbg_render2.json 2> gpurun_out/r2_dbg_render2.err
echo "rc $?"; grep -E "File \"|Thread|Timeout|Error|error" gpurun_out/r2_dbg_render2.err | head -60; tail -c 300 gpurun_out/r2_dbg_render2.json
