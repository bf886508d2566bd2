mkdir -p gpurun_out
export UCSA_BENCH_BACKEND=gloo UCSA_BENCH_WATCHDOG=150
for extra in "" "--replicated-adam"; do
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2953$((RANDOM % 10)) bench.py --gpus 2 --mode train --steps 3 --warmup 1 --pretrain-steps 30 $extra > gpurun_out/r2_dbg_train.json 2> gpurun_out/r2_dbg_train.err
  echo "== extra='$extra' rc $?"; grep -E "File \"|line [0-9]+ in|Thread|Timeout" gpurun_out/r2_dbg_train.err | head -60; tail -c 400 gpurun_out/r2_dbg_train.json
done
