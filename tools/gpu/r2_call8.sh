mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_cl_deeplab.py -q --tb=short -x > gpurun_out/r2_tests8.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2_tests8.log
export UCSA_BENCH_BACKEND=gloo
for dt in fp32 fp16 bf16; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$((RANDOM % 10)) bench.py --gpus 2 --steps 300 --warmup 2 --mode train --fresh --grad-comm-dtype $dt > gpurun_out/r2_train_payload_$dt.json 2> gpurun_out/r2_train_payload_$dt.err
done
unset UCSA_BENCH_BACKEND
bash tools/refresh_profiles.sh r02 > gpurun_out/r2_refresh.log 2>&1
python tools/pmc_traffic.py gpurun_out r02 > gpurun_out/r02_pmc_traffic.json 2> gpurun_out/r02_pmc_traffic.err
tail -3 gpurun_out/r2_tests8.log
python - <<'PY'
import json
for dt in ("fp32", "fp16", "bf16"):
    try:
        r = json.loads(open(f"gpurun_out/r2_train_payload_{dt}.json").read().strip().splitlines()[-1])["train_dp"]
        print(dt, {k: r.get(k) for k in ("eval_psnr_db", "eval_label_acc", "final_loss", "replicas_identical", "comm_bytes_per_step_per_rank", "ms_per_step")})
    except Exception as e:
        print(dt, "failed", repr(e)); print(open(f"gpurun_out/r2_train_payload_{dt}.err").read()[-800:])
PY
cat gpurun_out/r02_pmc_traffic.json | head -40
