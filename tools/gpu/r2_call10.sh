mkdir -p gpurun_out
for cfg in "--backbone resnet50" "--backbone resnet101" "--backbone resnet50 --seg-amp bf16 --nerf-precision fp16"; do
  timeout 900 python bench.py --mode cfg3 --steps 5 --warmup 2 $cfg >> gpurun_out/r2_bench_cfg3.json 2>> gpurun_out/r2_bench_cfg3.err
done
SEG_BENCHMARK=0 timeout 900 python tools/seg_bench.py > gpurun_out/r2_seg_bench0.log 2>&1
SEG_BENCHMARK=1 timeout 1500 python tools/seg_bench.py > gpurun_out/r2_seg_bench1.log 2>&1
python - <<'PY'
import json
for line in open("gpurun_out/r2_bench_cfg3.json"):
    line = line.strip()
    if line.startswith("{"):
        r = json.loads(line)
        print(r["config"]["backbone"], r["config"]["seg_precision"], r["config"]["nerf_render_nets"], "ms/step", round(r["ms_per_step"], 1), "rays/s", round(r["value"]), "img/s", round(r["config"]["seg_images_per_s"], 1))
PY
tail -3 gpurun_out/r2_bench_cfg3.err | cut -c1-300
grep -v amdgpu gpurun_out/r2_seg_bench0.log; grep -v amdgpu gpurun_out/r2_seg_bench1.log
