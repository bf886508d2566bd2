"""Run tests/test_gpu_configs.py::test_cfg2_whole_view_* on fields trained from
DIFFERENT fields (outside `-m gpu`, VERDICT r5 item 1a: inside the suite the test
sees ONE field, trained deterministically from seed 123 -- tests/util.bench_field;
here the init / ray-draw seed of bench.build_field varies: SEEDS=1,2,3,...)."""
import io, sys, os, contextlib, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import test_gpu_configs as t
for seed in [int(x) for x in os.environ.get("SEEDS", "1,2,3,4,5,6").split(",")]:
    os.environ["UCSA_TEST_FIELD_SEED"] = str(seed)
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            t.test_cfg2_whole_view_f16x2_every_ray_against_the_oracle()
        print(f"seed {seed}: PASS  " + buf.getvalue().strip().splitlines()[-1], flush=True)
    except AssertionError as e:
        import traceback
        lines = buf.getvalue().strip().splitlines()
        tb = traceback.extract_tb(e.__traceback__)[-1]
        print(f"seed {seed}: FAIL at {tb.filename.split('/')[-1]}:{tb.lineno} `{tb.line}`  {str(e)[:600]}", flush=True)
        for l in lines[-6:]:
            print("    " + l[:400])
