"""Run tests/test_gpu_configs.py::test_cfg2_whole_view_* on fields trained from
different RNG states (the bench field is re-trained by every run; inside the
suite the test inherits whatever RNG state the earlier tests left)."""
import io, sys, os, contextlib, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import test_gpu_configs as t
for seed in [int(x) for x in os.environ.get("SEEDS", "1,2,3,4,5,6").split(",")]:
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            t.test_cfg2_whole_view_f16x2_every_ray_against_the_oracle()
        print(f"seed {seed}: PASS  " + buf.getvalue().strip().splitlines()[-1], flush=True)
    except AssertionError as e:
        lines = buf.getvalue().strip().splitlines()
        print(f"seed {seed}: FAIL  {str(e)[:1500]}", flush=True)
        for l in lines[-12:]:
            print("    " + l[:300])
