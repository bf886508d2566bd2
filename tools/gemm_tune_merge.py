"""Merge the per-mode TunableOp result files of tools/gemm_tune.sh
(gpurun_out/tunable/*.csv) into the table the package ships."""
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
val, rows = [], []
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "tunable", "*.csv"))):
    for ln in open(f):
        ln = ln.rstrip("\n")
        if not ln:
            continue
        if ln.startswith("Validator"):
            if ln not in val:
                val.append(ln)
        elif ln not in rows:
            rows.append(ln)
out = os.path.join(ROOT, "ucsa_neural_rendering_amd", "gemm_tuning", "tunableop_gfx950.csv")
open(out, "w").write("\n".join(val + rows) + "\n")
print(f"{out}: {len(rows)} GEMM shapes")
