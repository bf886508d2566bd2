"""Merge the three per-step PMC summaries written by tools/train_pmc.sh /
tools/seg_pmc.sh (<prefix>_0.txt FETCH_SIZE, _1.txt WRITE_SIZE + TCC hits,
_2.txt SQ_* + GRBM_GUI_ACTIVE) into the dict bench.py reads.

    python tools/pmc_step_json.py gpurun_out/r03_seg_pmc_fp32_cl
"""
import json
import sys


def load(path):
    txt = open(path).read()
    return json.loads(txt[txt.index("{"):])


def merge(prefix):
    f, w, s = (load(f"{prefix}_{i}.txt") for i in range(3))
    fs, ws, ss = f["per_step"], w["per_step"], s["per_step"]
    # FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3
    fetch, write = fs["FETCH_SIZE"] * 1024.0, ws["WRITE_SIZE"] * 1024.0
    hit, miss = ws.get("TCC_HIT_sum", 0.0), ws.get("TCC_MISS_sum", 0.0)
    cyc = ss["GRBM_GUI_ACTIVE"] / 8.0       # summed over the 8 XCDs
    out = {
        "fetch_bytes_per_step": fetch, "write_bytes_per_step": write,
        "hbm_bytes_per_step": fetch + write,
        "tcc_hit_rate": hit / (hit + miss) if hit + miss else None,
        "mfma_busy_cycles_per_step": ss["SQ_VALU_MFMA_BUSY_CYCLES"],
        "valu_active_quad_cycles_per_step": ss["SQ_ACTIVE_INST_VALU"],
        "gpu_cycles_per_step": cyc,
        # busy cycles of all 1024 SIMDs / (cycles x 1024)
        "mfma_busy_frac": ss["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0),
        "valu_issue_frac": ss["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0),
        "steps_in_window": [f["steps_in_window"], w["steps_in_window"], s["steps_in_window"]],
        "fetch_by_kernel_bytes": {k: v["FETCH_SIZE"] * 1024.0
                                  for k, v in f["per_step_by_kernel"].items()},
        "write_by_kernel_bytes": {k: v["WRITE_SIZE"] * 1024.0
                                  for k, v in w["per_step_by_kernel"].items() if "WRITE_SIZE" in v},
        "note": "raw counters (gfx950: FETCH_SIZE under-reports wide streaming reads by up "
                "to 2x, MI355X_MICROARCH.md HBM section); separate --pmc passes",
    }
    return out


if __name__ == "__main__":
    print(json.dumps(merge(sys.argv[1]), indent=1))
