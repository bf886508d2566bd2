"""profiles/rNN_encoder_binding.json from the text summaries tools/encode_pmc.sh
leaves in gpurun_out/ (one file per PASS and counter set).  Round 5: a pass is
TWO encoder launches (levels 9-15 one level per grid row, levels 0-8 in the
several-levels kernel) -- counters and durations are kept per kernel and summed
per pass (the launches run back to back).
   python tools/encode_binding_json.py <tag> [out.json]"""
import glob
import json
import re
import sys

tag = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else f"profiles/{tag}_encoder_binding.json"
CUS, SIMDS, L2_PEAK_GBS = 256, 1024, 34500.0
KERNELS = {"coarse": ("k_hashgrid_encode_tiled<", "k_hashgrid_encode_tiled_ml<"),
           "fine": ("k_hashgrid_encode_sorted<", "k_hashgrid_encode_sorted_ml<")}
res = {}
for path in sorted(glob.glob(f"gpurun_out/{tag}_enc_pmc*.txt")):
    txt = open(path).read()
    m = re.search(r"# PASS=(\w+)\s+pmc:(.*)", txt)
    if not m:
        continue
    per_kernel = res.setdefault(m.group(1), {})
    for line in txt.splitlines():
        for key in KERNELS[m.group(1)] + ("k_tile_depth_order",):
            if key not in line:
                continue
            d = per_kernel.setdefault(key.rstrip("<"), {})
            if not m.group(2).strip():      # the plain kernel trace: launch times
                t = re.search(r"\s(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
                if t:
                    d["launch_us"] = float(t.group(3))
            else:
                c = re.search(r"\s(\d+)\s+(\w+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s*$", line)
                if c:
                    d[c.group(2)] = float(c.group(5))
            break


def derive(d):
    cyc = d["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs
    acc, miss = d["TCP_TOTAL_CACHE_ACCESSES_sum"], d["TCP_TCC_READ_REQ_sum"]
    r = {
        "launch_us": d["launch_us"],
        "gpu_cycles": cyc,
        "tcp_line_accesses": acc,
        "tcp_accesses_per_clock_per_cu": acc / (cyc * CUS),
        "l1_hit_rate": 1.0 - miss / acc,
        "l1_misses": miss,
        "l2_latency_cycles": d["TCP_TCC_READ_REQ_LATENCY_sum"] / max(miss, 1.0),
        "misses_in_flight_per_tcp": d["TCP_TCC_READ_REQ_LATENCY_sum"] / (cyc * CUS),
        "tcp_pending_stall_frac": d["TCP_PENDING_STALL_CYCLES_sum"] / (cyc * CUS),
        "tcp_tagconflict_stall_frac": d["TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"] / (cyc * CUS),
        "tcp_clock_enabled_frac": d["TCP_GATE_EN1_sum"] / (cyc * CUS),
        "valu_issue_frac": d["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * SIMDS),
        "l2_requests": d["TCC_REQ_sum"],
        "l2_hit_rate": d["TCC_HIT_sum"] / d["TCC_REQ_sum"],
        "l2_request_gbs": d["TCC_REQ_sum"] * 128.0 / (d["launch_us"] * 1e-6) / 1e9,
    }
    r["l2_request_frac_of_34500"] = r["l2_request_gbs"] / L2_PEAK_GBS
    return r


SUMMED = ("launch_us", "GRBM_GUI_ACTIVE", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum",
          "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_PENDING_STALL_CYCLES_sum",
          "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_GATE_EN1_sum", "SQ_ACTIVE_INST_VALU",
          "TCC_REQ_sum", "TCC_HIT_sum")
js = {"source": f"tools/encode_pmc.sh {tag}: PASS={{coarse,fine}} rocprofv3 --kernel-trace [--pmc <set>] -- "
                "python3 tools/encode_only.py, one run per counter set; the encoder alone on the "
                "bench's 61 440-ray chunk (5.9 M samples), fp32 table; per launch; a pass = the two "
                "encoder launches back to back (their counters summed); tools/encode_binding_json.py",
      "peak_note": "the TCP (per-CU vector L1) looks up one 128-B line per clock: "
                   "TCP_TOTAL_CACHE_ACCESSES / (256 CUs x cycles) is its utilisation; L2 -> L1 fills "
                   "are 128-B lines (l2_request_gbs against ~34.5 TB/s)"}
for name, per_kernel in res.items():
    keys = [k.rstrip("<") for k in KERNELS[name]]
    tot = {c: sum(per_kernel[k][c] for k in keys) for c in SUMMED}
    js[name] = derive(tot)
    js[name]["kernels"] = {k: derive(per_kernel[k]) for k in keys}
    sort = per_kernel.get("k_tile_depth_order")
    if sort and "launch_us" in sort:
        js[name]["sort_launch_us"] = sort["launch_us"]
json.dump(js, open(out, "w"), indent=1)
print(json.dumps(js, indent=1))
