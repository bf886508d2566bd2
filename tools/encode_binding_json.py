"""profiles/rNN_encoder_binding.json from the text summaries tools/encode_pmc.sh
leaves in gpurun_out/ (one file per PASS and counter set).
   python tools/encode_binding_json.py <tag> [out.json]"""
import glob
import json
import re
import sys

tag = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else f"profiles/{tag}_encoder_binding.json"
CUS, SIMDS, L2_PEAK_GBS = 256, 1024, 34500.0
res = {}
for path in sorted(glob.glob(f"gpurun_out/{tag}_enc_pmc*.txt")):
    txt = open(path).read()
    m = re.search(r"# PASS=(\w+)\s+pmc:(.*)", txt)
    if not m:
        continue
    d = res.setdefault(m.group(1), {})
    if not m.group(2).strip():  # the plain kernel trace: the launch time
        t = re.search(r"k_hashgrid_encode_tiled<.*?\s(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", txt)
        d["launch_us"] = float(t.group(3))
        continue
    for c in re.finditer(r"k_hashgrid_encode_tiled\S*\s+\d+\s+(\w+)\s+(\d+)\s+(\d+)\s+([\d.]+)", txt):
        d[c.group(1)] = float(c.group(4))
js = {"source": f"tools/encode_pmc.sh {tag}: PASS={{coarse,fine}} rocprofv3 --kernel-trace [--pmc <set>] -- "
                "python3 tools/encode_only.py, one run per counter set; k_hashgrid_encode_tiled alone on the "
                "bench's 61 440-ray chunk (5.9 M samples), fp32 table, plain 8-load gather; per launch; "
                "tools/encode_binding_json.py",
      "peak_note": "the TCP (per-CU vector L1) looks up one 128-B line per clock: "
                   "TCP_TOTAL_CACHE_ACCESSES / (256 CUs x cycles) is its utilisation"}
for name, d in res.items():
    cyc = d["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs
    acc, miss = d["TCP_TOTAL_CACHE_ACCESSES_sum"], d["TCP_TCC_READ_REQ_sum"]
    js[name] = {
        "launch_us": d["launch_us"],
        "gpu_cycles": cyc,
        "tcp_line_accesses": acc,
        "tcp_accesses_per_clock_per_cu": acc / (cyc * CUS),
        "l1_hit_rate": 1.0 - miss / acc,
        "l1_misses": miss,
        "l2_latency_cycles": d["TCP_TCC_READ_REQ_LATENCY_sum"] / miss,
        "misses_in_flight_per_tcp": d["TCP_TCC_READ_REQ_LATENCY_sum"] / (cyc * CUS),
        "tcp_pending_stall_frac": d["TCP_PENDING_STALL_CYCLES_sum"] / (cyc * CUS),
        "tcp_tagconflict_stall_frac": d["TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"] / (cyc * CUS),
        "tcp_clock_enabled_frac": d["TCP_GATE_EN1_sum"] / (cyc * CUS),
        "valu_issue_frac": d["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * SIMDS),
        "l2_requests": d["TCC_REQ_sum"],
        "l2_hit_rate": d["TCC_HIT_sum"] / d["TCC_REQ_sum"],
        "l2_request_gbs": d["TCC_REQ_sum"] * 128.0 / (d["launch_us"] * 1e-6) / 1e9,
    }
    js[name]["l2_request_frac_of_34500"] = js[name]["l2_request_gbs"] / L2_PEAK_GBS
json.dump(js, open(out, "w"), indent=1)
print(json.dumps(js, indent=1))
