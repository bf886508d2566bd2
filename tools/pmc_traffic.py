"""Distil per-launch HBM traffic / MFMA busy of the bench-size launches of the
two dominant kernels from the PMC summaries written by
tools/refresh_profiles.sh.   python tools/pmc_traffic.py gpurun_out r02 >
profiles/r02_pmc_traffic.json"""
import json
import re
import sys


def counters(path):
    out = {}
    on = False
    for line in open(path):
        if line.startswith("# counters:"):
            on = True
            continue
        if not on or not line.strip():
            continue
        m = re.match(r"\s+(.*?)\s+(\d+)\s+(\S+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s*$", line)
        if m:
            name, grid, cn, _, nd, per = m.groups()
            out.setdefault(name.strip(), {}).setdefault(int(grid), {})[cn] = float(per)
    return out


# launches of a kernel that the SHIPPED render path makes, where the bench also issues
# larger ones of the same kernel for its encoder-only stage times (round 6: the
# per-level encoder runs levels 12-15 = 4 grid rows of 61 440 x 96 / 4 threads; the
# unfused stage-time launches run 7)
PREFER_GRID = {"void k_hashgrid_encode_sorted<": 4 * 61440 * 96 // 4}


def pick(tab, prefix):
    """counters of the largest launch (or the PREFER_GRID one) of the first kernel
    whose name starts with `prefix`."""
    best = None
    for name, grids in tab.items():
        if name.startswith(prefix):
            want = PREFER_GRID.get(prefix)
            g = want if want in grids else max(grids)
            if best is None or g > best[0]:
                best = (g, grids[g])
    return best


def main(d, tag):
    fetch = counters(f"{d}/{tag}_pmc1.txt")
    wr = counters(f"{d}/{tag}_pmc2.txt")
    sq = counters(f"{d}/{tag}_pmc3.txt")
    try:
        l2 = counters(f"{d}/{tag}_pmc4.txt")
    except OSError:
        l2 = {}
    res = {
        "pretrain_steps": 200,
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                  "TCC_HIT_sum TCC_MISS_sum / --pmc SQ_* (three separate passes, "
                  "tools/refresh_profiles.sh) of `python3 bench.py --steps 2 "
                  "--warmup 1 --no-cpu-baseline --no-train-bench` (the default "
                  "200 pretrain steps = the field bench.py times); per-dispatch averages of the largest launches of "
                  "each kernel = the bench's chunks (grid_threads / 96 samples / 16 levels "
                  "= rays per launch for the encoder)",
        "units": "bytes per launch; FETCH_SIZE/WRITE_SIZE are reported in KiB by "
                 "rocprofv3 and multiplied by 1024 here.  gfx950 note "
                 "(MI355X_MICROARCH.md HBM section): FETCH_SIZE under-reports "
                 "wide coalesced streaming reads by 2x; these kernels read by "
                 "8/16-byte gathers, for which the counter is uncalibrated, so "
                 "the raw value is given and 2x raw is the upper bound",
    }
    for key, prefix in (("k_composite", "void k_composite<3, 2, false, false>"),
                        ("k_hashgrid_encode_tiled", "void k_hashgrid_encode_tiled<HIP_vector_t"),
                        # round 5: the encoder is four kernels (coarse pass: _tiled levels
                        # 9-15 + _tiled_ml levels 0-8; fine pass: _sorted + _sorted_ml on the
                        # depth-ordered samples) and the per-tile sort
                        ("k_hashgrid_encode_tiled_ml", "void k_hashgrid_encode_tiled_ml<"),
                        ("k_hashgrid_encode_sorted", "void k_hashgrid_encode_sorted<"),
                        ("k_hashgrid_encode_sorted_ml", "void k_hashgrid_encode_sorted_ml<"),
                        ("k_tile_depth_order2", "k_tile_depth_order2("),
                        # round 6: levels 0-7 inside the sigma MLP, both passes depth-ordered
                        ("k_density_sorted", "void k_density_sorted<3,"),
                        ("k_weights_compact", "k_weights_compact"),
                        ("k_shade16_f16", "void k_shade16<3, 1, 1,"),
                        ("k_shade16_x3", "void k_shade16<3, 1, 2,"),
                        ("k_shade16_h2", "void k_shade16<3, 1, 3,"),
                        # (two column blocks per group since round 4's last step)
                        ("k_shade16_h2", "void k_shade16<3, 2, 3,")):
        f, w, s = pick(fetch, prefix), pick(wr, prefix), pick(sq, prefix)
        if not (f and w):
            continue
        e = {"grid_threads": f[0],
             "fetch_bytes": int(f[1].get("FETCH_SIZE", 0) * 1024),
             "write_bytes": int(w[1].get("WRITE_SIZE", 0) * 1024)}
        hit, miss = w[1].get("TCC_HIT_sum", 0), w[1].get("TCC_MISS_sum", 0)
        if hit + miss:
            e["tcc_hit_rate"] = hit / (hit + miss)
        if s and s[1].get("SQ_INSTS_VALU"):
            e["valu_wave_instructions"] = int(s[1]["SQ_INSTS_VALU"])
        if s and s[1].get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles, the MFMA counter
            # the busy cycles of all 1024 SIMDs: busy / (cycles * 1024)
            e["mfma_busy_frac"] = (s[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) /
                                   (s[1]["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0))
        if s and s[1].get("GRBM_GUI_ACTIVE") and s[1].get("SQ_ACTIVE_INST_VALU"):
            # quad-cycles in which a SIMD issues a VALU-class instruction (MFMA
            # issue slots included) / SIMD cycles
            e["valu_issue_frac"] = (s[1]["SQ_ACTIVE_INST_VALU"] * 4.0 /
                                    (s[1]["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0))
        q = pick(l2, prefix)
        if q:
            # requests arriving at the L2 (all XCDs); one request = one 128-B
            # line on gfx950's TCP->TCC path for these 8/16-byte gathers
            e["l2_requests"] = int(q[1].get("TCC_REQ_sum", 0))
            e["l2_read_requests"] = int(q[1].get("TCC_READ_sum", 0))
            e["l2_request_bytes"] = int(e["l2_requests"] * 128)
        res[key] = e
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
