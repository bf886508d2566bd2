"""Static instruction histogram of one kernel of a device assembly file
(`hipcc --cuda-device-only -S`), weighted by the measured gfx950 issue costs
(tools/ubench/valu_rates.hip -> profiles/r03_valu_rates.txt; cycles per wave
instruction per SIMD at 4 waves / SIMD, 2.4 GHz nominal).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude \
          --cuda-device-only -S ucsa_neural_rendering_amd/csrc/composite_split.hip -o /tmp/cs.s
    python tools/isa_cost.py /tmp/cs.s 'k_shade16ILi3ELi1ELi1ELi16ELb1'

The count is STATIC (every instruction of the kernel once, loop bodies as
unrolled by the compiler): use it to compare two builds of one kernel, not as
a time."""
import collections
import re
import sys

FULL = 2.45   # v_add/sub/mul/fma_f32, v_and/or/xor, v_mov
HALF = 4.25   # conversions, shifts, perm, integer min/max/add, cndmask, med3, fma_mix, packed
QUART = 8.25  # transcendentals
COST = {}
for n in ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32",
          "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_max_f32", "v_min_f32"):
    COST[n] = FULL
for n in ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32", "v_mul_lo_u32",
          "v_mul_hi_u32"):
    COST[n] = QUART
MFMA16 = 17.0


def kernel_body(path, key):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\w*:", l) and key in l:
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            return lines[start:i]
    raise SystemExit(f"no kernel matching {key!r}")


def main():
    body = kernel_body(sys.argv[1], sys.argv[2])
    hist = collections.Counter()
    for l in body:
        m = re.match(r"^\s+((?:v|s|ds|global|buffer|flat)_\w+)", l)
        if m:
            op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", m.group(1))
            hist[op] += 1
    valu = sum(c for o, c in hist.items() if o.startswith("v_") and "mfma" not in o)
    mfma = sum(c for o, c in hist.items() if "mfma" in o)
    cyc = sum(c * COST.get(o, HALF) for o, c in hist.items()
              if o.startswith("v_") and "mfma" not in o)
    print(f"kernel {sys.argv[2]}: {sum(hist.values())} instructions; VALU {valu} "
          f"(~{cyc:.0f} issue cycles), MFMA {mfma} (~{mfma * MFMA16:.0f} cycles), "
          f"SALU {sum(c for o, c in hist.items() if o.startswith('s_'))}, "
          f"LDS {sum(c for o, c in hist.items() if o.startswith('ds_'))}, "
          f"VMEM {sum(c for o, c in hist.items() if o.startswith(('global', 'buffer', 'flat')))}")
    for o, c in hist.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 45):
        print(f"  {c:6d}  {o}")
    for l in body:
        if "vgpr_count" in l or "NumVgprs" in l or "Occupancy" in l or "ScratchSize" in l or "NumAgprs" in l:
            print(l.strip())


if __name__ == "__main__":
    main()
