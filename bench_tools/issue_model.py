#!/usr/bin/env python3
"""Instruction-issue model of one loop of a gfx950 kernel: classify every VALU instruction of the loop by the issue cost the
instruction-rate probes measured (bench_tools/microbench.hip -> profiles/r04_microbench_instr_rates.txt) and add them up.

What the probes say (cycles per wave-instruction per SIMD at 2.4 GHz, one wave64 instruction stream per SIMD saturating the VALU):
    v_mad_u64_u32                                   4.7    (also v_fma_f64 4.8)
    any VOP3-encoded / 64-bit / SGPR-operand form   4.1    v_lshrrev_b64, v_lshl_add_u64, v_mov_b64, v_alignbit_b32, v_add3_u32, v_cndmask e64,
                                                           v_addc_co, v_mul_lo/hi_u32, v_add_u32 with an SGPR operand
    VOP2 / VOP1 e32 on VGPRs, inline constants      2.3    v_add_u32, v_sub_u32, v_and_b32, v_xor_b32, v_lshrrev_b32, v_mov_b32
    and 32-bit LITERALS (measured: 2.2)
Only the last class runs at the "full" rate, so a multiplier's true ceiling is NOT its v_mad count alone.

usage: issue_model.py <listing.s> <kernel name substring> [units per loop trip = 1] [whole]
       `whole`: census of the WHOLE kernel body, every instruction counted once (for straight-line kernels such as the unrolled NTT pass:
       units = elements per lane; the few instructions inside short loops are under-counted, stated in the output)
       hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only lurk_beta_amd/csrc/msm_acc.hip -o acc.s   (same flags as the Makefile)
prints the class census of the LARGEST loop of the kernel, cycles per trip, and the bound in trips per second for the whole device.
"""
import collections
import re
import sys

COST = {"mad64": 4.7, "slow": 4.1, "fast": 2.3}
VOP2_FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32",
             "v_mov_b32", "v_min_u32", "v_max_u32", "v_not_b32", "v_add_f32", "v_mul_f32"}
SIMDS, CLOCK_GHZ, LANES = 1024, 2.15, 64  # 256 CUs x 4; the clock the chip sustains under this kernel (bench.py: valu_roofline)


def classify(line):
    op, _, rest = line.partition(" ")
    if not op.startswith("v_"):
        return None
    if op.startswith("v_mad_u64_u32") or op.startswith("v_fma_f64"):
        return "mad64"
    base = op[:-4] if op.endswith(("_e32", "_e64")) else op
    if op.endswith("_e64") or base not in VOP2_FAST:
        return "slow"
    operands = [x.strip() for x in rest.split(",")]
    for x in operands[1:]:  # sources: an SGPR or VCC takes the instruction off the fast path (a literal or an inline constant does not)
        if re.match(r"^(s\d+|s\[|vcc|exec|m0|ttmp)", x):
            return "slow"
    return "fast"


def main(path, needle, units=1, mode="loop"):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if "Begin function" in l and needle in l)
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or "; -- End function" in lines[i])
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i:
                loops.append((labels[t], i))
    if mode == "whole":
        a, b = 0, len(body) - 1
    else:
        # the innermost loop that holds the bulk of the work: the shortest of the loops within 5 % of the longest
        longest = max(b - a for a, b in loops)
        a, b = min((x for x in loops if x[1] - x[0] >= 0.95 * longest), key=lambda x: x[1] - x[0])
    census, ops = collections.Counter(), collections.defaultdict(collections.Counter)
    other = collections.Counter()
    for l in body[a:b + 1]:
        l = l.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        if l.startswith(("ds_", "global_", "scratch_", "buffer_", "flat_")):
            other[l.split()[0]] += 1
        c = classify(l)
        if c:
            census[c] += 1
            ops[c][l.split()[0]] += 1
    cycles = sum(COST[c] * n for c, n in census.items())
    what = "whole body (straight-line count)" if mode == "whole" else f"loop at listing lines {a}-{b}"
    print(f"kernel *{needle}*: {what}, {sum(census.values())} VALU instructions per trip")
    if other:
        print("  memory / LDS instructions (not in the VALU model): " + ", ".join(f"{o} {n}" for o, n in other.most_common(8)))
    for c in ("mad64", "slow", "fast"):
        top = ", ".join(f"{o} {n}" for o, n in ops[c].most_common(8))
        print(f"  {c:6s} {census[c]:5d} x {COST[c]} cycles = {COST[c] * census[c]:8.0f}   ({top})")
    units = float(units)
    bound = SIMDS * CLOCK_GHZ * 1e9 * LANES / cycles * units
    print(f"  issue cycles per wave-trip: {cycles:.0f}  ->  bound {bound / 1e9:.2f} G units/s on {SIMDS} SIMDs at {CLOCK_GHZ} GHz ({units:g} unit(s) per lane-trip)")
    mad_only = SIMDS * CLOCK_GHZ * 1e9 * LANES / (COST['mad64'] * census['mad64']) * units
    print(f"  v_mad-only ceiling (the figure rounds 1-3 quoted): {mad_only / 1e9:.2f} G units/s; the mads are {COST['mad64'] * census['mad64'] / cycles:.1%} of the issue cycles")


def census_of(lines, needle, mode):
    """(VALU instructions, issue cycles) of the largest loop (mode "loop") or the whole body (mode "whole") of every kernel whose
    mangled name contains `needle`, summed over those kernels"""
    tot_n, tot_c = 0, 0.0
    starts = [i for i, l in enumerate(lines) if "Begin function" in l and needle in l]
    for start in starts:
        end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or "; -- End function" in lines[i])
        body = lines[start:end]
        a, b = 0, len(body) - 1
        if mode == "loop":
            labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
            loops = []
            for i, l in enumerate(body):
                m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
                if m:
                    t = m.group(1) or m.group(2)
                    if t in labels and labels[t] < i:
                        loops.append((labels[t], i))
            if loops:
                longest = max(y - x for x, y in loops)
                a, b = min((x for x in loops if x[1] - x[0] >= 0.95 * longest), key=lambda x: x[1] - x[0])
        for l in body[a:b + 1]:
            l = l.split(";")[0].strip()
            if not l or l.endswith(":") or l.startswith("."):
                continue
            c = classify(l)
            if c:
                tot_n += 1
                tot_c += COST[c]
    return tot_n, tot_c


def mix(out_json, *listings):
    """`issue_model.py mix <out.json> <listing.s> ...`: the mean issue cost per VALU instruction of every stage of a commitment
    (bench_workloads/msm.py: KERNEL_GROUPS), from the static instruction mix of the stage's kernels - the accumulation by its loop,
    the short kernels by their whole bodies (their loops are a few dozen instructions: the static mix IS the dynamic one to within
    the trip-count weights).  bench.py multiplies these by the SQ_INSTS_VALU it measures per stage (roofline_valu.pipeline)."""
    import json
    import os

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench_workloads.msm import KERNEL_GROUPS

    texts = [open(p).read().split("\n") for p in listings]
    out = {}
    for stage, needles in KERNEL_GROUPS.items():
        n, c = 0, 0.0
        for lines in texts:
            for needle in needles:
                dn, dc = census_of(lines, needle, "loop" if stage == "accumulate" else "whole")
                n, c = n + dn, c + dc
        if n:
            out[stage] = {"valu_instructions_static": n, "cycles_per_valu": round(c / n, 3)}
    out["_source"] = {"listings": [os.path.basename(p) for p in listings], "rates": COST,
                      "note": "hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only of the listed translation units of lurk_beta_amd/csrc "
                              "(msm_acc, msm_acc_persistent, msm_sort, msm, msm_reduce, msm_finalize)"}
    with open(out_json, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


def poseidon(path, needle, dyn_valu_per_wave_hash=None, measured_mhps=None):
    """`issue_model.py poseidon <listing.s> <kernel substring> [SQ_INSTS_VALU per wave and hash] [measured M hashes/s]`: the kernel's round
    loops nest (rounds x rows), so a static census cannot weight them; the issue cost per VALU instruction comes from the static mix of the
    whole body (the loops' mixes agree to 1 %), the instruction COUNT from the PMC pass (SQ_INSTS_VALU of a poseidon_tree run / waves /
    hashes per lane) - their product is the issue budget of one hash."""
    lines = open(path).read().split("\n")
    n, c = census_of(lines, needle, "whole")
    mads = sum(1 for l in lines[next(i for i, l in enumerate(lines) if "Begin function" in l and needle in l):] if l.strip().startswith("v_mad_u64_u32"))
    cpi = c / n
    print(f"kernel *{needle}*: {n} VALU instructions in the body, mean issue cost {cpi:.3f} cycles per instruction")
    if dyn_valu_per_wave_hash:
        d = float(dyn_valu_per_wave_hash)
        cycles = d * cpi
        bound = SIMDS * CLOCK_GHZ * 1e9 * LANES / cycles
        print(f"  dynamic: {d:.0f} VALU wave-instructions per hash -> {cycles:.0f} issue cycles per wave-hash -> bound {bound / 1e6:.1f} M hashes/s on {SIMDS} SIMDs at {CLOCK_GHZ} GHz")
        if measured_mhps:
            print(f"  measured {float(measured_mhps):.1f} M hashes/s = {float(measured_mhps) * 1e6 / bound:.1%} of the issue bound")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "poseidon":
        poseidon(*sys.argv[2:])
    elif len(sys.argv) > 1 and sys.argv[1] == "mix":
        mix(*sys.argv[2:])
    else:
        main(*sys.argv[1:])
