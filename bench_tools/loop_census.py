#!/usr/bin/env python3
"""Static census of the instructions inside the loops of one kernel of a `hipcc -S --cuda-device-only` listing: which opcodes the
accumulate loop is made of.  usage: loop_census.py <listing.s> <kernel name substring> [min instructions]"""
import collections
import re
import sys


def main(path, needle, min_len=200):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if "Begin function" in l and needle in l)
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or "; -- End function" in lines[i])
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i:
                loops.append((labels[t], i, t))
    for a, b, t in loops:
        ops = collections.Counter()
        for l in body[a:b + 1]:
            l = l.split(";")[0].strip()
            if not l or l.endswith(":") or l.startswith("."):
                continue
            ops[l.split()[0]] += 1
        n = sum(ops.values())
        if n < int(min_len):
            continue
        valu = sum(c for o, c in ops.items() if o.startswith("v_"))
        print(f"loop {t}: lines {a}-{b}, {n} instructions, {valu} VALU, "
              f"{sum(c for o, c in ops.items() if o.startswith('s_'))} SALU/ctl, "
              f"{sum(c for o, c in ops.items() if o.startswith(('global_', 'scratch_', 'buffer_', 'ds_', 'flat_')))} memory")
        for o, c in ops.most_common(40):
            print(f"   {c:6d}  {o}")


if __name__ == "__main__":
    main(*sys.argv[1:])
