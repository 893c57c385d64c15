#!/usr/bin/env python3
"""Text timeline of a rocprofv3 --kernel-trace CSV: per kernel start/end (us, relative), queue, and for every kernel how much of
its duration another queue's kernel was running.  python bench_tools/timeline.py <kernel_trace.csv> [first_us last_us]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lurk::", "")[:34],
                     r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0 = rows[0][0]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
for s, e, n, q, st in rows:
    a, b = (s - t0) / 1e3, (e - t0) / 1e3
    if b < lo or a > hi:
        continue
    ov = 0
    for s2, e2, n2, q2, _ in rows:
        if q2 != q and s2 < e and e2 > s:
            ov += min(e, e2) - max(s, s2)
    print(f"{a:10.1f} {b:10.1f} {b - a:8.1f} us  q{q:>3} s{st:>3}  ovl {100.0 * ov / max(e - s, 1):5.0f}%  {n}")
