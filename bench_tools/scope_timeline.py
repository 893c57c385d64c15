"""Print the device-side scope timeline (LURK_PROF_TIMELINE file) of the last whole step: times in microseconds relative to its cross term."""
import sys

L = [x.split() for x in open(sys.argv[1]) if not x.startswith("#")]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
idx = [i for i, x in enumerate(L) if x[4].startswith("r1cs_cross")]
streams0 = {}
for x in L:
    streams0.setdefault(x[3], len(streams0))
# the roofline leg's cross terms run on the default stream at the very end: skip streams that only ever carry cross terms
t0 = float(L[idx[-back - 3]][0])
t1 = float(L[idx[-back - 2]][0])
for x in sorted(L, key=lambda x: float(x[0])):
    t = float(x[0]) - t0
    if -900 < t < (t1 - t0) + 100:
        print("%8.0f %8.0f %7.0f  s%-2d %s" % (t, float(x[1]) - t0, float(x[2]), streams0[x[3]], x[4]))
