"""Print the device-side scope timeline (LURK_PROF_TIMELINE file) of the last whole step: times in microseconds relative to its cross term."""
import sys

L = [x.split() for x in open(sys.argv[1]) if not x.startswith("#")]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
idx = [i for i, x in enumerate(L) if x[4].startswith("r1cs_cross")]
# the bench's stand-alone legs (the cross term alone on the device, the uniform-column shape) launch cross terms on another stream at
# the very end: a step's cross terms are the ones on the stream of the FIRST cross term of the file
idx = [i for i in idx if L[i][3] == L[idx[0]][3]]
streams0 = {}
for x in L:
    streams0.setdefault(x[3], len(streams0))
t0 = float(L[idx[-back]][0])
t1 = float(L[idx[-back + 1]][0])
for x in sorted(L, key=lambda x: float(x[0])):
    t = float(x[0]) - t0
    if -900 < t < (t1 - t0) + 100:
        print("%8.0f %8.0f %7.0f  s%-2d %s" % (t, float(x[1]) - t0, float(x[2]), streams0[x[3]], x[4]))
