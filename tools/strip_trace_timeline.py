#!/usr/bin/env python3
"""tools/strip_trace_timeline.py KERNEL_TRACE_CSV [FIRST [COUNT]] -- the kernels of frames FIRST .. FIRST+COUNT-1 before the last
(by k_temporal launches) of a tools/strip_trace_c.py trace, one line per kernel: start, end (us, relative), queue, name, duration;
then the period over the last 30 frames and per queue the idle gaps."""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 20
count = int(sys.argv[3]) if len(sys.argv) > 3 else 5
def short(n):
    m = re.search(r"(k_[a-z_0-9]+)", n)
    return m.group(1) if m else n[:30]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows)
tp = [e for e in ev if e[2] == "k_temporal"]
print("period over the last 30 frames: %.1f us" % ((tp[-1][0] - tp[-31][0]) / 30 / 1e3))
t0, t1 = tp[-first][0], tp[-first + count][0]
for a, b, k, q in ev:
    if t0 <= a < t1:
        print("%8.1f %8.1f  q%s  %-22s %6.1f" % ((a - t0) / 1e3, (b - t0) / 1e3, q, k, (b - a) / 1e3))
