#!/usr/bin/env python3
"""tools/strip_trace_report.py KERNEL_TRACE_CSV -- timeline summary of tools/strip_trace.py: the last 30 frames (by k_send_image_to_pbo
launches, one per frame; the spatial pass runs up to three times per frame: interior rows and the two border bands); per kernel mean
duration and per queue the busy fraction of the span."""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    m = re.search(r"(k_[a-z_0-9]+)", n)
    return m.group(1) if m else n[:30]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows]
ev.sort()
sp = [e for e in ev if e[2] == "k_send_image_to_pbo"]
t0, t1 = sp[-31][1], sp[-1][1]
sel = [e for e in ev if e[0] >= t0 and e[1] <= t1]
span = (t1 - t0) / 30
print("frame period %.1f us (last 30 frames)" % (span / 1e3))
dur = defaultdict(list); busy = defaultdict(int)
for a, b, k, q in sel:
    dur[k].append(b - a); busy[q] += b - a
for k in sorted(dur):
    print("  %-24s n/frame %.1f  mean %7.1f us" % (k, len(dur[k]) / 30, sum(dur[k]) / len(dur[k]) / 1e3))
for q in sorted(busy):
    print("  queue %s busy %.1f us per frame (%.0f %%)" % (q, busy[q] / 30 / 1e3, 100 * busy[q] / (t1 - t0)))
# union busy time of the GPU (any kernel running)
iv = sorted((a, b) for a, b, _, _ in sel)
u, ce = 0, iv[0][0]
for a, b in iv:
    if b > ce:
        u += b - max(a, ce); ce = b
print("  some kernel running %.0f %% of the span" % (100 * u / (t1 - t0)))
