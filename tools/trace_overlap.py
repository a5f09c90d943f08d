#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel mean duration, busy time (union of kernel
intervals), mean concurrency, and per-queue busy fractions over the last 50% of the trace."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Queue_Id", "?")) for r in rows]
ev.sort()
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * 0.5
ev = [e for e in ev if e[0] >= t_lo]
span = ev[-1][1] - ev[0][0]
def union(evs):
    pts = []
    for s, e, *_ in evs:
        pts.append((s, 1)); pts.append((e, -1))
    pts.sort()
    busy = 0; area = 0; cur = 0; last = pts[0][0]
    for t, d in pts:
        if cur > 0: busy += t - last
        area += cur * (t - last); last = t; cur += d
    return busy, area
busy, area = union(ev)
print("span %.3f ms  busy %.1f%%  mean concurrency while busy %.2f  kernels %d" % (span / 1e6, 100 * busy / span, area / max(busy, 1), len(ev)))
byq = collections.defaultdict(list)
for e in ev: byq[e[3]].append(e)
for q, v in sorted(byq.items()):
    b, _ = union(v)
    print("queue %s: %d kernels, busy %.1f%% of span" % (q, len(v), 100 * b / span))
dur = collections.defaultdict(list)
for s, e, n, _ in ev: dur[n].append(e - s)
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-42s n=%5d mean %.1f us  total %.2f ms" % (n, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
