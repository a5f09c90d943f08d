#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel mean duration, busy time (union of kernel
intervals) and mean concurrency over the last 60% of the trace."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in rows]
ev.sort()
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * 0.4
ev = [e for e in ev if e[0] >= t_lo]
span = ev[-1][1] - ev[0][0]
pts = []
for s, e, _ in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = 0; area = 0; cur = 0; last = pts[0][0]
for t, d in pts:
    if cur > 0: busy += t - last
    area += cur * (t - last); last = t; cur += d
dur = collections.defaultdict(list)
for s, e, n in ev: dur[n].append(e - s)
print("span %.3f ms  busy %.1f%%  mean concurrency while busy %.2f" % (span / 1e6, 100 * busy / span, area / max(busy, 1)))
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-42s n=%5d mean %.1f us  total %.2f ms" % (n, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
