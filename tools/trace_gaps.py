#!/usr/bin/env python3
"""Idle time between kernels from a rocprofv3 --kernel-trace CSV: for every kernel, the gap between the latest end of
anything launched before it and its own start (gaps above 200 us are host-side pauses and are left out), grouped by kernel
name.  usage: trace_gaps.py <kernel_trace.csv> [rows]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")) for r in rows)
gap = defaultdict(lambda: [0, 0.0, 0.0])
busy = idle = 0.0
last_end = ev[0][0]
for s, e, name in ev:
    g = s - last_end
    if 0 < g <= 200_000:
        idle += g
        gap[name][0] += 1
        gap[name][1] += g
    if e > last_end:
        busy += e - max(s, last_end)
        last_end = e
    gap[name][2] += e - s
print(f"kernels {len(ev)}  busy (union) {busy / 1e6:.3f} ms  idle in gaps <= 200 us {idle / 1e6:.3f} ms  ({100 * idle / (busy + idle):.1f} % of busy + idle)")
for name, (c, t, d) in sorted(gap.items(), key=lambda kv: -kv[1][1])[:top]:
    if c:
        print(f"{name[:64]:64s} gaps {c:6d}  avg gap {t / c / 1e3:7.2f} us  total {t / 1e6:8.3f} ms   (kernel time {d / 1e6:8.3f} ms)")
