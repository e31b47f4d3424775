#!/usr/bin/env python3
"""ONE timed step of bench.py as a timeline, from a rocprofv3 --kernel-trace [--memory-copy-trace] run:

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras
    python tools/step_timeline.py DIR

The step is cut between two launches of the clip encoder's seg GEMM (one per step).  Every kernel / copy with its offset, its
duration and the idle gap in front of it; then the totals: kernel time, copies, idle."""
import csv
import glob
import re
import sys

d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", ""))))
ev.sort()
anchors = [i for i, e in enumerate(ev) if e[2].startswith("vfr::gemm_nt_mfma<true, 2>")]
if len(anchors) < 3:
    sys.exit("fewer than three steps in the trace")
a0, a1 = anchors[-2], anchors[-1]
# the step starts at the first kernel after the previous step's last scorer kernel: walk back from the anchor over the clip encoder's prologue
step = ev[a0:a1]
t0 = step[0][0]
last_end = t0
busy = idle = copies = 0.0
agg = {}
print(f"{'offset us':>10s} {'dur us':>9s} {'gap us':>8s}  kernel")
for s, e, n in step:
    g = max(0, s - last_end)
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {g / 1e3:8.1f}  {n[:90]}")
    idle += g
    if e > last_end:
        busy += e - max(s, last_end)
        last_end = e
    if n.startswith("COPY"):
        copies += e - s
    k = n[:60]
    agg[k] = (agg.get(k, (0, 0.0))[0] + 1, agg.get(k, (0, 0.0))[1] + (e - s))
print(f"step {(step[-1][1] - t0) / 1e6:.3f} ms (anchor to anchor {(ev[a1][0] - t0) / 1e6:.3f}): busy {busy / 1e6:.3f} ms, idle {idle / 1e6:.3f} ms, copies {copies / 1e6:.3f} ms, {len(step)} events")
print("by name:")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {k:60s} x{c:4d} {t / 1e6:8.3f} ms")
