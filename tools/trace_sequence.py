#!/usr/bin/env python3
"""The kernels of ONE request in launch order (gap before each, duration) from a rocprofv3 --kernel-trace CSV of
`tools/small_batch.py <Nq>`: the third-last embedding kernel onward.  usage: trace_sequence.py <kernel_trace.csv> [kernels]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 34
idx = [i for i, r in enumerate(rows) if "embed_" in r["Kernel_Name"]]
i0 = idx[-3]
prev_end = None
for r in rows[i0:i0 + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:90]
    print(f"gap {gap:6.1f} us  dur {(e - s) / 1e3:7.1f} us  {name}")
    prev_end = e
