#!/usr/bin/env python3
"""Per-kernel sums of every counter of a rocprofv3 --pmc pass.  usage: pmc_raw.py <counter_collection.csv> [name filter]"""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][:60]
    if flt not in k: continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k, c in agg.items():
    print(k, "launches", len(n[k]))
    for name, v in sorted(c.items()): print(f"   {name:32s} {v / len(n[k]):14.4g} per launch")
