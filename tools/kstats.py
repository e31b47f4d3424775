#!/usr/bin/env python3
"""Short table of a rocprofv3 *_kernel_stats.csv: python tools/kstats.py <csv> [n]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:n]:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    print(f"{name[:70]:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:10.1f} us  min {float(r['MinNs']) / 1e3:9.1f}  max {float(r['MaxNs']) / 1e3:9.1f}  {float(r['Percentage']):5.1f} %")
