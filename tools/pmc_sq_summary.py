#!/usr/bin/env python3
"""Per-kernel summary of the SQ counter pass of tools/run_round_profiles.sh: matrix-pipe busy, VALU instructions, wave-cycle
split.  usage: pmc_sq_summary.py <dir with pmc_sq/>  ->  <dir>/pmc_sq.txt"""
import collections
import csv
import glob
import sys

o = sys.argv[1]
f = glob.glob(o + '/pmc_sq/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][:64]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    n[k].add(r['Dispatch_Id'])
lines = ["rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras",
         "per launch; matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); SQ_INSTS_VALU includes the MFMAs; SQ_WAVE_CYCLES / SQ_WAIT_* in quad-cycles", ""]
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0))[:12]:
    l = len(n[k])
    cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8 / l
    busy = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / l / (1024 * cyc) if cyc else 0
    wc = max(c.get('SQ_WAVE_CYCLES', 1), 1)
    lines.append(f"{k:64s} launches {l:3d}  kernel cycles {cyc:10.4g}  matrix pipe busy {100 * busy:5.1f} %  VALU instr/launch {c.get('SQ_INSTS_VALU', 0) / l:10.4g}"
                 f"  wave-cycles: issue-stalled {100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:4.1f} %  waiting {100 * c.get('SQ_WAIT_ANY', 0) / wc:4.1f} %")
open(o + '/pmc_sq.txt', 'w').write("\n".join(lines) + "\n")
print("\n".join(lines))
