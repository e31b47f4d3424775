#!/usr/bin/env python3
"""Key fields of a bench.py JSON line: python tools/show_line.py <file>"""
import json
import sys

lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not lines:
    print(sys.argv[1], "no JSON line")
    sys.exit(1)
d = json.loads(lines[-1])
r = d.get("roofline", {})
print(f"n_gpus {d.get('n_gpus')}  ms/step {d.get('ms_per_step'):.3f}  value {d.get('value'):.4g}  checksums {d.get('ranks_checksum')} / {d.get('topk_checksum')}  "
      f"dominant {r.get('kernel')} frac {r.get('frac')} exec {r.get('frac_executed')}  notes {d.get('accounting_notes')}")
