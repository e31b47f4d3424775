#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py video-fragments-retrieval_amd/csrc/score.hip [filter]
"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
       "-fvisibility=hidden", "-fno-slp-vectorize", *sys.argv[3:], "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kr.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print(f"{'kernel':70s} VGPR AGPR  SGPR scratch occ  vspill")
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:70]:70s} {r.get('VGPRs','?'):>4} {r.get('AGPRs','?'):>4} {r.get('SGPRs','?'):>5} {r.get('ScratchSize [bytes/lane]','?'):>7} "
              f"{r.get('Occupancy [waves/SIMD]','?'):>3} {r.get('VGPRs Spill','?'):>6}")
