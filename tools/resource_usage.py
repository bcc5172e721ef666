#!/usr/bin/env python3
"""Compile-time resource usage (VGPR / AGPR / scratch / occupancy / LDS) of the kernels of one csrc file (no GPU needed):
    python tools/resource_usage.py conv.hip [name-filter] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dycon_paper_replication_amd", "csrc")
src, flt, extra = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else ""), sys.argv[3:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
                      *extra, "-c", src, "-o", "/dev/null"], cwd=csrc, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark:\s*([A-Za-z \[\]/]+?): (.+?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt and flt not in name:
        continue
    g = lambda k: r.get(k, "?")  # noqa: E731
    print(f"{name[-60:]:60s} VGPR {g('VGPRs'):>4} AGPR {g('AGPRs'):>4} scratch {g('ScratchSize [bytes/lane]'):>4} "
          f"occ {g('Occupancy [waves/SIMD]'):>2} LDS {g('LDS Size [bytes/block]'):>6}")
