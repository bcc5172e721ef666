#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV for ONE steady-state DyCON step (the span between the 2nd and 3rd
add_noise launches), by kernel and grid.  Usage: analyze_trace.py <kernel_trace.csv> [--by-grid]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "add_noise" in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
seg = rows[a:b]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3  # noqa: E731
busy = sum(dur(r) for r in seg)
wall = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
print(f"step wall {wall:.1f} us  busy {busy:.1f} us  launches {len(seg)}")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    name = r["Kernel_Name"].split("(")[0][-52:]
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]) if "--by-grid" in sys.argv else name
    agg[key][0] += 1
    agg[key][1] += dur(r)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{v[1]:9.1f} us {v[0]:4d} calls  avg {v[1] / v[0]:8.1f} us  {k}")
