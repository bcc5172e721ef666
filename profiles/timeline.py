#!/usr/bin/env python3
"""Per-stream timeline of ONE steady-state DyCON step from a rocprofv3 --kernel-trace CSV (the span between the last two
add_noise launches).  Prints, per HIP stream (queue), launches / busy time / summed gaps, and for the stream that carries the
most launches (the student's dependent chain) every kernel with its start offset, duration and the gap to its predecessor.
Usage: timeline.py <kernel_trace.csv> [--full]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "add_noise" in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
qkey = "Queue_Id" if "Queue_Id" in seg[0] else "Stream_Id"
byq = collections.defaultdict(list)
for r in seg:
    byq[r[qkey]].append(r)
wall = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
print(f"step wall {wall:.1f} us, {len(seg)} launches, {len(byq)} queues")
main = max(byq, key=lambda q: len(byq[q]))
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
    span = (int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])) / 1e3
    print(f"  queue {q}: {len(rs):4d} launches, busy {busy:8.1f} us, first..last span {span:8.1f} us{'   <- main chain' if q == main else ''}")
rs = byq[main]
prev_end = None
gaps = []
short = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rs:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    gaps.append(gap)
    name = r["Kernel_Name"].split("(")[0][-44:]
    short[name][0] += 1
    short[name][1] += (e - s) / 1e3
    short[name][2] += max(gap, 0.0)
    if "--full" in sys.argv:
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:7.1f}  grid {r['Grid_Size_X']:>8}  {name}")
    prev_end = max(e, prev_end or e)
pos = [g for g in gaps if g > 0]
print(f"main chain: {len(rs)} launches, kernel time {sum(v[1] for v in short.values()):.1f} us, positive gaps {sum(pos):.1f} us "
      f"(median {sorted(pos)[len(pos) // 2] if pos else 0:.1f} us)")
for k, v in sorted(short.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:30]:
    print(f"{v[1]:9.1f} us kernel + {v[2]:8.1f} us gap-before  {v[0]:4d} calls  {k}")
