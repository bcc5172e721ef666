"""Per-dispatch averages of the counters collected by tools/pmc_conv.sh.  usage: pmc_summary.py <tag> <kernel substring>"""
import collections
import csv
import glob
import sys

tag, pat = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(f"gpurun_out/pmc_{tag}_?")):
    try:
        rows = list(csv.DictReader(open(f"{d}/run_counter_collection.csv")))
    except OSError:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        vals = list(v.values())
        print(f"{k:34s} {sum(vals) / len(vals):12.4e}  (n={len(vals)})")
