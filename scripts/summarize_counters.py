"""Per-kernel sums of every counter of a rocprofv3 counter-collection CSV (one or more passes), with the per-launch mean:
    python scripts/summarize_counters.py KERNEL_SUBSTRING run_counter_collection.csv [more.csv ...]"""
import collections
import csv
import sys

key = sys.argv[1]
tot = collections.defaultdict(float)
disp = collections.defaultdict(set)
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path, newline="")):
        if key in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            disp[r["Counter_Name"]].add((path, r["Dispatch_Id"]))
for k in sorted(tot):
    n = max(1, len(disp[k]))
    print(f"{k:34s} {tot[k] / n:16.0f} per launch  ({n} launches)")
