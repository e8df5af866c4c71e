"""Averages rocprofv3 --pmc counter_collection CSVs per kernel.  Usage: pmc_summary.py <dir with g*/...>"""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per_dispatch[key] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for (disp, ctr), v in per_dispatch.items():
        k = names[disp].split("(")[0][:60]
        acc[k][ctr].append(v)
out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} | {"_dispatches": max(len(v) for v in cs.values())}
       for k, cs in acc.items() if k.startswith(("void mgx", "mgx"))}
json.dump(out, sys.stdout, indent=1)
