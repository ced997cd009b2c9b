#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (p1,p2,p3 under a directory) per kernel: sums over dispatches + total duration."""
import collections, csv, glob, json, sys
d = sys.argv[1]
out = {}
for p in sorted(glob.glob(f"{d}/p*")):
    f = glob.glob(f"{p}/**/*_counter_collection.csv", recursive=True)
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, v in agg.items():
        o = out.setdefault(k, {})
        o["dispatches"] = len(disp[k])
        o.setdefault("total_ms_" + p.split("/")[-1], sum(disp[k].values()) / 1e6)
        o.update(v)
json.dump(out, open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout, indent=1)
