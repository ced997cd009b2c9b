#!/usr/bin/env python3
"""profiles/<round>_traffic.json from the PMC passes written by profiles/pmc_pass.sh.

  python tools/make_traffic.py gpurun_out/pmc profiles/r01_traffic.json

Per MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are collected in separate passes, are in
KiB, and on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B, so it is doubled.  The figure is the average over every
conv launch (conv3x3_mfma_kernel and conv3x3_pair_kernel) of `bench.py --steps 1 --warmup 1`.
"""
import csv, glob, json, sys

src, dst = sys.argv[1], sys.argv[2]
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
launches, dur_ns = {}, {}
for f in glob.glob(f"{src}/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3x3" not in r["Kernel_Name"] or r["Counter_Name"] not in tot:
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE":
            launches[r["Dispatch_Id"]] = 1
            dur_ns[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
n = len(launches)
fetch = tot["FETCH_SIZE"] * 1024 * 2 / n
write = tot["WRITE_SIZE"] * 1024 / n
avg_ms = sum(dur_ns.values()) / n / 1e6
json.dump({
    "kernel": "conv3x3_mfma_kernel + conv3x3_pair_kernel (all instantiations)",
    "launches": n,
    "hbm_bytes_per_launch": fetch + write,
    "fetch_bytes_per_launch_corrected": fetch,
    "write_bytes_per_launch": write,
    "avg_launch_ms_under_pmc": avg_ms,
    "method": "rocprofv3 --kernel-trace --pmc, separate passes for FETCH_SIZE and WRITE_SIZE (profiles/pmc_pass.sh); "
              "KiB -> bytes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); averaged "
              "over every conv launch of bench.py --steps 1 --warmup 1",
    "hbm_tb_per_s": (fetch + write) / (avg_ms * 1e-3) / 1e12,
}, open(dst, "w"), indent=1)
print(open(dst).read())
