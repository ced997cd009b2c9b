#!/usr/bin/env python3
"""profiles/<round>_traffic.json from the PMC passes written by profiles/pmc_pass.sh.

  python tools/make_traffic.py gpurun_out/pmc profiles/r02_traffic.json [dtype]

Per MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are collected in separate passes, are in
KiB, and on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B, so it is doubled.  The figures are the average over every
conv launch (conv3x3_mfma_kernel, conv3x3_pair*_kernel, conv_up2x_phase_kernel) of `bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-path`
and the sum over the 213 conv launches of one frame.  The file is tagged with the digest of the library it was measured on:
bench.py quotes it only when that is the library it runs.
"""
import csv, glob, hashlib, json, sys
from pathlib import Path

src, dst = sys.argv[1], sys.argv[2]
dtype = sys.argv[3] if len(sys.argv) > 3 else "f16"
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
per_kernel = {}
launches, dur_ns = {}, {}
for f in glob.glob(f"{src}/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if not ("conv3x3" in r["Kernel_Name"] or "conv_up2x_phase" in r["Kernel_Name"]) or r["Counter_Name"] not in tot:
            continue
        v = float(r["Counter_Value"])
        tot[r["Counter_Name"]] += v
        k = per_kernel.setdefault(r["Kernel_Name"][:80], {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
        k[r["Counter_Name"]] += v
        if r["Counter_Name"] == "FETCH_SIZE":
            k["n"] += 1
            launches[r["Dispatch_Id"]] = 1
            dur_ns[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
n = len(launches)
LAUNCHES_PER_FRAME = 213          # 138 fused pairs + 69 conv5 + first / body / up1 / up2 / hr / last
frames = n / LAUNCHES_PER_FRAME
fetch = tot["FETCH_SIZE"] * 1024 * 2 / n
write = tot["WRITE_SIZE"] * 1024 / n
avg_ms = sum(dur_ns.values()) / n / 1e6
root = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
from framewright_amd import build as _build   # the digest names the build by its sources and flags
json.dump({
    "kernel": "conv3x3_mfma_kernel + conv3x3_pair_slide_kernel + conv_up2x_phase_kernel (all instantiations)",
    "lib_digest": _build.source_digest(),
    "dtype": dtype,
    "launches": n,
    "frames": frames,
    "hbm_bytes_per_launch": fetch + write,
    "fetch_bytes_per_launch_corrected": fetch,
    "write_bytes_per_launch": write,
    "hbm_gb_per_frame": (fetch + write) * LAUNCHES_PER_FRAME / 1e9,
    "avg_launch_ms_under_pmc": avg_ms,
    "per_kernel_mb_per_launch": {k: {"fetch": v["FETCH_SIZE"] * 2048 / max(v["n"], 1) / 1e6, "write": v["WRITE_SIZE"] * 1024 / max(v["n"], 1) / 1e6,
                                     "launches": v["n"]} for k, v in per_kernel.items()},
    "method": "rocprofv3 --kernel-trace --pmc, separate passes for FETCH_SIZE and WRITE_SIZE (profiles/pmc_pass.sh); "
              "KiB -> bytes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); averaged "
              "over every conv launch of bench.py --steps 1 --warmup 1",
    "hbm_tb_per_s": (fetch + write) / (avg_ms * 1e-3) / 1e12,
}, open(dst, "w"), indent=1)
print(open(dst).read())
