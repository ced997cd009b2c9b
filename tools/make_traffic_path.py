#!/usr/bin/env python3
"""HBM traffic of one forward of a secondary path from two PMC passes (profiles/pmc_pass_path.sh with FETCH_SIZE and WRITE_SIZE):

  python tools/make_traffic_path.py gpurun_out/r02/pmc_tap profiles/r02_traffic_tap.json FORWARDS [dtype]

FORWARDS = forwards the profile target ran (tools/profile_nafnet.py: 4, tools/profile_restormer.py: 3).  Units and the gfx950
correction as in tools/make_traffic.py (KiB; FETCH_SIZE doubled).  Every kernel of the library is counted (the blit kernels of the
weight upload are not).  Tagged with the digest of the library it was measured on."""
import csv, glob, hashlib, json, sys
from pathlib import Path

src, dst, forwards = sys.argv[1], sys.argv[2], int(sys.argv[3])
dtype = sys.argv[4] if len(sys.argv) > 4 else "f16"
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
per_kernel = {}
for f in glob.glob(f"{src}/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] not in tot or r["Kernel_Name"].startswith("__amd_rocclr"):
            continue
        v = float(r["Counter_Value"])
        tot[r["Counter_Name"]] += v
        k = per_kernel.setdefault(r["Kernel_Name"][:70], {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
        k[r["Counter_Name"]] += v
        if r["Counter_Name"] == "FETCH_SIZE":
            k["n"] += 1
fetch = tot["FETCH_SIZE"] * 1024 * 2 / forwards
write = tot["WRITE_SIZE"] * 1024 / forwards
root = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
from framewright_amd import build as _build   # the digest names the build by its sources and flags
json.dump({
    "lib_digest": _build.source_digest(),
    "dtype": dtype,
    "forwards": forwards,
    "hbm_bytes_per_forward": fetch + write,
    "fetch_bytes_per_forward_corrected": fetch,
    "write_bytes_per_forward": write,
    "per_kernel_mb_per_forward": {k: {"fetch": v["FETCH_SIZE"] * 2048 / forwards / 1e6, "write": v["WRITE_SIZE"] * 1024 / forwards / 1e6,
                                      "launches": v["n"] / forwards} for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["FETCH_SIZE"])},
    "method": "rocprofv3 --kernel-trace --pmc, separate passes for FETCH_SIZE and WRITE_SIZE (profiles/pmc_pass_path.sh); KiB -> bytes; "
              "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); summed over every kernel of the forwards",
}, open(dst, "w"), indent=1)
print(open(dst).read()[:1500])
