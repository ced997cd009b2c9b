#!/usr/bin/env python3
"""Joules per frame and per launch class of the headline path, from what profiles/refresh.sh leaves behind:

  python tools/energy_table.py POWER.log POWER_bench.json KERNEL_TRACE.csv OUT_power.json OUT_energy.json

POWER.log: tools/power_trace.sh (rocm-smi every 0.5 s while bench.py --steps 150 runs); KERNEL_TRACE.csv: rocprofv3 --kernel-trace of
bench.py (one row per launch).  The conv kernels all run at the chip's power cap, so a class's energy is the socket power under load
times the time its launches take; algorithmic FLOPs per class follow the reference's layer shapes (fw_rrdbnet_flops), executed MACs
the kernels' tile geometry (ring columns, warm-up tiles, identity MFMAs of the split trunk, 4-of-9-tap phase up-convs)."""
import csv, json, re, statistics as st, sys

plog, pbench, trace, out_power, out_energy = sys.argv[1:6]
t = open(plog).read()
pw = [float(x) for x in re.findall(r"Package Power \(W\): ([\d.]+)", t)]
sc = [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", t)]
tj = [float(x) for x in re.findall(r"junction\) \(C\): ([\d.]+)", t)]
load = [p for p in pw if p > 0.9 * max(pw)]
load_sc = [s for p, s in zip(pw, sc) if p > 0.9 * max(pw)] or sc
bench = json.loads(open(pbench).read().strip().splitlines()[-1])
ms = bench["ms_per_step"]
watts = st.median(load)
power = {"samples": len(pw), "samples_under_load": len(load), "socket_power_w": {"min": min(pw), "median": st.median(pw), "max": max(pw)},
         "under_load": {"power_w_median": watts, "sclk_mhz_median": st.median(load_sc), "sclk_mhz_min": min(load_sc), "sclk_mhz_max": max(load_sc)},
         "junction_c_max": max(tj) if tj else None, "bench_frames_per_s": bench["value"], "bench_ms_per_frame": ms,
         "joules_per_frame": watts * ms * 1e-3,
         "pj_per_algorithmic_flop": watts * ms * 1e-3 / (bench["config"]["frame_tflop"] * 1e12) * 1e12,
         "how": "tools/power_trace.sh: rocm-smi --showpower --showclocks sampled every 0.5 s while bench.py --steps 150 runs; under load = samples "
                "above 0.9 of the maximum (the first samples fall into engine set-up)"}
json.dump(power, open(out_power, "w"), indent=1)

rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if "fw" in r["Kernel_Name"]]
starts = [i for i, k in enumerate(ks) if "u8_to_nhwc" in k[0]]
frames = [ks[a:b] for a, b in zip(starts[1:-1], starts[2:])]   # whole frames, the first (warm-up) one dropped
H, W = 1080, 1920
px = H * W
def gflop(cin, cout, pixels=px): return 2.0 * 9 * cin * cout * pixels / 1e9
cls = {}
def add(name, us, alg, exe):
    c = cls.setdefault(name, {"launches": 0, "us": 0.0, "alg_gflop": 0.0, "exe_gflop": 0.0})
    c["launches"] += 1; c["us"] += us; c["alg_gflop"] += alg; c["exe_gflop"] += exe
tiles_pair = 64 * 68          # 30-column tiles: ceil(1920 / 30) x (1080 + 16) // 16
warm = 192                    # 3 of the 4 workgroups of a column start mid-column: conv_a of the tile above, by one wave (1/16 of a tile's conv_a)
for fr in frames:
    is_body = lambda nm: "pair_slide" in nm or "Li2ELi3" in nm or "wino_split" in nm
    body = [k for k in fr if is_body(k[0])]
    rest = [k for k in fr if not is_body(k[0])]
    for i, (nm, d) in enumerate(body):
        m = i % 3
        if m < 2:
            na = 2 + 2 * m
            alg = gflop(32 * na, 32) + gflop(32 * na + 32, 32)
            exe = alg * (tiles_pair * 16 * 32) / px + gflop(32 * na, 32) * (warm / 16) * (16 * 32) / px
            add("pair conv1+conv2" if m == 0 else "pair conv3+conv4", d / 1e3, alg, exe)
        else:
            rdb3 = (i // 3) % 3 == 2
            alg = gflop(192, 64)
            tiles5 = 60 * 68
            ident = (2 + (4 if rdb3 else 0)) * 8 / (6 * 144.0)      # identity MFMAs per tile and wave against 6 x 144 conv MFMAs
            wino = "wino_split" in nm                               # row-wise Winograd F(2, 3): 96 instead of 144 MFMAs per item and wave
            exe = alg * (tiles5 * 512) / px * ((2.0 / 3.0 if wino else 1.0) + ident)
            add("conv5 of rdb3 (+ R hi, R lo; writes hi + lo)" if rdb3 else ("conv5 of rdb1 / rdb2, Winograd F(2, 3) rows" if wino else "conv5 of rdb1 / rdb2"), d / 1e3, alg, exe)
    names = ["u8 -> NHWC", "conv_first", "conv_body", "conv_up1 (phase)", "conv_up2 (phase)", "conv_hr", "conv_last"]
    algs = [0, gflop(3, 64), gflop(64, 64), gflop(64, 64, 4 * px), gflop(64, 64, 16 * px), gflop(64, 64, 16 * px), gflop(64, 3, 16 * px)]
    exes = [0, gflop(32, 64), gflop(64, 64), gflop(64, 64, 4 * px) * 4 / 9, gflop(64, 64, 16 * px) * 4 / 9, gflop(64, 64, 16 * px), gflop(64, 32, 16 * px)]
    if len(rest) == len(names):
        for (nm, d), n, a, e in zip(rest, names, algs, exes):
            add(n, d / 1e3, a, e)
nf = len(frames)
table = []
tot_us = sum(c["us"] for c in cls.values()) / nf
for n, c in cls.items():
    us, alg, exe = c["us"] / nf, c["alg_gflop"] / nf, c["exe_gflop"] / nf
    j = watts * us * 1e-6
    table.append({"class": n, "launches_per_frame": c["launches"] / nf, "ms_per_frame": us / 1e3, "avg_launch_us": c["us"] / c["launches"],
                  "algorithmic_gflop_per_frame": alg, "executed_over_algorithmic": (exe / alg) if alg else None,
                  "tflops_algorithmic": alg / us * 1e3 if alg else None, "frac_of_mfma_peak": alg / us * 1e3 / 2500 if alg else None,
                  "joules_per_frame": j, "pj_per_algorithmic_flop": j / (alg * 1e9) * 1e12 if alg else None})
json.dump({"frames_in_trace": nf, "socket_power_w": watts, "kernel_ms_per_frame": tot_us / 1e3, "joules_per_frame_kernels": watts * tot_us * 1e-6,
           "classes": table,
           "reference_points": {"hipblaslt_f16_gemm_pj_per_flop": 0.98, "bare_mfma_loop_random_data_pj_per_flop": 1400 / 1954e12 * 1e12,
                                "source": "profiles/r02_gemm_ceiling.json, profiles/r02_mfma_peak.json (1261 W / 1282 TFLOP/s; 1400 W / 1954 TFLOP/s)"}},
          open(out_energy, "w"), indent=1)
print("| launch class | per frame | ms | avg launch us | of MFMA peak | executed / algorithmic MACs | J | pJ / FLOP |")
print("|---|---|---|---|---|---|---|---|")
for r in table:
    f = lambda v, fmt: "-" if v is None else format(v, fmt)
    print(f"| {r['class']} | {r['launches_per_frame']:.0f} | {r['ms_per_frame']:.2f} | {r['avg_launch_us']:.0f} | {f(r['frac_of_mfma_peak'], '.3f')} | "
          f"{f(r['executed_over_algorithmic'], '.3f')} | {r['joules_per_frame']:.1f} | {f(r['pj_per_algorithmic_flop'], '.2f')} |")
print(f"\nframe: {tot_us / 1e3:.2f} ms of kernels, {watts:.0f} W under load -> {watts * tot_us * 1e-6:.1f} J of kernels; bench {ms:.2f} ms -> {watts * ms * 1e-3:.1f} J per frame, "
      f"{power['pj_per_algorithmic_flop']:.2f} pJ per algorithmic FLOP")
