#!/usr/bin/env python3
"""Rewrites the per-path numbers of DESIGN.md section 6.4 and README from profiles/r03_{secondary_paths,bench_rife,bench_tap}.json (run after
tools/update_docs_numbers.py)."""
import json
import re
from pathlib import Path
R = Path(__file__).resolve().parent.parent
sec = json.loads((R / "profiles/r03_secondary_paths.json").read_text())
br = json.loads((R / "profiles/r03_bench_rife.json").read_text())
bt = json.loads((R / "profiles/r03_bench_tap.json").read_text())
for name in ("DESIGN.md", "README.md"):
    p = R / name
    s = p.read_text()
    s = re.sub(r"\| NAFNet-width64 1080p forward \| [\d.]+ ms alone \(4\.03 TFLOP: 0\.12 of MFMA peak; 51\.3 GB: 3\.8 TB/s\), [\d.]+ ms per frame",
               f"| NAFNet-width64 1080p forward | {sec['nafnet_1080p_whole_frame_ms']:.2f} ms alone (4.03 TFLOP: 0.12 of MFMA peak; 51.3 GB: 3.8 TB/s), {bt['ms_per_step']:.2f} ms per frame", s)
    s = re.sub(r"\| IFNet v4\.6 1080p pair \| [\d.]+ ms alone, \*\*[\d.]+ ms per pair with three pairs in flight\*\*",
               f"| IFNet v4.6 1080p pair | {sec['rife_1080p_pair_ms']:.2f} ms alone, **{br['ms_per_step']:.2f} ms per pair with three pairs in flight**", s)
    s = re.sub(r"NAFNet temporal denoise [\d.]+ ms per 1080p forward \([\d.]+ frames/s with two\nframes in flight\)",
               f"NAFNet temporal denoise {sec['nafnet_1080p_whole_frame_ms']:.1f} ms per 1080p forward ({bt['value']:.1f} frames/s with two\nframes in flight)", s)
    s = re.sub(r"RIFE ×2 [\d.]+ ms per 1080p pair alone, [\d.]+ ms \(\d+ pairs/s\) with three in flight",
               f"RIFE ×2 {sec['rife_1080p_pair_ms']:.2f} ms per 1080p pair alone, {br['ms_per_step']:.2f} ms ({br['value']:.0f} pairs/s) with three in flight", s)
    p.write_text(s)
print(f"nafnet {sec['nafnet_1080p_whole_frame_ms']:.2f} / {bt['ms_per_step']:.2f} ms, ifnet {sec['rife_1080p_pair_ms']:.2f} / {br['ms_per_step']:.2f} ms")
