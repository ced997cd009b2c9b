#!/usr/bin/env python3
"""Diagnostic (build with FW_EXTRA_CXXFLAGS=-DFW_PAIR_STAMP): where wave 0 of each workgroup of the fused pair kernel
spends its cycles.  Shares only — a stamped build is slower than the product build."""
import ctypes as C, json, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from framewright_amd import build as B
B.build()
from framewright_amd import _lib
from framewright_amd.realesrgan import RRDBNetEngine
from framewright_amd.synth import synthetic_frames, synthetic_rrdbnet_state
eng = RRDBNetEngine(23, 4, "bf16"); eng.load_state_dict(synthetic_rrdbnet_state(23, 4))
d = torch.from_numpy(synthetic_frames(1, 1080, 1920, seed=2)[0]).cuda(); o = torch.empty((4320, 7680, 3), dtype=torch.uint8, device="cuda")
lib = _lib.load()
buf = (C.c_ulonglong * 64)()
eng.upscale_device(d, out=o); torch.cuda.synchronize(); lib.fw_debug_pair_stamps(buf)
for _ in range(3):
    eng.upscale_device(d, out=o)
torch.cuda.synchronize(); lib.fw_debug_pair_stamps(buf)
v = np.array(list(buf), dtype=np.float64).reshape(8, 8)
names = ["barrier", "shared-chunk item compute", "x_a item compute", "emit (x_a + x_b)", "tile setup", "vmcnt wait"]
launches = 3 * 138 * 256
for w in range(8):
    cyc = v[w, :6].sum()
    print(f"wave {w}: " + "  ".join(f"{n.split()[0]} {x / cyc:.3f}" for n, x in zip(names, v[w, :6])) +
          f"  | cycles/launch {cyc / launches:.0f}  us/launch {v[w, 7] / launches / 100:.1f}  clock {cyc / v[w, 7] * 0.1:.3f} GHz")
