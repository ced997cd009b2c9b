#!/usr/bin/env python3
"""Diagnostic (build with FW_EXTRA_CXXFLAGS=-DFW_PAIR_STAMP): where the waves of the fused pair kernel and of the
64-channel residual conv spend their cycles.  Shares only — a stamped build is slower than the product build."""
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
buf = (C.c_ulonglong * 128)()
eng.upscale_device(d, out=o); torch.cuda.synchronize()
for k in (0, 1):
    lib.fw_debug_stamps(k, buf)
for _ in range(3):
    eng.upscale_device(d, out=o)
torch.cuda.synchronize()
NAMES = [["barrier", "shared-item", "xa-item", "xa-tile+carry", "tile-setup", "vmcnt", "convert+store"],
         ["barrier", "item", "residual-plane", "f32-epi", "tile-setup", "vmcnt", "typed-store"]]
for k, (title, per_frame) in enumerate([("fused pair kernel", 138), ("64-channel residual conv (conv5, conv_body)", 70)]):
    lib.fw_debug_stamps(k, buf)
    raw = np.array(list(buf), dtype=np.float64)
    v = raw[:64].reshape(8, 8)
    mx, sq = raw[64:72], raw[72:80]
    launches = 3 * per_frame * 256
    print(title)
    for w in range(8):
        cyc = v[w, :7].sum()
        print(f"  wave {w}: " + "  ".join(f"{n} {x / cyc:.3f}" for n, x in zip(NAMES[k], v[w, :7]) if n != "-") +
              f"  | cycles/launch {cyc / launches:.0f}  us/launch {v[w, 7] / launches / 100:.1f}  clock {cyc / v[w, 7] * 0.1:.3f} GHz"
              f"  slowest workgroup {mx[w] / 100:.1f} us, rms {np.sqrt(sq[w] / launches) / 100:.1f} us")
