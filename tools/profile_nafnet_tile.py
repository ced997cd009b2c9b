#!/usr/bin/env python3
"""NAFNet-width64 on one 512x512 tile (5 iterations) for rocprofv3 --kernel-trace --stats: kernel time vs wall time."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from framewright_amd import tap_denoise as T
from framewright_amd.synth import synthetic_frames, synthetic_nafnet_state
f = torch.from_numpy(synthetic_frames(1, 512, 512, seed=4)[0]).cuda()
eng = T.NAFNetEngine(dtype="f16", **T.NAFNET_ARGS); eng.load_state_dict(synthetic_nafnet_state(**T.NAFNET_ARGS))
out = torch.empty_like(f)
eng.denoise_device(f, out=out); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    eng.denoise_device(f, out=out)
torch.cuda.synchronize()
print("wall ms per tile", (time.perf_counter() - t0) / 5 * 1e3)
