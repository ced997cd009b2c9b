#!/usr/bin/env python3
"""Restormer on one 512x512 tile (3 iterations) for rocprofv3 --kernel-trace --stats."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from framewright_amd import restormer as RS
from framewright_amd.synth import synthetic_frames
f = torch.from_numpy(synthetic_frames(1, 512, 512, seed=4)[0]).cuda()
eng = RS.RestormerEngine(dtype="f16", **RS.RESTORMER_ARGS); eng.load_state_dict(RS.synthetic_restormer_state(**RS.RESTORMER_ARGS))
out = torch.empty_like(f)
for _ in range(3):
    eng.denoise_device(f, out=out)
torch.cuda.synchronize()
