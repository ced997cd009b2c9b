#!/usr/bin/env python3
"""One NAFNet-width64 1080p forward (3 iterations) for rocprofv3 --kernel-trace --stats."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from framewright_amd import tap_denoise as T
from framewright_amd.synth import synthetic_frames, synthetic_nafnet_state
f = torch.from_numpy(synthetic_frames(1, 1080, 1920, seed=4)[0]).cuda()
eng = T.NAFNetEngine(dtype="f16", **T.NAFNET_ARGS); eng.load_state_dict(synthetic_nafnet_state(**T.NAFNET_ARGS))
out = torch.empty_like(f)
for _ in range(4):
    eng.denoise_device(f, out=out)
torch.cuda.synchronize()
