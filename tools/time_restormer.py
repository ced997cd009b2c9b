#!/usr/bin/env python3
"""Restormer / NAFNet 512x512 tile timing (ms) on one MI355X."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from framewright_amd import restormer as RS, tap_denoise as T
from framewright_amd.synth import synthetic_frames, synthetic_nafnet_state
def timed(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
tile = torch.from_numpy(synthetic_frames(1, 512, 512, seed=4)[0]).cuda(); out = torch.empty_like(tile)
re_ = RS.RestormerEngine(dtype="f16", **RS.RESTORMER_ARGS); re_.load_state_dict(RS.synthetic_restormer_state(**RS.RESTORMER_ARGS))
res = {"restormer_512_tile_ms": timed(lambda: re_.denoise_device(tile, out=out))}
torch.cuda.synchronize(); t0 = time.perf_counter(); re_.denoise_device(tile, out=out); res["host_issue_ms"] = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
print(json.dumps(res))
