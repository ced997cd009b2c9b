#!/usr/bin/env python3
"""Restormer on a whole 1080p frame (the TAP driver with tile_size 0): fused fronts / merged projection / direct q-k layout against
the staged kernels (FW_REST_FUSE_FRONT=0 FW_REST_MERGE_PROJ=0), one process each; prints ms per frame and the difference."""
import json, os, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
CHILD = r"""
import json, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from framewright_amd import restormer as RS
from framewright_amd.synth import synthetic_frames
f = torch.from_numpy(synthetic_frames(1, 1080, 1920, seed=4)[0]).cuda()
eng = RS.RestormerEngine(dtype="f16", **RS.RESTORMER_ARGS); eng.load_state_dict(RS.synthetic_restormer_state(**RS.RESTORMER_ARGS))
out = torch.empty_like(f); rgb = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda")
eng.denoise_device(f, out=out, out_rgb_f32=rgb); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): eng.denoise_device(f, out=out, out_rgb_f32=rgb)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
np.save(sys.argv[1], rgb.cpu().numpy())
print(json.dumps({"ms": ms, "finite": bool(torch.isfinite(rgb).all()), "mean": float(rgb.mean())}))
"""
outs = []
for name, env in (("fused", {}), ("staged", {"FW_REST_FUSE_FRONT": "0", "FW_REST_MERGE_PROJ": "0"})):
    path = f"/tmp/rest_{name}.npy"
    r = subprocess.run([sys.executable, "-c", CHILD % str(ROOT), path], capture_output=True, text=True, env=dict(os.environ, **env))
    print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:], flush=True)
    outs.append(np.load(path))
d = np.abs(outs[0] - outs[1])
print(json.dumps({"max_abs_diff": float(d.max()), "mean_abs_diff": float(d.mean())}))
