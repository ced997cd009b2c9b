#!/usr/bin/env python3
"""1080p frame through the TAP driver with the reference's 512 / 32 tiling (12 tiles), NAFNet and Restormer, for FW_TAP_TILE_STREAMS =
1, 2, 3, 4, 6 (ms per frame, one process per setting)."""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
CHILD = r"""
import json, sys, time
sys.path.insert(0, %r)
import torch
from framewright_amd import tap_denoise as T, restormer as RS
from framewright_amd.synth import synthetic_frames, synthetic_nafnet_state
f = torch.from_numpy(synthetic_frames(1, 1080, 1920, seed=4)[0]).cuda()
res = {}
for name in ("nafnet", "restormer"):
    if name == "nafnet":
        eng = T.NAFNetEngine(dtype="f16", **T.NAFNET_ARGS); eng.load_state_dict(synthetic_nafnet_state(**T.NAFNET_ARGS))
    else:
        eng = RS.RestormerEngine(dtype="f16", **RS.RESTORMER_ARGS); eng.load_state_dict(RS.synthetic_restormer_state(**RS.RESTORMER_ARGS))
    dn = T.TAPDenoiser(T.TAPDenoiseConfig(model=name, tile_size=512, tile_overlap=32, temporal_window=1), engine=eng)
    for _ in range(2): dn._denoise_frame_tiled_device(f)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 5
    for _ in range(N): out = dn._denoise_frame_tiled_device(f)
    torch.cuda.synchronize(); res[name] = (time.perf_counter() - t0) / N * 1e3
    res[name + "_checksum"] = int(out[::7, ::5].to(torch.int64).sum())
    dn.clear_cache()
print(json.dumps(res))
"""
for k in sys.argv[1:] or ("1", "2", "3", "4", "6"):
    r = subprocess.run([sys.executable, "-c", CHILD % str(ROOT)], capture_output=True, text=True, env=dict(os.environ, FW_TAP_TILE_STREAMS=k))
    print("tile_streams", k, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-600:], flush=True)
