#!/usr/bin/env python3
"""SRVGGNetCompact x4 on a 1080p frame: realesr-animevideov3 (16 convs) and realesr-general-x4v3 (32 convs); ms per frame."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from framewright_amd.srvgg import SRVGGNetEngine, synthetic_srvgg_state
from framewright_amd.synth import synthetic_frames
f = torch.from_numpy(synthetic_frames(1, 1080, 1920, seed=2)[0]).cuda()
res = {}
for name, nc in (("animevideov3_16", 16), ("general_x4v3_32", 32)):
    eng = SRVGGNetEngine(nc, 4, "bf16"); eng.load_state_dict(synthetic_srvgg_state(nc, 4))
    for _ in range(2): eng.upscale_device(f)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): eng.upscale_device(f)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 8 * 1e3
    # 64 -> 64 3x3 convs at 1080p: 2 * 9 * 64 * 64 * pixels each, + the 3 -> 64 head and the 64 -> 48 tail
    flop = 2 * 9 * 2073600 * (3 * 64 + nc * 64 * 64 + 64 * 48)
    res[name] = {"ms": ms, "tflops": flop / ms / 1e9}
print(json.dumps(res))
