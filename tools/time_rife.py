#!/usr/bin/env python3
"""IFNet v4.6 on one 1080p pair: ms per forward (FW_IFNET_FUSE_GLUE / FW_IFNET_MERGE_GROUPS / FW_IFNET_GRAPH for the A/B)."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from framewright_amd import rife as RF
from framewright_amd.synth import synthetic_frames, synthetic_ifnet_state
fr = synthetic_frames(2, 1080, 1920, seed=3)
a, b = torch.from_numpy(fr[0]).cuda(), torch.from_numpy(fr[1]).cuda()
eng = RF.IFNetEngine("f16"); eng.load_state_dict(synthetic_ifnet_state())
out = torch.empty_like(a)
for _ in range(5): eng.interpolate_device(a, b, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 50
for _ in range(N): eng.interpolate_device(a, b, out=out)
torch.cuda.synchronize(); print(json.dumps({"ms": (time.perf_counter() - t0) / N * 1e3, "checksum": int(out[::7, ::5].to(torch.int64).sum())}))
