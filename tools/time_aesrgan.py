#!/usr/bin/env python3
"""AESRGAN(num_block=23, scale=2, num_attention=4) - the reference's defaults - on face crops; ms per crop."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from framewright_amd.aesrgan import AESRGANEngine
from framewright_amd.synth import synthetic_attention_state, synthetic_rrdbnet_state
eng = AESRGANEngine(23, 2, 4, "f16")
eng.load_state_dict(synthetic_rrdbnet_state(23, 4, seed=1), synthetic_attention_state(23, 4, seed=2))
res = {}
for side in (64, 128, 192):
    x = torch.rand((side, side, 3), device="cuda")
    for _ in range(2): eng.forward_rgb(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): eng.forward_rgb(x)
    torch.cuda.synchronize(); res[f"crop_{side}_ms"] = (time.perf_counter() - t0) / 5 * 1e3
print(json.dumps(res))
