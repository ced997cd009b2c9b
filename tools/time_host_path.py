#!/usr/bin/env python3
"""PCIe-inclusive rate of the Real-ESRGAN x4 1080p path: host uint8 frame in (pageable and pinned), host frame out."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ctypes as C, numpy as np, torch
from framewright_amd import _lib
from framewright_amd.realesrgan import RRDBNetEngine
from framewright_amd.synth import synthetic_frames, synthetic_rrdbnet_state
eng = RRDBNetEngine(23, 4, "bf16"); eng.load_state_dict(synthetic_rrdbnet_state(23, 4))
f = synthetic_frames(1, 1080, 1920, seed=2)[0]
res = {}
eng.upscale(f)
t0 = time.perf_counter()
for _ in range(3): eng.upscale(f)
res["pageable_ms_per_frame"] = (time.perf_counter() - t0) / 3 * 1e3
pin_in = torch.from_numpy(f).pin_memory(); pin_out = torch.empty((4320, 7680, 3), dtype=torch.uint8).pin_memory()
lib = _lib.load()
def run():
    _lib.check(lib.fw_rrdbnet_upscale_u8(eng._h, C.c_void_p(pin_in.data_ptr()), _lib.FW_HOST, 1080, 1920, C.c_void_p(pin_out.data_ptr()), _lib.FW_HOST, None, None))
run()
t0 = time.perf_counter()
for _ in range(3): run()
res["pinned_ms_per_frame"] = (time.perf_counter() - t0) / 3 * 1e3
d = torch.from_numpy(f).cuda(); o = torch.empty((4320, 7680, 3), dtype=torch.uint8, device="cuda")
eng.upscale_device(d, out=o); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): eng.upscale_device(d, out=o)
torch.cuda.synchronize()
res["hbm_resident_ms_per_frame"] = (time.perf_counter() - t0) / 3 * 1e3
frames = [f] * 16
for _ in eng.upscale_stream(frames[:2]): pass
t0 = time.perf_counter()
n = sum(1 for _ in eng.upscale_stream(frames))
res["pipelined_host_ms_per_frame"] = (time.perf_counter() - t0) / n * 1e3
print(json.dumps(res))
