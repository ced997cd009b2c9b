#!/usr/bin/env python3
"""Two frames in flight on one GPU: two engines, two streams, frames alternating.  FW_CONV_GRID=128 gives each launch half of the
CUs so that the two streams' kernels run side by side.  Prints frames/s."""
import json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from framewright_amd.realesrgan import RRDBNetEngine
from framewright_amd.synth import synthetic_frames, synthetic_rrdbnet_state
n_eng = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sd = synthetic_rrdbnet_state(23, 4)
engs = [RRDBNetEngine(23, 4, os.environ.get("FW_AB_DTYPE", "f16")) for _ in range(n_eng)]
for e in engs: e.load_state_dict(sd)
streams = [torch.cuda.Stream() for _ in range(n_eng)]
frames = [torch.from_numpy(f).cuda() for f in synthetic_frames(4, 1080, 1920, seed=2)]
outs = [torch.empty((4320, 7680, 3), dtype=torch.uint8, device="cuda") for _ in range(n_eng)]
def run(n):
    for i in range(n):
        k = i % n_eng
        engs[k].upscale_device(frames[i % 4], out=outs[k], stream=streams[k].cuda_stream)
    torch.cuda.synchronize()
run(2 * n_eng)
N = 24
t0 = time.perf_counter(); run(N); dt = time.perf_counter() - t0
print(json.dumps({"engines": n_eng, "grid": os.environ.get("FW_CONV_GRID", "256"), "fps": N / dt, "ms_per_frame": dt / N * 1e3,
                  "checksum": int(outs[0][::97, ::89].to(torch.int64).sum())}))
