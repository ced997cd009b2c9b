import sys, time; sys.path.insert(0, ".")
import torch
from framewright_amd import tap_denoise as T
from framewright_amd.synth import synthetic_frames, synthetic_nafnet_state
f = torch.from_numpy(synthetic_frames(1, 1080, 1920, seed=4)[0]).cuda()
eng = T.NAFNetEngine(dtype="f16", **T.NAFNET_ARGS); eng.load_state_dict(synthetic_nafnet_state(**T.NAFNET_ARGS))
out = torch.empty_like(f)
for _ in range(2): eng.denoise_device(f, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): eng.denoise_device(f, out=out)
torch.cuda.synchronize(); print("nafnet 1080p ms", (time.perf_counter() - t0) / 5 * 1e3)
