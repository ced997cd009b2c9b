#!/usr/bin/env python3
"""Timing of the secondary BASELINE configs on one MI355X (not the headline bench): NAFNet temporal denoise 1080p
(reference tiling 512/32 and whole-frame) and RIFE x2 1080p pairs.  Prints one JSON line."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from framewright_amd import tap_denoise as T, rife as RF
from framewright_amd.synth import synthetic_frames, synthetic_nafnet_state, synthetic_ifnet_state

def timed(fn, n=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

frames = synthetic_frames(2, 1080, 1920, seed=4)
dev = [torch.from_numpy(f).cuda() for f in frames]
res = {}
eng = T.NAFNetEngine(dtype="f16", **T.NAFNET_ARGS); eng.load_state_dict(synthetic_nafnet_state(**T.NAFNET_ARGS))
out = torch.empty_like(dev[0])
res["nafnet_1080p_whole_frame_ms"] = timed(lambda: eng.denoise_device(dev[0], out=out))
res["nafnet_1080p_tflop"] = eng.flops(1080, 1920) / 1e12
dn = T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet", tile_size=512, tile_overlap=32), engine=eng)
res["tap_1080p_tiled512_per_frame_ms"] = timed(lambda: dn._denoise_frame_tiled_device(dev[0]))
tile = torch.from_numpy(np.ascontiguousarray(frames[0][:512, :512])).cuda(); tout = torch.empty_like(tile)
res["nafnet_512_tile_ms"] = timed(lambda: eng.denoise_device(tile, out=tout), n=5)
eng.close()
from framewright_amd import restormer as RS
re_ = RS.RestormerEngine(dtype="f16", **RS.RESTORMER_ARGS); re_.load_state_dict(RS.synthetic_restormer_state(**RS.RESTORMER_ARGS))
res["restormer_512_tile_ms"] = timed(lambda: re_.denoise_device(tile, out=tout), n=3)
dr = T.TAPDenoiser(T.TAPDenoiseConfig(tile_size=512, tile_overlap=32), engine=re_)
res["tap_restormer_1080p_tiled512_per_frame_ms"] = timed(lambda: dr._denoise_frame_tiled_device(dev[0]), n=2)
re_.close()
ie = RF.IFNetEngine("f16"); ie.load_state_dict(synthetic_ifnet_state())
o2 = torch.empty_like(dev[0])
res["rife_1080p_pair_ms"] = timed(lambda: ie.interpolate_device(dev[0], dev[1], out=o2))
# BASELINE configs[4] on one GPU: temporal denoise (NAFNet, window 5, whole frame) -> Real-ESRGAN x4plus (bf16) -> RIFE x2 on
# the 8K frames, device-resident hand-off (pipeline.py); per INPUT frame, 4-frame clip
try:
    from framewright_amd import pipeline as P, realesrgan as R
    from framewright_amd.synth import synthetic_rrdbnet_state
    clip = [torch.from_numpy(f).cuda() for f in synthetic_frames(4, 1080, 1920, seed=5)]
    naf = T.NAFNetEngine(dtype="f16", **T.NAFNET_ARGS); naf.load_state_dict(synthetic_nafnet_state(**T.NAFNET_ARGS))
    dnc = T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet", tile_size=0, temporal_window=5), engine=naf)
    sr = R.RRDBNetEngine(23, 4, "bf16"); sr.load_state_dict(synthetic_rrdbnet_state(23, 4))
    pipe = P.DeviceRestorationPipeline(dnc, sr, ie, interp_passes=1)
    def chain():
        return pipe.run_device(clip)
    res["chain_1080p_denoise_x4_rife_per_input_frame_ms"] = timed(chain, n=2, warm=1) / len(clip)
    res["chain_output"] = "7 frames of 7680x4320 per 4 input frames"
except Exception as e:  # noqa: BLE001 - record, do not lose the other numbers
    res["chain_error"] = f"{type(e).__name__}: {e}"[:300]
print(json.dumps(res))
