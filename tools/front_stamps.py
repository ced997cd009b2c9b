#!/usr/bin/env python3
"""Cycles per phase of the fused NAFBlock front kernel (pw_dw_fused.hip built with -DFW_FRONT_STAMP): one 1080p forward, the kernel
prints the phase totals of two waves of two workgroups per launch.  Rebuilds the default library at the end."""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import ab_front as A
CHILD = r"""
import sys
sys.path.insert(0, %r)
import torch
from framewright_amd import tap_denoise as T
from framewright_amd.synth import synthetic_frames, synthetic_nafnet_state
f = torch.from_numpy(synthetic_frames(1, 1080, 1920, seed=4)[0]).cuda(); out = torch.empty_like(f)
eng = T.NAFNetEngine(dtype="f16", **T.NAFNET_ARGS); eng.load_state_dict(synthetic_nafnet_state(**T.NAFNET_ARGS))
eng.denoise_device(f, out=out); torch.cuda.synchronize()
"""
CHILD_RESTORMER = r"""
import sys
sys.path.insert(0, %r)
import torch
from framewright_amd import restormer as RS
from framewright_amd.synth import synthetic_frames
t = torch.from_numpy(synthetic_frames(1, 512, 512, seed=4)[0]).cuda(); out = torch.empty_like(t)
eng = RS.RestormerEngine(dtype="f16", **RS.RESTORMER_ARGS); eng.load_state_dict(RS.synthetic_restormer_state(**RS.RESTORMER_ARGS))
eng.denoise_device(t, out=out); torch.cuda.synchronize()
"""
if os.environ.get("FW_STAMP_CHILD") == "restormer":   # one Restormer 512 x 512 tile instead (the 48- / 96-channel qkv and GDFN fronts)
    CHILD = CHILD_RESTORMER
A.rebuild("pw_dw_fused.hip", ["-DFW_FRONT_STAMP", *sys.argv[1:]])
r = subprocess.run([sys.executable, "-c", CHILD % str(ROOT)], capture_output=True, text=True)
print(r.stdout[-int(os.environ.get('FW_STAMP_TAIL', '6000')):], r.stderr[-2000:])
A.rebuild("pw_dw_fused.hip", [])
