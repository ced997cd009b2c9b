#!/usr/bin/env python3
"""NAFNet-width64 1080p forward (ms) on one box: default, without the pipelined GEMM kernel of the deep levels (FW_NAF_GEMM=0), without
the fused conv3..conv5 kernel of the width-64 blocks (FW_NAF_FUSE_TAIL=0), without the fused norm1..gate kernel of the 64- / 128-channel
blocks (FW_NAF_FUSE_FRONT=0), and with none of them (the round-1 kernels)."""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
CHILD = r"""
import json, sys, time
sys.path.insert(0, %r)
import torch
from framewright_amd import tap_denoise as T
from framewright_amd.synth import synthetic_frames, synthetic_nafnet_state
f = torch.from_numpy(synthetic_frames(1, 1080, 1920, seed=4)[0]).cuda(); out = torch.empty_like(f)
eng = T.NAFNetEngine(dtype="f16", **T.NAFNET_ARGS); eng.load_state_dict(synthetic_nafnet_state(**T.NAFNET_ARGS))
for _ in range(3): eng.denoise_device(f, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 20
for _ in range(N): eng.denoise_device(f, out=out)
torch.cuda.synchronize(); print(json.dumps({"ms": (time.perf_counter() - t0) / N * 1e3, "checksum": int(out[::7, ::5].to(torch.int64).sum())}))
"""
R1 = {"FW_NAF_GEMM": "0", "FW_NAF_FUSE_TAIL": "0", "FW_NAF_FUSE_FRONT": "0"}
for name, env in (("default", {}), ("no_fused_front", {"FW_NAF_FUSE_FRONT": "0"}), ("no_gemm", {"FW_NAF_GEMM": "0"}), ("no_fused_tail", {"FW_NAF_FUSE_TAIL": "0"}),
                  ("round1", R1), ("default_again", {})):
    r = subprocess.run([sys.executable, "-c", CHILD % str(ROOT)], capture_output=True, text=True, env=dict(os.environ, **env))
    print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:], flush=True)
