#!/usr/bin/env python3
"""A/B of build variants of one kernel file on ONE box: python tools/ab_front.py FILE.hip NAME[=FLAGS] ...
Each variant recompiles csrc/FILE.hip with FLAGS, relinks the library and times a NAFNet 1080p forward in a child process
(FW_AB_CHILD=restormer: a Restormer 512x512 tile instead).  The default library is rebuilt at the end."""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from framewright_amd import build as B

NAF = r"""
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
REST = r"""
import json, sys, time
sys.path.insert(0, %r)
import torch
from framewright_amd import restormer as RS
from framewright_amd.synth import synthetic_frames
f = torch.from_numpy(synthetic_frames(1, 512, 512, seed=4)[0]).cuda(); out = torch.empty_like(f)
eng = RS.RestormerEngine(dtype="f16", **RS.RESTORMER_ARGS); eng.load_state_dict(RS.synthetic_restormer_state(**RS.RESTORMER_ARGS))
for _ in range(3): eng.denoise_device(f, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 20
for _ in range(N): eng.denoise_device(f, out=out)
torch.cuda.synchronize(); print(json.dumps({"ms": (time.perf_counter() - t0) / N * 1e3, "checksum": int(out[::7, ::5].to(torch.int64).sum())}))
"""


def rebuild(name: str, flags: list[str]) -> None:
    cc = B.hipcc()
    B.build()
    subprocess.run([cc, *flags, *B.CXXFLAGS, f"-I{B.INCLUDE}", f"-I{B.CSRC}", "-c", str(B.CSRC / name), "-o", str(B.OBJ_DIR / (Path(name).stem + ".o"))], check=True)
    objs = [str(B.OBJ_DIR / (s.stem + ".o")) for s in B.sources()]
    subprocess.run([cc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-fno-gpu-rdc", *objs, "-o", str(B.LIB_PATH)], check=True)


def main():
    fname = sys.argv[1]
    child = REST if os.environ.get("FW_AB_CHILD") == "restormer" else NAF
    for spec in sys.argv[2:]:
        name, _, flags = spec.partition("=")
        toks = flags.split()
        env = dict(os.environ, **dict(t[4:].split("=", 1) for t in toks if t.startswith("ENV:")))
        rebuild(fname, [t for t in toks if not t.startswith("ENV:")])
        r = subprocess.run([sys.executable, "-c", child % str(ROOT)], capture_output=True, text=True, env=env)
        print(f"{name:24s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]}", flush=True)
    rebuild(fname, [])


if __name__ == "__main__":
    main()
