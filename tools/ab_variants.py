#!/usr/bin/env python3
"""A/B of conv-kernel variants on ONE box in ONE call (boxes differ by +-4 % on this path, so only same-box numbers compare).

  python tools/ab_variants.py NAME[:SRCDIR][=FLAGS] ...

Each variant recompiles csrc/conv3x3_mfma.hip and csrc/conv3x3_pair.hip (from SRCDIR when given: a directory holding
alternative copies of csrc/, e.g. the previous commit's, exported under build/) with FW_EXTRA flags FLAGS, relinks the
library with the other objects of the in-tree build, and times Real-ESRGAN x4 1080p frames in a child process.  The default
library is rebuilt at the end.  Prints one line per variant: ms/frame, conv TFLOP/s, per-kernel-class averages.
"""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from framewright_amd import build as B

CHILD = r"""
import json, os, sys, time
sys.path.insert(0, %r)
import torch
from framewright_amd.realesrgan import RRDBNetEngine
from framewright_amd.synth import synthetic_frames, synthetic_rrdbnet_state
eng = RRDBNetEngine(23, 4, os.environ.get("FW_AB_DTYPE", "f16")); eng.load_state_dict(synthetic_rrdbnet_state(23, 4))
d = torch.from_numpy(synthetic_frames(1, 1080, 1920, seed=2)[0]).cuda(); o = torch.empty((4320, 7680, 3), dtype=torch.uint8, device="cuda")
for _ in range(3): eng.upscale_device(d, out=o)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 8
for _ in range(N): eng.upscale_device(d, out=o)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / N * 1e3
eng.profile_enable(True)
for _ in range(2): eng.upscale_device(d, out=o)
launches, conv_ms, conv_flops = eng.profile_read()
up = {}
if os.environ.get("FW_AB_TIME_UP"):
    import ctypes as C, numpy as np
    from framewright_amd import _lib
    lib = _lib.load()
    w = (np.random.default_rng(0).standard_normal((64, 64, 3, 3)) / 24).astype(np.float32)
    n = lib.fw_pack_conv_up2x_phase(1, None, None); pk = np.zeros(n, np.uint16)
    lib.fw_pack_conv_up2x_phase(1, C.c_void_p(w.ctypes.data), C.c_void_p(pk.ctypes.data))
    wp = torch.from_numpy(pk.view(np.int16)).cuda(); bias = torch.zeros(64, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    for nm, (h, w_) in {"up1": (1080, 1920), "up2": (2160, 3840)}.items():
        x = torch.randn((2, h, w_, 32), device="cuda").half(); out = torch.empty((2, 2 * h, 2 * w_, 32), dtype=torch.float16, device="cuda")
        run = lambda: _lib.check(lib.fw_conv_up2x_phase_nhwc(1, p(x), 32, h * w_ * 32, h, w_, p(wp), p(bias), 1, p(out), 32, 4 * h * w_ * 32, None))
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        up[nm + "_ms"] = e0.elapsed_time(e1) / 20
print(json.dumps({"ms_per_frame": ms, "conv_tflops": conv_flops / conv_ms / 1e9, "checksum": int(o[::97, ::89].to(torch.int64).sum()), **up}))
"""


def rebuild(srcdir: Path, flags: list[str]) -> None:
    cc = B.hipcc()
    B.build()  # the other objects
    for name in ("conv3x3_mfma", "conv3x3_pair", "conv3x3_pair_slide", "conv3x3_pair_slide32", "conv_up2x_phase", "conv3x3_wino"):
        src = srcdir / f"{name}.hip"
        if not src.exists():
            src = B.CSRC / f"{name}.hip"
        cmd = [cc, *flags, *B.CXXFLAGS, f"-I{B.INCLUDE}", f"-I{srcdir}", f"-I{B.CSRC}", "-c", str(src), "-o", str(B.OBJ_DIR / f"{name}.o")]
        subprocess.run(cmd, check=True)
    objs = [str(B.OBJ_DIR / (s.stem + ".o")) for s in B.sources()]
    subprocess.run([cc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-fno-gpu-rdc", *objs, "-o", str(B.LIB_PATH)], check=True)


def main():
    for spec in sys.argv[1:]:
        name, _, flags = spec.partition("=")
        name, _, sd = name.partition(":")
        srcdir = (ROOT / sd) if sd else B.CSRC
        toks = flags.split()
        env = dict(os.environ, **dict(t[4:].split("=", 1) for t in toks if t.startswith("ENV:")))  # ENV:NAME=VALUE tokens
        rebuild(srcdir, [t for t in toks if not t.startswith("ENV:")])
        r = subprocess.run([sys.executable, "-c", CHILD % str(ROOT)], capture_output=True, text=True, env=env)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
        print(f"{name:24s} {line}", flush=True)
    rebuild(B.CSRC, [])


if __name__ == "__main__":
    main()
