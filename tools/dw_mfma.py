#!/usr/bin/env python3
"""Depthwise 3x3 on the matrix cores (tools/diag/diag_dw_mfma.hip, its own shared object): checks one 64-channel x 16 x 32 tile against
numpy and prints the cycles a workgroup of 8 waves spends per 64-channel chunk - to be set against the ~8200 cycles per chunk and SIMD
(two waves of ~4100) of the VALU depthwise phase in pw_dw_fused.hip (`tools/front_stamps.py`).  One JSON line."""
import ctypes as C, json, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from framewright_amd import build as B
def load(flags, tag):
    so = ROOT / "tools" / "diag" / f"libdiag_dw_mfma{tag}.so"
    subprocess.run([B.hipcc(), "-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={B.ARCH}", f"-I{B.INCLUDE}", f"-I{B.CSRC}", *flags,
                    str(ROOT / "tools" / "diag" / "diag_dw_mfma.hip"), "-o", str(so)], check=True)
    l = C.CDLL(str(so))
    l.fw_debug_dw_mfma.restype = C.c_int
    l.fw_debug_dw_mfma.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
    return l
lib = load([], "")
rng = np.random.default_rng(3)
NC = 32   # channels per launch tile (half a 64-channel chunk)
y = rng.standard_normal((NC, 18, 56)).astype(np.float16)
w = (rng.standard_normal((NC, 9)) / 3).astype(np.float32)
out = np.zeros((NC, 16, 32), np.float16)
cyc = C.c_double()
assert lib.fw_debug_dw_mfma(y.ctypes.data, w.ctypes.data, 1, 1, out.ctypes.data, C.byref(cyc)) == 0
wf = w.astype(np.float16).astype(np.float32).reshape(NC, 3, 3)
yf = y.astype(np.float32)
want = np.zeros((NC, 16, 32), np.float32)
for dy in range(3):
    for d in range(3):
        want += wf[:, dy, d][:, None, None] * yf[:, dy:dy + 16, d:d + 32]
err = float(np.abs(out.astype(np.float32) - want).max())
res = {"max_abs_err_vs_numpy": err, "ok": bool(err < 2e-2)}
for blocks in (1, 256):
    assert lib.fw_debug_dw_mfma(y.ctypes.data, w.ctypes.data, blocks, 2000, out.ctypes.data, C.byref(cyc)) == 0
    res[f"cycles_per_64_channels_blocks{blocks}"] = 2 * cyc.value
# where the time goes (timing-only builds: wrong results)
for tag, flags in (("_noread", ["-DFW_DWM_NOREAD"]), ("_nowrite", ["-DFW_DWM_NOWRITE"]), ("_neither", ["-DFW_DWM_NOREAD", "-DFW_DWM_NOWRITE"])):
    l2 = load(flags, tag)
    assert l2.fw_debug_dw_mfma(y.ctypes.data, w.ctypes.data, 256, 2000, out.ctypes.data, C.byref(cyc)) == 0
    res[f"cycles_per_64_channels{tag}"] = 2 * cyc.value
print(json.dumps(res))
