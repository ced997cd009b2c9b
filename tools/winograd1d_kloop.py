#!/usr/bin/env python3
"""The K loop of a ROW-WISE Winograd F(2, 3) trunk item (tools/diag/diag_winograd1d.hip, its own shared object) against the direct item
of the product kernels on the same harness - static data in LDS, no DMA, no epilogue: correctness of the formulation on one 16 x 32 tile
(against numpy), cycles per 32-channel item with 8 waves on every CU, with and without the per-item barrier.  One JSON object."""
import ctypes as C, json, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from framewright_amd import build as B
so = ROOT / "tools" / "diag" / "libdiag_winograd1d.so"
subprocess.run([B.hipcc(), "-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={B.ARCH}", f"-I{B.INCLUDE}", f"-I{B.CSRC}",
                str(ROOT / "tools" / "diag" / "diag_winograd1d.hip"), "-o", str(so)], check=True)
lib = C.CDLL(str(so))
lib.fw_debug_winograd1d.restype = C.c_int
lib.fw_debug_winograd1d.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_ulonglong)]
rng = np.random.default_rng(7)
x = rng.standard_normal((18, 34, 32)).astype(np.float16)
w = (rng.standard_normal((64, 32, 3, 3)) / 17).astype(np.float32)
xf, wf = x.astype(np.float64), w.astype(np.float16).astype(np.float64)
want = np.zeros((16, 32, 64))
for dy in range(3):
    for dx in range(3):
        want += np.einsum("hwc,oc->hwo", xf[dy:dy + 16, dx:dx + 32], wf[:, :, dy, dx])
res = {}
for mode, name in ((0, "winograd_rows"), (1, "direct")):
    y = np.zeros((16, 32, 64), np.float32)
    ms, clk = C.c_float(), (C.c_ulonglong * 2)()
    assert lib.fw_debug_winograd1d(mode, x.ctypes.data, w.ctypes.data, 1, 1, 1, y.ctypes.data, C.byref(ms), clk) == 0
    res[name] = {"max_abs_err_vs_numpy": float(np.abs(y - want).max()), "output_max_abs": float(np.abs(want).max())}
    blocks, iters = 256, 4000
    for sync in (0, 1):
        assert lib.fw_debug_winograd1d(mode, x.ctypes.data, w.ctypes.data, blocks, iters, sync, None, C.byref(ms), clk) == 0
        ghz = clk[0] / max(clk[1], 1) * 0.1
        res[name][f"sync{sync}"] = {"ms": ms.value, "cycles_per_item": clk[0] / blocks / iters, "clock_ghz": ghz,
                                    "direct_equivalent_tflops": blocks * iters * (16 * 32 * 64 * 32 * 9 * 2) / (ms.value * 1e-3) / 1e12}
print(json.dumps(res, indent=1))
