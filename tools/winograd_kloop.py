#!/usr/bin/env python3
"""The K loop of the Winograd F(2x2, 3x3) trunk kernel sketched in DESIGN.md section 8, measured on static data (tools/diag/
diag_winograd.hip, its own shared object): cycles per 32-channel chunk of an 8 x 32-pixel tile into 64 output channels, with one and
two waves per SIMD, with and without the per-chunk barrier; and the rate that would be in direct-convolution FLOPs."""
import ctypes as C, json, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from framewright_amd import build as B
so = ROOT / "tools" / "diag" / "libdiag_winograd.so"
subprocess.run([B.hipcc(), "-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={B.ARCH}", f"-I{B.INCLUDE}", f"-I{B.CSRC}",
                str(ROOT / "tools" / "diag" / "diag_winograd.hip"), "-o", str(so)], check=True)
lib = C.CDLL(str(so))
lib.fw_debug_winograd_kloop.restype = C.c_int
lib.fw_debug_winograd_kloop.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_ulonglong)]
res = {}
blocks, iters = 256, 20000
direct_flop = 256 * 32 * 64 * 9 * 2          # what one chunk is worth as a direct 3x3 convolution
for waves in (4, 8):
    for sync in (0, 1):
        ms, clk = C.c_float(), (C.c_ulonglong * 2)()
        assert lib.fw_debug_winograd_kloop(waves, sync, blocks, iters, C.byref(ms), clk) == 0
        res[f"waves{waves}_sync{sync}"] = {"ms": ms.value, "cycles_per_chunk": clk[0] / blocks / iters, "clock_ghz": clk[0] / max(clk[1], 1) * 0.1,
                                          "direct_equivalent_tflops": blocks * iters * direct_flop / (ms.value * 1e-3) / 1e12}
print(json.dumps(res, indent=1))
