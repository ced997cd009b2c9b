#!/usr/bin/env python3
"""What the matrix cores sustain on random bf16 data with nothing else in the way (tools/diag/diag_mfma.hip, built here into its
own shared object - it is a microbenchmark, not part of libframewright_hip.so): TFLOP/s and the clock the chip holds, for both
MFMA shapes, 2 waves per SIMD on every CU.  Context for roofline.frac, which is quoted against the 2.5 PFLOP/s spec figure."""
import ctypes as C, json, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from framewright_amd import build as B
so = ROOT / "tools" / "diag" / "libdiag_mfma.so"
subprocess.run([B.hipcc(), "-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={B.ARCH}", f"-I{B.INCLUDE}", f"-I{B.CSRC}",
                str(ROOT / "tools" / "diag" / "diag_mfma.hip"), "-o", str(so)], check=True)
lib = C.CDLL(str(so))
lib.fw_debug_mfma_peak.restype = C.c_int
lib.fw_debug_mfma_peak.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_ulonglong)]
res = {}
import os
BL = [int(x) for x in os.environ.get("FW_PEAK_BLOCKS", "256").split(",")]
for blocks in BL:
  for shape, per_iter, flop in ((16, 16, 16 * 16 * 32 * 2), (32, 8, 32 * 32 * 16 * 2)):
    iters = 100000
    ms, clk = C.c_float(), (C.c_ulonglong * 2)()
    assert lib.fw_debug_mfma_peak(shape, 0, 0, blocks, iters, C.byref(ms), clk) == 0
    total = blocks * 8 * iters * per_iter * flop
    res[f"mfma_{shape}_b{blocks}"] = {"ms": ms.value, "tflops": total / (ms.value * 1e-3) / 1e12,
                            "clock_ghz": clk[0] / max(clk[1], 1) * 0.1,
                            # wave 0 of a workgroup is the OLDER wave of its SIMD: it wins every MFMA arbitration, runs at the
                            # full rate and retires after about half of the launch (its partner then runs alone)
                            "older_wave_ms": clk[1] / blocks / 1e5,
                            "older_wave_cycles_per_mfma": clk[0] / blocks / (iters * per_iter)}
# 16x16x32 on all CUs: the order the fragments are walked in, and all-zero operands
for name, order, zeros in (("a_stays", 1, 0), ("both_change", 2, 0), ("one_pair", 3, 0), ("zeros", 0, 1)):
    iters = 100000
    ms, clk = C.c_float(), (C.c_ulonglong * 2)()
    assert lib.fw_debug_mfma_peak(16, order, zeros, 256, iters, C.byref(ms), clk) == 0
    res[f"mfma_16_b256_{name}"] = {"ms": ms.value, "tflops": 256 * 8 * iters * 16 * (16 * 16 * 32 * 2) / (ms.value * 1e-3) / 1e12,
                                   "clock_ghz": clk[0] / max(clk[1], 1) * 0.1}
print(json.dumps(res))
