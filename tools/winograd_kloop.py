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
lib.fw_debug_winograd_stream.restype = C.c_int
lib.fw_debug_winograd_stream.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_ulonglong)]
# with the data movement: weight slices re-fetched from an L2-resident buffer every chunk, raw tiles from one tile (L2) or a 2-GiB buffer (HBM)
# ("wreg": the weight slices by plain 16-byte loads straight into registers, one chunk ahead, instead of LDS-DMA)
# ("stag": workgroup b starts at weight chunk b % 6, so that the CUs of an XCD do not all ask for the same lines at the same time)
for name, wreg, stag, mib in (("stream_wdma_raw_l2", 0, 0, 0), ("stream_wdma_raw_hbm", 0, 0, 2048), ("stream_wreg_raw_l2", 1, 0, 0), ("stream_wreg_raw_hbm", 1, 0, 2048),
                              ("stream_wdma_stag_raw_hbm", 0, 1, 2048), ("stream_wreg_stag_raw_hbm", 1, 1, 2048)):
    ms, clk = C.c_float(), (C.c_ulonglong * 2)()
    it = 4000
    assert lib.fw_debug_winograd_stream(wreg, stag, blocks, it, mib, C.byref(ms), clk) == 0
    ghz = clk[0] / max(clk[1], 1) * 0.1
    res[name] = {"ms": ms.value, "cycles_per_chunk": ms.value * 1e-3 / it * ghz * 1e9, "clock_ghz": ghz,
                 "direct_equivalent_tflops": blocks * it * direct_flop / (ms.value * 1e-3) / 1e12,
                 "l2_to_lds_tb_per_s": blocks * it * (65536 + 22 * 1024) / (ms.value * 1e-3) / 1e12}
# is that rate a limit of the chip or of one CU?  The same kernel on 32 .. 256 workgroups (round-robin over the XCDs: 4 .. 32 CUs each)
for nb in (32, 64, 128, 256):
    ms, clk = C.c_float(), (C.c_ulonglong * 2)()
    it = 4000
    assert lib.fw_debug_winograd_stream(0, 0, nb, it, 2048, C.byref(ms), clk) == 0
    ghz = clk[0] / max(clk[1], 1) * 0.1
    res[f"stream_wdma_raw_hbm_blocks{nb}"] = {"ms": ms.value, "cycles_per_chunk": ms.value * 1e-3 / it * ghz * 1e9, "clock_ghz": ghz,
                                               "gb_per_s_per_cu": it * (65536 + 22 * 1024) / (ms.value * 1e-3) / 1e9}
print(json.dumps(res, indent=1))
