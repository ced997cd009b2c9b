#!/usr/bin/env python3
"""What a library GEMM sustains on this chip with real data (the practical MFMA ceiling the conv kernels should be priced against):
torch.matmul (hipBLASLt / rocBLAS) on f16 / bf16 square matrices, random normal operands and all-zero operands, with the socket power and
the clock rocm-smi reports half-way through.  Prints one JSON line."""
import json, re, subprocess, sys, time
import torch
def smi():
    try:
        t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        p = re.search(r"Package Power \(W\): ([\d.]+)", t); s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", t)
        return (float(p.group(1)) if p else None, int(s.group(1)) if s else None)
    except Exception:
        return (None, None)
res = {}
for dt_name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
    for n in (8192, 16384):
        for data in ("normal", "zeros"):
            a = torch.randn(n, n, device="cuda").to(dt) if data == "normal" else torch.zeros(n, n, device="cuda", dtype=dt)
            b = torch.randn(n, n, device="cuda").to(dt) if data == "normal" else torch.zeros(n, n, device="cuda", dtype=dt)
            c = torch.empty(n, n, device="cuda", dtype=dt)
            for _ in range(3): torch.matmul(a, b, out=c)
            torch.cuda.synchronize()
            iters = 200 if n == 8192 else 30
            t0 = time.perf_counter()
            for i in range(iters):
                torch.matmul(a, b, out=c)
            mid = smi()                       # the queue is still full of GEMMs while rocm-smi samples
            torch.cuda.synchronize()
            dt_s = time.perf_counter() - t0
            res[f"{dt_name}_{n}_{data}"] = {"tflops": 2.0 * n ** 3 * iters / dt_s / 1e12, "power_w": mid[0], "sclk_mhz": mid[1]}
            del a, b, c
print(json.dumps(res))
