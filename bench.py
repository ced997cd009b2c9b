#!/usr/bin/env python3
"""Benchmarks of the hot path on synthetic 1080p clips (BASELINE.json configs[1..4]); the default is the headline:
frames/sec of Real-ESRGAN x4 (RRDBNet-x4plus, 23 blocks), 1920x1080 -> 7680x4320.

  python bench.py [--gpus N --steps K --warmup W] [--config sr|rife|tap|chain]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one INPUT frame through the configured path, uint8 BGR in HBM -> uint8 BGR in HBM (`value`; PCIe is never in
it).  Rank 0 prints ONE JSON line.

  sr     configs[1]  Real-ESRGAN x4plus; frames are independent: rank r takes frames r, r+N, ... (no data-path collective)
  rife   configs[2]  IFNet v4.6 x2: a step = one pair -> its mid frame; block partition, one INPUT frame from rank r+1
  tap    configs[3]  NAFNet-width64 temporal denoise, window 5, whole frame: every frame denoised once by its owner, the two
                     edge frames of each block sent to the neighbours as CUDA tensors (ncclSend / ncclRecv), then the window
                     average locally (sharding.sharded_tap_denoise_device)
  chain  configs[4]  temporal denoise -> Real-ESRGAN x4 -> RIFE x2 per block of frames, device-resident hand-off

Extra objects on the line: `roofline` (measured live with HIP events on the launch stream; MFMA-bound convs for sr / chain, HBM
byte model of the engine's own dataflow for tap / rife), `cpu_baseline` (the fp32 CPU oracle on a bounded crop, rank 0 at N=1),
`parity` (the engine against that same oracle output: max-abs on the [0,1] float image before quantisation, PSNR on uint8),
and for sr at N=1 `host_to_host` (pinned host frame in, host frame out through the 3-stream pipeline: PCIe-inclusive).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import platform
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_TFLOPS = 2500.0   # dense bf16 / f16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0  # HBM3E spec (≈6300 GB/s achievable per the same guide)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="frames per GPU (default: 300 for sr, 100 for rife / tap, 24 for chain)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="sr", choices=["sr", "rife", "tap", "chain"])
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--model", default="RealESRGAN_x4plus")
    ap.add_argument("--dtype", default="f16", choices=["bf16", "f16"],
                    help="MFMA operand type; f16 is what the reference's own GPU path runs (half=True) and what meets the 1e-3 bar")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-crop", type=int, default=192, help="side of the square crop timed on the CPU oracle (also the parity crop)")
    ap.add_argument("--no-host-path", action="store_true")
    return ap.parse_args()


# ---- small helpers ----------------------------------------------------------------------------------------------------------
def cpu_info():
    """(model string, physical cores, threads torch uses)."""
    import torch
    model, phys = platform.processor() or "unknown", None
    try:
        cores = set()
        cur = {}
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if ":" in line:
                k, v = [s.strip() for s in line.split(":", 1)]
                cur[k] = v
                if k == "model name":
                    model = v
            elif not line.strip() and cur:
                cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor", "0"))))
                cur = {}
        if cur:
            cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor", "0"))))
        phys = len(cores) or None
    except OSError:
        pass
    return model, phys or os.cpu_count(), torch.get_num_threads()


def psnr_u8(a, b):
    import numpy as np
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 99.0 if mse == 0 else 10 * math.log10(255.0 ** 2 / mse)


def lib_digest():
    from framewright_amd import build as B
    return B.source_digest()


def nafnet_design_bytes(H, W, width=64, enc=(2, 2, 4, 8), mid=12, dec=(2, 2, 2, 2)):
    """HBM bytes one NAFNet forward moves in THIS engine's dataflow, every tensor counted once per kernel that reads or writes
    it.  Per NAFBlock and pixel, c channels: 16c at c = 64 and 128 (two kernels: the fused front reads the
    fp32 stream and writes the gated tensor, the fused tail reads both and writes the stream), 48c from 256 channels up (seven passes: fp32 stream read twice and written
    twice, five typed tensors - the round-1 figure for every level); plus the 2x2 down convs, the 1x1 + PixelShuffle ups, intro
    and ending."""
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    px, c, total = Hp * Wp, width, 0.0
    per_block = lambda ch: 16 if ch <= 128 else 48
    total += px * (2 * 32 + 4 * c)                       # intro: typed frame in, fp32 stream out
    for n in enc:
        total += px * per_block(c) * c * n
        total += px * 4 * c + px // 4 * 4 * 2 * c         # down: read fp32 stream, write the next level's
        px, c = px // 4, 2 * c
    total += px * per_block(c) * c * mid
    for n in dec:
        total += px * 4 * c + 4 * px * (4 * c // 2) * 2   # up: read stream, write + read skip, write stream
        px, c = 4 * px, c // 2
        total += px * per_block(c) * c * n
    total += px * (2 * c + 2 * c + 12) + H * W * 6        # ending: planar copy, conv, image in / out
    return total


def ifnet_design_bytes(H, W):
    """HBM bytes of one IFNet v4.6 forward in this engine's dataflow: the full-resolution fp32 maps (two images, flow, mask, the
    8-channel block input) dominate; the convolutions run at 1/32 ... 1/4 resolution."""
    Hp, Wp = (H + 31) // 32 * 32, (W + 31) // 32 * 32
    px, total = Hp * Wp, 0.0
    total += 2 * (H * W * 3 + px * 12)                    # uint8 -> fp32 RGB, twice
    for i, (c, s) in enumerate(zip((192, 128, 96, 64), (8, 4, 2, 1))):
        cin = 7 if i == 0 else 12
        total += px * (24 + (0 if i == 0 else 2 * 4 * 12 + 20) + 32)   # build_x: images (+ 4-tap warps, flow, mask) -> X
        ps = px // (s * s)
        total += px * 32 / (1 if s == 1 else 4) + ps * 4 * cin          # resize X (and flow) down to the block's scale
        total += ps * (4 * cin + 2 * 4 * cin) / 1                       # unshuffle + cast
        pf = ps // 16
        cp = (c + 63) // 64 * 64
        total += ps // 4 * (2 * 4 * cin + 2 * c // 2) + pf * (2 * 2 * c + 2 * cp + 4 * cp)   # conv0
        total += 8 * pf * (2 * cp + 4 * cp + 2 * cp + 4 * cp)                                 # 8 ResConvs: typed + fp32 in and out
        total += pf * (2 * cp + 4 * 96) + ps * (4 * 6 + 4 * 6) + px * (16 + 20)               # lastconv, depth-to-space, accumulate
    total += px * (24 + 16 + 4 + 8 * 4 * 3) / 1 + H * W * 3                                    # blend: two 4-tap warps, uint8 out
    return total


class PowerSampler:
    """Socket power and sclk of one GPU while the timed region runs: `rocm-smi` as a child process every half second (it does not touch
    this process's HIP state).  stop() -> {"power_w", "sclk_mhz", "samples"} (medians over the samples within 10 % of the highest power:
    the region under load), or None when rocm-smi is missing or the run was too short for two samples."""

    def __init__(self, device: int):
        import shutil
        import threading
        self._exe = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
        self._dev, self._rows, self._stop = device, [], threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)
        if self._exe:
            self._th.start()

    def _run(self):
        import re
        import subprocess
        while not self._stop.is_set():
            try:
                t = subprocess.run([self._exe, "-d", str(self._dev), "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                pw = re.search(r"Power \(W\): ([\d.]+)", t)
                sc = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", t)
                if pw:
                    self._rows.append((float(pw.group(1)), int(sc.group(1)) if sc else None))
            except Exception:  # noqa: BLE001 - a sampler must never take the bench down
                return
            self._stop.wait(0.5)

    def stop(self):
        import statistics as st
        self._stop.set()
        if self._exe:
            self._th.join(timeout=15)
        if len(self._rows) < 2:
            return None
        top = max(p for p, _ in self._rows)
        load = [(p, c) for p, c in self._rows if p > 0.9 * top]
        clk = [c for _, c in load if c]
        return {"power_w": st.median(p for p, _ in load), "sclk_mhz": st.median(clk) if clk else None, "samples": len(load)}


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # one rank per GPU, started by torch.distributed.run: `--gpus N` alone must not silently measure one GPU and report N
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch as `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...`")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # rehearsal of the multi-rank path on a 1-GPU box only: FW_BENCH_FORCE_DEVICE=0 puts every rank on one card
    dev_ord = int(os.environ.get("FW_BENCH_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(dev_ord)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one rank per GPU over RCCL; the single-card rehearsal (FW_BENCH_FORCE_DEVICE) uses gloo unless told otherwise
        backend = os.environ.get("FW_BENCH_BACKEND", "gloo" if "FW_BENCH_FORCE_DEVICE" in os.environ else "nccl")
        dist.init_process_group(backend)
    on_nccl = world > 1 and dist.get_backend() == "nccl"

    from framewright_amd import build as fw_build
    if rank == 0:
        fw_build.build()

    def barrier():
        if world > 1:
            if on_nccl:
                dist.barrier(device_ids=[dev_ord])
            else:
                dist.barrier()

    barrier()
    from framewright_amd import rife as RF
    from framewright_amd import sharding as S
    from framewright_amd import tap_denoise as T
    from framewright_amd.realesrgan import RRDBNetEngine
    from framewright_amd.synth import (RRDB_MODELS, synthetic_frames, synthetic_ifnet_state, synthetic_nafnet_state,
                                       synthetic_rrdbnet_state)

    cfg = args.config
    if cfg == "chain":
        # BASELINE configs[4]: "hipGraph-captured per-frame stages" - each stage's forward is captured once per (frame size,
        # buffers) and replayed (identical frames: tests/test_fullsize_gpu.py); FW_*_GRAPH=0 in the environment turns it off
        for k in ("FW_NAF_GRAPH", "FW_RRDB_GRAPH", "FW_IFNET_GRAPH"):
            os.environ.setdefault(k, "1")
    steps = args.steps if args.steps is not None else {"sr": 300, "rife": 100, "tap": 100, "chain": 24}[cfg]
    if cfg in ("tap", "chain") and world > 1:
        steps = max(steps, 2)      # a block holds at least window // 2 frames (sharding.block_partition)
    H, W = args.height, args.width
    dev = torch.device("cuda", dev_ord)
    halo_dev = dev if (world == 1 or on_nccl) else torch.device("cpu")
    num_block, scale = RRDB_MODELS[args.model]

    # ---- engines ----------------------------------------------------------------------------------------------------------
    sr = ifn = tap = naf = None
    sd_sr = sd_if = sd_naf = None
    if cfg in ("sr", "chain"):
        sd_sr = synthetic_rrdbnet_state(num_block, scale, seed=1234)
        sr = RRDBNetEngine(num_block, scale, args.dtype, device_id=dev_ord)
        sr.load_state_dict(sd_sr)
    if cfg in ("rife", "chain"):
        sd_if = synthetic_ifnet_state()
        ifn = RF.IFNetEngine(args.dtype, dev_ord)
        ifn.load_state_dict(sd_if)
    if cfg in ("tap", "chain"):
        sd_naf = synthetic_nafnet_state(**T.NAFNET_ARGS)
        naf = T.NAFNetEngine(dtype=args.dtype, device_id=dev_ord, **T.NAFNET_ARGS)
        naf.load_state_dict(sd_naf)
        tap = T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet", tile_size=0, temporal_window=5, gpu_id=dev_ord, dtype=args.dtype),
                            engine=naf)

    # ---- this rank's frames: a few distinct ones resident in HBM, cycled ------------------------------------------------------
    seed = {"sr": 2, "rife": 3, "tap": 4, "chain": 5}[cfg]
    n_distinct = max(2, min(4, steps))
    clip = synthetic_frames(n_distinct, H, W, seed=seed + 100 * rank)
    res_frames = [torch.from_numpy(np.ascontiguousarray(f)).to(dev) for f in clip]
    frame_of = lambda i: res_frames[i % n_distinct]

    # ---- one pass over this rank's `k` frames ----------------------------------------------------------------------------------
    out_sr = torch.empty((H * scale, W * scale, 3), dtype=torch.uint8, device=dev) if sr is not None else None
    out_if = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)

    def global_list(block):
        """This rank's block placed at its position in a clip of world * len(block) frames (other entries None): what the
        sharded helpers take."""
        k = len(block)
        full = [None] * (world * k)
        full[rank * k:(rank + 1) * k] = block
        return full

    def run_pass(k):
        if cfg == "sr":
            for i in range(k):
                sr.upscale_device(frame_of(i), out=out_sr)
        elif cfg == "rife":
            # k pairs = k + 1 frames of which this rank owns k; the halo (the next rank's first frame) crosses the link once
            frames = global_list([frame_of(i) for i in range(k)])
            if world == 1:
                frames.append(frame_of(k))
            else:
                frames.append(frame_of(k) if rank == world - 1 else None)
            S.sharded_interpolate_device(ifn, frames, halo_device=halo_dev)
        elif cfg == "tap":
            S.sharded_tap_denoise_device(tap, global_list([frame_of(i) for i in range(k)]), halo_device=halo_dev)
        else:
            den = S.sharded_tap_denoise_device(tap, global_list([frame_of(i) for i in range(k)]), halo_device=halo_dev)
            ups = [None] * (world * k)
            for i, t in den.items():
                ups[i] = sr.upscale_device(t)
            S.sharded_pairs(ups, lambda a, b: ifn.interpolate_device(a.to(dev), b.to(dev), 0.5), device=halo_dev)

    chunk = {"sr": steps, "rife": steps, "tap": min(steps, 20), "chain": min(steps, 4)}[cfg]   # frames resident per pass

    def run_steps(n):
        done = 0
        while done < n:
            k = min(chunk, n - done)
            if cfg in ("tap", "chain") and world > 1 and n - done - k == 1:
                k += 1            # never leave a one-frame block behind
            run_pass(k)
            done += k

    warm = max(1, args.warmup)
    if cfg in ("tap", "chain") and world > 1:
        warm = max(warm, 2)        # a block holds at least window // 2 frames, in the warm-up pass too
    if cfg != "sr":
        # these passes hand back one output tensor per frame: the warm-up has to be one whole pass, or the first timed pass pays the
        # hipMallocs behind torch's caching allocator (620 MB of mid frames on the rife line: 1.24 against 0.83 ms per pair -
        # profiles/r03_ab/rife_output_buffers.txt) - a cost a clip pays once, not per pass.  `sr` writes into one buffer: W as given.
        warm = max(warm, min(chunk, steps))
    run_steps(warm)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sampler = PowerSampler(dev_ord) if rank == 0 else None       # rocm-smi in a child process, twice a second: joules per step
    t0 = time.perf_counter()
    ev0.record()
    run_steps(steps)
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    power = sampler.stop() if sampler is not None else None
    t = torch.tensor([wall], dtype=torch.float64, device=dev if (world == 1 or on_nccl) else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())
    dev_ms = ev0.elapsed_time(ev1)

    # ---- roofline --------------------------------------------------------------------------------------------------------
    roof = None
    if rank == 0:
        if cfg in ("sr", "chain"):
            # HIP events around every conv launch of the Real-ESRGAN forward, on the launch stream (the MFMA implicit-GEMM
            # kernels are > 99 % of its FLOPs)
            sr.profile_enable(True)
            for i in range(2):
                sr.upscale_device(frame_of(i), out=out_sr)
            launches, conv_ms, conv_flops = sr.profile_read()
            sr.profile_enable(False)
            achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
            traffic, note = None, None
            tf = newest_profile("traffic.json")
            if tf is not None and (H, W, args.model) == (1080, 1920, "RealESRGAN_x4plus"):
                tj = json.loads(tf.read_text())
                if tj.get("lib_digest") == lib_digest() and tj.get("dtype") == args.dtype:
                    traffic = tj["hbm_bytes_per_launch"]
                    note = f"profiles/{tf.name}, measured on this build ({tj['lib_digest']}): {tj['hbm_gb_per_frame']:.0f} GB per frame"
                else:
                    note = f"profiles/{tf.name} was measured on another build or dtype: not quoted"
            roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS,
                    "traffic": traffic, "traffic_source": note,
                    "kernel": "conv3x3_pair_slide_kernel + conv3x3_wino_split_kernel + conv3x3_mfma_kernel + conv_up2x_phase_kernel (every conv launch of the Real-ESRGAN forward; the pair kernel is 53 % of the frame)",
                    "launches_timed": launches, "avg_launch_ms": conv_ms / max(launches, 1),
                    "avg_launch_gflop": conv_flops / max(launches, 1) / 1e9}
        else:
            # HBM byte model of the engine's own dataflow per forward (functions above), divided by the
            # forward's duration measured with HIP events on the launch stream
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            o = torch.empty_like(res_frames[0])
            fwd = (lambda: naf.denoise_device(res_frames[0], out=o)) if cfg == "tap" else \
                  (lambda: ifn.interpolate_device(res_frames[0], res_frames[1], 0.5, out=o))
            fwd()
            e0.record()
            for _ in range(reps):
                fwd()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            nbytes = nafnet_design_bytes(H, W) if cfg == "tap" else ifnet_design_bytes(H, W)
            gbs = nbytes / (ms * 1e-3) / 1e9
            traffic, note = None, None
            tf = newest_profile("traffic_tap.json" if cfg == "tap" else "traffic_rife.json")
            if tf is not None and (H, W) == (1080, 1920):
                tj = json.loads(tf.read_text())
                if tj.get("lib_digest") == lib_digest() and tj.get("dtype") == args.dtype:
                    traffic = tj["hbm_bytes_per_forward"]
                    note = f"profiles/{tf.name}, PMC counters of this build ({tj['lib_digest']}): {traffic / 1e9:.1f} GB per forward"
                else:
                    note = f"profiles/{tf.name} was measured on another build or dtype: not quoted"
            roof = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": traffic,
                    "traffic_source": note,
                    "kernel": "one NAFNet-width64 forward (all kernels)" if cfg == "tap" else "one IFNet v4.6 forward (all kernels)",
                    "forward_ms": ms, "bytes_per_forward": nbytes,
                    "flops_per_forward": (naf.flops(H, W) if cfg == "tap" else ifn.flops(H, W))}

    # ---- parity + CPU baseline: the engine and the fp32 oracle on one crop (outside every timed region) --------------------------
    parity = cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        parity, cpu = parity_and_cpu(cfg, args, clip, sr, ifn, naf, sd_sr, sd_if, sd_naf, num_block, scale, H * W)

    # ---- PCIe-inclusive rate (sr, one GPU): pinned host frames through the 3-stream pipeline -----------------------------------
    host = None
    if rank == 0 and world == 1 and cfg == "sr" and not args.no_host_path:
        n_h = min(steps, 30)
        for _ in sr.upscale_stream([clip[i % n_distinct] for i in range(3)], depth=2):   # pins the staging slots (tens of ms, once)
            pass
        torch.cuda.synchronize()
        # The clock covers the WHOLE stream: first frame handed over -> last frame back in host memory, pipeline fill and drain
        # included, and every one of the n_h frames is computed inside it - so this figure can never exceed the resident rate
        # (round 2 started the clock after two frames had been computed and still counted them).
        th = time.perf_counter()
        got = 0
        for _ in sr.upscale_stream((clip[i % n_distinct] for i in range(n_h)), depth=2):
            got += 1
        torch.cuda.synchronize()
        dt_h = time.perf_counter() - th
        assert got == n_h
        host = {"value": n_h / dt_h, "unit": "frames/s", "ms_per_frame": dt_h / n_h * 1e3, "frames": n_h,
                "what": "pinned host uint8 frame in -> host uint8 frame out (6.2 MB up, 99.5 MB down per frame), upload / compute / "
                        "download on three streams (RRDBNetEngine.upscale_stream); clock: first frame in -> last frame out, fill and "
                        "drain of the pipeline included"}

    if rank == 0:
        flops_frame = sr.flops(H, W) if sr is not None else None
        fps = world * steps / wall_max
        workloads = {
            "sr": f"{args.model} x{scale}, {W}x{H} uint8 BGR frames resident in HBM -> {W * scale}x{H * scale} uint8 BGR in HBM",
            "rife": f"IFNet v4.6 x2, {W}x{H} pairs resident in HBM -> mid frame (timestep 0.5), 2-frame window",
            "tap": f"NAFNet-width64 temporal denoise, {W}x{H}, window 5, whole frame, every frame denoised once",
            "chain": f"temporal denoise (NAFNet, window 5) -> {args.model} x{scale} -> RIFE x2 on the {W * scale}x{H * scale} frames, "
                     f"{W}x{H} input frames resident in HBM, device-resident hand-off",
        }
        sharding = {"sr": f"round-robin frames over {world} rank(s), no collective",
                    "rife": f"block partition over {world} rank(s), one input frame from rank r+1 (isend/irecv)",
                    "tap": f"block partition over {world} rank(s), 2 denoised frames each way between neighbours (isend/irecv)",
                    "chain": f"block partition over {world} rank(s): denoised halos, then one upscaled frame from rank r+1"}[cfg]
        metric = {"sr": "frames/sec Real-ESRGAN x4 1080p (RRDBNet-x4plus, 1920x1080 -> 7680x4320)",
                  "rife": "frames/sec RIFE x2 1080p (input pairs per second)",
                  "tap": "frames/sec NAFNet temporal denoise 1080p (window 5)",
                  "chain": "frames/sec preset chain 1080p (input frames per second)"}[cfg]
        res = {
            "metric": metric, "value": fps, "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warm,
            "ms_per_step": wall_max / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workloads[cfg] + f", seeded synthetic weights, {steps} frames per GPU", "sharding": sharding},
            "device_ms_per_step": dev_ms / steps, "roofline": roof, "lib_digest": lib_digest(),
        }
        if flops_frame is not None:
            res["config"]["frame_tflop"] = flops_frame / 1e12
            if cfg == "sr":
                res["whole_path_tflops_per_gpu"] = flops_frame * steps / (dev_ms * 1e-3) / 1e12
        if power is not None:
            # rank 0's GPU under load during the timed region: the conv kernels run pinned at the board's power limit, so
            # joules per step, not cycles, is what a change has to lower
            res["energy"] = {"power_w": power["power_w"], "sclk_mhz": power["sclk_mhz"], "samples": power["samples"],
                             "joules_per_step_per_gpu": power["power_w"] * wall_max / steps}
            if flops_frame is not None and cfg == "sr":
                res["energy"]["pj_per_algorithmic_flop"] = power["power_w"] * (wall_max / steps) / flops_frame * 1e12
        if parity is not None:
            res["parity"] = parity
        if cpu is not None:
            res["cpu_baseline"] = cpu
        if host is not None:
            # the PCIe-inclusive figure contains everything the resident one does: it cannot be the faster of the two
            host["consistent_with_value"] = bool(host["value"] <= fps * 1.01)
            if not host["consistent_with_value"]:
                print(f"bench: host_to_host {host['value']:.3f} frames/s exceeds the resident rate {fps:.3f}: mis-timed", file=sys.stderr)
            res["host_to_host"] = host
            res["value_is"] = ("frames resident in HBM at the start of the timed region (the bench contract); host_to_host is the "
                               "PCIe-inclusive rate of the same path")
        print(json.dumps(res), flush=True)
    for e in (sr, ifn, naf):
        if e is not None:
            e.close()
    if world > 1:
        barrier()
        dist.destroy_process_group()


def newest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that has one (the measured-traffic files carry the digest of the build they were taken
    on: a line only quotes the file when the digests agree)."""
    found = sorted((ROOT / "profiles").glob(f"r[0-9][0-9]_{suffix}"))
    return found[-1] if found else None


def parity_and_cpu(cfg, args, clip, sr, ifn, naf, sd_sr, sd_if, sd_naf, num_block, scale, frame_px):
    """The fp32 CPU oracle on a bounded crop of frame 0, timed (cpu_baseline), and the engine on the same crop compared with
    it (parity).  chain: the three oracles on the crop in sequence."""
    import numpy as np
    import torch
    from oracle import ifnet_ref, nafnet_ref
    from oracle import rrdbnet_ref as ref

    model, phys, threads = cpu_info()
    c = args.cpu_crop if cfg != "chain" else min(args.cpu_crop, 96)
    a, b = np.ascontiguousarray(clip[0][:c, :c]), np.ascontiguousarray(clip[1][:c, :c])
    to_t = lambda f: torch.from_numpy(np.ascontiguousarray(f[:, :, ::-1]).astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    sdt = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}
    dev = torch.device("cuda", torch.cuda.current_device())
    up = lambda f: torch.from_numpy(f).to(dev)
    t_cpu, what = 0.0, []
    with torch.no_grad():
        ref.rrdbnet_forward(sdt(synth_small()), torch.zeros(1, 3, 16, 16), 1, 4)     # warm the thread pool
        if cfg in ("tap", "chain"):
            t0 = time.perf_counter()
            from framewright_amd.tap_denoise import NAFNET_ARGS
            y = nafnet_ref.nafnet_forward(sdt(sd_naf), to_t(a), NAFNET_ARGS["middle_blk_num"], NAFNET_ARGS["enc_blk_nums"],
                                          NAFNET_ARGS["dec_blk_nums"])
            t_cpu += time.perf_counter() - t0
            want_tap = y.squeeze(0).permute(1, 2, 0).numpy()
            what.append("NAFNet-width64 forward")
        if cfg in ("sr", "chain"):
            t0 = time.perf_counter()
            y = ref.rrdbnet_forward(sdt(sd_sr), to_t(a), num_block, scale)
            t_cpu += time.perf_counter() - t0
            want_sr = y.squeeze(0).permute(1, 2, 0).numpy()
            what.append(f"all {num_block} RRDB blocks + tail")
        if cfg in ("rife", "chain"):
            t0 = time.perf_counter()
            y = ifnet_ref.ifnet_forward(sdt(sd_if), to_t(a), to_t(b), 0.5)
            dt_if = time.perf_counter() - t0
            # in the chain the interpolation runs on the upscaled frames: 16 x the pixels, 0.75 pairs per input frame
            t_cpu += dt_if * (scale * scale * 0.75 if cfg == "chain" else 1.0)
            want_if = y.squeeze(0).permute(1, 2, 0).numpy()
            what.append("IFNet v4.6 pair" + (" (scaled to the upscaled frame size)" if cfg == "chain" else ""))
    cpu = {"value": (c * c / frame_px) / t_cpu, "unit": "frames/s", "cores": threads, "physical_cores": phys, "cpu_model": model,
           "kind": "port",
           "sample": f"{c}x{c} crop of frame 0 through {' + '.join(what)} in {t_cpu:.1f} s on {threads} torch threads, scaled by pixel "
                     f"count to a {frame_px}-pixel frame (fp32 torch CPU oracle, oracle/*.py)"}
    # parity of the configured net(s) on that crop
    par = {"vs": f"fp32 CPU oracle, {c}x{c} crop of frame 0", "dtype": args.dtype}
    if cfg in ("sr", "chain"):
        rgb = torch.empty((c * scale, c * scale, 3), dtype=torch.float32, device=dev)
        u8 = torch.empty((c * scale, c * scale, 3), dtype=torch.uint8, device=dev)
        sr.upscale_device(up(a), out=u8, out_rgb_f32=rgb)
        torch.cuda.synchronize()
        want_u8 = (np.clip(want_sr, 0, 1) * 255.0).round().astype(np.uint8)[:, :, ::-1]
        par.update(max_abs=float(np.abs(rgb.cpu().numpy() - want_sr).max()), psnr_db=psnr_u8(u8.cpu().numpy(), want_u8),
                   max_lsb=int(np.abs(u8.cpu().numpy().astype(int) - want_u8.astype(int)).max()), tolerance="max-abs <= 1e-3, PSNR >= 50 dB")
    if cfg == "tap":
        rgb = torch.empty((c, c, 3), dtype=torch.float32, device=dev)
        u8 = torch.empty((c, c, 3), dtype=torch.uint8, device=dev)
        naf.denoise_device(up(a), out=u8, out_rgb_f32=rgb)
        torch.cuda.synchronize()
        want_u8 = np.clip(want_tap * 255.0, 0, 255).astype(np.uint8)[:, :, ::-1]       # truncation, tap_denoise.py:399-415
        par.update(max_abs=float(np.abs(rgb.cpu().numpy() - want_tap).max()), psnr_db=psnr_u8(u8.cpu().numpy(), want_u8),
                   max_lsb=int(np.abs(u8.cpu().numpy().astype(int) - want_u8.astype(int)).max()), tolerance="max-abs <= 2e-3 (f16)")
    if cfg == "rife":
        rgb = torch.empty((c, c, 3), dtype=torch.float32, device=dev)
        u8 = torch.empty((c, c, 3), dtype=torch.uint8, device=dev)
        ifn.interpolate_device(up(a), up(b), 0.5, out=u8, out_rgb_f32=rgb)
        torch.cuda.synchronize()
        want_u8 = (np.clip(want_if, 0, 1) * 255.0).round().astype(np.uint8)[:, :, ::-1]
        par.update(max_abs=float(np.abs(rgb.cpu().numpy() - want_if).max()), psnr_db=psnr_u8(u8.cpu().numpy(), want_u8),
                   max_lsb=int(np.abs(u8.cpu().numpy().astype(int) - want_u8.astype(int)).max()), tolerance="max-abs <= 4e-3 (f16), PSNR >= 50 dB")
    return par, cpu


def synth_small():
    from framewright_amd.synth import synthetic_rrdbnet_state
    return synthetic_rrdbnet_state(1, 4, seed=1)


if __name__ == "__main__":
    main()
