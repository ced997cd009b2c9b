#!/usr/bin/env python3
"""Headline benchmark: frames/sec of Real-ESRGAN x4 (RRDBNet-x4plus, 23 blocks) on synthetic 1080p frames.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one frame through the whole hot path (uint8 BGR 1920x1080 resident in HBM -> uint8 BGR 7680x4320 in
HBM).  Frames are independent, so N ranks each process K frames of their round-robin shard (weak scaling, no
data-path collective); rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around every conv
launch (the MFMA implicit-GEMM kernels are >99 % of the FLOPs); `cpu_baseline` times the fp32 CPU oracle on a bounded
crop on rank 0 at N=1.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0}  # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--model", default="RealESRGAN_x4plus")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-crop", type=int, default=192, help="side of the square crop timed on the CPU oracle")
    return ap.parse_args()


def cpu_baseline(sd, num_block, scale, frame, crop, frame_px):
    """fp32 oracle on the host cores, bounded sample: one crop x crop window of frame 0, whole network."""
    import numpy as np
    import torch
    from oracle import rrdbnet_ref as ref

    cores = torch.get_num_threads()
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    c = frame[:crop, :crop]
    x = torch.from_numpy(np.ascontiguousarray(c[:, :, ::-1]).astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    with torch.no_grad():
        ref.rrdbnet_forward(sdt, x[:, :, :32, :32], num_block, scale)  # warm the thread pool
        t0 = time.perf_counter()
        ref.rrdbnet_forward(sdt, x, num_block, scale)
        dt = time.perf_counter() - t0
    fps = (crop * crop / frame_px) / dt
    return {"value": fps, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{crop}x{crop} crop of frame 0 through all {num_block} RRDB blocks + tail in {dt:.1f} s, "
                      f"scaled by pixel count to a {frame_px}-pixel frame (fp32 torch CPU oracle)"}


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # rehearsal of the multi-rank path on a 1-GPU box only: FW_BENCH_FORCE_DEVICE=0 puts every rank on one card
    dev_ord = int(os.environ.get("FW_BENCH_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(dev_ord)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one rank per GPU over RCCL; the single-card rehearsal (FW_BENCH_FORCE_DEVICE) uses gloo unless told otherwise
        backend = os.environ.get("FW_BENCH_BACKEND", "gloo" if "FW_BENCH_FORCE_DEVICE" in os.environ else "nccl")
        dist.init_process_group(backend)

    from framewright_amd import build as fw_build
    if rank == 0:
        fw_build.build()
    def barrier():
        if world > 1:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[dev_ord])
            else:
                dist.barrier()

    barrier()
    from framewright_amd.realesrgan import RRDBNetEngine
    from framewright_amd.synth import RRDB_MODELS, synthetic_frames, synthetic_rrdbnet_state

    num_block, scale = RRDB_MODELS[args.model]
    H, W = args.height, args.width
    sd = synthetic_rrdbnet_state(num_block, scale, seed=1234)
    eng = RRDBNetEngine(num_block, scale, args.dtype, device_id=dev_ord)
    eng.load_state_dict(sd)

    # this rank's shard of the clip: frames rank, rank+world, ... (round-robin); a few distinct frames are kept
    # resident and cycled
    n_distinct = max(1, min(4, args.steps))
    clip = synthetic_frames(n_distinct * world, H, W, seed=2)
    mine = [torch.from_numpy(np.ascontiguousarray(clip[i])).cuda() for i in range(rank, n_distinct * world, world)]
    out = torch.empty((H * scale, W * scale, 3), dtype=torch.uint8, device="cuda")

    def step(i):
        eng.upscale_device(mine[i % len(mine)], out=out)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(i)
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    t = torch.tensor([wall], dtype=torch.float64,
                     device="cuda" if (world > 1 and dist.get_backend() == "nccl") or world == 1 else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())
    dev_ms = ev0.elapsed_time(ev1)

    # ---- roofline: HIP events around every conv launch, on the launch stream -----------------------------
    eng.profile_enable(True)
    prof_frames = min(2, args.steps)
    for i in range(prof_frames):
        step(i)
    launches, conv_ms, conv_flops = eng.profile_read()
    eng.profile_enable(False)

    if rank == 0:
        flops_frame = eng.flops(H, W)
        fps = world * args.steps / wall_max
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        peak = PEAK_TFLOPS[args.dtype]
        # HBM traffic per conv launch from the committed PMC run of this same command (separate rocprofv3 --pmc passes,
        # gfx950 FETCH_SIZE correction applied) — bench.py cannot collect PMC counters itself.
        traffic, traffic_note = None, None
        tf = ROOT / "profiles" / "r01_traffic.json"
        if tf.exists() and (H, W, args.model) == (1080, 1920, "RealESRGAN_x4plus"):
            tj = json.loads(tf.read_text())
            traffic = tj["hbm_bytes_per_launch"]
            traffic_note = f"profiles/r01_traffic.json ({tj['hbm_tb_per_s']:.2f} TB/s of real HBM traffic while profiled)"
        res = {
            "metric": "frames/sec Real-ESRGAN x4 1080p (RRDBNet-x4plus, 1920x1080 -> 7680x4320)",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.model} x{scale}, {W}x{H} uint8 BGR frames resident in HBM -> "
                                   f"{W * scale}x{H * scale} uint8 BGR in HBM, seeded synthetic weights, "
                                   f"{args.steps} frames per GPU", "frame_tflop": flops_frame / 1e12,
                       "sharding": f"round-robin frames over {world} rank(s), no collective"},
            "device_ms_per_step": dev_ms / args.steps,
            "whole_path_tflops_per_gpu": flops_frame * args.steps / (dev_ms * 1e-3) / 1e12,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": "conv3x3_mfma_kernel + conv3x3_pair_slide_kernel (all instantiations)",
                         "launches_timed": launches, "avg_launch_ms": conv_ms / max(launches, 1),
                         "avg_launch_gflop": conv_flops / max(launches, 1) / 1e9},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sd, num_block, scale, clip[0], args.cpu_crop, H * W)
        print(json.dumps(res), flush=True)
    eng.close()
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
