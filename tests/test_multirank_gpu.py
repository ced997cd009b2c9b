"""The multi-rank paths END TO END with the real engines (BASELINE configs[2..3]): two child processes on
device 0 (the GPU box has one card; the rendezvous is gloo on 127.0.0.1, the halos cross as CPU tensors exactly where RCCL
would move device memory), NAFNet temporal denoise with the 2-frame denoised halo and IFNet pairs with the 1-frame input halo,
each compared bit for bit with the single-process clip."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from framewright_amd import rife as RF
from framewright_amd import sharding as S
from framewright_amd import tap_denoise as T
from framewright_amd.synth import synthetic_frames, synthetic_ifnet_state, synthetic_nafnet_state

pytestmark = pytest.mark.gpu

SMALL_NAF = dict(width=32, middle_blk_num=1, enc_blk_nums=(1, 1), dec_blk_nums=(1, 1))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_tap(strength=0.8):
    eng = T.NAFNetEngine(dtype="f16", device_id=0, **SMALL_NAF)
    eng.load_state_dict(synthetic_nafnet_state(seed=3, **SMALL_NAF))
    return T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet", tile_size=32, tile_overlap=8, temporal_window=5, strength=strength), engine=eng)


def _rank_main(rank, world, port, frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        tap = _make_tap()
        den = S.sharded_tap_denoise_device(tap, frames, halo_device="cpu")
        ifn = RF.IFNetEngine("f16", 0)
        ifn.load_state_dict(synthetic_ifnet_state())
        mids = S.sharded_interpolate_device(ifn, frames, halo_device="cpu")
        torch.cuda.synchronize()
        q.put((rank, {k: v.cpu().numpy() for k, v in den.items()}, {k: v.cpu().numpy() for k, v in mids.items()}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [9, 4])
def test_two_ranks_with_real_engines_equal_the_single_process_clip(hip_lib, n_frames):
    frames = list(synthetic_frames(n_frames, 48, 64, seed=5))
    tap = _make_tap()
    want_den = tap.denoise_clip(frames)
    ifn = RF.IFNetEngine("f16", 0)
    ifn.load_state_dict(synthetic_ifnet_state())
    want_mid = [ifn.interpolate(frames[i], frames[i + 1]) for i in range(n_frames - 1)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    den, mids = {}, {}
    for rank, d, m in got:
        lo, hi = S.block_partition(n_frames, 2, 2)[rank]
        assert sorted(d) == list(range(lo, hi))
        den.update(d)
        mids.update(m)
    assert sorted(den) == list(range(n_frames)) and sorted(mids) == list(range(n_frames - 1))
    for i in range(n_frames):
        assert np.array_equal(den[i], want_den[i]), f"denoised frame {i} differs across the rank boundary"
    for i in range(n_frames - 1):
        assert np.array_equal(mids[i], want_mid[i]), f"mid frame {i}"


@pytest.mark.parametrize("config", ["sr", "tap", "rife", "chain"])
def test_bench_two_ranks_under_torch_distributed_run(hip_lib, config, tmp_path):
    """The command the driver uses for its scaling run, with two fresh ranks (`python -m torch.distributed.run ... bench.py --gpus 2`)
    - both on the box's one card (FW_BENCH_FORCE_DEVICE=0: gloo rendezvous, halos as CPU tensors; on a node each rank has its own
    GPU and the same branch moves device tensors over RCCL).  Must exit 0 and print ONE JSON line that says n_gpus 2; and
    `--gpus 2` WITHOUT the launcher must refuse instead of measuring one GPU and reporting two."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, FW_BENCH_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    size = ["--height", "136", "--width", "240"] if config in ("sr", "chain") else ["--height", "270", "--width", "480"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "2", "--config", config, "--steps", "4", "--warmup", "1",
           "--no-cpu-baseline", *size]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(root))
    assert r.returncode == 0, "\n".join(ln for ln in r.stderr.splitlines() if "socket.cpp" not in ln and "amdgpu.ids" not in ln)[-6000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 4 and res["scaling"] == "weak" and res["value"] > 0
    assert abs(res["value"] - 2 * 4 / (res["ms_per_step"] * 4 * 1e-3)) < 1e-6 * res["value"]      # whole-job rate: both ranks' frames over the slowest rank's time
    if config == "sr":
        r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True,
                           timeout=300, cwd=str(root))
        assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
