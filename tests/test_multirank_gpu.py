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
