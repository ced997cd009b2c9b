"""Engines on a device other than the process's current one (every test ran on device 0).  Needs two visible GPUs;
skipped on the one-GPU boxes of this pool, runs where the driver has a whole node.  Each engine on cuda:1, with cuda:0 left as the
current device of the calling thread, must give the frames the same engine gives on cuda:0."""
import numpy as np
import pytest
import torch

from framewright_amd import realesrgan as R
from framewright_amd import restormer as RS
from framewright_amd import rife as RF
from framewright_amd import tap_denoise as T
from framewright_amd.synth import synthetic_frames, synthetic_ifnet_state, synthetic_nafnet_state, synthetic_rrdbnet_state

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible GPUs")]

NAF = dict(width=32, middle_blk_num=1, enc_blk_nums=(1, 1), dec_blk_nums=(1, 1))
REST = dict(dim=48, num_blocks=(1, 1, 1, 1), num_refinement_blocks=1, heads=(1, 2, 4, 8), ffn_expansion_factor=2.66)


def _both(make, run):
    outs = []
    for dev in (0, 1):
        torch.cuda.set_device(0)                      # the caller's current device stays 0
        eng = make(dev)
        outs.append(run(eng, torch.device("cuda", dev)))
        torch.cuda.synchronize(dev)
        eng.close()
    assert np.array_equal(outs[0], outs[1])


def test_rrdbnet_on_device_1(hip_lib):
    sd = synthetic_rrdbnet_state(2, 4, seed=3)
    f = synthetic_frames(1, 40, 56, seed=1)[0]

    def make(d):
        e = R.RRDBNetEngine(2, 4, "f16", device_id=d)
        e.load_state_dict(sd)
        return e
    _both(make, lambda e, dev: e.upscale_device(torch.from_numpy(f).to(dev)).cpu().numpy())


def test_nafnet_and_tap_driver_on_device_1(hip_lib):
    sd = synthetic_nafnet_state(**NAF)
    frames = list(synthetic_frames(3, 40, 56, seed=2))

    def make(d):
        e = T.NAFNetEngine(dtype="f16", device_id=d, **NAF)
        e.load_state_dict(sd)
        return e

    def run(e, dev):
        dn = T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet", tile_size=0, temporal_window=3, gpu_id=dev.index), engine=e)
        return np.stack([t.cpu().numpy() for t in dn.denoise_clip_device(frames)])
    _both(make, run)


def test_ifnet_on_device_1(hip_lib):
    sd = synthetic_ifnet_state()
    fr = synthetic_frames(2, 64, 96, seed=3)

    def make(d):
        e = RF.IFNetEngine("f16", d)
        e.load_state_dict(sd)
        return e
    _both(make, lambda e, dev: e.interpolate_device(torch.from_numpy(fr[0]).to(dev), torch.from_numpy(fr[1]).to(dev), 0.5).cpu().numpy())


def test_restormer_on_device_1(hip_lib):
    sd = RS.synthetic_restormer_state(seed=4, **REST)
    f = synthetic_frames(1, 40, 56, seed=5)[0]

    def make(d):
        e = RS.RestormerEngine(dtype="f16", device_id=d, **REST)
        e.load_state_dict(sd)
        return e
    _both(make, lambda e, dev: e.denoise_device(torch.from_numpy(f).to(dev)).cpu().numpy())
