"""Device-resident stage hand-off (framewright_amd/pipeline.py) against the three stage drivers run one after the other on
host arrays — the same engine calls, so the outputs must be bit-identical."""
import numpy as np
import pytest
import torch

from framewright_amd import pipeline as P
from framewright_amd import realesrgan as R
from framewright_amd import rife as RF
from framewright_amd import tap_denoise as T
from framewright_amd.synth import synthetic_frames, synthetic_ifnet_state, synthetic_nafnet_state, synthetic_rrdbnet_state

pytestmark = pytest.mark.gpu


def test_pipeline_equals_stage_by_stage(hip_lib):
    frames = list(synthetic_frames(4, 40, 56, seed=21))
    small = dict(width=32, middle_blk_num=1, enc_blk_nums=(1, 1), dec_blk_nums=(1, 1))
    naf = T.NAFNetEngine(dtype="f16", **small)
    naf.load_state_dict(synthetic_nafnet_state(**small))
    dn = T.TAPDenoiser(T.TAPDenoiseConfig(tile_size=0, temporal_window=3, strength=0.8), engine=naf)
    sr = R.RRDBNetEngine(2, 2, "f16")
    sr.load_state_dict(synthetic_rrdbnet_state(2, 2, seed=5))
    ie = RF.IFNetEngine("f16")
    ie.load_state_dict(synthetic_ifnet_state())

    got = P.DeviceRestorationPipeline(dn, sr, ie, interp_passes=1).run(frames)

    den = dn.denoise_clip(frames)
    ups = [sr.upscale(f) for f in den]
    want = RF.FrameInterpolator(engine=ie).double(ups)
    assert len(got) == len(want) == 2 * len(frames) - 1
    for g, w in zip(got, want):
        assert g.shape == w.shape == (80, 112, 3) and np.array_equal(g, w)

    # stages are optional; CUDA tensors are accepted as input
    only_sr = P.DeviceRestorationPipeline(upscaler=sr).run_device([torch.from_numpy(frames[0]).cuda()])
    torch.cuda.synchronize()
    assert np.array_equal(only_sr[0].cpu().numpy(), sr.upscale(frames[0]))
    with pytest.raises(ValueError):
        P.DeviceRestorationPipeline().run_device(frames)
    for e in (naf, sr):
        e.close()
