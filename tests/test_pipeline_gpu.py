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


@pytest.mark.parametrize("n,block,passes,window", [(11, 4, 1, 5), (3, 8, 2, 5), (9, 2, 1, 3), (1, 4, 1, 5)])
def test_streaming_pipeline_through_the_codec_edge_equals_the_in_memory_pipeline(hip_lib, tmp_path, n, block, passes, window):
    """SURVEY §8 f1, the codec edge: a child process standing in for `ffmpeg -f rawvideo -pix_fmt bgr24 -` feeds RawVideoReader,
    `run_stream` runs denoise -> upscale -> interpolate block by block with decode, GPU work, download and encode overlapped, and
    RawVideoWriter feeds a child process standing in for the encoder: the bytes that arrive there equal the in-memory pipeline's
    frames (whole clip at once) bit for bit - across block boundaries of the temporal window, the pair boundary of the
    interpolation, a clip shorter than a block and a one-frame clip."""
    import sys

    from framewright_amd import codec as K
    Hs, Ws = 40, 56
    frames = synthetic_frames(n, Hs, Ws, seed=33)
    raw_in = tmp_path / "in.bgr"
    raw_in.write_bytes(np.ascontiguousarray(frames).tobytes())
    small = dict(width=32, middle_blk_num=1, enc_blk_nums=(1, 1), dec_blk_nums=(1, 1))
    naf = T.NAFNetEngine(dtype="f16", **small)
    naf.load_state_dict(synthetic_nafnet_state(**small))
    dn = T.TAPDenoiser(T.TAPDenoiseConfig(tile_size=0, temporal_window=window, strength=0.9), engine=naf)
    sr = R.RRDBNetEngine(2, 2, "f16")
    sr.load_state_dict(synthetic_rrdbnet_state(2, 2, seed=5))
    ie = RF.IFNetEngine("f16")
    ie.load_state_dict(synthetic_ifnet_state())
    pipe = P.DeviceRestorationPipeline(dn, sr, ie, interp_passes=passes)
    want = pipe.run(list(frames))

    decoder = [sys.executable, "-c", "import sys, shutil; shutil.copyfileobj(open(sys.argv[1], 'rb'), sys.stdout.buffer, 4096)", str(raw_in)]
    out = tmp_path / "out.bgr"
    encoder = [sys.executable, "-c", "import sys, shutil; shutil.copyfileobj(sys.stdin.buffer, open(sys.argv[1], 'wb'), 1 << 16)", str(out)]
    with K.RawVideoReader(decoder, Hs, Ws, depth=3) as reader, K.RawVideoWriter(encoder, depth=2) as writer:
        written = pipe.run_stream(reader, writer, block=block, slots=3)
    assert written == len(want) == (n - 1) * 2 ** passes + 1 and reader.frames_read == n
    assert out.read_bytes() == b"".join(np.ascontiguousarray(w).tobytes() for w in want)

    # the generator form on CUDA tensors, no codec: the same frames
    got = [t.cpu().numpy() for t in pipe.stream_device([torch.from_numpy(f).cuda() for f in frames], block=block)]
    assert len(got) == len(want) and all(np.array_equal(g, w) for g, w in zip(got, want))
    for e in (naf, sr):
        e.close()
