"""BASELINE configs[2..4] at their FULL size (1920x1080), where the fp32 oracle cannot run whole in seconds: size-independent
properties plus oracle checks on the pieces the reference's own tiling makes small.

* configs[2] RIFE 1080p pair     : determinism, hipGraph replay == direct launches, and the oracle on a 192x256 pair run whole
* configs[3] NAFNet TAP 1080p    : the reference's 512 / 32 tiling - one 512x512 tile of the frame against oracle/nafnet_ref.py on
                                   that tile, and the blended frame against the reference's blend arithmetic (oracle/tap_ref.py,
                                   pinned on reference-run vectors) applied to the engine's own tiles: bit-exact
* configs[4] chain 1080p, 4 frames: the device-resident pipeline == the three stage drivers run one after the other
"""
import numpy as np
import pytest
import torch

from framewright_amd import pipeline as P
from framewright_amd import realesrgan as R
from framewright_amd import rife as RF
from framewright_amd import tap_denoise as T
from framewright_amd.synth import synthetic_frames, synthetic_ifnet_state, synthetic_nafnet_state, synthetic_rrdbnet_state
from oracle import ifnet_ref, nafnet_ref, tap_ref

pytestmark = pytest.mark.gpu
H, W = 1080, 1920


def test_rife_1080p_pair_properties_and_oracle_crop(hip_lib, monkeypatch):
    sd = synthetic_ifnet_state()
    fr = synthetic_frames(2, H, W, seed=3)
    a, b = torch.from_numpy(fr[0]).cuda(), torch.from_numpy(fr[1]).cuda()
    eng = RF.IFNetEngine("f16")
    eng.load_state_dict(sd)
    o1 = eng.interpolate_device(a, b).clone()
    o2 = eng.interpolate_device(a, b).clone()
    torch.cuda.synchronize()
    assert o1.shape == (H, W, 3) and torch.equal(o1, o2)                      # deterministic (1080 pads to 1088 inside)
    mid = ((fr[0].astype(np.int32) + fr[1]) // 2)
    d = np.abs(o1.cpu().numpy().astype(np.int32) - mid)
    assert d.mean() < 40 and d.max() > 0                                       # an interpolation, not an average and not garbage
    assert np.array_equal(eng.interpolate(fr[0], fr[1]), o1.cpu().numpy())     # host-buffer entry == device entry
    eng.close()
    # the same forward replayed from a captured hipGraph (BASELINE configs[4]: "hipGraph-captured per-frame stages")
    monkeypatch.setenv("FW_IFNET_GRAPH", "1")
    g = RF.IFNetEngine("f16")
    g.load_state_dict(sd)
    out = torch.empty_like(a)
    for _ in range(3):                                                          # first call direct, second captures, third replays
        g.interpolate_device(a, b, out=out)
    torch.cuda.synchronize()
    assert torch.equal(out, o1)
    g.close()
    monkeypatch.delenv("FW_IFNET_GRAPH")
    # oracle on a pair small enough to run whole
    ca, cb = np.ascontiguousarray(fr[0][300:492, 700:956]), np.ascontiguousarray(fr[1][300:492, 700:956])
    e2 = RF.IFNetEngine("f16")
    e2.load_state_dict(sd)
    rgb = torch.empty((192, 256, 3), dtype=torch.float32, device="cuda")
    e2.interpolate_device(torch.from_numpy(ca).cuda(), torch.from_numpy(cb).cuda(), 0.5, out_rgb_f32=rgb)
    torch.cuda.synchronize()
    t = lambda f: torch.from_numpy(f[:, :, ::-1].astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    with torch.no_grad():
        want = ifnet_ref.ifnet_forward({k: torch.from_numpy(v) for k, v in sd.items()}, t(ca), t(cb), 0.5)[0].permute(1, 2, 0).numpy()
    assert np.abs(rgb.cpu().numpy() - want).max() < 4e-3
    e2.close()


def test_tap_1080p_reference_tiling_tile_vs_oracle_and_bit_exact_blend(hip_lib):
    sd = synthetic_nafnet_state(**T.NAFNET_ARGS)
    frame = synthetic_frames(1, H, W, seed=4)[0]
    eng = T.NAFNetEngine(dtype="f16", **T.NAFNET_ARGS)
    eng.load_state_dict(sd)
    dn = T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet", tile_size=512, tile_overlap=32, temporal_window=1), engine=eng)
    got = dn.denoise_clip([frame])[0]
    assert got.shape == frame.shape
    # (a) one interior tile against the fp32 oracle on that tile (SCA pools over the tile: the tile geometry is part of the result)
    tiles = tap_ref.tile_grid(H, W, 512, 32)
    assert len(tiles) == 12                                                    # 3 x 4 tiles, tap_denoise.py:435-450
    y1, x1 = tiles[5]
    tile = np.ascontiguousarray(frame[y1:y1 + 512, x1:x1 + 512])
    rgb = torch.empty((512, 512, 3), dtype=torch.float32, device="cuda")
    u8 = torch.empty((512, 512, 3), dtype=torch.uint8, device="cuda")
    eng.denoise_device(torch.from_numpy(tile).cuda(), out=u8, out_rgb_f32=rgb)
    torch.cuda.synchronize()
    x = torch.from_numpy(tile[:, :, ::-1].astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    with torch.no_grad():
        want = nafnet_ref.nafnet_forward({k: torch.from_numpy(v) for k, v in sd.items()}, x, T.NAFNET_ARGS["middle_blk_num"],
                                         T.NAFNET_ARGS["enc_blk_nums"], T.NAFNET_ARGS["dec_blk_nums"])[0].permute(1, 2, 0).numpy()
    assert np.abs(rgb.cpu().numpy() - want).max() < 2e-3
    want_u8 = np.clip(want * 255.0, 0, 255).astype(np.uint8)[:, :, ::-1]
    assert np.abs(u8.cpu().numpy().astype(int) - want_u8.astype(int)).max() <= 1
    # (b) the whole frame: the reference's blend (float32 accumulate of ramp-weighted uint8 tiles, / max(w, 1e-8), truncation)
    #     applied to the engine's own tiles reproduces the device result bit for bit
    out = np.zeros((H, W, 3), np.float32)
    wsum = np.zeros((H, W, 1), np.float32)
    ramp = np.linspace(0, 1, 32)
    for ty, tx in tiles:
        t8 = eng.denoise(np.ascontiguousarray(frame[ty:ty + 512, tx:tx + 512])).astype(np.float32)
        tw = np.ones((512, 512, 1), np.float32)
        if ty > 0:
            tw[:32] *= ramp.reshape(-1, 1, 1)
        if ty + 512 < H:
            tw[-32:] *= ramp[::-1].reshape(-1, 1, 1)
        if tx > 0:
            tw[:, :32] *= ramp.reshape(1, -1, 1)
        if tx + 512 < W:
            tw[:, -32:] *= ramp[::-1].reshape(1, -1, 1)
        out[ty:ty + 512, tx:tx + 512] += t8 * tw
        wsum[ty:ty + 512, tx:tx + 512] += tw
    want_frame = (out / np.maximum(wsum, 1e-8)).astype(np.uint8)
    assert np.array_equal(got, want_frame)
    dn.clear_cache()


def test_nafnet_1080p_hipgraph_replay_equals_direct_launches(hip_lib, monkeypatch):
    """BASELINE configs[4] names hipGraph-captured per-frame stages: the NAFNet forward captured once (FW_NAF_GRAPH=1) and replayed
    produces the frame the direct launches produce, also after the weights were replaced (captures are dropped with them)."""
    sd = synthetic_nafnet_state(**T.NAFNET_ARGS)
    f = torch.from_numpy(synthetic_frames(1, H, W, seed=6)[0]).cuda()
    eng = T.NAFNetEngine(dtype="f16", **T.NAFNET_ARGS)
    eng.load_state_dict(sd)
    want = torch.empty_like(f)
    eng.denoise_device(f, out=want)
    torch.cuda.synchronize()
    eng.close()
    monkeypatch.setenv("FW_NAF_GRAPH", "1")
    g = T.NAFNetEngine(dtype="f16", **T.NAFNET_ARGS)
    g.load_state_dict(sd)
    out = torch.empty_like(f)
    for _ in range(3):                                                          # first call direct, second captures, third replays
        out.zero_()
        g.denoise_device(f, out=out)
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    sd2 = synthetic_nafnet_state(seed=9, **T.NAFNET_ARGS)
    g.load_state_dict(sd2)                                                      # new weights: the captured graph must not be replayed
    g.denoise_device(f, out=out)
    g.denoise_device(f, out=out)
    torch.cuda.synchronize()
    g.close()
    monkeypatch.delenv("FW_NAF_GRAPH")
    e2 = T.NAFNetEngine(dtype="f16", **T.NAFNET_ARGS)
    e2.load_state_dict(sd2)
    want2 = torch.empty_like(f)
    e2.denoise_device(f, out=want2)
    torch.cuda.synchronize()
    assert torch.equal(out, want2) and not torch.equal(want2, want)
    e2.close()


@pytest.mark.parametrize("graphs", [False, True])
def test_chain_1080p_equals_stage_by_stage(hip_lib, monkeypatch, graphs):
    if graphs:   # every stage replays (or re-captures, where the pipeline hands over fresh buffers) a hipGraph of its forward
        for k in ("FW_NAF_GRAPH", "FW_RRDB_GRAPH", "FW_IFNET_GRAPH"):
            monkeypatch.setenv(k, "1")
    frames = list(synthetic_frames(4, H, W, seed=5))
    naf = T.NAFNetEngine(dtype="f16", **T.NAFNET_ARGS)
    naf.load_state_dict(synthetic_nafnet_state(**T.NAFNET_ARGS))
    dn = T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet", tile_size=0, temporal_window=5), engine=naf)
    sr = R.RRDBNetEngine(6, 4, "f16")                                          # the 6-block x4 model: 8K frames, a quarter of the time
    sr.load_state_dict(synthetic_rrdbnet_state(6, 4, seed=5))
    ie = RF.IFNetEngine("f16")
    ie.load_state_dict(synthetic_ifnet_state())
    got = P.DeviceRestorationPipeline(dn, sr, ie, interp_passes=1).run_device(frames)
    torch.cuda.synchronize()
    assert len(got) == 7 and tuple(got[0].shape) == (4 * H, 4 * W, 3)
    den = dn.denoise_clip(frames)
    for i in range(4):
        up = sr.upscale(den[i])
        assert np.array_equal(got[2 * i].cpu().numpy(), up), f"upscaled frame {i}"
        if i == 1:                                                              # one 8K pair through the interpolator on the host path
            up0 = got[0].cpu().numpy()
            assert np.array_equal(got[1].cpu().numpy(), ie.interpolate(up0, up))
    for e in (naf, sr, ie):
        e.close()
