"""TAP temporal denoise on the GPU: NAFNet forward vs the fp32 CPU oracle (oracle/nafnet_ref.py), and the driver
arithmetic (tile ramp blend, temporal average, strength blend) BIT-EXACT vs the numpy transcription of reference
tap_denoise.py (oracle/tap_ref.py).  Tolerance for the network: uint8 output within 1 LSB (f16) / 2 LSB (bf16) of the
oracle and max-abs on the float output < 2e-3 (f16, the default dtype) / 1.5e-2 (bf16: opt-in, a regression bound and not the parity
bar); parity vs upstream NAFNet itself is unpinned."""
import ctypes as C

import numpy as np
import pytest
import torch

from framewright_amd import _lib
from framewright_amd import tap_denoise as T
from framewright_amd.synth import synthetic_frames, synthetic_nafnet_state
from oracle import nafnet_ref, tap_ref

pytestmark = pytest.mark.gpu

SMALL = dict(width=32, middle_blk_num=2, enc_blk_nums=(1, 2), dec_blk_nums=(1, 1))
FULL = dict(width=64, middle_blk_num=12, enc_blk_nums=(2, 2, 4, 8), dec_blk_nums=(2, 2, 2, 2))


def _oracle_model(sd, args):
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    return lambda x: nafnet_ref.nafnet_forward(sdt, x, args["middle_blk_num"], args["enc_blk_nums"], args["dec_blk_nums"])


@pytest.mark.parametrize("dtype,tol,lsb", [("f16", 2e-3, 1), ("bf16", 1.5e-2, 3)])
@pytest.mark.parametrize("args,H,W", [(SMALL, 40, 56), (SMALL, 37, 51), (FULL, 64, 80), (FULL, 45, 70)])
def test_nafnet_vs_oracle(hip_lib, dtype, tol, lsb, args, H, W):
    sd = synthetic_nafnet_state(seed=7, **args)
    eng = T.NAFNetEngine(dtype=dtype, **args)
    eng.load_state_dict(sd)
    frame = synthetic_frames(1, H, W, seed=H + W)[0]
    t = torch.from_numpy(frame).cuda()
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    u8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    eng.denoise_device(t, out=u8, out_rgb_f32=rgb)
    torch.cuda.synchronize()
    with torch.no_grad():
        want = _oracle_model(sd, args)(tap_ref.preprocess(frame))
    want_rgb = want.squeeze(0).permute(1, 2, 0).numpy()
    err = np.abs(rgb.cpu().numpy() - want_rgb).max()
    want_u8 = tap_ref.postprocess(want)
    d = np.abs(u8.cpu().numpy().astype(int) - want_u8.astype(int)).max()
    print(f"{dtype} {args['width']}w {H}x{W}: max-abs {err:.2e}, uint8 max diff {d}, |out-in| mean "
          f"{np.abs(want_u8.astype(int) - frame.astype(int)).mean():.2f}")
    assert err < tol and d <= lsb
    assert np.array_equal(eng.denoise(frame), u8.cpu().numpy())
    eng.close()


def _p(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("h,w,ts,ov", [(100, 130, 64, 16), (64, 64, 32, 8), (97, 65, 48, 1), (80, 80, 40, 0)])
def test_tile_ramp_blend_bit_exact(hip_lib, h, w, ts, ov):
    """fw_u8_crop + fw_tile_blend_accumulate + fw_tile_blend_finish vs tap_denoise.py:435-486 with the SAME tile
    outputs (a deterministic stand-in for the model, applied on the CPU for both sides)."""
    rng = np.random.default_rng(h * w)
    frame = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    fake = lambda x: x * 0.93 + 0.031 * torch.sin(40 * x)
    want = tap_ref.denoise_frame_tiled(fake, frame, ts, ov)
    dev = torch.from_numpy(frame).cuda()
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    wsum = torch.zeros((h, w), dtype=torch.float32, device="cuda")
    tile = torch.empty((ts, ts, 3), dtype=torch.uint8, device="cuda")
    for y1, x1 in T.tile_grid(h, w, ts, ov):
        _lib.check(hip_lib.fw_u8_crop(_p(dev), h, w, y1, x1, ts, ts, _p(tile), None))
        torch.cuda.synchronize()
        assert np.array_equal(tile.cpu().numpy(), frame[y1:y1 + ts, x1:x1 + ts])
        with torch.no_grad():
            tout = torch.from_numpy(tap_ref.postprocess(fake(tap_ref.preprocess(tile.cpu().numpy())))).cuda()
        _lib.check(hip_lib.fw_tile_blend_accumulate(_p(acc), _p(wsum), h, w, _p(tout), y1, x1, ts, ts, ov, None))
    out = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    _lib.check(hip_lib.fw_tile_blend_finish(_p(acc), _p(wsum), h, w, _p(out), None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("n,center,window", [(7, 3, 5), (7, 0, 5), (7, 6, 5), (3, 1, 3), (9, 4, 7)])
def test_temporal_average_bit_exact(hip_lib, n, center, window):
    rng = np.random.default_rng(n * 10 + center)
    den = [rng.integers(0, 256, size=(33, 47, 3), dtype=np.uint8) for _ in range(n)]
    s, e, ws = T.temporal_window(n, center, window)
    want = tap_ref.temporal_average(den[s:e], ws)
    d = T.TAPDenoiser(T.TAPDenoiseConfig(temporal_window=window))
    got = d._temporal_average_device([torch.from_numpy(x).cuda() for x in den[s:e]], ws)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("s", [0.0, 0.3, 0.5, 0.8, 0.999])
def test_strength_blend_bit_exact(hip_lib, s):
    rng = np.random.default_rng(int(s * 1000))
    o = rng.integers(0, 256, size=(50, 70, 3), dtype=np.uint8)
    dn = rng.integers(0, 256, size=(50, 70, 3), dtype=np.uint8)
    d = T.TAPDenoiser(T.TAPDenoiseConfig(strength=s))
    got = d._strength_blend_device(torch.from_numpy(o).cuda(), torch.from_numpy(dn).cuda())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy(), tap_ref.strength_blend(o, dn, s))


def test_clip_driver_matches_reference_schedule(hip_lib):
    """denoise_clip (each frame denoised once, cached) == the reference's per-centre re-denoising schedule run through
    the oracle driver with the GPU engine as the model stand-in is not possible on the CPU; instead check it against
    the oracle driver fed by the GPU's own per-frame outputs (isolates the driver logic), tiled and windowed."""
    args = SMALL
    sd = synthetic_nafnet_state(seed=11, **args)
    eng = T.NAFNetEngine(dtype="f16", **args)
    eng.load_state_dict(sd)
    cfg = T.TAPDenoiseConfig(model="nafnet", temporal_window=5, strength=0.8, tile_size=48, tile_overlap=8)
    dn = T.TAPDenoiser(cfg, engine=eng)
    frames = list(synthetic_frames(6, 72, 100, seed=5))
    got = dn.denoise_clip(frames)
    per_frame = [dn._denoise_frame_tiled(f) for f in frames]
    for i in range(len(frames)):
        s, e, ws = tap_ref.temporal_weights(len(frames), i, 5)
        want = tap_ref.strength_blend(frames[i], tap_ref.temporal_average(per_frame[s:e], ws), 0.8)
        np.testing.assert_array_equal(got[i], want)
    # halo form (multi-GPU block partition): splitting the clip in two blocks with exchanged denoised halos gives the same
    a, b = frames[:3], frames[3:]
    halo_for_a = dn.denoise_halo_frames(b, 2, head=True)
    halo_for_b = dn.denoise_halo_frames(a, 2, head=False)
    ga = dn.denoise_clip(a, halo_after=halo_for_a)
    gb = dn.denoise_clip(b, halo_before=halo_for_b)
    for i, g in enumerate(ga + gb):
        np.testing.assert_array_equal(g, got[i])
    eng.close()


def test_denoise_frames_directory_contract(hip_lib, tmp_path, monkeypatch):
    from PIL import Image
    monkeypatch.setenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", "1")
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    for i, f in enumerate(synthetic_frames(3, 48, 64, seed=9)):
        Image.fromarray(f[:, :, ::-1]).save(src / f"frame_{i + 1:08d}.png")
    seen = []
    dn = T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet", tile_size=0, temporal_window=3), model_dir=tmp_path / "none")
    res = dn.denoise_frames(src, dst, progress_callback=seen.append)
    assert res.frames_processed == 3 and res.frames_failed == 0 and res.output_dir == dst
    assert sorted(p.name for p in dst.glob("*.png")) == [f"frame_{i:08d}.png" for i in (1, 2, 3)]
    assert seen == pytest.approx([1 / 3, 2 / 3, 1.0])
    dn.clear_cache()


def test_temporal_average_matches_reference_run(hip_lib):
    """The device temporal average against outputs of the reference's own `_denoise_with_temporal_window`
    (tests/golden/tap_reference.npz: identity per-frame step, every centre index of five clip/window shapes)."""
    import json
    from pathlib import Path
    g = Path(__file__).parent / "golden"
    arrs, meta = np.load(g / "tap_reference.npz"), json.loads((g / "tap_reference.json").read_text())
    for w in meta["windows"]:
        frames = list(arrs[w["key"] + "_frames"])
        want = arrs[w["key"] + "_out"]
        d = T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet", temporal_window=w["window"]))
        for i in range(w["n"]):
            s, e, ws = T.temporal_window(w["n"], i, w["window"])
            got = d._temporal_average_device([torch.from_numpy(x).cuda() for x in frames[s:e]], ws)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(got.cpu().numpy(), want[i])


def test_motion_adaptive_driver(hip_lib, tmp_path, monkeypatch):
    from PIL import Image
    monkeypatch.setenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", "1")
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    frames = list(synthetic_frames(3, 48, 64, seed=9))
    for i, f in enumerate(frames):
        Image.fromarray(f[:, :, ::-1]).save(src / f"frame_{i + 1:08d}.png")
    tap = T.TAPDenoiseConfig(model="nafnet", tile_size=0, temporal_window=3)
    m = T.MotionAdaptiveTAPDenoiser(T.MotionAdaptiveConfig(base_strength=1.0, motion_sensitivity=1.0), tap, model_dir=tmp_path / "none")
    res = m.denoise_frames_motion_aware(src, dst, [T.MotionLevel.STATIC, T.MotionLevel.EXTREME])     # third level defaults to MODERATE
    assert res.frames_processed == 3 and res.frames_failed == 0
    outs = [np.asarray(Image.open(dst / f"frame_{i + 1:08d}.png"))[:, :, ::-1] for i in range(3)]
    full = T.TAPDenoiser(tap, model_dir=tmp_path / "none").denoise_clip(frames)
    assert np.array_equal(outs[0], full[0])                                          # STATIC at base 1.0 -> strength 1.0
    for i, s in ((1, 0.4), (2, 0.8)):
        want = tap_ref.strength_blend(frames[i], full[i], s)
        assert np.array_equal(outs[i], want)
    m.clear_cache()


def test_tile_blend_matches_reference_run(hip_lib):
    """The device crop / ramp-blend / finish kernels against outputs of the reference's own `_denoise_frame_tiled` (identity
    pre / model / post steps; tests/golden/tile_flow_reference.npz)."""
    import json
    from pathlib import Path
    g = Path(__file__).parent / "golden"
    arrs, meta = np.load(g / "tile_flow_reference.npz"), json.loads((g / "tile_flow_reference.json").read_text())

    class Identity:
        device_id = 0

        def denoise_device(self, frame, out=None, out_rgb_f32=None, stream=None):
            if out is None:
                return frame.clone()
            out.copy_(frame)
            return out

        def clone(self):
            return Identity()

        def close(self):
            pass

    for c in meta["tiled"]:
        frame, want = arrs[c["key"] + "_in"], arrs[c["key"] + "_out"]
        d = T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet", tile_size=c["tile_size"], tile_overlap=c["overlap"]), engine=Identity())
        got = d._denoise_frame_tiled_device(torch.from_numpy(np.ascontiguousarray(frame)).cuda())
        torch.cuda.synchronize()
        np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("h,w", [(48, 64), (37, 51), (5, 7), (1, 1), (12, 200)])
@pytest.mark.parametrize("factor", [0.3, 0.12])
def test_grain_addback_bit_exact(hip_lib, h, w, factor):
    """preserve_grain (tap_denoise.py:621-632) on the device == the numpy restatement of OpenCV's 8-bit arithmetic."""
    rng = np.random.default_rng(h * 1000 + w)
    orig = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    den = np.clip(orig.astype(np.int64) + rng.integers(-20, 21, orig.shape), 0, 255).astype(np.uint8)
    if h >= 5:
        den[:2] = 250                                                               # exercises the saturating add
    dn = T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet"), engine=object())
    got = dn._grain_addback_device(torch.from_numpy(orig).cuda(), torch.from_numpy(den).cuda(), factor).cpu().numpy()
    np.testing.assert_array_equal(got, tap_ref.grain_addback(orig, den, factor))


def test_clip_driver_with_preserve_grain(hip_lib, tmp_path, monkeypatch):
    """strength blend, then the grain of the ORIGINAL added back (reference order, :614-632); the motion-aware driver scales
    the factor by adjusted / base strength (:1015-1023)."""
    from PIL import Image
    args = SMALL
    eng = T.NAFNetEngine(dtype="f16", **args)
    eng.load_state_dict(synthetic_nafnet_state(seed=11, **args))
    frames = list(synthetic_frames(4, 40, 56, seed=6))
    plain = T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet", temporal_window=3, strength=0.7, tile_size=0), engine=eng).denoise_clip(frames)
    grain = T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet", temporal_window=3, strength=0.7, tile_size=0, preserve_grain=True),
                          engine=eng).denoise_clip(frames)
    for f, p, g in zip(frames, plain, grain):
        np.testing.assert_array_equal(g, tap_ref.grain_addback(f, p, 0.3))
    eng.close()
    monkeypatch.setenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", "1")
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    for i, f in enumerate(frames[:2]):
        Image.fromarray(f[:, :, ::-1]).save(src / f"frame_{i + 1:08d}.png")
    tap = T.TAPDenoiseConfig(model="nafnet", tile_size=0, temporal_window=3, preserve_grain=True)
    m = T.MotionAdaptiveTAPDenoiser(T.MotionAdaptiveConfig(base_strength=1.0, motion_sensitivity=1.0), tap, model_dir=tmp_path / "none")
    res = m.denoise_frames_motion_aware(src, dst, [T.MotionLevel.STATIC, T.MotionLevel.EXTREME])
    assert res.frames_processed == 2
    outs = [np.asarray(Image.open(dst / f"frame_{i + 1:08d}.png"))[:, :, ::-1] for i in range(2)]
    full = T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet", tile_size=0, temporal_window=3), model_dir=tmp_path / "none").denoise_clip(frames[:2])
    np.testing.assert_array_equal(outs[0], tap_ref.grain_addback(frames[0], full[0], 0.3 * 1.0))
    np.testing.assert_array_equal(outs[1], tap_ref.grain_addback(frames[1], tap_ref.strength_blend(frames[1], full[1], 0.4), 0.3 * 0.4))
    m.clear_cache()


@pytest.mark.parametrize("H,W,dw_mfma", [(272, 400, "0"), (272, 400, "1"), (1080, 1920, "0")])
def test_deep_level_gemm_kernel_equals_the_staged_kernel(hip_lib, monkeypatch, H, W, dw_mfma):
    """The pipelined GEMM kernel of the >= 256-channel levels (pointwise_gemm.hip: 256 x 256 tiles, LDS-DMA, SimpleGate / residual
    epilogues, SCA folded into scaled weights) against the register-staged pointwise kernel it replaces there (FW_NAF_GEMM=0):
    the same fp32 accumulation of the same rounded operands up to summation order, except for conv3, whose SCA factor multiplies
    the weights instead of the activations (one more rounding of an operand).  At 272 x 400 the deep levels have 425 / 119 / 34
    pixels (ragged pixel tiles, rows clamped); 1080p is the BASELINE size (several tiles per workgroup)."""
    monkeypatch.setenv("FW_PW_DW_MFMA", dw_mfma)   # "1": the fused front with its depthwise phase on the matrix cores (SimpleGate + SCA sums)
    sd = synthetic_nafnet_state(seed=11, **FULL)
    frame = synthetic_frames(1, H, W, seed=3)[0]
    t = torch.from_numpy(frame).cuda()
    outs = []
    for mode, tail in (("1", "1"), ("0", "0")):
        monkeypatch.setenv("FW_NAF_GEMM", mode)
        monkeypatch.setenv("FW_NAF_FUSE_TAIL", tail)   # and the fused conv3..conv5 kernel of the width-64 blocks against the unfused kernels
        monkeypatch.setenv("FW_NAF_FUSE_FRONT", tail)  # and norm1 + conv1 + depthwise conv + gate of the 64- / 128-channel blocks (pw_dw_fused.hip)
        eng = T.NAFNetEngine(dtype="f16", **FULL)
        eng.load_state_dict(sd)
        rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        u8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
        eng.denoise_device(t, out=u8, out_rgb_f32=rgb)
        eng.denoise_device(t, out=u8, out_rgb_f32=rgb)            # twice: deterministic, no state left behind
        torch.cuda.synchronize()
        outs.append((rgb.cpu().numpy(), u8.cpu().numpy()))
        eng.close()
    (ra, ua), (rb, ub) = outs
    assert np.isfinite(ra).all()
    assert np.abs(ra - rb).max() < 1e-3 and np.abs(ra - rb).mean() < 5e-5
    assert np.abs(ua.astype(int) - ub.astype(int)).max() <= 1
    if H <= 300:
        with torch.no_grad():
            want = _oracle_model(sd, FULL)(tap_ref.preprocess(frame)).squeeze(0).permute(1, 2, 0).numpy()
        assert np.abs(ra - want).max() < 2e-3
