"""Real-ESRGAN generator on the GPU (fw_rrdbnet_* through the C-ABI) against the fp32 CPU oracle
(oracle/rrdbnet_ref.py + oracle/realesrganer_ref.py) and against the reference-generated golden vectors.

Tolerances (stated on the [0,1] image range):
  * f16 operands (the default of every entry point and of bench.py): max-abs <= 1e-3 on the un-clamped float output - the
    north-star bar - and PSNR >= 60 dB on uint8
  * bf16 operands (opt-in): PSNR >= 50 dB; max-abs measures 2.6e-3 (x4) ... 3.6e-3 (x2) on the 23-block nets (8 mantissa bits), so the 1e-3 bar is an expected
    failure for this dtype (test_bf16_operands_miss_the_1e3_bar, strict xfail) and 4e-3 is kept as a regression bound
  * vs an oracle whose weights AND activations are rounded like the kernel's: <= 2e-4 — isolates kernel bugs from
    operand rounding.
"""
import math
import threading
import time

import numpy as np
import pytest
import torch

from framewright_amd import _lib
from framewright_amd import realesrgan as R
from framewright_amd.synth import synthetic_frames, synthetic_rrdbnet_state
from oracle import realesrganer_ref as oref
from oracle import rrdbnet_ref as ref

pytestmark = pytest.mark.gpu


def _sd_t(sd):
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def _psnr_u8(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * math.log10(255.0 ** 2 / mse)


def _gpu_rgb_f32(engine, frame_bgr):
    t = torch.from_numpy(frame_bgr).cuda()
    s = engine.scale
    rgb = torch.empty((frame_bgr.shape[0] * s, frame_bgr.shape[1] * s, 3), dtype=torch.float32, device="cuda")
    u8 = torch.empty(rgb.shape, dtype=torch.uint8, device="cuda")
    engine.upscale_device(t, out=u8, out_rgb_f32=rgb)
    torch.cuda.synchronize()
    return rgb.cpu().numpy(), u8.cpu().numpy()


def _oracle_rgb_f32(sd, frame_bgr, num_block, scale):
    x = torch.from_numpy(frame_bgr[:, :, ::-1].astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    if scale == 2:  # reflect mod-pad, as RealESRGANer.pre_process
        ph, pw = x.shape[2] % 2, x.shape[3] % 2
        x = torch.nn.functional.pad(x, (0, pw, 0, ph), "reflect")
    with torch.no_grad():
        y = ref.rrdbnet_forward(_sd_t(sd), x, num_block, scale)
    y = y[:, :, :frame_bgr.shape[0] * scale, :frame_bgr.shape[1] * scale]
    return y.squeeze(0).permute(1, 2, 0).numpy()


@pytest.mark.parametrize("dtype,tol", [("f16", 1e-3), pytest.param("bf16", 4e-3, id="bf16-optin-regression-bound")])
def test_golden_trunk_and_tail(hip_lib, golden_dir, dtype, tol):
    """Input/output recorded from the reference's AESRGAN module (oracle/gen_golden.py)."""
    g = np.load(golden_dir / "rrdb_reference.npz")
    sd = synthetic_rrdbnet_state(2, 4, seed=int(g["net_seed"]))
    # the golden input is float; quantise to uint8 for the frame path and re-run the oracle on the same bytes
    frame = np.ascontiguousarray((g["net_in"][0].transpose(1, 2, 0)[:, :, ::-1] * 255).round().astype(np.uint8))
    eng = R.RRDBNetEngine(2, 4, dtype)
    eng.load_state_dict(sd)
    rgb, _ = _gpu_rgb_f32(eng, frame)
    want = _oracle_rgb_f32(sd, frame, 2, 4)
    assert np.abs(rgb - want).max() < tol
    # and the oracle on the original float input reproduces the reference's recorded output
    with torch.no_grad():
        y = ref.rrdbnet_forward(_sd_t(sd), torch.from_numpy(g["net_in"]), 2, 4).numpy()
    assert np.abs(y - g["net_out"]).max() < 2e-5
    eng.close()


@pytest.mark.parametrize("dtype,max_abs,min_psnr", [("f16", 1e-3, 60.0), pytest.param("bf16", 4e-3, 50.0, id="bf16-optin-regression-bound")])
@pytest.mark.parametrize("num_block,scale,H,W", [(23, 4, 40, 56), (6, 4, 33, 47), (23, 2, 41, 57), (3, 2, 64, 64)])
def test_rrdbnet_vs_oracle(hip_lib, dtype, max_abs, min_psnr, num_block, scale, H, W):
    sd = synthetic_rrdbnet_state(num_block, scale, seed=1234)
    frame = synthetic_frames(1, H, W, seed=H * W)[0]
    eng = R.RRDBNetEngine(num_block, scale, dtype)
    eng.load_state_dict(sd)
    rgb, u8 = _gpu_rgb_f32(eng, frame)
    want = _oracle_rgb_f32(sd, frame, num_block, scale)
    assert rgb.shape == want.shape == (H * scale, W * scale, 3)
    err = np.abs(rgb - want).max()
    want_u8 = (np.clip(want, 0, 1) * 255.0).round().astype(np.uint8)[:, :, ::-1]
    psnr = _psnr_u8(u8, want_u8)
    print(f"{dtype} nb={num_block} x{scale} {H}x{W}: max-abs {err:.2e} psnr {psnr:.1f} dB "
          f"out range [{want.min():.2f},{want.max():.2f}] std {want.std():.3f}")
    assert err < max_abs
    assert psnr >= min_psnr
    assert np.abs(u8.astype(int) - want_u8.astype(int)).max() <= (1 if dtype == "f16" else 3)
    eng.close()


@pytest.mark.xfail(strict=True, reason="bf16 operands are opt-in: 2.6e-3 max-abs on the 23-block x4 net (8 mantissa bits); the 1e-3 bar "
                                      "is met by f16, the default dtype of every entry point and of bench.py")
def test_bf16_operands_miss_the_1e3_bar(hip_lib):
    sd = synthetic_rrdbnet_state(23, 4, seed=1234)
    frame = synthetic_frames(1, 40, 56, seed=40 * 56)[0]
    eng = R.RRDBNetEngine(23, 4, "bf16")
    eng.load_state_dict(sd)
    rgb, _ = _gpu_rgb_f32(eng, frame)
    eng.close()
    assert np.abs(rgb - _oracle_rgb_f32(sd, frame, 23, 4)).max() < 1e-3


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_kernel_vs_operand_rounded_oracle(hip_lib, dtype):
    """Same rounding points as the kernels (weights and every conv input rounded to the operand type, fp32
    accumulate, fp32 trunk): what is left is accumulation order only."""
    tdt = torch.float16 if dtype == "f16" else torch.bfloat16
    nb, H, W = 4, 36, 44
    sd = synthetic_rrdbnet_state(nb, 4, seed=77)
    frame = synthetic_frames(1, H, W, seed=9)[0]
    eng = R.RRDBNetEngine(nb, 4, dtype)
    eng.load_state_dict(sd)
    rgb, _ = _gpu_rgb_f32(eng, frame)
    eng.close()

    import torch.nn.functional as F
    sdt = _sd_t(sd)
    q = lambda t: t.to(tdt).float()
    conv = lambda k, x: F.conv2d(q(x), q(sdt[k + ".weight"]), sdt[k + ".bias"], 1, 1)
    lr = lambda t: F.leaky_relu(t, 0.2)
    x = torch.from_numpy(frame[:, :, ::-1].astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    with torch.no_grad():
        feat = conv("conv_first", x)
        t = feat
        for b in range(nb):
            r_in = t
            for r in (1, 2, 3):
                p = f"body.{b}.rdb{r}"
                x0 = t
                x1 = lr(conv(p + ".conv1", x0))
                x2 = lr(conv(p + ".conv2", torch.cat([x0, x1], 1)))
                x3 = lr(conv(p + ".conv3", torch.cat([x0, x1, x2], 1)))
                x4 = lr(conv(p + ".conv4", torch.cat([x0, x1, x2, x3], 1)))
                t = conv(p + ".conv5", torch.cat([x0, x1, x2, x3, x4], 1)) * 0.2 + x0
            t = t * 0.2 + r_in
        feat = feat + conv("conv_body", t)
        feat = lr(conv("conv_up1", F.interpolate(feat, scale_factor=2, mode="nearest")))
        feat = lr(conv("conv_up2", F.interpolate(feat, scale_factor=2, mode="nearest")))
        y = conv("conv_last", lr(conv("conv_hr", feat)))
    want = y.squeeze(0).permute(1, 2, 0).numpy()
    # a rounding flip of one operand is possible where accumulation order moves a value across a rounding
    # boundary, hence not bit-exact
    assert np.abs(rgb - want).max() < 2e-4


@pytest.mark.parametrize("dtype,tol_max,tol_mean", [("f16", 5e-4, 1e-4), ("bf16", 4e-3, 8e-4)])
@pytest.mark.parametrize("nb,scale,H,W", [(23, 4, 40, 56), (2, 2, 37, 45)])
def test_split_trunk_matches_fp32_trunk(hip_lib, monkeypatch, dtype, tol_max, tol_mean, nb, scale, H, W):
    """The residual trunk kept as typed hi + typed lo planes (the default) against the fp32 trunk it replaces
    (FW_RRDB_SPLIT_TRUNK=0).  hi + lo carries 16 (bf16) / 22 (f16) mantissa bits, so the trunks agree to ~1e-6; what
    the outputs show is operands whose rounding to the operand type flips, amplified through 69 dense blocks: a
    fraction of the operand-rounding error of the path itself.  So the second check is against the fp32 oracle: the
    split trunk must be as close to it as the fp32 trunk is.  Edge tiles, ragged sizes and the in-place update of the
    RRDB input by rdb3 are all exercised."""
    sd = synthetic_rrdbnet_state(nb, scale, seed=5)
    frame = synthetic_frames(1, H, W, seed=6)[0]
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("FW_RRDB_SPLIT_TRUNK", mode)
        eng = R.RRDBNetEngine(nb, scale, dtype)
        eng.load_state_dict(sd)
        outs.append(_gpu_rgb_f32(eng, frame))
        eng.close()
    (rgb_s, u8_s), (rgb_f, u8_f) = outs
    d = np.abs(rgb_s - rgb_f)
    assert d.max() < tol_max and d.mean() < tol_mean, (d.max(), d.mean())
    assert np.abs(u8_s.astype(int) - u8_f.astype(int)).max() <= 1
    want = _oracle_rgb_f32(sd, frame, nb, scale)
    e_s, e_f = np.abs(rgb_s - want).mean(), np.abs(rgb_f - want).mean()
    assert e_s < 1.25 * e_f + 1e-6, (e_s, e_f)


@pytest.mark.parametrize("H,W", [(1, 1), (2, 3), (7, 5), (14, 30), (15, 31), (16, 32), (17, 33), (29, 61), (100, 100), (333, 517)])
def test_fused_split_path_equals_plain_path_on_ragged_sizes(hip_lib, monkeypatch, H, W):
    """The default build (fused conv pairs on 14x30 tiles, split trunk, DMA batches with the border-tile fallback) against
    the plain one-conv-per-launch fp32-trunk path over sizes that put every kind of border tile in play: below one tile,
    exactly one pair tile, one pixel more than a tile in each direction, many tiles.  The two paths round at the same
    points except for the trunk representation, so they agree to rounding-flip noise; an addressing bug would not."""
    sd = synthetic_rrdbnet_state(2, 4, seed=H * 1000 + W)
    frame = synthetic_frames(1, H, W, seed=W)[0]
    outs = []
    for fuse, split in (("1", "1"), ("0", "0")):
        monkeypatch.setenv("FW_RRDB_FUSE_PAIRS", fuse)
        monkeypatch.setenv("FW_RRDB_SPLIT_TRUNK", split)
        eng = R.RRDBNetEngine(2, 4, "f16")
        eng.load_state_dict(sd)
        outs.append(_gpu_rgb_f32(eng, frame))
        eng.close()
    (rgb_a, u8_a), (rgb_b, u8_b) = outs
    assert np.isfinite(rgb_a).all()
    assert np.abs(rgb_a - rgb_b).max() < 2e-4
    assert np.abs(u8_a.astype(int) - u8_b.astype(int)).max() <= 1


def test_upscale_stream_overlapped_copies_equal_frame_by_frame(hip_lib):
    """The 3-stream host pipeline (upload / compute / download overlapped, pinned staging) returns, in order, exactly what
    the synchronous per-frame call returns."""
    sd = synthetic_rrdbnet_state(2, 4, seed=8)
    eng = R.RRDBNetEngine(2, 4, "bf16")
    eng.load_state_dict(sd)
    frames = list(synthetic_frames(7, 33, 47, seed=4))
    want = [eng.upscale(f) for f in frames]
    got = [o.copy() for o in eng.upscale_stream(frames, depth=2)]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert list(eng.upscale_stream([])) == []
    with pytest.raises(ValueError):
        list(eng.upscale_stream([frames[0], frames[1][:20]]))
    eng.close()


def test_upscale_host_buffers_and_determinism(hip_lib):
    sd = synthetic_rrdbnet_state(2, 4, seed=5)
    frame = synthetic_frames(1, 30, 50, seed=3)[0]
    eng = R.RRDBNetEngine(2, 4, "bf16")
    eng.load_state_dict(sd)
    a = eng.upscale(frame)
    b = eng.upscale(frame)
    assert a.shape == (120, 200, 3) and a.dtype == np.uint8
    assert np.array_equal(a, b)
    _, u8 = _gpu_rgb_f32(eng, frame)
    assert np.array_equal(a, u8)
    eng.close()


def test_enhance_matches_realesrganer_oracle_with_tiles(hip_lib):
    """HipRealESRGANer.enhance (tile / tile_pad / alpha / gray) vs oracle/realesrganer_ref.enhance."""
    nb = 2
    sd = synthetic_rrdbnet_state(nb, 4, seed=21)
    eng = R.RRDBNetEngine(nb, 4, "f16")
    eng.load_state_dict(sd)
    model = oref.make_model(_sd_t(sd), nb, 4)
    frame = synthetic_frames(1, 45, 70, seed=8)[0]
    for tile in (0, 32):
        up = R.HipRealESRGANer(4, eng, tile=tile, tile_pad=10)
        out, mode = up.enhance(frame, outscale=4)
        want, wmode = oref.enhance(model, frame, 4, outscale=4, tile=tile, tile_pad=10)
        assert mode == wmode == "RGB" and out.shape == want.shape == (180, 280, 3)
        assert np.abs(out.astype(int) - want.astype(int)).max() <= 1
    rgba = np.concatenate([frame, frame[:, :, :1]], axis=2)
    out, mode = R.HipRealESRGANer(4, eng).enhance(rgba, outscale=4)
    want, _ = oref.enhance(model, rgba, 4, outscale=4)
    assert mode == "RGBA" and out.shape == want.shape
    assert np.abs(out.astype(int) - want.astype(int)).max() <= 2
    gray = frame[:, :, 0]
    out, mode = R.HipRealESRGANer(4, eng).enhance(gray, outscale=4)
    want, _ = oref.enhance(model, gray, 4, outscale=4)
    assert mode == "L" and out.shape == want.shape == (180, 280)
    assert np.abs(out.astype(int) - want.astype(int)).max() <= 2
    eng.close()


def test_enhance_frame_pytorch_roundtrip(hip_lib, tmp_path, monkeypatch):
    """The reference's function boundary B1 end to end on PNG files (pytorch_realesrgan.py:176-247)."""
    from PIL import Image
    monkeypatch.setenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("FRAMEWRIGHT_MODEL_DIR", str(tmp_path / "nomodels"))
    frame = synthetic_frames(1, 24, 40, seed=4)[0]
    src = tmp_path / "frame_00000001.png"
    Image.fromarray(frame[:, :, ::-1]).save(src)
    cfg = R.PyTorchESRGANConfig(model_name="RealESRGAN_x4plus_anime_6B", scale_factor=4)
    ok, msg = R.enhance_frame_pytorch(src, tmp_path / "out.png", cfg)
    assert ok and msg is None
    out = np.asarray(Image.open(tmp_path / "out.png"))
    assert out.shape == (96, 160, 3)
    ok, msg = R.enhance_frame_pytorch(tmp_path / "nope.png", tmp_path / "o2.png", cfg)
    assert not ok and "Failed to read image" in msg
    R.clear_upsampler_cache()


def test_baseline_config0_x2_on_eight_256x256_frames(hip_lib, tmp_path, monkeypatch):
    """BASELINE.json configs[0]: Real-ESRGAN x2 (RealESRGAN_x2plus, 23 blocks) on 8 synthetic 256x256 frames through the
    reference's file boundary (`enhance_frame_pytorch`, one PNG in, one PNG out).  Every frame must come back 512x512;
    two of them are checked against the fp32 CPU oracle (PSNR >= 50 dB, the north-star bar, bf16 operands)."""
    from PIL import Image
    monkeypatch.setenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("FRAMEWRIGHT_MODEL_DIR", str(tmp_path / "nomodels"))
    frames = synthetic_frames(8, 256, 256, seed=17)
    cfg = R.PyTorchESRGANConfig(model_name="RealESRGAN_x2plus", scale_factor=2, dtype="bf16")
    outs = []
    for i, f in enumerate(frames):
        src, dst = tmp_path / f"frame_{i + 1:08d}.png", tmp_path / f"out_{i + 1:08d}.png"
        Image.fromarray(f[:, :, ::-1]).save(src)
        ok, msg = R.enhance_frame_pytorch(src, dst, cfg)
        assert ok and msg is None
        outs.append(np.asarray(Image.open(dst))[:, :, ::-1])
        assert outs[-1].shape == (512, 512, 3)
    sd = synthetic_rrdbnet_state(23, 2)      # the seeded weights FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS substitutes
    for i in (0, 5):
        want = _oracle_rgb_f32(sd, frames[i], 23, 2)
        want_u8 = np.rint(np.clip(want, 0, 1) * 255.0).astype(np.uint8)[:, :, ::-1]
        assert _psnr_u8(outs[i], want_u8) >= 50.0
    R.clear_upsampler_cache()


@pytest.mark.parametrize("H,W,y0,x0,dtype", [(1080, 1920, 400, 800, "f16"), (1080, 1920, 400, 800, "bf16"), (2160, 3840, 1900, 3500, "f16")])
def test_full_size_properties_x4(hip_lib, H, W, y0, x0, dtype):
    """BASELINE size (1920x1080 -> 7680x4320) and 4K (3840x2160 -> 15360x8640: 46 GB of workspace, byte offsets beyond
    2^32), 6-block model to keep the test short: properties that need no oracle.  f16 is the benched operand type (bench.py,
    every entry point's default); bf16 the opt-in one.
    (a) a crop far from the borders equals the same crop upscaled on its own with enough context (receptive field of
    the 6-block net + tail is < 100 px), (b) determinism."""
    nb = 6
    sd = synthetic_rrdbnet_state(nb, 4, seed=99)
    eng = R.RRDBNetEngine(nb, 4, dtype)
    eng.load_state_dict(sd)
    frame = synthetic_frames(1, H, W, seed=2)[0]
    t = torch.from_numpy(frame).cuda()
    full = eng.upscale_device(t)
    torch.cuda.synchronize()
    assert tuple(full.shape) == (4 * H, 4 * W, 3)
    sz, ctx = 64, 100
    crop = np.ascontiguousarray(frame[y0 - ctx:y0 + sz + ctx, x0 - ctx:x0 + sz + ctx])
    small = eng.upscale(crop)[ctx * 4:(ctx + sz) * 4, ctx * 4:(ctx + sz) * 4]
    big = full[y0 * 4:(y0 + sz) * 4, x0 * 4:(x0 + sz) * 4].cpu().numpy()
    assert np.abs(small.astype(int) - big.astype(int)).max() <= 1
    full2 = eng.upscale_device(t)
    torch.cuda.synchronize()
    assert torch.equal(full, full2)
    eng.close()


def test_out_of_memory_is_reported_with_the_word_memory(hip_lib, tmp_path, monkeypatch):
    """restorer.py:1746 retries with a smaller tile when the error text contains "memory"/"vram": a frame whose
    workspace cannot be allocated must come back as (False, "GPU out of memory ...") — not as a crash."""
    sd = synthetic_rrdbnet_state(1, 4, seed=1)
    eng = R.RRDBNetEngine(1, 4, "bf16")
    eng.load_state_dict(sd)
    assert eng.workspace_bytes(16384, 16384) > 400e9            # > 288 GB of HBM
    big = torch.empty((16384, 16384, 3), dtype=torch.uint8, device="cuda")
    out = torch.empty((8, 8, 3), dtype=torch.uint8, device="cuda")  # never written: the allocation fails first
    with pytest.raises(_lib.FramewrightOutOfMemory, match="memory"):
        eng._lib  # noqa: B018
        _lib.check(eng._lib.fw_rrdbnet_upscale_u8(eng._h, big.data_ptr(), _lib.FW_DEVICE, 16384, 16384, out.data_ptr(),
                                                  _lib.FW_DEVICE, None, None))
    # the engine is still usable afterwards
    small = synthetic_frames(1, 16, 16, seed=1)[0]
    assert eng.upscale(small).shape == (64, 64, 3)
    eng.close()


@pytest.mark.parametrize("scale", [4, 2])
def test_16bit_frames_follow_the_65535_branch(hip_lib, scale):
    """RealESRGANer.enhance on a 16-bit image (max > 256): /65535 in, clamp * 65535 round -> uint16 out, RGB / gray / RGBA;
    checked against the CPU oracle (oracle/realesrganer_ref.py takes the same branch)."""
    from oracle import realesrganer_ref as oref
    sd = synthetic_rrdbnet_state(2, scale, seed=21)
    eng = R.RRDBNetEngine(2, scale, "f16")
    eng.load_state_dict(sd)
    model = oref.make_model({k: torch.from_numpy(v) for k, v in sd.items()}, 2, scale)
    f8 = synthetic_frames(1, 37, 50, seed=13)[0]
    rng = np.random.default_rng(5)
    f16 = (f8.astype(np.uint16) << 8) | rng.integers(0, 256, f8.shape, dtype=np.uint16)      # uses all 16 bits
    up = R.HipRealESRGANer(scale, eng)
    out, mode = up.enhance(f16, outscale=scale)
    want, wmode = oref.enhance(model, f16, scale, outscale=scale)
    assert out.dtype == np.uint16 and mode == wmode == "RGB" and out.shape == want.shape == (37 * scale, 50 * scale, 3)
    d = np.abs(out.astype(np.int64) - want.astype(np.int64))
    assert d.max() <= 1.0e-3 * 65535, d.max()                   # 1e-3 on the [0,1] scale (f16 operands)
    mse = float(np.mean(d.astype(np.float64) ** 2))
    assert 10 * np.log10(65535.0 ** 2 / max(mse, 1e-12)) >= 60.0
    # the same picture as 8-bit data differs from the 16-bit result only by the input quantisation
    out8, _ = up.enhance(f8, outscale=scale)
    assert out8.dtype == np.uint8 and np.abs(out8.astype(np.int64) - (out.astype(np.int64) + 128) // 257).max() <= 16
    g16 = f16[:, :, 1]
    og, mg = up.enhance(g16, outscale=scale)
    wg, _ = oref.enhance(model, g16, scale, outscale=scale)
    assert mg == "L" and og.dtype == np.uint16 and np.abs(og.astype(np.int64) - wg.astype(np.int64)).max() <= 1.0e-3 * 65535 + 1
    rgba = np.concatenate([f16, f16[:, :, :1]], axis=2)
    oa, ma = up.enhance(rgba, outscale=scale)
    assert ma == "RGBA" and oa.shape[2] == 4 and oa.dtype == np.uint16 and np.array_equal(oa[:, :, :3], out)
    # outscale != netscale on the 16-bit frame: the uint16 result resized by OpenCV's float Lanczos path (round 1 raised here)
    from oracle import lanczos_ref
    half, mh = up.enhance(f16, outscale=scale / 2)
    hw = (int(37 * scale / 2), int(50 * scale / 2))
    assert mh == "RGB" and half.dtype == np.uint16 and half.shape == (hw[0], hw[1], 3)
    assert np.array_equal(half, lanczos_ref.resize_lanczos4_u16(out, hw[1], hw[0]))
    eng.close()


def test_hipgraph_capture_gives_identical_frames(hip_lib, monkeypatch):
    """FW_RRDB_GRAPH: the per-frame forward replayed from a captured hipGraph (small frames by default) == launched directly;
    host and device buffers, a changed frame size and a re-used graph."""
    import time
    sd = synthetic_rrdbnet_state(3, 4, seed=8)
    frames = synthetic_frames(3, 48, 64, seed=3)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("FW_RRDB_GRAPH", mode)
        eng = R.RRDBNetEngine(3, 4, "bf16")
        eng.load_state_dict(sd)
        res = [eng.upscale(f) for f in frames] + [eng.upscale(frames[0][:40, :52].copy())] + [eng.upscale(f) for f in frames[:2]]
        d = torch.from_numpy(frames[1]).cuda()
        o = torch.empty((192, 256, 3), dtype=torch.uint8, device="cuda")
        for _ in range(3):
            eng.upscale_device(d, out=o)
        torch.cuda.synchronize()
        res.append(o.cpu().numpy())
        t0 = time.perf_counter()
        for _ in range(20):
            eng.upscale_device(d, out=o)
        torch.cuda.synchronize()
        outs[mode] = (res, (time.perf_counter() - t0) / 20 * 1e3)
        eng.close()
    for a, b in zip(outs["0"][0], outs["1"][0]):
        assert np.array_equal(a, b)
    print(f"48x64 x4, 3 blocks: direct {outs['0'][1]:.3f} ms, graph {outs['1'][1]:.3f} ms")


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_sliding_window_pair_kernel_equals_ring_kernel_at_1080p(hip_lib, monkeypatch, dtype):
    """Three independent implementations of the fused conv pair (conv3x3_pair_slide32.hip, conv3x3_pair_slide.hip - the
    default at this size -, conv3x3_pair.hip) through the whole 23-block x4 network on a BASELINE-size frame: every byte of the
    7680x4320 output must agree (same per-pixel accumulation order)."""
    sd = synthetic_rrdbnet_state(23, 4, seed=1234)
    eng = R.RRDBNetEngine(23, 4, dtype)      # f16: exactly the network, size and operand type bench.py times
    eng.load_state_dict(sd)
    d = torch.from_numpy(synthetic_frames(1, 1080, 1920, seed=2)[0]).cuda()
    outs = []
    for mode in ("2", "1", "0"):
        monkeypatch.setenv("FW_PAIR_SLIDE", mode)
        o = torch.empty((4320, 7680, 3), dtype=torch.uint8, device="cuda")
        eng.upscale_device(d, out=o)
        torch.cuda.synchronize()
        outs.append(o)
    assert torch.equal(outs[0], outs[2])
    assert torch.equal(outs[1], outs[2])
    assert int(outs[0].max()) > int(outs[0].min())          # not a constant image
    eng.close()


def test_thread_pool_on_one_shared_upsampler_with_an_injected_oom(hip_lib, tmp_path, monkeypatch):
    """The reference drives enhance_frame_pytorch from ThreadPoolExecutor(parallel_frames) against ONE shared upsampler
    (restorer.py:1830-1973, B1 of SURVEY.md section 8(b)): every frame equals the serial result, and an out-of-memory frame on
    one worker - whose handler clears the upsampler cache (pytorch_realesrgan.py:237-244) - neither crashes the workers that
    are inside the engine at that moment nor leaves them with a destroyed handle: they finish or report, the next call
    rebuilds the upsampler."""
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    monkeypatch.setenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("FRAMEWRIGHT_MODEL_DIR", str(tmp_path / "none"))
    R.clear_upsampler_cache()
    cfg = R.PyTorchESRGANConfig(model_name="RealESRGAN_x4plus_anime_6B", scale_factor=4)
    frames = synthetic_frames(12, 36, 52, seed=77)
    for i, f in enumerate(frames):
        Image.fromarray(f[:, :, ::-1]).save(tmp_path / f"in_{i:02d}.png")
    up = R.get_upsampler(cfg)
    serial = [up.enhance(f, outscale=4)[0] for f in frames]

    # (1) one handle, many threads through .enhance
    with ThreadPoolExecutor(max_workers=4) as ex:
        got = list(ex.map(lambda f: up.enhance(f, outscale=4)[0], list(frames) * 2))
    for k, g in enumerate(got):
        assert np.array_equal(g, serial[k % len(frames)])

    # (2) enhance_frame_pytorch from a pool, one frame failing with OOM in the middle
    real_enhance = R.HipRealESRGANer.enhance
    hit = {"n": 0}

    def flaky(self, img, outscale=None, alpha_upsampler="realesrgan"):
        hit["n"] += 1
        if hit["n"] == 5:
            raise _lib.FramewrightOutOfMemory(_lib.FW_ERR_OOM, "GPU out of memory: injected")
        return real_enhance(self, img, outscale, alpha_upsampler)

    monkeypatch.setattr(R.HipRealESRGANer, "enhance", flaky)

    def job(i):
        return i, R.enhance_frame_pytorch(tmp_path / f"in_{i:02d}.png", tmp_path / f"out_{i:02d}.png", cfg)

    with ThreadPoolExecutor(max_workers=4) as ex:
        results = dict(ex.map(job, range(len(frames))))
    failed = [i for i, (ok, _) in results.items() if not ok]
    assert len(failed) >= 1
    msgs = [results[i][1] for i in failed]
    assert any("memory" in m.lower() for m in msgs)            # restorer.py:1746's tile downshift keys on this word
    for i, (ok, err) in results.items():
        if ok:
            out = np.asarray(Image.open(tmp_path / f"out_{i:02d}.png"))[:, :, ::-1]
            assert np.array_equal(out, serial[i]), i
        else:
            assert isinstance(err, str) and err                   # reported, never raised
    # the cache was cleared by the OOM handler; the next call builds a fresh upsampler and works
    monkeypatch.setattr(R.HipRealESRGANer, "enhance", real_enhance)
    ok, err = R.enhance_frame_pytorch(tmp_path / "in_00.png", tmp_path / "again.png", cfg)
    assert ok and err is None
    assert np.array_equal(np.asarray(Image.open(tmp_path / "again.png"))[:, :, ::-1], serial[0])
    # clearing the cache only drops it (pytorch_realesrgan.py:250-260): a worker that already holds the upsampler finishes its
    # frames on it, the next get_upsampler builds a new one; an engine closed explicitly refuses calls instead of touching freed memory
    held = R.get_upsampler(cfg)
    R.clear_upsampler_cache()
    assert np.array_equal(held.enhance(frames[1], outscale=4)[0], serial[1])
    fresh = R.get_upsampler(cfg)
    assert fresh is not held and np.array_equal(fresh.enhance(frames[1], outscale=4)[0], serial[1])
    eng = held.engine
    eng.close()
    with pytest.raises(_lib.FramewrightHipError, match="closed"):
        eng.upscale(frames[0])
    R.clear_upsampler_cache()

    # one worker runs out of memory while its siblings are QUEUED on the shared upsampler's lock (they fetched it before the
    # failing frame cleared the cache): every queued frame still comes back, from the object they hold
    up = R.get_upsampler(cfg)
    gate = threading.Event()
    hit["n"] = 0

    def oom_first(self, img, outscale=None, alpha_upsampler="realesrgan"):
        hit["n"] += 1
        if hit["n"] == 1:
            gate.wait(5.0)                       # the siblings pile up behind this frame ...
            raise _lib.FramewrightOutOfMemory(_lib.FW_ERR_OOM, "GPU out of memory: injected")
        return real_enhance(self, img, outscale, alpha_upsampler)

    monkeypatch.setattr(R.HipRealESRGANer, "enhance", oom_first)
    with ThreadPoolExecutor(max_workers=4) as ex:
        futs = [ex.submit(job, i) for i in range(4)]
        time.sleep(0.3)
        gate.set()
        results = dict(f.result() for f in futs)
    assert sum(1 for ok, _ in results.values() if not ok) == 1
    for i, (ok, err) in results.items():
        if ok:
            assert np.array_equal(np.asarray(Image.open(tmp_path / f"out_{i:02d}.png"))[:, :, ::-1], serial[i]), i
        else:
            assert "memory" in err.lower()
    del up
    R.clear_upsampler_cache()


def test_gfpgan_background_upsampler(hip_lib, tmp_path, monkeypatch):
    """`FaceRestorer._get_bg_upsampler` (face_restore.py:379-401) hands GFPGAN a RealESRGANer(scale=4, x4plus, tile=400, tile_pad=10,
    pre_pad=0, half=True) or None: the drop-in shares the x4plus engine of `get_upsampler` but keeps its own tiling."""
    monkeypatch.setenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("FRAMEWRIGHT_MODEL_DIR", str(tmp_path / "none"))
    R.clear_upsampler_cache()
    assert R.get_bg_upsampler(None) is None and R.get_bg_upsampler("none") is None and R.get_bg_upsampler("") is None
    bg = R.get_bg_upsampler("realesrgan")
    main = R.get_upsampler(R.PyTorchESRGANConfig(model_name="RealESRGAN_x4plus", scale_factor=4, tile_size=0))
    assert isinstance(bg, R.HipRealESRGANer) and bg.engine is main.engine
    assert (bg.scale, bg.tile_size, bg.tile_pad, bg.pre_pad, bg.half) == (4, 400, 10, 0, True) and main.tile_size == 0
    frame = synthetic_frames(1, 40, 56, seed=5)[0]
    a, mode = bg.enhance(frame, outscale=4)
    b, _ = main.enhance(frame, outscale=4)
    assert mode == "RGB" and a.shape == (160, 224, 3) and np.array_equal(a, b)      # below the tile size: one whole-frame forward
    R.clear_upsampler_cache()


@pytest.mark.parametrize("wino", ["1", "2"])
@pytest.mark.parametrize("num_block,scale,H,W", [(23, 4, 40, 56), (6, 4, 33, 47), (23, 2, 41, 57), (4, 4, 96, 150)])
def test_winograd_conv5_vs_oracle_and_vs_the_direct_kernel(hip_lib, monkeypatch, num_block, scale, H, W, wino):
    """conv5 of rdb1 / rdb2 as a row-wise Winograd F(2, 3) (conv3x3_wino.hip, FW_RRDB_C5_WINO=1: transformed weights rounded to f16 once,
    pixel differences formed in f16): the same 1e-3 / 60 dB bar against the fp32 oracle as the direct kernels, and within rounding of
    them (ragged sizes, odd widths, several tiles per workgroup row)."""
    sd = synthetic_rrdbnet_state(num_block, scale, seed=1234)
    frame = synthetic_frames(1, H, W, seed=H * W)[0]
    outs = {}
    for mode in (wino, "0"):        # "1": conv5 of rdb1 / rdb2, "2": rdb3's as well (residual planes R hi, R lo in frequencies 0 / 3)
        monkeypatch.setenv("FW_RRDB_C5_WINO", mode)
        eng = R.RRDBNetEngine(num_block, scale, "f16")
        eng.load_state_dict(sd)
        outs[mode] = _gpu_rgb_f32(eng, frame)
        eng.close()
    want = _oracle_rgb_f32(sd, frame, num_block, scale)
    (rgb, u8), (rgb_d, u8_d) = outs[wino], outs["0"]
    err, err_d = np.abs(rgb - want).max(), np.abs(rgb_d - want).max()
    print(f"nb={num_block} x{scale} {H}x{W}: winograd {err:.2e}, direct {err_d:.2e}, between them {np.abs(rgb - rgb_d).max():.2e}")
    assert not np.array_equal(rgb, rgb_d)                      # the knob took effect
    assert err < 1e-3 and np.abs(rgb - rgb_d).max() < 1e-3
    want_u8 = (np.clip(want, 0, 1) * 255.0).round().astype(np.uint8)[:, :, ::-1]
    assert _psnr_u8(u8, want_u8) >= 60.0 and np.abs(u8.astype(int) - want_u8.astype(int)).max() <= 1
