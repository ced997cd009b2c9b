"""AESRGANFaceRestorer on the GPU (reference src/framewright/processors/aesrgan_face.py:383-728): the two device kernels of its
paste-back against the oracle's restatement (bit for bit), `restore_frame` with injected face boxes against the oracle composed
around the engine's own enhanced crops, the directory driver."""
import ctypes as C

import numpy as np
import pytest
import torch

from framewright_amd import _lib
from framewright_amd import aesrgan as A
from framewright_amd.synth import synthetic_attention_state, synthetic_frames, synthetic_rrdbnet_state
from oracle import face_ref as F

pytestmark = pytest.mark.gpu
p = lambda t: C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("hs,ws,hd,wd,ch", [(24, 36, 24, 36, 3), (24, 36, 12, 18, 3), (64, 48, 16, 12, 3), (37, 53, 91, 70, 3), (91, 70, 37, 53, 3),
                                            (5, 7, 1, 1, 3), (1, 1, 9, 6, 3), (40, 40, 13, 77, 1), (33, 21, 50, 50, 4), (128, 96, 63, 49, 3)])
def test_resize_linear_u8_equals_the_oracle(hip_lib, hs, ws, hd, wd, ch):
    """cv2.resize's default (INTER_LINEAR) on 8-bit images: equal size, the 2:1 "area fast" case, 4:1, up, down, ragged, 1-4 channels."""
    rng = np.random.default_rng(hs * 100 + wd)
    img = rng.integers(0, 256, size=(hs, ws, ch), dtype=np.uint8)
    src = torch.from_numpy(img).cuda()
    dst = torch.zeros((hd, wd, ch), dtype=torch.uint8, device="cuda")
    _lib.check(hip_lib.fw_resize_linear_u8(p(src), hs, ws, ch, p(dst), hd, wd, None))
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), F.resize_linear_u8(img, wd, hd))


@pytest.mark.parametrize("H,W,region,strength", [(60, 80, (10, 5, 58, 45), 0.8), (60, 80, (0, 0, 80, 60), 1.0), (60, 80, (70, 50, 80, 60), 0.5),
                                                 (200, 300, (33, 21, 290, 199), 0.37), (48, 48, (8, 8, 15, 40), 0.8), (64, 64, (3, 4, 60, 11), 0.0)])
def test_face_paste_u8_equals_the_oracle(hip_lib, H, W, region, strength):
    """The feathered float32 blend, statement for statement (regions with and without a feather band: min(w, h) // 8 may be 0)."""
    rng = np.random.default_rng(H + region[2])
    frame = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    x1, y1, x2, y2 = region
    enh = rng.integers(0, 256, size=(y2 - y1, x2 - x1, 3), dtype=np.uint8)      # already at the region's size: the resize is the identity
    d = torch.from_numpy(frame.copy()).cuda()
    _lib.check(hip_lib.fw_face_paste_u8(p(d), H, W, x1, y1, x2, y2, p(torch.from_numpy(enh).cuda()), strength, None))
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), F.paste_face_back(frame, enh, region, strength))
    assert hip_lib.fw_face_paste_u8(p(d), H, W, x1, y1, W + 1, y2, p(d), 0.5, None) != _lib.FW_OK      # region outside the frame


@pytest.fixture(scope="module")
def small_engine(hip_lib):
    eng = A.AESRGANEngine(num_block=2, scale=2, num_attention=1, dtype="f16")
    eng.load_state_dict(synthetic_rrdbnet_state(2, 4, seed=31), synthetic_attention_state(2, 1))   # AESRGAN has no unshuffle front end: the x4 key set
    yield eng
    eng.close()


def test_restore_frame_with_injected_boxes(hip_lib, small_engine, tmp_path):
    """Two faces (one below the detection threshold, one clipped by the frame border): crop -> network -> truncating uint8 -> resize ->
    feathered blend, face by face on the running result, equals the oracle's arithmetic around the engine's own crops."""
    frame = synthetic_frames(1, 96, 128, seed=8)[0]
    boxes = [(20, 16, 52, 56, 0.95), (100, 60, 126, 94, 0.9), (5, 5, 20, 20, 0.3)]
    cfg = A.AESRGANFaceConfig(enhancement_strength=0.8, upscale_factor=2)
    r = A.AESRGANFaceRestorer(cfg, model_dir=tmp_path, detect_fn=lambda f: boxes, engine=small_engine)
    assert r.is_available()
    got, n = r.restore_frame(frame)
    assert n == 2 and got.shape == frame.shape and not np.array_equal(got, frame)
    want = frame.copy()
    for b in boxes[:2]:
        crop, region = F.extract_face(want, b[:4])
        x = torch.from_numpy(np.ascontiguousarray(crop[:, :, ::-1])).cuda().float() / 255.0
        enh = F.postprocess_truncating(small_engine.forward_rgb(x).cpu().numpy())
        assert enh.shape == (2 * crop.shape[0], 2 * crop.shape[1], 3)
        assert np.array_equal(r._enhance_face(crop), enh)
        want = F.paste_face_back(want, enh, region, 0.8)
    assert np.array_equal(got, want)
    # paste_back off: the faces are counted, the frame comes back unchanged
    r2 = A.AESRGANFaceRestorer(A.AESRGANFaceConfig(paste_back=False), model_dir=tmp_path, detect_fn=lambda f: boxes, engine=small_engine)
    same, n2 = r2.restore_frame(frame)
    assert n2 == 2 and np.array_equal(same, frame)
    # no detector back end (the reference without retinaface / cv2): no faces, the very same array
    r3 = A.AESRGANFaceRestorer(cfg, model_dir=tmp_path, engine=small_engine)
    out3, n3 = r3.restore_frame(frame)
    assert n3 == 0 and out3 is frame


def test_restore_faces_directory_driver(hip_lib, small_engine, tmp_path):
    from PIL import Image
    frames = synthetic_frames(3, 64, 80, seed=9)
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    for i, f in enumerate(frames):
        Image.fromarray(np.ascontiguousarray(f[:, :, ::-1])).save(src / f"frame_{i:08d}.png")
    (src / "frame_00000003.png").write_bytes(b"not a png")                      # a frame that fails is counted and copied through
    r = A.AESRGANFaceRestorer(A.AESRGANFaceConfig(), model_dir=tmp_path, detect_fn=lambda f: [(16, 12, 56, 52)], engine=small_engine)
    seen = []
    res = r.restore_faces(src, dst, progress_callback=seen.append)
    assert (res.frames_processed, res.frames_failed, res.faces_enhanced, res.output_dir) == (3, 1, 3, dst)
    assert seen == [0.25, 0.5, 0.75, 1.0] and (dst / "frame_00000003.png").read_bytes() == b"not a png"
    for i, f in enumerate(frames):
        got = np.asarray(Image.open(dst / f"frame_{i:08d}.png").convert("RGB"))[:, :, ::-1]
        assert np.array_equal(got, r.restore_frame(f)[0])
    assert r.restore_faces(tmp_path / "empty_missing", tmp_path / "o2").frames_processed == 0
