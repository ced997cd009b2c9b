"""cv2.INTER_LANCZOS4 resize (RealESRGANer.enhance's outscale != netscale tail): the numpy oracle's known answers on the
CPU, the device kernel bit-exact against the oracle on the GPU."""
import numpy as np
import pytest

from oracle import lanczos_ref as L


def test_oracle_identity_constant_and_phase_weights():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    assert np.array_equal(L.resize_lanczos4_u8(a, 30, 20), a)                 # same size: tap 3 has weight 1
    c = np.full((17, 23, 4), 201, np.uint8)
    assert np.all(L.resize_lanczos4_u8(c, 40, 9) == 201)                       # constants survive the 11-bit weights
    ofs, coef = L.tables(8, 4)                                                 # 2:1: every output sits at phase 0.5
    assert list(ofs) == [0, 2, 4, 6]
    assert coef[0].tolist() == [-26, 122, -340, 1267, 1267, -340, 122, -26]
    w = L.interpolate_lanczos4(np.float32(0.25))
    assert abs(float(w.sum()) - 1.0) < 1e-6 and int(np.argmax(w)) == 3
    g = rng.integers(0, 256, (9, 11), dtype=np.uint8)                          # gray images keep their rank
    assert L.resize_lanczos4_u8(g, 5, 4).shape == (4, 5)
    # the float path of 16-bit images: identity at equal size, constants within one unit of float rounding, shape, dtype
    a16 = rng.integers(0, 65536, (20, 30, 3), dtype=np.uint16)
    assert np.array_equal(L.resize_lanczos4_u16(a16, 30, 20), a16)
    c16 = np.full((17, 23, 3), 51234, np.uint16)
    r16 = L.resize_lanczos4_u16(c16, 40, 9)
    assert r16.dtype == np.uint16 and r16.shape == (9, 40, 3) and np.abs(r16.astype(int) - 51234).max() <= 1
    ofs, cf = L.float_tables(8, 4)
    assert list(ofs) == [0, 2, 4, 6] and abs(float(cf[0].sum()) - 1.0) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dst", [((64, 96, 3), (48, 32)), ((33, 47, 3), (94, 66)), ((40, 40, 1), (13, 71)),
                                       ((25, 31, 4), (31, 25)), ((1, 1, 3), (5, 4)), ((160, 224, 3), (112, 80))])
def test_device_resize_is_bit_exact(shape, dst):
    from framewright_amd.realesrgan import resize_lanczos4_u8
    rng = np.random.default_rng(sum(shape) + dst[0])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    if shape[2] == 1:
        img = img[:, :, 0]
    got = resize_lanczos4_u8(img, dst[0], dst[1])
    want = L.resize_lanczos4_u8(img, dst[0], dst[1])
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dst", [((64, 96, 3), (48, 32)), ((33, 47, 3), (94, 66)), ((40, 40, 1), (13, 71)), ((25, 31, 4), (31, 25)),
                                       ((1, 1, 3), (5, 4))])
def test_device_resize_of_16_bit_images_is_bit_exact(shape, dst):
    from framewright_amd.realesrgan import resize_lanczos4_u8
    rng = np.random.default_rng(sum(shape) + dst[0] + 7)
    img = rng.integers(0, 65536, shape, dtype=np.uint16)
    if shape[2] == 1:
        img = img[:, :, 0]
    got = resize_lanczos4_u8(img, dst[0], dst[1])
    want = L.resize_lanczos4_u16(img, dst[0], dst[1])
    assert got.dtype == np.uint16 and got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
def test_enhance_outscale_2_on_a_16_bit_frame():
    """RealESRGANer.enhance on a 16-bit image (max_range 65535) with outscale != netscale: uint16 out, resized by the float path."""
    from framewright_amd.realesrgan import HipRealESRGANer, RRDBNetEngine
    from framewright_amd.synth import synthetic_frames, synthetic_rrdbnet_state
    eng = RRDBNetEngine(2, 4, "f16")
    eng.load_state_dict(synthetic_rrdbnet_state(2, 4, seed=5))
    up = HipRealESRGANer(4, eng)
    frame = (synthetic_frames(1, 36, 52, seed=9)[0].astype(np.uint16) * 257)
    full, _ = up.enhance(frame, outscale=4)
    half, mode = up.enhance(frame, outscale=2)
    assert mode == "RGB" and full.dtype == half.dtype == np.uint16 and half.shape == (72, 104, 3)
    assert np.array_equal(half, L.resize_lanczos4_u16(full, 104, 72))
    eng.close()


@pytest.mark.gpu
def test_enhance_outscale_2_with_x4_model():
    """`upsampler.enhance(img, outscale=2)` on a x4 network = the x4 result resized by INTER_LANCZOS4 to 2x the input."""
    from framewright_amd.realesrgan import HipRealESRGANer, RRDBNetEngine
    from framewright_amd.synth import synthetic_frames, synthetic_rrdbnet_state
    eng = RRDBNetEngine(2, 4, "f16")
    eng.load_state_dict(synthetic_rrdbnet_state(2, 4, seed=5))
    up = HipRealESRGANer(4, eng)
    frame = synthetic_frames(1, 36, 52, seed=9)[0]
    full, _ = up.enhance(frame, outscale=4)
    half, mode = up.enhance(frame, outscale=2)
    assert mode == "RGB" and half.shape == (72, 104, 3)
    assert np.array_equal(half, L.resize_lanczos4_u8(full, 104, 72))
    rgba = np.concatenate([frame, frame[:, :, :1]], axis=2)
    out4, mode4 = up.enhance(rgba, outscale=3)
    assert mode4 == "RGBA" and out4.shape == (108, 156, 4)
    eng.close()
