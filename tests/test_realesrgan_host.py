"""Host-side mirror of reference src/framewright/processors/pytorch_realesrgan.py — the parts that do not
need a GPU (mirrors the reference's own tests/test_processors/test_pytorch_realesrgan.py config/enum checks)."""
import numpy as np
import pytest

from framewright_amd import realesrgan as R
from framewright_amd import _lib
from framewright_amd.synth import rrdbnet_conv_shapes, synthetic_rrdbnet_state, synthetic_frames


def test_config_defaults_match_reference():
    c = R.PyTorchESRGANConfig()
    assert (c.model_name, c.scale_factor, c.tile_size, c.tile_pad, c.pre_pad, c.half_precision, c.gpu_id) == \
        ("RealESRGAN_x4plus", 4, 0, 10, 0, True, 0)
    c.validate()


@pytest.mark.parametrize("name", ["RealESRGAN_x4plus", "RealESRGAN_x4plus_anime_6B", "RealESRGAN_x2plus",
                                  "realesr-animevideov3", "realesr-general-x4v3"])
def test_valid_models(name):
    R.PyTorchESRGANConfig(model_name=name).validate()


def test_invalid_model_and_scale():
    with pytest.raises(ValueError, match="Invalid model"):
        R.PyTorchESRGANConfig(model_name="nope").validate()
    with pytest.raises(ValueError, match="Scale factor must be 2 or 4"):
        R.PyTorchESRGANConfig(scale_factor=3).validate()


def test_ncnn_name_mapping():
    assert R.convert_ncnn_model_name("realesrgan-x4plus") == "RealESRGAN_x4plus"
    assert R.convert_ncnn_model_name("realesrgan-x2plus") == "RealESRGAN_x2plus"
    assert R.convert_ncnn_model_name("realesrnet-x4plus") == "realesr-general-x4v3"
    assert R.convert_ncnn_model_name("unknown") == "RealESRGAN_x4plus"
    assert len(R.NCNN_TO_PYTORCH_MODEL) == 5


def test_state_dict_shapes_are_basicsr():
    shapes = rrdbnet_conv_shapes(23, 4)
    assert len(shapes) == 1 + 23 * 15 + 5 == 351
    assert shapes[0] == ("conv_first", 64, 3)
    assert ("body.22.rdb3.conv5", 64, 192) in shapes
    assert shapes[-1] == ("conv_last", 3, 64)
    assert rrdbnet_conv_shapes(23, 2)[0] == ("conv_first", 64, 12)
    sd = synthetic_rrdbnet_state(1, 4, seed=3)
    n_params = sum(v.size for v in sd.values())
    assert n_params == 719424 + (64 * 3 * 9 + 64) + 4 * (64 * 64 * 9 + 64) + (3 * 64 * 9 + 3)
    sd2 = synthetic_rrdbnet_state(1, 4, seed=3)
    assert all(np.array_equal(sd[k], sd2[k]) for k in sd)


def test_synthetic_clip_is_deterministic_and_moving():
    a = synthetic_frames(3, 48, 64, seed=2)
    b = synthetic_frames(3, 48, 64, seed=2)
    assert a.shape == (3, 48, 64, 3) and a.dtype == np.uint8
    assert np.array_equal(a, b)
    assert not np.array_equal(a[0], a[1])
    assert 20 < a.std() < 90


def test_no_gpu_means_loud_failure_not_fallback(hip_lib, tmp_path):
    if hip_lib.fw_device_count() > 0:
        pytest.skip("GPU present")
    assert R.is_pytorch_esrgan_available() is False
    with pytest.raises(_lib.FramewrightHipError, match="no CPU fallback"):
        R.RRDBNetEngine(2, 4)
    # reference contract: enhance_frame_pytorch never raises, it reports (False, message)
    from PIL import Image
    src = tmp_path / "frame_00000001.png"
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(src)
    ok, msg = R.enhance_frame_pytorch(src, tmp_path / "out.png", R.PyTorchESRGANConfig())
    assert ok is False and msg
    ok, msg = R.enhance_frame_pytorch(tmp_path / "missing.png", tmp_path / "out.png", R.PyTorchESRGANConfig())
    assert ok is False and "Failed to read image" in msg
