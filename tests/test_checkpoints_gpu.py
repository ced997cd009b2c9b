"""Checkpoints ON DISK through the reference's own entry points (round 2 exercised them with FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS=1
only, except SRVGG): the wrappers the upstream loaders unwrap, keys a `strict=False` load tolerates, the DataParallel prefix.

  RRDBNet   `params_ema` / `params` / raw     RealESRGANer's loader (third-party; call site pytorch_realesrgan.py:160-170)
  NAFNet    `params` / `state_dict` / raw     tap_denoise.py:348-355 (`strict=False`)
  Restormer the same                          tap_denoise.py:317-325
  IFNet     `module.` prefixed keys           interpolation.py:106-124 (flownet.pkl of rife-v4.6)
"""
import numpy as np
import pytest
import torch

from framewright_amd import realesrgan as R
from framewright_amd import rife as RF
from framewright_amd import tap_denoise as T
from framewright_amd.synth import synthetic_frames, synthetic_ifnet_state, synthetic_nafnet_state, synthetic_rrdbnet_state

pytestmark = pytest.mark.gpu


def _t(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


@pytest.mark.parametrize("wrap", ["params_ema", "params", "raw"])
def test_rrdbnet_pth_through_get_upsampler(hip_lib, tmp_path, monkeypatch, wrap):
    monkeypatch.delenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", raising=False)
    monkeypatch.setenv("FRAMEWRIGHT_MODEL_DIR", str(tmp_path))
    R.clear_upsampler_cache()
    sd = synthetic_rrdbnet_state(6, 4, seed=4321)            # NOT the seed the synthetic switch would substitute
    ck = _t(sd)
    if wrap == "params_ema":
        ck = {"params": _t(synthetic_rrdbnet_state(6, 4, seed=1)), "params_ema": ck}     # params_ema wins, as in RealESRGANer
    elif wrap == "params":
        ck = {"params": ck}
    torch.save(ck, str(tmp_path / R.MODEL_FILES["RealESRGAN_x4plus_anime_6B"]))
    frame = synthetic_frames(1, 32, 44, seed=8)[0]
    got = R.get_upsampler(R.PyTorchESRGANConfig(model_name="RealESRGAN_x4plus_anime_6B", scale_factor=4)).enhance(frame, outscale=4)[0]
    eng = R.RRDBNetEngine(6, 4, "f16")
    eng.load_state_dict(sd)
    assert np.array_equal(got, eng.upscale(frame))
    eng.close()
    R.clear_upsampler_cache()
    # no file and no synthetic switch: a FileNotFoundError that names the path, not a silent fallback
    monkeypatch.setenv("FRAMEWRIGHT_MODEL_DIR", str(tmp_path / "empty"))
    with pytest.raises(FileNotFoundError, match="RealESRGAN_x4plus_anime_6B.pth"):
        R.get_upsampler(R.PyTorchESRGANConfig(model_name="RealESRGAN_x4plus_anime_6B", scale_factor=4))
    R.clear_upsampler_cache()


@pytest.mark.parametrize("wrap", ["params", "state_dict", "raw"])
def test_nafnet_pth_through_tap_denoiser(hip_lib, tmp_path, monkeypatch, wrap):
    monkeypatch.delenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", raising=False)
    sd = synthetic_nafnet_state(seed=77, **T.NAFNET_ARGS)
    ck = _t(sd)
    ck["not_a_nafnet_key.weight"] = torch.zeros(3)          # strict=False: unexpected keys are ignored (tap_denoise.py:355)
    if wrap != "raw":
        ck = {wrap: ck}
    torch.save(ck, str(tmp_path / T.TAPDenoiser.MODEL_FILES[T.TAPModel.NAFNET]))
    frame = synthetic_frames(1, 48, 64, seed=9)[0]
    den = T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet", tile_size=0, temporal_window=1), model_dir=tmp_path)
    got = den._denoise_frame_tiled(frame)
    eng = T.NAFNetEngine(dtype="f16", **T.NAFNET_ARGS)
    eng.load_state_dict(sd)
    assert np.array_equal(got, eng.denoise(frame))
    eng.close()
    den.clear_cache()
    with pytest.raises(FileNotFoundError, match="NAFNet-SIDD-width64.pth"):
        T.TAPDenoiser(T.TAPDenoiseConfig(model="nafnet"), model_dir=tmp_path / "empty")._denoise_frame_tiled(frame)


@pytest.mark.parametrize("wrap", ["params", "state_dict"])
def test_restormer_pth_through_tap_denoiser(hip_lib, tmp_path, monkeypatch, wrap):
    from framewright_amd.restormer import RESTORMER_ARGS, RestormerEngine, synthetic_restormer_state
    monkeypatch.delenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", raising=False)
    sd = synthetic_restormer_state(seed=5, **RESTORMER_ARGS)
    ck = _t(sd)
    ck["some.unexpected.buffer"] = torch.zeros(1)
    torch.save({wrap: ck}, str(tmp_path / T.TAPDenoiser.MODEL_FILES[T.TAPModel.RESTORMER]))
    frame = synthetic_frames(1, 40, 56, seed=10)[0]
    den = T.TAPDenoiser(T.TAPDenoiseConfig(tile_size=0, temporal_window=1), model_dir=tmp_path)      # RESTORMER is the default model
    assert den.config.model == T.TAPModel.RESTORMER
    got = den._denoise_frame_tiled(frame)
    eng = RestormerEngine(dtype="f16", **RESTORMER_ARGS)
    eng.load_state_dict(sd)
    assert np.array_equal(got, eng.denoise(frame))
    eng.close()
    den.clear_cache()


def test_ifnet_flownet_pkl_with_module_prefix(hip_lib, tmp_path, monkeypatch):
    monkeypatch.delenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", raising=False)
    monkeypatch.setenv("FRAMEWRIGHT_MODEL_DIR", str(tmp_path))
    sd = synthetic_ifnet_state(seed=1357)
    (tmp_path / "rife-v4.6").mkdir()
    torch.save({"module." + k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, str(tmp_path / "rife-v4.6" / "flownet.pkl"))
    a, b = synthetic_frames(2, 64, 96, seed=11)
    fi = RF.FrameInterpolator("rife-v4.6", 0)
    got = fi._get_engine().interpolate(a, b, 0.5)
    eng = RF.IFNetEngine("f16", 0)
    eng.load_state_dict(sd)
    assert np.array_equal(got, eng.interpolate(a, b, 0.5))
    eng.close()
    monkeypatch.setenv("FRAMEWRIGHT_MODEL_DIR", str(tmp_path / "empty"))
    with pytest.raises(RF.InterpolationError, match="flownet.pkl"):
        RF.FrameInterpolator("rife-v4.6", 0)._get_engine()
