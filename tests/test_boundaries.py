"""Boundary mirrors: interface shape on the CPU, end-to-end behaviour on the GPU."""
import inspect

import numpy as np
import pytest

from framewright_amd import backends as B
from framewright_amd import plugins as PL
from framewright_amd import realesrgan as R
from framewright_amd.synth import synthetic_frames


def test_sr_backend_interface_matches_reference_abc():
    # reference processors/enhancement/super_resolution.py:237-311
    for name in ("name", "supported_scales", "is_available", "estimate_vram_usage", "upscale_frame", "upscale_frames"):
        assert name in B.SRBackend.__abstractmethods__
    assert list(inspect.signature(B.SRBackend.upscale_frames).parameters) == ["self", "input_dir", "output_dir", "scale",
                                                                              "progress_callback"]
    b = B.HipRealESRGANBackend(model_variant="x2plus")
    assert b.name == "realesrgan_hip_x2plus" and b.supported_scales == [2]
    assert B.HipRealESRGANBackend().estimate_vram_usage(1920, 1080, 4) > 10_000     # MB: the real buffer plan
    with pytest.raises(ValueError):
        B.HipRealESRGANBackend(model_variant="nope")
    r = B.SRResult()
    assert (r.frames_processed, r.backend_used, r.scale_factor, r.warnings) == (0, "unknown", 4, [])


def test_denoiser_backend_interface():
    assert {"name", "is_available", "process"} <= B.DenoiserBackend.__abstractmethods__
    b = B.HipTAPDenoiserBackend()
    assert b.name == "tap_hip"

    class Cfg:
        temporal_radius, strength, preserve_grain, half_precision, tile_size, gpu_id = 1, 0.5, False, True, 0, 0
    dn = b._get_denoiser(Cfg())
    assert dn.config.temporal_window == 3 and dn.config.strength == 0.5 and dn.config.tile_size == 512


def test_plugin_metadata_and_contract():
    m = PL.RealESRGANPlugin.get_metadata()
    assert m.name == "realesrgan_mi355x" and PL.PluginCapability.UPSCALE in m.capabilities and m.supports_cpu is False
    assert PL.PluginCapability.DENOISE in PL.TAPDenoisePlugin.get_metadata().capabilities
    p = PL.RealESRGANPlugin()
    assert p.is_initialized is False and p.get_temporal_radius() == 0
    assert p.estimate_output_size((1080, 1920)) == (4320, 7680)
    assert PL.TAPDenoisePlugin().get_temporal_radius() == 2 and PL.TAPDenoisePlugin().supports_batch()
    with pytest.raises(RuntimeError, match="GPU device string"):
        p.initialize("cpu")
    r = PL.RIFEInterpolatePlugin()
    assert PL.PluginCapability.INTERPOLATE in r.get_metadata().capabilities and r.get_temporal_radius() == 1 and r.supports_batch()


def test_shard_process_fn_reports_errors_like_reference(tmp_path, hip_lib):
    fn = B.make_shard_process_fn()
    out, ok, err = fn(tmp_path / "missing.png", tmp_path, 0)
    assert out == tmp_path / "missing.png" and ok is False and err


@pytest.mark.gpu
def test_backends_and_plugins_end_to_end(hip_lib, tmp_path, monkeypatch):
    from PIL import Image
    monkeypatch.setenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("FRAMEWRIGHT_MODEL_DIR", str(tmp_path / "nomodels"))
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    frames = synthetic_frames(2, 24, 32, seed=6)
    for i, f in enumerate(frames):
        Image.fromarray(f[:, :, ::-1]).save(src / f"frame_{i + 1:08d}.png")
    b = B.HipRealESRGANBackend(model_variant="anime")
    assert b.is_available()
    seen = []
    res = b.upscale_frames(src, dst, 4, seen.append)
    assert res.frames_processed == 2 and res.frames_failed == 0 and res.avg_fps > 0 and seen[-1] == 1.0
    direct = b.upscale_frame(frames[0])
    assert direct.shape == (96, 128, 3)
    assert np.array_equal(np.asarray(Image.open(dst / "frame_00000001.png"))[:, :, ::-1], direct)
    pl = PL.RealESRGANPlugin()
    pl.initialize("cuda:0", {"model_name": "RealESRGAN_x4plus_anime_6B"})
    assert np.array_equal(pl.process_frame(frames[0], 0), direct)
    assert len(pl.process_batch(list(frames), 0)) == 2
    pl.cleanup()
    fn = B.make_device_process_func(R.PyTorchESRGANConfig(model_name="RealESRGAN_x4plus_anime_6B"))
    assert np.array_equal(fn(frames[0], 0), direct)
    proc = B.RealESRGANProcessor("realesrgan-x4plus-anime", 4, 0)
    assert proc.process_frame(str(src / "frame_00000002.png"), str(tmp_path / "p.png"))
    b.clear_cache()


def test_rocm_compute_backend_interface():
    """`HipRocmBackend` has every abstract method of the reference's `Backend` ABC (infrastructure/gpu/backends/base.py:
    89-205) and behaves like it when nothing is loaded."""
    from framewright_amd import backends as B
    for m in ("backend_type", "name", "is_initialized", "initialize", "cleanup", "get_capabilities", "allocate_memory",
              "free_memory", "get_memory_info", "load_model", "unload_model", "run_inference", "__enter__", "__exit__"):
        assert hasattr(B.HipRocmBackend, m), m
    b = B.HipRocmBackend(device_id=0)
    assert b.backend_type == "rocm" and not b.is_initialized
    assert b.load_model("RealESRGAN_x4plus") is False                 # not initialised
    with pytest.raises(ValueError, match="not loaded"):
        b.run_inference("RealESRGAN_x4plus", np.zeros((4, 4, 3), np.uint8))
    caps = B.BackendCapabilities(name="x")
    assert set(caps.to_dict()) == {"name", "backend_type", "vendor", "supports_fp16", "supports_int8", "max_memory_mb", "max_batch_size"}


@pytest.mark.gpu
def test_rocm_compute_backend_runs_models(hip_lib, tmp_path, monkeypatch):
    from framewright_amd import backends as B, realesrgan as R
    from framewright_amd.synth import synthetic_frames
    monkeypatch.setenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("FRAMEWRIGHT_MODEL_DIR", str(tmp_path / "nomodels"))
    f0, f1 = synthetic_frames(2, 32, 48, seed=3)
    with B.HipRocmBackend(0) as b:
        assert b.is_initialized and b.get_capabilities().max_memory_mb > 100000 and b.allocate_memory(1024)
        assert b.load_model("RealESRGAN_x4plus_anime_6B") and b.load_model("rife-v4.6") and not b.load_model("no-such-model")
        up = b.run_inference("RealESRGAN_x4plus_anime_6B", f0)
        assert up.shape == (128, 192, 3) and up.dtype == np.uint8
        mid = b.run_inference("rife-v4.6", (f0, f1))
        assert mid.shape == f0.shape
        b.unload_model("rife-v4.6")
        with pytest.raises(ValueError):
            b.run_inference("rife-v4.6", (f0, f1))
    assert not b.is_initialized
    R.clear_upsampler_cache()


@pytest.mark.gpu
def test_rife_plugin_matches_the_engine(hip_lib, monkeypatch):
    import numpy as np
    from framewright_amd import rife as RF
    from framewright_amd.synth import synthetic_frames, synthetic_ifnet_state
    monkeypatch.setenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", "1")
    frames = list(synthetic_frames(3, 64, 96, seed=11))
    pl = PL.RIFEInterpolatePlugin()
    pl.initialize("cuda:0", {"dtype": "f16"})
    eng = RF.IFNetEngine("f16", 0)
    eng.load_state_dict(synthetic_ifnet_state())
    mid = pl.process_frame(frames[0], 0, {"next_frame": frames[1]})
    assert np.array_equal(mid, eng.interpolate(frames[0], frames[1]))
    assert np.array_equal(pl.process_frame(frames[2], 2), frames[2])          # end of the clip: the frame itself
    seq = pl.process_batch(frames, 0)
    assert len(seq) == 5 and np.array_equal(seq[1], mid) and np.array_equal(seq[2], frames[1])
    pl.cleanup()
