"""SRVGGNetCompact (realesr-animevideov3 / realesr-general-x4v3) on the GPU against the fp32 CPU oracle
(oracle/srvgg_ref.py; parity unpinned at the third-party boundary, see its header).  Tolerances as for the RRDBNet path:
f16 operands (the default) 1e-3 max-abs on the [0,1] float output; bf16 operands are opt-in: 1e-2 and PSNR >= 50 dB are regression
bounds for that dtype, not the parity bar."""
import math

import numpy as np
import pytest
import torch

from framewright_amd import srvgg as S
from framewright_amd.synth import synthetic_frames
from oracle import srvgg_ref as ref

pytestmark = pytest.mark.gpu


def _oracle(sd, frame, num_conv, scale):
    x = torch.from_numpy(frame[:, :, ::-1].astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    with torch.no_grad():
        y = ref.srvgg_forward(sdt, x, num_conv, scale)
    return y.squeeze(0).permute(1, 2, 0).numpy()


@pytest.mark.parametrize("dtype,max_abs,min_psnr", [("f16", 1e-3, 60.0), ("bf16", 1e-2, 50.0)])
@pytest.mark.parametrize("model,H,W", [("realesr-animevideov3", 37, 53), ("realesr-general-x4v3", 20, 33)])
def test_srvgg_vs_oracle(hip_lib, dtype, max_abs, min_psnr, model, H, W):
    num_conv, scale = S.SRVGG_MODELS[model]
    sd = S.synthetic_srvgg_state(num_conv, scale, seed=3)
    frame = synthetic_frames(1, H, W, seed=8)[0]
    eng = S.SRVGGNetEngine(num_conv, scale, dtype)
    eng.load_state_dict(sd)
    t = torch.from_numpy(frame).cuda()
    rgb = torch.empty((H * scale, W * scale, 3), dtype=torch.float32, device="cuda")
    u8 = eng.upscale_device(t, out_rgb_f32=rgb)
    torch.cuda.synchronize()
    eng.close()
    want = _oracle(sd, frame, num_conv, scale)
    got = rgb.cpu().numpy()
    assert np.abs(got - want).max() < max_abs
    want_u8 = np.rint(np.clip(want, 0, 1) * 255.0).astype(np.uint8)[:, :, ::-1]
    mse = np.mean((u8.cpu().numpy().astype(np.float64) - want_u8.astype(np.float64)) ** 2)
    assert (99.0 if mse == 0 else 10 * math.log10(255.0 ** 2 / mse)) >= min_psnr
    assert want.std() > 0.05  # the synthetic net is not degenerate


def test_srvgg_16_bit_frame_through_the_engine_and_the_upsampler(hip_lib):
    """A 16-bit frame (RealESRGANer.enhance: max_range 65535) on an SRVGG checkpoint: uint16 in, uint16 out, the same network on
    input / 65535; equal to the fp32 oracle within the f16 bar, and what HipRealESRGANer.enhance returns for it."""
    from framewright_amd.realesrgan import HipRealESRGANer
    num_conv, scale = S.SRVGG_MODELS["realesr-animevideov3"]
    sd = S.synthetic_srvgg_state(num_conv, scale, seed=3)
    f8 = synthetic_frames(1, 24, 40, seed=8)[0]
    f16 = f8.astype(np.uint16) * 257 + 13                                         # not a multiple of 257: real 16-bit content
    eng = S.SRVGGNetEngine(num_conv, scale, "f16")
    eng.load_state_dict(sd)
    got = eng.upscale(f16)
    assert got.dtype == np.uint16 and got.shape == (24 * scale, 40 * scale, 3)
    x = torch.from_numpy(f16[:, :, ::-1].astype(np.float32) / 65535.0).permute(2, 0, 1).unsqueeze(0)
    with torch.no_grad():
        want = ref.srvgg_forward({k: torch.from_numpy(v) for k, v in sd.items()}, x, num_conv, scale).squeeze(0).permute(1, 2, 0).numpy()
    want16 = np.rint(np.clip(want, 0, 1) * 65535.0).astype(np.int64)[:, :, ::-1]
    assert np.abs(got.astype(np.int64) - want16).max() <= 66                      # 1e-3 of the range
    out, mode = HipRealESRGANer(scale, eng).enhance(f16)
    assert mode == "RGB" and np.array_equal(out, got)
    eng.close()


def test_srvgg_rejects_wrong_state(hip_lib):
    eng = S.SRVGGNetEngine(16, 4, "bf16")
    sd = S.synthetic_srvgg_state(16, 4)
    del sd["body.3.weight"]
    with pytest.raises(S.FramewrightHipError, match="missing body.3.weight"):
        eng.load_state_dict(sd)
    sd = S.synthetic_srvgg_state(16, 4)
    sd["body.2.weight"] = sd["body.2.weight"][:, :32]
    with pytest.raises(S.FramewrightHipError, match="expected shape"):
        eng.load_state_dict(sd)
    eng.close()


def test_get_upsampler_routes_srvgg_checkpoints(hip_lib, tmp_path, monkeypatch):
    """A checkpoint with SRVGG keys under the name realesr-animevideov3 gets the SRVGG engine (the published file is one);
    an RRDB-shaped checkpoint under the same name still gets the RRDBNet the reference declares."""
    from framewright_amd import realesrgan as R
    num_conv, scale = S.SRVGG_MODELS["realesr-animevideov3"]
    sd = S.synthetic_srvgg_state(num_conv, scale, seed=11)
    path = tmp_path / "realesr-animevideov3.pth"
    torch.save({"params": {k: torch.from_numpy(v) for k, v in sd.items()}}, str(path))
    R.clear_upsampler_cache()
    cfg = R.PyTorchESRGANConfig(model_name="realesr-animevideov3", dtype="f16", model_path=str(path))
    up = R.get_upsampler(cfg)
    assert isinstance(up.engine, S.SRVGGNetEngine)
    frame = synthetic_frames(1, 24, 31, seed=2)[0]
    out, mode = up.enhance(frame)
    assert mode == "RGB" and out.shape == (96, 124, 3) and out.dtype == np.uint8
    want = np.rint(np.clip(_oracle(sd, frame, num_conv, scale), 0, 1) * 255.0).astype(np.uint8)[:, :, ::-1]
    assert np.abs(out.astype(int) - want.astype(int)).max() <= 1
    R.clear_upsampler_cache()
    monkeypatch.setenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", "1")
    up2 = R.get_upsampler(R.PyTorchESRGANConfig(model_name="realesr-animevideov3", dtype="f16",
                                                model_path=str(tmp_path / "absent.pth")))
    assert isinstance(up2.engine, R.RRDBNetEngine)
    R.clear_upsampler_cache()
