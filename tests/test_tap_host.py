"""TAP driver host logic (no GPU): tile grid, temporal weights, config validation — mirrors the reference's
tests/test_processors/test_tap_denoise.py (defaults / validation / MODEL_FILES) and pins the integer bookkeeping of
reference tap_denoise.py:435-450,508-528 on values worked out by hand from those formulas."""
import numpy as np
import pytest
import torch

from framewright_amd import tap_denoise as T
from oracle import tap_ref


def test_config_defaults_and_validation():
    c = T.TAPDenoiseConfig()
    assert (c.temporal_window, c.strength, c.preserve_grain, c.half_precision, c.tile_size, c.tile_overlap, c.gpu_id,
            c.batch_size) == (5, 1.0, False, True, 512, 32, 0, 1)
    assert T.TAPDenoiseConfig(model="nafnet").model is T.TAPModel.NAFNET
    for bad in (dict(temporal_window=0), dict(strength=1.5), dict(strength=-0.1), dict(tile_size=-1), dict(tile_overlap=-1)):
        with pytest.raises(ValueError):
            T.TAPDenoiseConfig(**bad)


def test_model_files_match_reference():
    assert T.TAPDenoiser.MODEL_FILES[T.TAPModel.NAFNET] == "NAFNet-SIDD-width64.pth"
    assert T.TAPDenoiser.MODEL_FILES[T.TAPModel.RESTORMER] == "restormer_deraining.pth"
    assert T.TAPDenoiser.DEFAULT_MODEL_DIR.parts[-3:] == (".framewright", "models", "tap")
    assert [m.value for m in T.TAPModel] == ["restormer", "nafnet", "tap"]


def test_tile_grid_1080p_is_3_by_4():
    g = T.tile_grid(1080, 1920, 512, 32)           # stride 480: h: (1048//480)+1 = 3, w: (1888//480)+1 = 4
    assert len(g) == 12
    assert sorted({y for y, _ in g}) == [0, 480, 568]     # last row clamps to h - tile
    assert sorted({x for _, x in g}) == [0, 480, 960, 1408]
    assert g == tap_ref.tile_grid(1080, 1920, 512, 32)
    assert T.tile_grid(512, 512, 512, 32) == [(0, 0)]
    assert T.tile_grid(992, 512, 512, 32) == [(0, 0), (480, 0)]   # (960 % 480 == 0) -> exactly 2 rows
    with pytest.raises(ValueError):
        T.tile_grid(100, 100, 32, 32)


def test_temporal_weights():
    s, e, w = T.temporal_window(10, 5, 5)
    assert (s, e) == (3, 8)
    raw = np.array([0.5, 1 / 1.5, 1.0, 1 / 1.5, 0.5])
    assert np.allclose(w, raw / raw.sum())
    s, e, w = T.temporal_window(10, 0, 5)          # clip start clamps the window (tap_denoise.py:510)
    assert (s, e) == (0, 3) and np.allclose(w, np.array([1.0, 1 / 1.5, 0.5]) / (1 + 1 / 1.5 + 0.5))
    s, e, w = T.temporal_window(10, 9, 5)
    assert (s, e) == (7, 10)
    assert T.temporal_window(4, 2, 1) == (2, 3, [1.0])
    assert tap_ref.temporal_weights(10, 5, 5) == T.temporal_window(10, 5, 5)


def test_oracle_truncates_like_the_reference():
    t = torch.tensor([0.9999, 0.5, -0.2, 1.3]).view(1, 1, 2, 2).repeat(1, 3, 1, 1)
    out = tap_ref.postprocess(t)
    assert out[0, 0, 0] == 254 and out[0, 1, 0] == 127 and out[1, 0, 0] == 0 and out[1, 1, 0] == 255


def test_unaccelerated_models_are_reported_unavailable():
    assert T.TAPDenoiseConfig().model is T.TAPModel.RESTORMER          # the reference's default (tap_denoise.py:110)
    d = T.TAPDenoiser(T.TAPDenoiseConfig(model=T.TAPModel.TAP))
    assert d.is_available() is False


# ---- pinned on outputs of the reference's own code (tests/golden/tap_reference.*, oracle/gen_golden.py tap_logic) ----
def _tap_golden():
    import json
    from pathlib import Path
    g = Path(__file__).parent / "golden"
    return np.load(g / "tap_reference.npz"), json.loads((g / "tap_reference.json").read_text())


def test_oracle_temporal_window_matches_reference_run():
    """oracle/tap_ref.py's window selection, weights, float32 accumulate and truncating cast against what the reference's
    `_denoise_with_temporal_window` returned in the build container (identity per-frame step)."""
    arrs, meta = _tap_golden()
    identity = lambda t: t
    for w in meta["windows"]:
        frames = list(arrs[w["key"] + "_frames"])
        want = arrs[w["key"] + "_out"]
        for i in range(w["n"]):
            s, e, ws = tap_ref.temporal_weights(w["n"], i, w["window"])
            got = tap_ref.temporal_average(frames[s:e], ws) if w["window"] > 1 else frames[i]
            np.testing.assert_array_equal(got, want[i])
            s2, e2, ws2 = T.temporal_window(w["n"], i, w["window"])
            assert (s2, e2) == (s, e) and np.allclose(ws2, ws, rtol=0, atol=0)


def test_mirrors_match_reference_tables():
    _, meta = _tap_golden()
    c = T.TAPDenoiseConfig()
    got = {"model": c.model.value, "temporal_window": c.temporal_window, "strength": c.strength, "preserve_grain": c.preserve_grain,
           "half_precision": c.half_precision, "tile_size": c.tile_size, "tile_overlap": c.tile_overlap, "gpu_id": c.gpu_id,
           "batch_size": c.batch_size}
    assert got == meta["config_defaults"]
    assert {k.value: v for k, v in T.TAPDenoiser.MODEL_FILES.items()} == meta["model_files"]
    assert {k.value: v for k, v in T.TAPDenoiser.MODEL_VRAM.items()} == meta["model_vram"]
    assert [lv.value for lv in T.MotionLevel] == meta["motion_levels"]
    for row in meta["motion_strength"]:
        m = T.MotionAdaptiveTAPDenoiser(T.MotionAdaptiveConfig(**row["config"]))
        for lv in T.MotionLevel:
            assert m.get_motion_adjusted_strength(lv) == row["strength"][lv.value]      # bit for bit
    assert T.create_tap_denoiser("nafnet", strength=0.5).config.model is T.TAPModel.NAFNET
    with pytest.raises(ValueError):
        T.MotionAdaptiveConfig(base_strength=1.5)


def _tile_flow_golden():
    import json
    from pathlib import Path
    g = Path(__file__).parent / "golden"
    return np.load(g / "tile_flow_reference.npz"), json.loads((g / "tile_flow_reference.json").read_text())


def test_oracle_tile_blend_matches_reference_run():
    """oracle/tap_ref.py's tile grid, ramps, float32 accumulate and truncating cast against the reference's
    `_denoise_frame_tiled` run with identity pre / model / post steps (tests/golden/tile_flow_reference.npz)."""
    arrs, meta = _tile_flow_golden()
    to_t = lambda f: torch.from_numpy(np.ascontiguousarray(f))
    for c in meta["tiled"]:
        frame, want = arrs[c["key"] + "_in"], arrs[c["key"] + "_out"]
        # identity "model" on the oracle's own pre/post: postprocess(preprocess(tile)) returns the tile for uint8 input
        got = tap_ref.denoise_frame_tiled(lambda t: t, frame, c["tile_size"], c["overlap"])
        np.testing.assert_array_equal(got, want)


def test_oracle_flow_accumulate_matches_reference_run():
    """oracle/temporal_ref.py's weighting / accumulate / cast against the reference's `_denoise_with_flow` and
    `_denoise_simple` (flow fields, aligned frames and one failing neighbour supplied through a stand-in estimator)."""
    from oracle import temporal_ref
    arrs, meta = _tile_flow_golden()
    for c in meta["flow"]:
        k, n = c["key"], c["n"]
        frames = [arrs[f"{k}_frame{i}"] for i in range(n)]
        # the oracle warps by itself; to feed it the recorded aligned frames, pass zero flow and the aligned frame as "frame"
        window, flows = [], []
        for i in range(n):
            if i == c["center"]:
                window.append(frames[i]); flows.append(None)
            elif i == c["failing"]:
                window.append(frames[i]); flows.append(None)
            else:
                z = np.zeros(frames[i].shape[:2], np.float32)
                window.append(arrs[f"{k}_aligned{i}"])
                flows.append(dict(flow_x=z, flow_y=z, magnitude=arrs[f"{k}_mag{i}"], confidence=arrs[f"{k}_conf{i}"]))
        np.testing.assert_array_equal(temporal_ref.denoise_with_flow(c["center"], window, flows, c["decay"]), arrs[k + "_out"])
        np.testing.assert_array_equal(temporal_ref.denoise_simple(frames, c["decay"]), arrs[k + "_simple"])


def test_grain_oracle_known_answers():
    """The numpy restatement of the preserve_grain arithmetic (OpenCV 8-bit paths; unpinned: cv2 is absent)."""
    import numpy as np
    from oracle import tap_ref
    k = tap_ref.gaussian_kernel_fixed_point()
    assert k.tolist() == [0, 1, 3, 4, 9, 14, 20, 28, 32, 34, 32, 28, 20, 14, 9, 4, 3, 1, 0] and int(k.sum()) == 256
    flat = np.full((9, 13, 3), 77, np.uint8)
    assert np.all(tap_ref.bgr2gray_u8(flat) == 77)                                    # the 14-bit weights sum to 2^14
    assert np.all(tap_ref.gaussian_blur_u8_sigma3(tap_ref.bgr2gray_u8(flat)) == 77)  # taps sum to 256: constants survive
    den = np.full_like(flat, 10)
    assert np.array_equal(tap_ref.grain_addback(flat, den), den)                     # no high-pass content, nothing added
    spike = flat.copy()
    spike[4, 6] = 255
    out = tap_ref.grain_addback(spike, den)
    gray = tap_ref.bgr2gray_u8(spike)
    blurred = tap_ref.gaussian_blur_u8_sigma3(gray)
    # the spike: (255*34*34 + 77*(65536 - 34*34) + 2^15) >> 16 = 80 -> grain 175 -> int(175 * 0.3) = 52
    assert int(blurred[4, 6]) == 80 and out[4, 6].tolist() == [62, 62, 62]
    assert np.array_equal(out[0, 0], den[0, 0])                                      # away from the spike: blurred >= gray
    assert tap_ref.bgr2gray_u8(np.array([[[255, 0, 0]]], np.uint8))[0, 0] == 29      # blue weight 1868 / 16384


def test_restated_architectures_match_the_published_sizes():
    """The third-party nets are absent (parity unpinned); the only upstream facts available offline are their published
    sizes.  The shape tables the engines AND the oracles are built from must reproduce them with the constructor arguments the
    reference passes (tap_denoise.py:340-346, :304-315): NAFNet-SIDD-width64 115.98 M parameters and 63.6 GMACs at 256 x 256
    (NAFNet paper, table 1: convolutions; a profiler's count adds the element-wise ops), Restormer 26.13 M parameters."""
    import re

    import numpy as np

    from framewright_amd.restormer import RESTORMER_ARGS, restormer_tensor_shapes
    from framewright_amd.synth import nafnet_tensor_shapes
    from framewright_amd.tap_denoise import NAFNET_ARGS

    naf = nafnet_tensor_shapes(**NAFNET_ARGS)
    assert abs(sum(int(np.prod(s)) for _, s in naf) / 1e6 - 115.98) < 0.005

    def level(key):      # resolution level a tensor's convolution runs at (level l = 1 / 2^l of the frame)
        for pat, f in ((r"encoders\.(\d+)\.", lambda i: i), (r"downs\.(\d+)\.", lambda i: i + 1), (r"ups\.(\d+)\.", lambda i: 4 - i),
                       (r"decoders\.(\d+)\.", lambda i: 3 - i)):
            m = re.match(pat, key)
            if m:
                return f(int(m.group(1)))
        return 4 if key.startswith("middle_blks") else 0

    macs = sum(int(np.prod(s)) * (1 if ".sca." in k else (256 >> level(k)) ** 2) for k, s in naf if k.endswith("weight") and len(s) == 4)
    assert 63.0 < macs / 1e9 < 64.0, macs / 1e9          # 63.24 G in the convolutions; published 63.6
    rest = restormer_tensor_shapes(**RESTORMER_ARGS)
    assert abs(sum(int(np.prod(s)) for _, s in rest) / 1e6 - 26.13) < 0.005
