"""Classical motion-compensated temporal denoise kernels against oracle/temporal_ref.py (bit-exact: integer remap
arithmetic, float64 accumulation in the reference's order)."""
import numpy as np
import pytest

from framewright_amd import temporal_denoise as TD
from framewright_amd.synth import synthetic_frames
from oracle import temporal_ref as ref

pytestmark = pytest.mark.gpu


def _flow(rng, h, w, amp):
    fx = (rng.standard_normal((h, w)) * amp).astype(np.float32)
    fy = (rng.standard_normal((h, w)) * amp).astype(np.float32)
    mag = np.sqrt(fx ** 2 + fy ** 2)
    conf = rng.random((h, w)).astype(np.float32)
    return TD.FlowField(fx, fy, mag, conf)


@pytest.mark.parametrize("h,w,amp,inverse", [(37, 53, 1.5, False), (16, 16, 6.0, True), (5, 3, 12.0, False), (1, 9, 2.0, False)])
def test_warp_frame_bit_exact(hip_lib, h, w, amp, inverse):
    rng = np.random.default_rng(h * 100 + w)
    frame = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    fl = _flow(rng, h, w, amp)
    fl.flow_x[0, 0] = 0.0          # an exact grid point
    fl.flow_y[0, 0] = 0.0
    acc = TD.DeviceTemporalAccumulator()
    got = acc.warp_frame(frame, fl, inverse=inverse)
    np.testing.assert_array_equal(got, ref.warp_frame(frame, fl.flow_x, fl.flow_y, inverse=inverse))
    assert tuple(got[0, 0]) == tuple(frame[0, 0])


@pytest.mark.parametrize("n,center,decay", [(7, 3, 0.5), (4, 0, 0.2), (3, 2, 1.0)])
def test_denoise_with_flow_bit_exact(hip_lib, n, center, decay):
    rng = np.random.default_rng(n * 10 + center)
    frames = list(synthetic_frames(n, 41, 59, seed=n))
    flows = [None if i == center else _flow(rng, 41, 59, 2.0) for i in range(n)]
    failing = 1 if center != 1 else 0                 # flow estimation "fails" for one neighbour: unaligned fallback
    by_id = {id(f): fl for f, fl in zip(frames, flows)}

    def flow_fn(frame, center_frame):
        if frame is frames[failing]:
            raise RuntimeError("flow failed")
        return by_id[id(frame)]

    acc = TD.DeviceTemporalAccumulator(temporal_weight_decay=decay, flow_fn=flow_fn)
    got = acc.denoise_with_flow(center, frames)
    oflows = [None if (fl is None or i == failing) else dict(flow_x=fl.flow_x, flow_y=fl.flow_y, magnitude=fl.magnitude,
                                                              confidence=fl.confidence) for i, fl in enumerate(flows)]
    np.testing.assert_array_equal(got, ref.denoise_with_flow(center, frames, oflows, decay))


def test_denoise_simple_bit_exact_and_missing_flow_estimator(hip_lib):
    frames = list(synthetic_frames(5, 30, 44, seed=3))
    acc = TD.DeviceTemporalAccumulator(temporal_weight_decay=0.5)
    np.testing.assert_array_equal(acc.denoise_simple(frames), ref.denoise_simple(frames, 0.5))
    # no estimator: every neighbour falls back to the unaligned frame with the temporal weight only, as in the reference
    got = acc.denoise_with_flow(2, frames)
    np.testing.assert_array_equal(got, ref.denoise_with_flow(2, frames, [None] * 5, 0.5))


def test_flow_accumulate_matches_reference_run(hip_lib):
    """The device accumulate against outputs of the reference's own `_denoise_with_flow` / `_denoise_simple`
    (tests/golden/tile_flow_reference.npz; the aligned frames are the recorded ones, so cv2.remap is not involved)."""
    import json
    from pathlib import Path
    g = Path(__file__).parent / "golden"
    arrs, meta = np.load(g / "tile_flow_reference.npz"), json.loads((g / "tile_flow_reference.json").read_text())
    for c in meta["flow"]:
        k, n = c["key"], c["n"]
        frames = [arrs[f"{k}_frame{i}"] for i in range(n)]
        window = [frames[i] if i in (c["center"], c["failing"]) else arrs[f"{k}_aligned{i}"] for i in range(n)]
        z = np.zeros(frames[0].shape[:2], np.float32)
        by_id = {id(window[i]): TD.FlowField(z, z, arrs[f"{k}_mag{i}"], arrs[f"{k}_conf{i}"]) for i in range(n)}

        def flow_fn(frame, center, by_id=by_id, bad=window[c["failing"]]):
            if frame is bad:
                raise RuntimeError("flow failed")
            return by_id[id(frame)]

        acc = TD.DeviceTemporalAccumulator(temporal_weight_decay=c["decay"], flow_fn=flow_fn)
        np.testing.assert_array_equal(acc.denoise_with_flow(c["center"], window), arrs[k + "_out"])
        np.testing.assert_array_equal(acc.denoise_simple(frames), arrs[k + "_simple"])


@pytest.mark.parametrize("h,w,thr", [(40, 56, 30), (33, 70, 30), (64, 64, 10), (7, 9, 30), (1, 12, 30), (130, 97, 60)])
def test_preserve_edges_bit_exact(hip_lib, h, w, thr):
    """fw_preserve_edges_u8 (gray, Sobel, non-maximum suppression, hysteresis across 32x32 tiles, dilate, 5x5 Gaussian, blend)
    against oracle/temporal_ref.preserve_edges; shapes include several hysteresis tiles, ragged edges and a one-row image."""
    rng = np.random.default_rng(h * 1000 + w)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 128 + 70 * np.sin(xx / 5.0) * np.cos(yy / 7.0) + 60 * ((xx // 9 + yy // 11) % 2)
    orig = np.clip(np.stack([base + 10 * c for c in range(3)], 2) + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
    den = np.clip(orig.astype(int) + rng.integers(-25, 25, orig.shape), 0, 255).astype(np.uint8)
    acc = TD.DeviceTemporalAccumulator()
    got = acc.preserve_edges(orig, den, thr)
    want = ref.preserve_edges(orig, den, thr)
    np.testing.assert_array_equal(got, want)
    edges = ref.canny_u8(ref.bgr2gray_u8(orig), thr, 3 * thr)
    if h > 4:
        assert 0 < edges.mean() < 255                      # there are edges and non-edges: both blend branches ran


def test_preserve_edges_long_weak_chain_crosses_tiles(hip_lib):
    """One strong pixel at the left end of a long weak ridge: the hysteresis has to carry the edge through every 32-pixel tile."""
    h, w = 20, 200
    orig = np.full((h, w, 3), 100, np.uint8)
    orig[10:, :, :] = 112                                   # a weak horizontal step (L1 magnitude 4 * 12 = 48: above 30, below 90)
    orig[10:, :3, :] = 160                                  # ... that is strong at its left end (4 * 60 > 90)
    den = np.full_like(orig, 50)
    gray = ref.bgr2gray_u8(orig)
    e = ref.canny_u8(gray, 30, 90)
    assert e[:, 150:].max() == 255 and ref.canny_u8(gray[:, 8:], 30, 90).max() == 0     # the chain only exists through its strong end
    np.testing.assert_array_equal(TD.DeviceTemporalAccumulator().preserve_edges(orig, den, 30), ref.preserve_edges(orig, den, 30))
