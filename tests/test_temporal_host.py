"""Known answers for the cv2 restatements behind `_preserve_edges` (oracle/temporal_ref.py; OpenCV is absent, parity unpinned):
the published constants and a few hand-computable cases."""
import numpy as np

from oracle import temporal_ref as ref


def test_bgr2gray_weights():
    assert ref.bgr2gray_u8(np.array([[[255, 255, 255]]], np.uint8))[0, 0] == 255           # the 14-bit weights sum to 2^14
    assert ref.bgr2gray_u8(np.array([[[255, 0, 0]]], np.uint8))[0, 0] == 29                # 0.114 * 255
    assert ref.bgr2gray_u8(np.array([[[0, 255, 0]]], np.uint8))[0, 0] == 150               # 0.587 * 255
    assert ref.bgr2gray_u8(np.array([[[0, 0, 255]]], np.uint8))[0, 0] == 76                # 0.299 * 255


def test_canny_on_a_vertical_step():
    g = np.zeros((9, 12), np.uint8)
    g[:, 6:] = 200
    e = ref.canny_u8(g, 30, 90)
    # Sobel responds on the two columns next to the step with equal magnitude; "m > left and m >= right" keeps the left one
    assert (e[:, 5] == 255).all() and e[:, :5].max() == 0 and e[:, 6:].max() == 0
    assert ref.canny_u8(np.full((5, 5), 77, np.uint8), 30, 90).max() == 0
    assert ref.canny_u8(g, 5000, 9000).max() == 0                                          # thresholds above 4 * 255 * 2


def test_hysteresis_needs_a_strong_seed():
    g = np.full((12, 40), 100, np.uint8)
    g[6:, :] = 112                       # weak step: magnitude 48
    assert ref.canny_u8(g, 30, 90).max() == 0
    g[6:, :2] = 160                      # strong at one end: the whole chain becomes an edge
    assert (ref.canny_u8(g, 30, 90)[5:7, 5:35].max(axis=0) == 255).all()


def test_dilate_gaussian_and_blend():
    e = np.zeros((7, 7), np.uint8)
    e[3, 3] = 255
    d = ref.dilate3x3_u8(e)
    assert d[2:5, 2:5].min() == 255 and d.sum() == 9 * 255
    m = ref.gaussian5_f32(np.ones((6, 8), np.float32))
    assert np.array_equal(m, np.ones((6, 8), np.float32))                                   # the kernel sums to exactly 1
    imp = np.zeros((9, 9), np.float32)
    imp[4, 4] = 1
    k = ref.gaussian5_f32(imp)[4, 2:7]
    assert np.array_equal(k, np.array([1, 4, 6, 4, 1], np.float32) / 16 * np.float32(0.375))
    o, dn = np.full((8, 8, 3), 200, np.uint8), np.full((8, 8, 3), 100, np.uint8)
    assert np.array_equal(ref.preserve_edges(o, dn), dn)                                    # no edges: the denoised frame
