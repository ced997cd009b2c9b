"""The host side of AESRGANFaceRestorer (reference src/framewright/processors/aesrgan_face.py:51-136, 270-760) without a GPU: the mirror's
interface against what `inspect` read off the reference's own classes (tests/golden/face_reference.json, oracle/gen_golden.py
face_restorer_logic), `_extract_face` against the reference's own run, and the oracle's restatement of `cv2.resize` / the feathered
blend (oracle/face_ref.py, unpinned: no cv2) against known answers."""
import dataclasses
import inspect
import json

import numpy as np
import pytest

from framewright_amd import aesrgan as A
from oracle import face_ref as F


@pytest.fixture(scope="module")
def golden(golden_dir):
    return json.loads((golden_dir / "face_reference.json").read_text())


def test_interface_matches_the_reference_classes(golden):
    for name, want in golden["classes"].items():
        cls = getattr(A, name)
        if "fields" in want:
            got = {f.name: (f.default.value if hasattr(f.default, "value") else f.default) for f in dataclasses.fields(cls)}
            got = {k: ("<required>" if v is dataclasses.MISSING else v) for k, v in got.items()}
            assert got == want["fields"], name
        for meth, params in want["methods"].items():
            if params == ["<property>"]:
                assert isinstance(getattr(cls, meth), property), (name, meth)
                continue
            if meth in ("_detect_retinaface", "_detect_opencv"):
                continue                      # the two third-party detector back ends: replaced by an injected callable
            assert hasattr(cls, meth), (name, meth)
            ours = list(inspect.signature(getattr(cls, meth)).parameters)
            assert ours[:len(params)] == params, (name, meth, ours)       # extra keyword parameters (detect_fn, engine) come last
    assert {m.name: m.value for m in A.FaceDetectorType} == golden["FaceDetectorType"]
    assert A.AESRGANFaceRestorer.MODEL_FILE == golden["constants"]["MODEL_FILE"]
    assert list(A.AESRGANFaceRestorer.DEFAULT_MODEL_DIR.parts[-3:]) == golden["constants"]["DEFAULT_MODEL_DIR_tail"]
    assert list(inspect.signature(A.create_aesrgan_restorer).parameters) == golden["factory_params"]


def test_config_validation_messages(golden):
    for kw_json, want in golden["validate"].items():
        kw = json.loads(kw_json)
        if want["ok"]:
            assert A.AESRGANFaceConfig(**kw).face_detector.value == want["face_detector"]
        else:
            with pytest.raises(ValueError) as e:
                A.AESRGANFaceConfig(**kw)
            assert f"ValueError: {e.value}" == want["error"]


def test_extract_face_equals_the_reference_run(golden, tmp_path):
    r = A.AESRGANFaceRestorer(A.AESRGANFaceConfig(), model_dir=tmp_path)      # no GPU here: the restorer reports itself unavailable, like the reference without cv2
    rng = np.random.default_rng(5)
    for case in golden["extract"]:
        frame = rng.integers(0, 256, size=(case["h"], case["w"], 3), dtype=np.uint8)
        fb = A.FaceBox(*case["box"], confidence=0.9)
        assert (fb.width, fb.height, list(fb.center)) == (case["width"], case["height"], case["center"])
        for crop, region in (r._extract_face(frame, fb, padding=case["padding"]), F.extract_face(frame, tuple(case["box"]), case["padding"])):
            assert [int(v) for v in region] == case["region"] and list(crop.shape) == case["crop_shape"]
            assert int(crop.astype(np.int64).sum()) == case["crop_sum"]


def test_resize_linear_known_answers():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(24, 36, 3), dtype=np.uint8)
    assert np.array_equal(F.resize_linear_u8(img, 36, 24), img)                                    # equal size: a copy
    const = np.full((10, 14, 3), 77, np.uint8)
    for dw, dh in ((7, 5), (28, 20), (9, 33), (1, 1)):
        assert (F.resize_linear_u8(const, dw, dh) == 77).all()                                     # constants stay constant
    half = F.resize_linear_u8(img, 18, 12)                                                         # exact 2:1 -> the 2 x 2 mean, rounded half up
    want = (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.array_equal(half, want.astype(np.uint8))
    row = np.arange(0, 64, dtype=np.uint8)[None, :].repeat(8, 0)                                   # 4:1 sits half way between pixels 4d+1 and 4d+2
    q = F.resize_linear_u8(row, 16, 2)
    assert np.array_equal(q[0], ((row[0, 1::4].astype(int) + row[0, 2::4] + 1) // 2).astype(np.uint8))
    up = F.resize_linear_u8(np.array([[0, 100]], np.uint8), 4, 1)                                  # 1:2 up: 0, 25, 75, 100
    assert up.tolist() == [[0, 25, 75, 100]]
    assert F.resize_linear_u8(img[:, :, 0], 20, 9).shape == (9, 20)                                # 2-D images keep their rank


def test_feather_mask_and_blend_properties():
    m = F.feather_mask(40, 64)
    assert m.dtype == np.float32 and m.shape == (40, 64)
    assert (m[0] == 0).all() and (m[-1] == 0).all() and (m[:, 0] == 0).all() and (m[:, -1] == 0).all()     # i = 0 -> alpha 0
    assert (m[5:-5, 5:-5] == 1).all() and np.array_equal(m, m[::-1]) and np.array_equal(m, m[:, ::-1])        # feather = 40 // 8 = 5
    assert m[2, 20] == np.float32(2 / 5) and m[2, 3] == np.float32(np.float32(2 / 5) * np.float32(3 / 5))
    assert (F.feather_mask(7, 7) == 1).all()                                                                  # min(w, h) // 8 == 0: no feather
    rng = np.random.default_rng(2)
    frame = rng.integers(0, 256, size=(60, 80, 3), dtype=np.uint8)
    enh = rng.integers(0, 256, size=(80, 96, 3), dtype=np.uint8)
    out = F.paste_face_back(frame, enh, (10, 5, 58, 45), 0.8)
    assert np.array_equal(out[:5], frame[:5]) and np.array_equal(out[:, :10], frame[:, :10])                 # nothing outside the region
    assert np.array_equal(out[5, 10:58], frame[5, 10:58])                                                     # mask 0 on the region's border
    assert np.array_equal(F.paste_face_back(frame, enh, (10, 5, 58, 45), 0.0), frame)                         # strength 0: the frame


def test_checkpoint_keys_and_detector_boxes():
    """AESRGAN.state_dict() numbers RRDBs and AttentionBlocks in one ModuleList (aesrgan_face.py:228-235): 4 blocks with 2 attention
    blocks behind RRDB 0 and 2 -> modules 0 R, 1 A, 2 R, 3 R, 4 A, 5 R."""
    sd = {"conv_first.weight": 1}
    layout = ["R", "A", "R", "R", "A", "R"]
    for m, kind in enumerate(layout):
        if kind == "R":
            sd[f"body.{m}.rdb1.conv1.weight"] = f"rrdb{m}"
        else:
            sd[f"body.{m}.query.weight"] = f"attn{m}"
            sd[f"body.{m}.gamma"] = f"gamma{m}"
    trunk, attn = A.split_aesrgan_checkpoint(sd, 4, 2)
    assert trunk["conv_first.weight"] == 1
    assert [trunk[f"body.{i}.rdb1.conv1.weight"] for i in range(4)] == ["rrdb0", "rrdb2", "rrdb3", "rrdb5"]
    assert attn == {"attn.0.query.weight": "attn1", "attn.0.gamma": "gamma1", "attn.2.query.weight": "attn4", "attn.2.gamma": "gamma4"}
    det = A.FaceDetector(A.FaceDetectorType.OPENCV, 0, lambda f: [(1, 2, 30, 40), (5, 6, 7, 8, 0.25), A.FaceBox(0, 0, 9, 9, 0.5)])
    boxes = det.detect(np.zeros((50, 50, 3), np.uint8))
    assert [(b.x1, b.y1, b.x2, b.y2, b.confidence) for b in boxes] == [(1, 2, 30, 40, 1.0), (5, 6, 7, 8, 0.25), (0, 0, 9, 9, 0.5)]
    assert A.FaceDetector(A.FaceDetectorType.RETINAFACE).detect(np.zeros((4, 4, 3), np.uint8)) == []     # no back end: no faces (the reference's fallback)
