"""The codec edge (framewright_amd/codec.py) on the CPU: the two pipe ends against Python child processes that stand
in for ffmpeg (absent from the image) and speak the same protocol - H * W * 3 bytes of bgr24 per frame, no header."""
import sys
import time

import numpy as np
import pytest

from framewright_amd import codec as K

H, W = 18, 26

PRODUCER = """
import sys, numpy as np
n, h, w, cut, rc = (int(a) for a in sys.argv[1:6])
rng = np.random.default_rng(5)
data = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8).tobytes()
sys.stdout.buffer.write(data[:len(data) - cut])
sys.stdout.buffer.flush()
sys.exit(rc)
"""

CONSUMER = """
import sys
with open(sys.argv[1], "wb") as f:
    while True:
        b = sys.stdin.buffer.read(1 << 16)
        if not b:
            break
        f.write(b)
sys.exit(int(sys.argv[2]))
"""


def _frames(n):
    return np.random.default_rng(5).integers(0, 256, size=(n, H, W, 3), dtype=np.uint8)


def test_command_lines_mirror_the_reference(tmp_path):
    # restorer.py:1110-1117 (input side) and :3000-3027 (framerate, flac audio, codec / crf / preset / pix_fmt / -y output)
    assert K.decode_command("in.mkv") == ["ffmpeg", "-i", "in.mkv", "-f", "rawvideo", "-pix_fmt", "bgr24", "-"]
    audio = tmp_path / "a.flac"
    audio.write_bytes(b"x")
    cmd = K.encode_command("out.mkv", 7680, 4320, 47.952, crf=16, preset="medium", audio_path=audio)
    assert cmd[:11] == ["ffmpeg", "-f", "rawvideo", "-pix_fmt", "bgr24", "-s", "7680x4320", "-framerate", "47.952", "-i", "-"]
    assert cmd[11:15] == ["-i", str(audio), "-c:a", "flac"]
    assert cmd[15:] == ["-c:v", "libx265", "-crf", "16", "-preset", "medium", "-pix_fmt", "yuv420p10le", "-y", "out.mkv"]
    assert "-c:a" not in K.encode_command("o.mkv", 4, 4, 24, audio_path=tmp_path / "absent.flac")


@pytest.mark.parametrize("n,depth", [(0, 1), (1, 1), (9, 2), (23, 4)])
def test_reader_yields_every_frame_of_the_pipe(n, depth):
    want = _frames(n)
    with K.RawVideoReader([sys.executable, "-c", PRODUCER, str(n), str(H), str(W), "0", "0"], H, W, depth=depth, pin=False) as r:
        got = []
        for f in r:
            assert f.shape == (H, W, 3) and f.dtype == np.uint8
            got.append(f.copy())
            if len(got) % 5 == 0:
                time.sleep(0.01)           # a slow consumer: the reader thread waits on its bounded queue, nothing is dropped
        assert r.frames_read == n
        with pytest.raises(StopIteration):
            next(r)
    assert len(got) == n and all(np.array_equal(a, b) for a, b in zip(got, want))


def test_reader_keeps_the_last_three_frames_valid():
    want = _frames(12)
    with K.RawVideoReader([sys.executable, "-c", PRODUCER, "12", str(H), str(W), "0", "0"], H, W, depth=2, pin=False) as r:
        held = []
        for i, f in enumerate(r):
            held.append((i, f))                # views, not copies
            for j, v in held[-3:]:
                assert np.array_equal(v, want[j])


def test_reader_reports_a_truncated_stream_and_a_failing_decoder():
    with K.RawVideoReader([sys.executable, "-c", PRODUCER, "4", str(H), str(W), "100", "0"], H, W, pin=False) as r:
        with pytest.raises(K.CodecError, match="inside frame 3"):
            list(r)
    with K.RawVideoReader([sys.executable, "-c", PRODUCER, "2", str(H), str(W), "0", "3"], H, W, pin=False) as r:
        with pytest.raises(K.CodecError, match="status 3"):
            list(r)
    with pytest.raises(ValueError):
        K.RawVideoReader([sys.executable, "-c", "pass"], 0, 4)


def test_writer_streams_frames_in_order_with_ready_and_release(tmp_path):
    frames = _frames(17)
    out = tmp_path / "raw.bgr"
    events = []
    with K.RawVideoWriter([sys.executable, "-c", CONSUMER, str(out), "0"], depth=2) as w:
        for i, f in enumerate(frames):
            w.write(f, ready=lambda i=i: events.append(("ready", i)), release=lambda i=i: events.append(("release", i)))
    assert w.frames_written == 17
    assert out.read_bytes() == frames.tobytes()
    assert events == [e for i in range(17) for e in (("ready", i), ("release", i))]      # per frame: wait for it, write it, free its slot
    with pytest.raises(ValueError):
        K.RawVideoWriter(open(tmp_path / "x", "wb")).write(np.zeros((4, 4), np.uint8))


def test_writer_reports_a_failing_encoder(tmp_path):
    w = K.RawVideoWriter([sys.executable, "-c", CONSUMER, str(tmp_path / "o.bgr"), "2"])
    w.write(_frames(1)[0])
    with pytest.raises(K.CodecError, match="status 2"):
        w.close()
    w = K.RawVideoWriter([sys.executable, "-c", "import sys; sys.exit(0)"])        # closes its stdin at once: broken pipe
    with pytest.raises(K.CodecError):
        for f in _frames(64):
            w.write(np.repeat(np.repeat(f, 8, 0), 8, 1))
            time.sleep(0.005)
        w.close()


def test_reader_into_writer_is_the_identity(tmp_path):
    out = tmp_path / "copy.bgr"
    with K.RawVideoReader([sys.executable, "-c", PRODUCER, "30", str(H), str(W), "0", "0"], H, W, depth=3, pin=False) as r, \
            K.RawVideoWriter([sys.executable, "-c", CONSUMER, str(out), "0"], depth=2) as w:
        for f in r:
            w.write(f.copy())
    assert out.read_bytes() == _frames(30).tobytes()
