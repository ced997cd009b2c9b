"""The codec edge of the hot path: decoded frames stream INTO the device pipeline and restored frames
stream OUT of it through pipes, overlapped with the GPU work - instead of the reference's PNG directories.

The reference extracts every frame to ``frames/frame_%08d.png`` with one ffmpeg run (restorer.py:1078-1117), moves PNG directories
between `tap_denoise -> enhance -> interpolate` (:3217-3329) and re-reads a PNG directory for the final encode (:2950-3027).  Here
the same two ffmpeg processes sit at the two ends of a pipe of raw ``bgr24`` frames:

    ffmpeg -i in.mkv -f rawvideo -pix_fmt bgr24 -  ->  RawVideoReader  ->  DeviceRestorationPipeline.run_stream  ->
    RawVideoWriter  ->  ffmpeg -f rawvideo -pix_fmt bgr24 -s WxH -framerate F -i - [-i audio -c:a flac] -c:v libx265 -crf .. out.mkv

A reader thread pulls whole frames from the decoder's stdout into a bounded queue of (pinned) slots, a writer thread drains a
bounded queue of finished frames into the encoder's stdin; the caller's thread only enqueues GPU work.  There is no ffmpeg in the
build image: the tests drive both ends with Python child processes that speak the same byte protocol (H * W * 3 bytes per frame,
no header), which is all the two classes know about their peers.
"""
from __future__ import annotations

import queue
import subprocess
import threading
from pathlib import Path
from typing import IO, Iterator, List, Optional, Sequence, Union

import numpy as np


class CodecError(RuntimeError):
    """A pipe ended in the middle of a frame, or a codec process failed."""


def decode_command(video_path: Union[str, Path], ffmpeg: str = "ffmpeg") -> List[str]:
    """The reference's extraction command (restorer.py:1110-1117: ``ffmpeg -i video ... frame_%08d.png``) with the PNG sink replaced
    by raw BGR frames on stdout - the layout ``cv2.imread`` hands the processors."""
    return [ffmpeg, "-i", str(video_path), "-f", "rawvideo", "-pix_fmt", "bgr24", "-"]


def encode_command(output_path: Union[str, Path], width: int, height: int, framerate: float, codec: str = "libx265", crf: int = 18,
                   preset: str = "slow", pix_fmt: str = "yuv420p10le", audio_path: Optional[Union[str, Path]] = None,
                   ffmpeg: str = "ffmpeg") -> List[str]:
    """The reference's reassembly command (restorer.py:3000-3027) with the PNG-pattern input replaced by raw BGR frames on stdin:
    same ``-framerate``, optional ``-i audio -c:a flac``, ``-c:v codec -crf -preset -pix_fmt -y output``."""
    cmd = [ffmpeg, "-f", "rawvideo", "-pix_fmt", "bgr24", "-s", f"{int(width)}x{int(height)}", "-framerate", str(framerate), "-i", "-"]
    if audio_path is not None and Path(audio_path).exists():
        cmd += ["-i", str(audio_path), "-c:a", "flac"]
    cmd += ["-c:v", codec, "-crf", str(crf), "-preset", preset, "-pix_fmt", pix_fmt, "-y", str(output_path)]
    return cmd


def _alloc_slot(height: int, width: int, pin: bool) -> np.ndarray:
    if pin:
        try:
            import torch
            if torch.cuda.is_available():
                return torch.empty((height, width, 3), dtype=torch.uint8).pin_memory().numpy()
        except Exception:  # noqa: BLE001 - pinning is an optimisation of the upload, never a requirement
            pass
    return np.empty((height, width, 3), np.uint8)


class RawVideoReader:
    """Iterator over H x W x 3 uint8 BGR frames read from a pipe of raw ``bgr24`` video.

    ``source``: a command line (list: started as a child process, its stdout is the pipe) or a binary file object.  A thread reads
    ``depth`` frames ahead into a ring of ``depth + 4`` slots (pinned host memory when a GPU is present: the upload that follows is
    then asynchronous); a yielded frame stays valid until the iterator has advanced twice more.  A stream that ends inside a frame,
    or a decoder that exits non-zero, raises CodecError from the consuming thread."""

    def __init__(self, source: Union[Sequence[str], IO[bytes]], height: int, width: int, depth: int = 4, pin: bool = True):
        if height < 1 or width < 1 or depth < 1:
            raise ValueError("RawVideoReader: bad geometry")
        self.height, self.width, self.depth = int(height), int(width), int(depth)
        self._proc: Optional[subprocess.Popen] = None
        if isinstance(source, (list, tuple)):
            self._proc = subprocess.Popen(list(source), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            self._f = self._proc.stdout
        else:
            self._f = source
        self._slots = [_alloc_slot(self.height, self.width, pin) for _ in range(self.depth + 4)]   # depth queued + 1 filling + 3 with the consumer
        self._free: "queue.Queue[int]" = queue.Queue()
        for k in range(len(self._slots)):
            self._free.put(k)
        self._full: "queue.Queue[object]" = queue.Queue(maxsize=self.depth)
        self._held: List[int] = []
        self._stop = threading.Event()
        self.frames_read = 0
        self._t = threading.Thread(target=self._pump, name="fw-decode", daemon=True)
        self._t.start()

    def _pump(self) -> None:
        nbytes = self.height * self.width * 3
        pumped = 0
        try:
            while not self._stop.is_set():
                k = self._free.get()
                if k is None:
                    return
                view = memoryview(self._slots[k]).cast("B")
                got = 0
                while got < nbytes:
                    n = self._f.readinto(view[got:])
                    if not n:
                        break
                    got += n
                if got == 0:
                    break                                            # clean end of stream
                if got < nbytes:
                    raise CodecError(f"raw video ended inside frame {pumped} ({got} of {nbytes} bytes)")
                pumped += 1
                self._put(k)
            rc = self._proc.wait() if self._proc is not None else 0
            if rc != 0:
                raise CodecError(f"decoder exited with status {rc}")
            self._put(None)
        except BaseException as e:  # noqa: BLE001 - handed to the consuming thread
            self._put(e)

    def _put(self, item) -> None:
        while not self._stop.is_set():
            try:
                self._full.put(item, timeout=0.1)
                return
            except queue.Full:
                continue

    def __iter__(self) -> Iterator[np.ndarray]:
        return self

    def __next__(self) -> np.ndarray:
        item = self._full.get()
        if item is None:
            self._full.put(None)       # a second next() after the end stops again
            raise StopIteration
        if isinstance(item, BaseException):
            self._full.put(item)
            raise item
        self._held.append(item)
        if len(self._held) > 2:        # the slot handed out three frames ago may be refilled
            self._free.put(self._held.pop(0))
        self.frames_read += 1
        return self._slots[item]

    def close(self) -> None:
        self._stop.set()
        self._free.put(None)
        if self._proc is not None:
            if self._proc.poll() is None:
                self._proc.kill()
            self._proc.wait()
            if self._proc.stdout:
                self._proc.stdout.close()
        self._t.join(timeout=5)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class RawVideoWriter:
    """Sink of H x W x 3 uint8 BGR frames into a pipe of raw ``bgr24`` video.

    ``sink``: a command line (started as a child process, its stdin is the pipe) or a binary file object.  ``write(frame, ready,
    release)`` queues a frame (at most ``depth`` waiting: the producer blocks, which is the back-pressure on the GPU stages); the
    writer thread calls ``ready()`` first (e.g. a HIP event's synchronize: the download of that frame has finished), writes the
    bytes and then ``release()`` (the staging slot may be reused).  ``close()`` drains the queue, closes the pipe and waits for the
    encoder; its failure (or a broken pipe) raises CodecError."""

    def __init__(self, sink: Union[Sequence[str], IO[bytes]], depth: int = 4):
        self._proc: Optional[subprocess.Popen] = None
        if isinstance(sink, (list, tuple)):
            self._proc = subprocess.Popen(list(sink), stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            self._f = self._proc.stdin
        else:
            self._f = sink
        self._q: "queue.Queue[object]" = queue.Queue(maxsize=max(1, int(depth)))
        self._err: Optional[BaseException] = None
        self.frames_written = 0
        self._t = threading.Thread(target=self._pump, name="fw-encode", daemon=True)
        self._t.start()

    def _pump(self) -> None:
        while True:
            item = self._q.get()
            if item is None:
                return
            frame, ready, release = item
            try:
                if self._err is None:
                    if ready is not None:
                        ready()
                    self._f.write(memoryview(np.ascontiguousarray(frame)).cast("B"))
                    self.frames_written += 1
            except BaseException as e:  # noqa: BLE001 - reported by write() / close(); keep draining so the producer never blocks
                self._err = e
            finally:
                if release is not None:
                    release()

    def write(self, frame: np.ndarray, ready=None, release=None) -> None:
        if self._err is not None:
            raise CodecError(f"encoder pipe failed: {self._err}") from self._err
        if frame.dtype != np.uint8 or frame.ndim != 3 or frame.shape[2] != 3:
            raise ValueError("RawVideoWriter.write expects an H x W x 3 uint8 BGR frame")
        self._q.put((frame, ready, release))

    def close(self) -> None:
        self._q.put(None)
        self._t.join()
        try:
            self._f.flush()
            if self._proc is not None:
                self._f.close()
        except BaseException as e:  # noqa: BLE001
            self._err = self._err or e
        if self._proc is not None:
            rc = self._proc.wait()
            if rc != 0 and self._err is None:
                self._err = CodecError(f"encoder exited with status {rc}")
        if self._err is not None:
            raise CodecError(f"encoder pipe failed: {self._err}") from self._err

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()
        else:                       # already failing: do not mask the first error
            try:
                self.close()
            except CodecError:
                pass
