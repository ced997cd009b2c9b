"""In-tree build of libframewright_hip.so for gfx950 (MI355X).

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container as well as on the GPU box.
The shared library is written next to the sources (lib/libframewright_hip.so): it is git-ignored but travels
with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
INCLUDE = PKG_DIR.parent / "include"
LIB_DIR = PKG_DIR / "lib"
OBJ_DIR = PKG_DIR / "build"
LIB_PATH = LIB_DIR / "libframewright_hip.so"
ARCH = "gfx950"
# frame_ops.hip restates float32 numpy arithmetic bit for bit (a*b + c rounds twice): no FMA contraction there
PER_FILE_FLAGS = {"frame_ops.hip": ["-ffp-contract=off"]}
EXTRA = os.environ.get("FW_EXTRA_CXXFLAGS", "").split()
CXXFLAGS = [*EXTRA, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm); the HIP path cannot be built")


def sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _digest(paths: list[Path]) -> str:
    h = hashlib.sha256()
    for p in paths:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(CXXFLAGS).encode())
    h.update(repr(sorted(PER_FILE_FLAGS.items())).encode())
    return h.hexdigest()


def needs_build() -> bool:
    deps = sources() + sorted(CSRC.glob("*.h")) + sorted(INCLUDE.glob("*.h"))
    stamp = LIB_DIR / ".digest"
    return not (LIB_PATH.exists() and stamp.exists() and stamp.read_text() == _digest(deps))


def source_digest() -> str:
    """16 hex digits that name a BUILD by what went into it (file names and bytes of csrc/*.hip, csrc/*.h, include/*.h, the compile
    flags) - what lib/.digest holds since the library was linked.  The measured-traffic files under profiles/ carry it, and bench.py
    quotes their numbers only for the build they were measured on.  (The bytes of the .so itself differ with the directory the tree
    is built in - __FILE__ strings - so they cannot name a build across checkouts.)"""
    f = LIB_DIR / ".digest"
    full = f.read_text().strip() if f.exists() else _digest(sources() + sorted(CSRC.glob("*.h")) + sorted(INCLUDE.glob("*.h")))
    return full[:16]


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every .hip under csrc/ for gfx950 and link lib/libframewright_hip.so."""
    deps = sources() + sorted(CSRC.glob("*.h")) + sorted(INCLUDE.glob("*.h"))
    if not force and not needs_build():
        return LIB_PATH
    cc = hipcc()
    OBJ_DIR.mkdir(exist_ok=True)
    LIB_DIR.mkdir(exist_ok=True)

    def compile_one(src: Path) -> Path:
        obj = OBJ_DIR / (src.stem + ".o")
        cmd = [cc, *CXXFLAGS, *PER_FILE_FLAGS.get(src.name, []), f"-I{INCLUDE}", f"-I{CSRC}", "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(sources()))) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", *map(str, objs), "-o", str(LIB_PATH)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    (LIB_DIR / ".digest").write_text(_digest(deps))
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
