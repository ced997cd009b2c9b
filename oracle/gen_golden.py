"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN MODULES in the build container.

Run here only (``python oracle/gen_golden.py``): /root/reference does not exist on the GPU box, and the reference
source never travels — only the small input/output vectors written by this script do.

``import framewright`` fails in the reference snapshot (framewright/__init__.py:6 -> restorer.py:178 ->
infrastructure/__init__.py:67 imports a directory that is not in the snapshot), so individual modules are loaded
through namespace stubs.

Fixture set 1 — residual-dense arithmetic (reference src/framewright/processors/aesrgan_face.py):
  * ``ResidualDenseBlock(64)``            :171-189   1x64x12x12 -> 1x64x12x12
  * ``RRDB(64)``                          :192-204   1x64x12x12 -> 1x64x12x12
  * ``AESRGAN(num_block=2, scale=4, num_attention=1)`` :207-269   1x3x16x16 -> 1x3x64x64
    (its single AttentionBlock has gamma = 0 at construction (:149), so ``gamma*out + x`` (:169) is the identity and
    the module is exactly conv_first -> 2xRRDB -> conv_body -> +feat -> up1 -> up2 -> conv_hr -> conv_last)
Weights are NOT stored: they are regenerated from a seed with framewright_amd.synth (numpy PCG64), loaded into the
reference modules here and into the oracle in the tests.
"""
from __future__ import annotations

import importlib
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.dont_write_bytecode = True
REF_SRC = "/root/reference/src"

from framewright_amd.synth import synthetic_rrdbnet_state  # noqa: E402


def load_reference(module: str):
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    for n in ["framewright", "framewright.processors", "framewright.processors.enhancement", "framewright.utils",
              "framewright.plugins", "framewright.engine", "framewright.infrastructure",
              "framewright.infrastructure.gpu"]:
        if n not in sys.modules:
            m = types.ModuleType(n)
            m.__path__ = [f"{REF_SRC}/{n.replace('.', '/')}"]
            sys.modules[n] = m
    return importlib.import_module(module)


def _load_rdb(rdb, sd, prefix):
    for c in range(1, 6):
        conv = getattr(rdb, f"conv{c}")
        conv.weight.data.copy_(torch.from_numpy(sd[f"{prefix}.conv{c}.weight"]))
        conv.bias.data.copy_(torch.from_numpy(sd[f"{prefix}.conv{c}.bias"]))


def _load_rrdb(rrdb, sd, prefix):
    for r in (1, 2, 3):
        _load_rdb(getattr(rrdb, f"rdb{r}"), sd, f"{prefix}.rdb{r}")


def main() -> None:
    a = load_reference("framewright.processors.aesrgan_face")
    assert a.HAS_TORCH
    out = {}
    torch.manual_seed(0)
    rng = np.random.default_rng(2024)

    # --- ResidualDenseBlock -----------------------------------------------------------------
    sd = synthetic_rrdbnet_state(1, 4, seed=11)
    rdb = a.ResidualDenseBlock(64).eval()
    _load_rdb(rdb, sd, "body.0.rdb1")
    x = rng.standard_normal((1, 64, 12, 12)).astype(np.float32)
    with torch.no_grad():
        y = rdb(torch.from_numpy(x).clone()).numpy()
    out["rdb_seed"], out["rdb_in"], out["rdb_out"] = np.int64(11), x, y

    # --- RRDB -----------------------------------------------------------------------------------
    sd = synthetic_rrdbnet_state(1, 4, seed=12)
    rrdb = a.RRDB(64).eval()
    _load_rrdb(rrdb, sd, "body.0")
    x = rng.standard_normal((1, 64, 12, 12)).astype(np.float32)
    with torch.no_grad():
        y = rrdb(torch.from_numpy(x).clone()).numpy()
    out["rrdb_seed"], out["rrdb_in"], out["rrdb_out"] = np.int64(12), x, y

    # --- AESRGAN trunk + tail (attention inert: gamma == 0) -----------------------------------------------
    sd = synthetic_rrdbnet_state(2, 4, seed=13)
    net = a.AESRGAN(num_in_ch=3, num_out_ch=3, num_feat=64, num_block=2, scale=4, num_attention=1).eval()
    rrdbs = [m for m in net.body if isinstance(m, a.RRDB)]
    attn = [m for m in net.body if isinstance(m, a.AttentionBlock)]
    assert len(rrdbs) == 2 and len(attn) == 1 and float(attn[0].gamma.abs().sum()) == 0.0
    for i, m in enumerate(rrdbs):
        _load_rrdb(m, sd, f"body.{i}")
    for name in ("conv_first", "conv_body", "conv_up1", "conv_up2", "conv_hr", "conv_last"):
        conv = getattr(net, name)
        conv.weight.data.copy_(torch.from_numpy(sd[name + ".weight"]))
        conv.bias.data.copy_(torch.from_numpy(sd[name + ".bias"]))
    x = rng.uniform(0, 1, size=(1, 3, 16, 16)).astype(np.float32)
    with torch.no_grad():
        y = net(torch.from_numpy(x).clone()).numpy()
    out["net_seed"], out["net_in"], out["net_out"] = np.int64(13), x, y

    dst = ROOT / "tests" / "golden" / "rrdb_reference.npz"
    dst.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(dst, **out)
    print(f"wrote {dst} ({dst.stat().st_size / 1024:.0f} KiB)")
    for k, v in out.items():
        if hasattr(v, "shape") and v.ndim:
            print(f"  {k}: {v.shape} mean {v.mean():+.4f} std {v.std():.4f}")


if __name__ == "__main__":
    main()
