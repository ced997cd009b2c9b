"""C-ABI surface: the library builds for gfx950, loads without a GPU, and exports every symbol that
include/framewright_hip.h declares.  No compute call is made here."""
import ctypes as C
import re

import numpy as np
import pytest

from framewright_amd import _lib


def _header_symbols():
    text = (_lib.PKG_DIR.parent / "include" / "framewright_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fw_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _header_symbols() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol(hip_lib):
    for name in _header_symbols():
        assert hasattr(hip_lib, name), f"{name} declared in the header but not exported"


def test_abi_version_and_error_string(hip_lib):
    assert hip_lib.fw_abi_version() >= 1
    assert isinstance(hip_lib.fw_last_error(), bytes)


def test_invalid_arguments_are_reported_not_crashed(hip_lib):
    h = C.c_void_p()
    assert hip_lib.fw_rrdbnet_create(0, 0, 4, 0, C.byref(h)) == _lib.FW_ERR_INVALID
    assert b"num_block" in hip_lib.fw_last_error()
    assert hip_lib.fw_rrdbnet_create(0, 23, 3, 0, C.byref(h)) == _lib.FW_ERR_INVALID
    assert hip_lib.fw_rrdbnet_create(0, 23, 4, 7, C.byref(h)) == _lib.FW_ERR_INVALID
    assert hip_lib.fw_rrdbnet_destroy(None) == _lib.FW_OK
    assert hip_lib.fw_rrdbnet_workspace_bytes(None, 10, 10) == 0


def _pack_ref(w, dtype, cout_tiles, cin_chunks):
    """numpy restatement of the MFMA A-fragment order documented in csrc/conv3x3_mfma.hip."""
    import torch
    cout, cin = w.shape[:2]
    wp = np.zeros((32 * cout_tiles, 32 * cin_chunks, 3, 3), np.float32)
    wp[:cout, :cin] = w
    t = torch.from_numpy(wp).to(torch.bfloat16 if dtype == _lib.FW_DTYPE_BF16 else torch.float16)
    bits = t.view(torch.int16).numpy().view(np.uint16)
    # [chunk][tap][16-channel cout tile][lane][j]: v_mfma_f32_16x16x32 A operand, row = lane & 15, k = 8*(lane >> 4) + j
    out = np.empty((cin_chunks, 9, 2 * cout_tiles, 64, 8), np.uint16)
    lane = np.arange(64)
    for c in range(cin_chunks):
        for tap in range(9):
            for wt in range(2 * cout_tiles):
                co = 16 * wt + (lane & 15)
                for j in range(8):
                    ci = 32 * c + 8 * (lane >> 4) + j
                    out[c, tap, wt, :, j] = bits[co, ci, tap // 3, tap % 3]
    return out.reshape(-1)


@pytest.mark.parametrize("dtype", [_lib.FW_DTYPE_BF16, _lib.FW_DTYPE_F16])
@pytest.mark.parametrize("cout,cin,ct,ch", [(32, 64, 1, 2), (64, 192, 2, 6), (3, 64, 1, 2), (64, 12, 2, 1)])
def test_weight_packing_matches_fragment_map(hip_lib, dtype, cout, cin, ct, ch):
    rng = np.random.default_rng(cout * 1000 + cin)
    w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
    n = hip_lib.fw_pack_conv3x3(dtype, None, cout, cin, ct, ch, None)
    assert n == ch * 9 * 2 * ct * 64 * 8
    dst = np.zeros(n, np.uint16)
    assert hip_lib.fw_pack_conv3x3(dtype, C.c_void_p(w.ctypes.data), cout, cin, ct, ch,
                                   C.c_void_p(dst.ctypes.data)) == n
    np.testing.assert_array_equal(dst, _pack_ref(w, dtype, ct, ch))


def test_pack_rejects_bad_shapes(hip_lib):
    assert hip_lib.fw_pack_conv3x3(0, None, 65, 64, 2, 2, None) == 0
    assert hip_lib.fw_pack_conv3x3(0, None, 64, 65, 2, 2, None) == 0


@pytest.mark.parametrize("dtype", [_lib.FW_DTYPE_BF16, _lib.FW_DTYPE_F16])
def test_phase_weight_packing_matches_its_fragment_map(hip_lib, dtype):
    """fw_pack_conv_up2x_phase: [row phase a][chunk c][step (tx, r, b) in use order][16-channel tile][lane][j] of the summed
    2x2 phase kernels of nearest-x2 + conv3x3 (csrc/conv_up2x_phase.hip), sums in fp64, rounded once."""
    import torch
    rng = np.random.default_rng(7)
    w = rng.standard_normal((64, 64, 3, 3)).astype(np.float32)
    n = hip_lib.fw_pack_conv_up2x_phase(dtype, None, None)
    assert n == 2 * 2 * 32 * 64 * 8
    dst = np.zeros(n, np.uint16)
    assert hip_lib.fw_pack_conv_up2x_phase(dtype, C.c_void_p(w.ctypes.data), C.c_void_p(dst.ctypes.data)) == n
    assert hip_lib.fw_pack_conv_up2x_phase(7, None, None) == 0
    fold = {0: ([0], [1, 2]), 1: ([0, 1], [2])}
    steps = [(0, 0, 0), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 0), (1, 1, 1), (2, 0, 1), (2, 1, 1)]   # (tx, r, b); column tap = tx - b
    want = np.empty((2, 2, 8, 4, 64, 8), np.uint16)
    lane = np.arange(64)
    tdt = torch.bfloat16 if dtype == _lib.FW_DTYPE_BF16 else torch.float16
    for a in range(2):
        for si, (tx, r, b) in enumerate(steps):
            ws = w.astype(np.float64)[:, :, fold[a][r]][:, :, :, fold[b][tx - b]].sum(axis=(2, 3)).astype(np.float32)
            bits = torch.from_numpy(ws).to(tdt).view(torch.int16).numpy().view(np.uint16)
            for c in range(2):
                for wt in range(4):
                    co = 16 * wt + (lane & 15)
                    for j in range(8):
                        want[a, c, si, wt, :, j] = bits[co, 32 * c + 8 * (lane >> 4) + j]
    np.testing.assert_array_equal(dst, want.reshape(-1))
