"""Parity of the MFMA conv3x3 kernel (csrc/conv3x3_mfma.hip) against a plain torch fp32 conv on the SAME
operand-rounded inputs, through the C-ABI (fw_conv3x3_nhwc).  With identical operands the only difference is
fp32 accumulation order, so the tolerance is tight."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from framewright_amd import _lib

pytestmark = pytest.mark.gpu

TDT = {_lib.FW_DTYPE_BF16: torch.bfloat16, _lib.FW_DTYPE_F16: torch.float16}


def _pack(lib, dtype, w, ct, ch):
    cout, cin = w.shape[:2]
    n = lib.fw_pack_conv3x3(dtype, None, cout, cin, ct, ch, None)
    dst = np.zeros(n, np.uint16)
    wc = np.ascontiguousarray(w, np.float32)
    assert lib.fw_pack_conv3x3(dtype, C.c_void_p(wc.ctypes.data), cout, cin, ct, ch, C.c_void_p(dst.ctypes.data)) == n
    return torch.from_numpy(dst.view(np.int16)).cuda()


def _run(lib, dtype, x_nhwc, cin, w, b, H, W, act=0, ups=0, res1=None, s1=1.0, res2=None, s2=1.0, out_cstride=None,
         out_coff=0, want_f32=False):
    cout = w.shape[0]
    ct, ch = (cout + 31) // 32, (cin + 31) // 32
    wp = _pack(lib, dtype, w, ct, ch)
    bias = torch.zeros(32 * ct, dtype=torch.float32, device="cuda")
    bias[:cout] = torch.from_numpy(b).cuda()
    out_cstride = out_cstride or 32 * ct
    out = torch.full((H, W, out_cstride), 7.0, dtype=TDT[dtype], device="cuda")
    out_f32 = torch.zeros((H, W, 32 * ct), dtype=torch.float32, device="cuda") if want_f32 else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = lib.fw_conv3x3_nhwc(dtype, p(x_nhwc), x_nhwc.shape[-1], 0, ch, H, W, p(wp), p(bias), ct, act, ups,
                             p(res1), s1, p(res2), s2, p(out), out_cstride, 0, out_coff, p(out_f32),
                             C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(st)
    torch.cuda.synchronize()
    return out, out_f32


def _ref(dtype, x_nhwc, cin, w, b, act=0, ups=0):
    x = x_nhwc[..., :cin].float().permute(2, 0, 1).unsqueeze(0)
    if ups:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    wq = torch.from_numpy(w).cuda().to(TDT[dtype]).float()
    y = F.conv2d(x, wq, torch.from_numpy(b).cuda(), 1, 1)
    if act:
        y = F.leaky_relu(y, 0.2)
    return y.squeeze(0).permute(1, 2, 0)  # HWC fp32


@pytest.mark.parametrize("dtype", [_lib.FW_DTYPE_BF16, _lib.FW_DTYPE_F16])
@pytest.mark.parametrize("cin,cout,H,W,cstride,act", [
    (64, 32, 16, 32, 192, 1),     # exactly one tile
    (96, 32, 23, 45, 192, 1),     # ragged edges
    (160, 32, 40, 70, 192, 1),
    (192, 64, 33, 65, 192, 0),    # conv5 shape
    (32, 64, 18, 34, 32, 0),      # conv_first (padded 3->32 input)
    (64, 64, 5, 3, 64, 1),        # smaller than a tile
    (64, 64, 1, 1, 64, 0),        # single pixel
])
def test_conv_store(hip_lib, dtype, cin, cout, H, W, cstride, act):
    rng = np.random.default_rng(cin * 7 + cout + H)
    x = torch.from_numpy(rng.standard_normal((H, W, cstride)).astype(np.float32)).cuda().to(TDT[dtype])
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    out, out_f32 = _run(hip_lib, dtype, x, cin, w, b, H, W, act=act, want_f32=True)
    ref = _ref(dtype, x, cin, w, b, act=act)
    assert torch.isfinite(out_f32).all()
    err32 = (out_f32[..., :cout] - ref).abs().max().item()
    assert err32 < 2e-5 * max(1.0, ref.abs().max().item()), err32
    # typed output = fp32 result rounded once to the operand type
    assert torch.equal(out[..., :cout], out_f32[..., :cout].to(TDT[dtype]))


@pytest.mark.parametrize("dtype", [_lib.FW_DTYPE_BF16, _lib.FW_DTYPE_F16])
def test_conv_writes_only_its_channel_slice(hip_lib, dtype):
    rng = np.random.default_rng(0)
    H, W = 20, 40
    x = torch.from_numpy(rng.standard_normal((H, W, 192)).astype(np.float32)).cuda().to(TDT[dtype])
    w = (rng.standard_normal((32, 96, 3, 3)) / 30).astype(np.float32)
    b = np.zeros(32, np.float32)
    out, _ = _run(hip_lib, dtype, x, 96, w, b, H, W, act=1, out_cstride=192, out_coff=96)
    ref = _ref(dtype, x, 96, w, b, act=1)
    assert (out[..., :96].float() == 7.0).all() and (out[..., 128:].float() == 7.0).all()
    assert (out[..., 96:128].float() - ref).abs().max().item() < 2e-2


@pytest.mark.parametrize("dtype", [_lib.FW_DTYPE_BF16, _lib.FW_DTYPE_F16])
def test_conv_in_place_concat_buffer(hip_lib, dtype):
    """conv k reads channels [0, 64+32(k-1)) and writes the NEXT slice of the same buffer."""
    rng = np.random.default_rng(1)
    H, W = 19, 37
    buf = torch.from_numpy(rng.standard_normal((H, W, 192)).astype(np.float32)).cuda().to(TDT[dtype])
    keep = buf.clone()
    w = (rng.standard_normal((32, 64, 3, 3)) / 24).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    ref = _ref(dtype, keep, 64, w, b, act=1)
    wp = _pack(hip_lib, dtype, w, 1, 2)
    bias = torch.from_numpy(b).cuda()
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(hip_lib.fw_conv3x3_nhwc(dtype, p(buf), 192, 0, 2, H, W, p(wp), p(bias), 1, 1, 0, None, 1.0, None, 1.0,
                                       p(buf), 192, 0, 64, None, None))
    torch.cuda.synchronize()
    assert torch.equal(buf[..., :64], keep[..., :64]) and torch.equal(buf[..., 96:], keep[..., 96:])
    assert (buf[..., 64:96].float() - ref).abs().max().item() < (2e-2 if dtype == _lib.FW_DTYPE_BF16 else 3e-3)


@pytest.mark.parametrize("dtype", [_lib.FW_DTYPE_BF16, _lib.FW_DTYPE_F16])
def test_conv_upsample2x(hip_lib, dtype):
    rng = np.random.default_rng(2)
    h, w_ = 13, 21
    x = torch.from_numpy(rng.standard_normal((h, w_, 64)).astype(np.float32)).cuda().to(TDT[dtype])
    w = (rng.standard_normal((64, 64, 3, 3)) / 24).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    _, out_f32 = _run(hip_lib, dtype, x, 64, w, b, 2 * h, 2 * w_, act=1, ups=1, want_f32=True)
    ref = _ref(dtype, x, 64, w, b, act=1, ups=1)
    assert (out_f32 - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def _phase_weights_ref(w):
    """The four 2x2 phase kernels of nearest-x2 + conv3x3 (csrc/conv_up2x_phase.hip): phase 0 folds taps {0} / {1, 2} onto source
    offsets 0 / 1, phase 1 folds {0, 1} / {2}.  Returns [a][b][cout][cin][2][2] in fp64."""
    fold = {0: ([0], [1, 2]), 1: ([0, 1], [2])}
    w = w.astype(np.float64)
    out = np.zeros((2, 2) + w.shape[:2] + (2, 2))
    for a in range(2):
        for b in range(2):
            for r in range(2):
                for s in range(2):
                    out[a, b, :, :, r, s] = w[:, :, fold[a][r]][:, :, :, fold[b][s]].sum(axis=(2, 3))
    return out


@pytest.mark.parametrize("dtype", [_lib.FW_DTYPE_BF16, _lib.FW_DTYPE_F16])
@pytest.mark.parametrize("h,w_,planar", [
    (16, 32, False),     # exactly one source tile
    (13, 21, True),      # ragged, chunk-planar in and out (the RRDBNet tail's layout)
    (1, 1, False),
    (5, 3, True),
    (33, 65, False),     # one row / column into the next tile
    (368, 736, True),    # 529 source tiles: two or three per workgroup - the alternating chunk order and the refills between tiles
])
def test_conv_up2x_phase(hip_lib, dtype, h, w_, planar):
    """conv_up1 / conv_up2 as four 2x2 phase convolutions == lrelu(conv3x3(nearest_x2(x))) (aesrgan_face.py:258-266)."""
    rng = np.random.default_rng(h * 1000 + w_)
    x = torch.from_numpy(rng.standard_normal((h, w_, 64)).astype(np.float32)).cuda().to(TDT[dtype])
    w = (rng.standard_normal((64, 64, 3, 3)) / 24).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    n = hip_lib.fw_pack_conv_up2x_phase(dtype, None, None)
    assert n == 2 * 2 * 32 * 64 * 8
    pk = np.zeros(n, np.uint16)
    assert hip_lib.fw_pack_conv_up2x_phase(dtype, C.c_void_p(w.ctypes.data), C.c_void_p(pk.ctypes.data)) == n
    wp = torch.from_numpy(pk.view(np.int16)).cuda()
    bias = torch.from_numpy(b).cuda()
    p = lambda t: C.c_void_p(t.data_ptr())
    if planar:
        xin = x.reshape(h, w_, 2, 32).permute(2, 0, 1, 3).contiguous()
        out = torch.full((2, 2 * h, 2 * w_, 32), 7.0, dtype=TDT[dtype], device="cuda")
        _lib.check(hip_lib.fw_conv_up2x_phase_nhwc(dtype, p(xin), 32, h * w_ * 32, h, w_, p(wp), p(bias), 1, p(out), 32, 4 * h * w_ * 32,
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        got = out.permute(1, 2, 0, 3).reshape(2 * h, 2 * w_, 64).float()
    else:
        out = torch.full((2 * h, 2 * w_, 64), 7.0, dtype=TDT[dtype], device="cuda")
        _lib.check(hip_lib.fw_conv_up2x_phase_nhwc(dtype, p(x), 64, 0, h, w_, p(wp), p(bias), 1, p(out), 64, 0,
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        got = out.float()
    # (1) against the same arithmetic in torch: the phase kernels with their ONCE-rounded summed weights, fp32 accumulation
    wph = torch.from_numpy(_phase_weights_ref(w)).cuda().float().to(TDT[dtype]).float()
    xs = x.float().permute(2, 0, 1).unsqueeze(0)
    xp = F.pad(xs, (1, 1, 1, 1))
    exact = torch.empty((2 * h, 2 * w_, 64), device="cuda")
    for a in range(2):
        for bb in range(2):
            y = F.conv2d(xp[:, :, a:a + h + 1, bb:bb + w_ + 1], wph[a, bb], torch.from_numpy(b).cuda())
            exact[a::2, bb::2] = F.leaky_relu(y, 0.2).squeeze(0).permute(1, 2, 0)
    scale = max(1.0, exact.abs().max().item())
    typed = exact.to(TDT[dtype]).float()
    assert (got - typed).abs().max().item() <= (1.6e-2 if dtype == _lib.FW_DTYPE_BF16 else 2e-3) * scale   # one output rounding flip at most
    assert (got - exact).abs().max().item() < (1.2e-2 if dtype == _lib.FW_DTYPE_BF16 else 1.5e-3) * scale
    # (2) against the reference's formulation with fp32 weights: only operand rounding apart
    wf = torch.from_numpy(w).cuda()
    ref = F.leaky_relu(F.conv2d(F.interpolate(xs, scale_factor=2, mode="nearest"), wf, torch.from_numpy(b).cuda(), 1, 1), 0.2)
    ref = ref.squeeze(0).permute(1, 2, 0)
    assert (got - ref).abs().max().item() < (5e-2 if dtype == _lib.FW_DTYPE_BF16 else 6e-3) * scale


@pytest.mark.parametrize("dtype", [_lib.FW_DTYPE_BF16, _lib.FW_DTYPE_F16])
@pytest.mark.parametrize("two", [False, True])
def test_conv_residual_epilogue(hip_lib, dtype, two):
    """x5*0.2 + x and (x5*0.2 + x)*0.2 + rrdb_in — reference aesrgan_face.py:189,204."""
    rng = np.random.default_rng(3)
    H, W = 21, 50
    x = torch.from_numpy(rng.standard_normal((H, W, 192)).astype(np.float32)).cuda().to(TDT[dtype])
    w = (rng.standard_normal((64, 192, 3, 3)) / 41).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    r1 = torch.from_numpy(rng.standard_normal((H, W, 64)).astype(np.float32)).cuda()
    r2 = torch.from_numpy(rng.standard_normal((H, W, 64)).astype(np.float32)).cuda() if two else None
    out, out_f32 = _run(hip_lib, dtype, x, 192, w, b, H, W, res1=r1, s1=0.2, res2=r2, s2=0.2, want_f32=True)
    ref = _ref(dtype, x, 192, w, b) * 0.2 + r1
    if two:
        ref = ref * 0.2 + r2
    assert (out_f32 - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(out, out_f32.to(TDT[dtype]))


def test_conv_rejects_bad_arguments(hip_lib):
    x = torch.zeros((4, 4, 64), dtype=torch.bfloat16, device="cuda")
    p = C.c_void_p(x.data_ptr())
    assert hip_lib.fw_conv3x3_nhwc(0, p, 64, 0, 3, 4, 4, p, p, 1, 0, 0, None, 1.0, None, 1.0, p, 32, 0, 0, None, None) \
        == _lib.FW_ERR_INVALID            # contracts 96 channels of a 64-channel buffer
    assert hip_lib.fw_conv3x3_nhwc(0, p, 64, 0, 2, 5, 4, p, p, 1, 0, 1, None, 1.0, None, 1.0, p, 32, 0, 0, None, None) \
        == _lib.FW_ERR_INVALID            # upsample needs even output
    assert hip_lib.fw_conv3x3_nhwc(0, p, 64, 0, 2, 4, 4, p, p, 3, 0, 0, None, 1.0, None, 1.0, p, 32, 0, 0, None, None) \
        == _lib.FW_ERR_INVALID


@pytest.mark.parametrize("dtype", [_lib.FW_DTYPE_BF16, _lib.FW_DTYPE_F16])
def test_conv_chunk_planar_layout(hip_lib, dtype):
    """The layout the RRDB trunk uses: 32-channel planes [chunk][H][W][32] in, two planes out."""
    rng = np.random.default_rng(5)
    H, W, cin = 27, 43, 160
    x = torch.from_numpy(rng.standard_normal((H, W, cin)).astype(np.float32)).cuda().to(TDT[dtype])
    planes = x.reshape(H, W, cin // 32, 32).permute(2, 0, 1, 3).contiguous()       # [5][H][W][32]
    w = (rng.standard_normal((64, cin, 3, 3)) / 38).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    wp = _pack(hip_lib, dtype, w, 2, 5)
    bias = torch.from_numpy(b).cuda()
    out = torch.zeros((2, H, W, 32), dtype=TDT[dtype], device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(hip_lib.fw_conv3x3_nhwc(dtype, p(planes), 32, H * W * 32, 5, H, W, p(wp), p(bias), 2, 1, 0, None, 1.0,
                                       None, 1.0, p(out), 32, H * W * 32, 0, None, None))
    torch.cuda.synchronize()
    ref = _ref(dtype, x, cin, w, b, act=1)
    got = out.permute(1, 2, 0, 3).reshape(H, W, 64).float()
    assert (got - ref).abs().max().item() < (2e-2 if dtype == _lib.FW_DTYPE_BF16 else 3e-3)


@pytest.mark.parametrize("dtype", [_lib.FW_DTYPE_BF16, _lib.FW_DTYPE_F16])
@pytest.mark.parametrize("slide", ["2", "1", "0"])
@pytest.mark.parametrize("na,H,W", [(2, 14, 30), (2, 33, 71), (4, 40, 64), (2, 3, 5), (4, 29, 91), (2, 520, 330), (4, 1000, 200),
                                    (2, 17, 1), (4, 16, 31), (2, 47, 32), (4, 130, 97)])
def test_conv_pair_fused(hip_lib, dtype, na, H, W, slide, monkeypatch):
    """conv_a + conv_b fused vs two torch convs with x_a rounded to the operand type in between, chunk-planar layout, tile
    borders and image borders (zero padding of x_a).  Both forms of the kernel: the sliding window (conv3x3_pair_slide.hip:
    16-row steps, carried x_a rows, warm-up tiles where a workgroup starts mid-column - the two tall shapes give every
    workgroup several tiles and column wraps), the window kernel with all 32 columns of a tile valid (conv3x3_pair_slide32.hip,
    "2": the two extra x_a columns come from column-shaped MFMAs) and the ring kernel (conv3x3_pair.hip: 14x30 valid outputs per
    tile)."""
    monkeypatch.setenv("FW_PAIR_SLIDE", slide)
    rng = np.random.default_rng(na * 100 + H)
    cin = 32 * na
    x = torch.from_numpy(rng.standard_normal((H, W, cin)).astype(np.float32)).cuda().to(TDT[dtype])
    planes = x.reshape(H, W, na, 32).permute(2, 0, 1, 3).contiguous()
    wa = (rng.standard_normal((32, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    wb = (rng.standard_normal((32, cin + 32, 3, 3)) / np.sqrt(9 * (cin + 32))).astype(np.float32)
    ba, bb = rng.standard_normal(32).astype(np.float32), rng.standard_normal(32).astype(np.float32)
    pa, pb = _pack(hip_lib, dtype, wa, 1, na), _pack(hip_lib, dtype, wb, 1, na + 1)
    ta, tb = torch.from_numpy(ba).cuda(), torch.from_numpy(bb).cuda()
    oa = torch.full((H, W, 32), 5.0, dtype=TDT[dtype], device="cuda")
    ob = torch.full((H, W, 32), 5.0, dtype=TDT[dtype], device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(hip_lib.fw_conv3x3_pair_nhwc(dtype, p(planes), 32, H * W * 32, na, H, W, p(pa), p(ta), p(pb), p(tb), p(oa),
                                            p(ob), 32, None))
    torch.cuda.synchronize()
    ref_a = _ref(dtype, x, cin, wa, ba, act=1)
    xa_t = ref_a.to(TDT[dtype])
    ref_b = _ref(dtype, torch.cat([x, xa_t], dim=2), cin + 32, wb, bb, act=1)
    tol = 2e-2 if dtype == _lib.FW_DTYPE_BF16 else 3e-3
    assert (oa.float() - ref_a).abs().max().item() < tol
    assert (ob.float() - ref_b).abs().max().item() < 2 * tol


@pytest.mark.parametrize("na,H,W", [(2, 530, 331), (4, 1000, 200), (2, 65, 2000)])
def test_conv_pair_three_kernels_bit_identical(hip_lib, na, H, W, monkeypatch):
    """The ring kernel, the 30-column window kernel and the 32-column window kernel accumulate every output pixel in the same order
    (chunks in order, taps dx-major, fp32): identical bytes, f16 (the benched type)."""
    dtype = _lib.FW_DTYPE_F16
    rng = np.random.default_rng(7 * na + H)
    cin = 32 * na
    x = torch.from_numpy(rng.standard_normal((na, H, W, 32)).astype(np.float32)).cuda().to(TDT[dtype])
    wa = (rng.standard_normal((32, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    wb = (rng.standard_normal((32, cin + 32, 3, 3)) / np.sqrt(9 * (cin + 32))).astype(np.float32)
    ba, bb = rng.standard_normal(32).astype(np.float32), rng.standard_normal(32).astype(np.float32)
    pa, pb = _pack(hip_lib, dtype, wa, 1, na), _pack(hip_lib, dtype, wb, 1, na + 1)
    ta, tb = torch.from_numpy(ba).cuda(), torch.from_numpy(bb).cuda()
    p = lambda t: C.c_void_p(t.data_ptr())
    outs = {}
    for slide in ("0", "1", "2"):
        monkeypatch.setenv("FW_PAIR_SLIDE", slide)
        oa = torch.full((H, W, 32), 5.0, dtype=TDT[dtype], device="cuda")
        ob = torch.full((H, W, 32), 5.0, dtype=TDT[dtype], device="cuda")
        _lib.check(hip_lib.fw_conv3x3_pair_nhwc(dtype, p(x), 32, H * W * 32, na, H, W, p(pa), p(ta), p(pb), p(tb), p(oa), p(ob), 32, None))
        torch.cuda.synchronize()
        outs[slide] = (oa, ob)
    for slide in ("1", "2"):
        assert torch.equal(outs[slide][0], outs["0"][0]), f"x_a differs: kernel {slide} vs ring"
        assert torch.equal(outs[slide][1], outs["0"][1]), f"x_b differs: kernel {slide} vs ring"
