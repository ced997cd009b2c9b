"""Restormer (the reference's default TAP model) on the GPU: building blocks against torch, the network against the fp32
CPU oracle (oracle/restormer_ref.py; parity unpinned at the third-party boundary, see its header)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from framewright_amd import _lib
from framewright_amd import restormer as RS
from framewright_amd.synth import synthetic_frames
from oracle import restormer_ref as ref

pytestmark = pytest.mark.gpu
TDT = {"f16": torch.float16, "bf16": torch.bfloat16}
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("C_,ld", [(48, 64), (96, 96), (384, 384)])
def test_layernorm(hip_lib, dtype, C_, ld):
    g = torch.Generator().manual_seed(C_)
    M = 777
    x = torch.zeros(M, ld)
    x[:, :C_] = torch.randn(M, C_, generator=g) * 2 + 0.3
    w, b = torch.randn(C_, generator=g), torch.randn(C_, generator=g)
    out = torch.full((M, ld), 7.0, dtype=TDT[dtype], device="cuda")
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    _lib.check(hip_lib.fw_layernorm_nhwc(_lib.DTYPES[dtype], P(xd), ld, M, C_, P(wd), P(bd), 1e-5, P(out), ld, ld, _st()))
    torch.cuda.synchronize()
    want = F.layer_norm(x[:, :C_], (C_,), w, b, 1e-5)
    got = out.float().cpu()
    tol = 2e-3 if dtype == "f16" else 2e-2
    assert (got[:, :C_] - want).abs().max() < tol * max(1.0, want.abs().max().item())
    assert (got[:, C_:] == 0).all()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("mode,Cc", [(0, 192), (1, 256), (0, 576), (0, 1152), (1, 1024), (1, 2048)])   # from 512 output channels: the channel-range kernel
def test_dwconv3x3(hip_lib, dtype, mode, Cc):
    g = torch.Generator().manual_seed(mode + Cc)
    H, W = 13, 21
    x = torch.randn(H, W, Cc, generator=g).to(TDT[dtype])
    w = torch.randn(Cc, 9, generator=g) / 3
    Co = Cc // 2 if mode else Cc
    out = torch.empty((H, W, Co), dtype=TDT[dtype], device="cuda")
    xd, wd = x.cuda(), w.cuda()
    _lib.check(hip_lib.fw_dwconv3x3_nhwc(_lib.DTYPES[dtype], P(xd), Cc, H, W, Cc, P(wd), mode, P(out), Co, _st()))
    torch.cuda.synchronize()
    y = F.conv2d(x.float().permute(2, 0, 1).unsqueeze(0), w.reshape(Cc, 1, 3, 3), None, 1, 1, 1, Cc)
    if mode:
        y1, y2 = y.chunk(2, 1)
        y = F.gelu(y1) * y2
    want = y.squeeze(0).permute(1, 2, 0)
    tol = 4e-3 if dtype == "f16" else 3e-2
    assert (out.float().cpu() - want).abs().max() < tol * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("heads,ch,M", [(1, 48, 1000), (2, 48, 333), (8, 48, 64), (1, 96, 500)])
def test_attention_matrix_and_apply(hip_lib, dtype, heads, ch, M):
    g = torch.Generator().manual_seed(heads * 100 + ch)
    dim = heads * ch
    cp = (dim + 31) // 32 * 32
    qkv = torch.zeros(M, 3 * cp)
    for t in range(3):
        qkv[:, t * cp:t * cp + dim] = torch.randn(M, dim, generator=g)
    qkv = qkv.to(TDT[dtype])
    temp = 1 + torch.rand(heads, generator=g)
    qd, td = qkv.cuda(), temp.cuda()
    ws = torch.empty(hip_lib.fw_attn_workspace_floats(heads, ch), dtype=torch.float32, device="cuda")
    attn = torch.empty((heads, ch, ch), dtype=torch.float32, device="cuda")
    dt = _lib.DTYPES[dtype]
    _lib.check(hip_lib.fw_attn_matrix(dt, P(qd), 3 * cp, M, cp, heads, ch, P(td), P(ws), P(attn), _st()))
    out = torch.full((M, cp), 7.0, dtype=TDT[dtype], device="cuda")
    _lib.check(hip_lib.fw_attn_apply(dt, P(qd), 3 * cp, M, 2 * cp, heads, ch, P(attn), P(out), cp, cp, _st()))
    # the same matrices from the MFMA Gram path (pixel-major q, k; what the engine runs)
    attn_m = torch.empty_like(attn)
    scratch = torch.empty(int(hip_lib.fw_attn_qk_scratch_elems(M, heads, ch)), dtype=TDT[dtype], device="cuda")
    _lib.check(hip_lib.fw_attn_matrix_mfma(dt, P(qd), 3 * cp, M, cp, heads, ch, P(td), P(ws), P(scratch), P(attn_m), _st()))
    torch.cuda.synchronize()
    assert (attn_m - attn).abs().max().item() < 1e-5
    f = qkv.float()
    q, k, v = (f[:, t * cp:t * cp + dim].T.reshape(heads, ch, M) for t in range(3))
    a = (F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)) * temp[:, None, None]
    a = a.softmax(-1)
    assert (attn.cpu() - a).abs().max() < 1e-5
    want = (a @ v).reshape(dim, M).T
    got = out.float().cpu()
    tol = 3e-3 if dtype == "f16" else 2e-2
    assert (got[:, :dim] - want).abs().max() < tol * max(1.0, want.abs().max().item())
    assert (got[:, dim:] == 0).all()
    # the same product through the MFMA GEMM with the block-diagonal packed matrix (what the engine runs)
    apk = torch.empty(int(hip_lib.fw_pack_pointwise(dt, None, cp, cp, None)), dtype=torch.int16, device="cuda")
    _lib.check(hip_lib.fw_attn_pack(dt, P(attn), heads, ch, cp, P(apk), _st()))
    out2 = torch.full((M, cp), 7.0, dtype=TDT[dtype], device="cuda")
    _lib.check(hip_lib.fw_pointwise_nhwc(dt, C.c_void_p(qd.data_ptr() + 2 * cp * 2), 0, 3 * cp, M, cp, P(apk), None, cp // 32,
                                         P(out2), cp, None, 0, None, None, _st()))
    torch.cuda.synchronize()
    got2 = out2.float().cpu()
    assert (got2[:, :dim] - want).abs().max() < 3 * tol * max(1.0, want.abs().max().item())   # attn rounded to the operand type
    assert (got2[:, dim:] == 0).all()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("M,K,N,lda,ld", [(777, 96, 96, 128, 96), (4096, 1024, 384, 1024, 384), (5000, 384, 1152, 384, 1152), (300, 192, 192, 576, 192),
                                          (20000, 96, 96, 128, 96), (17000, 256, 96, 256, 96), (16500, 64, 64, 192, 64), (16400, 192, 576, 192, 576)])
def test_pointwise_1x1_typed_store_and_residual(hip_lib, dtype, M, K, N, lda, ld):
    """fw_pointwise_nhwc on a typed operand: the few-pixel kernel (M <= 16384: pointwise_small_kernel, nn_ops.hip) and the large one, typed
    store and fp32 residual epilogue (the stream updated in place), against torch fp32 on the operand-rounded inputs; lanes behind N untouched."""
    g = torch.Generator().manual_seed(M + K + N)
    dt = _lib.DTYPES[dtype]
    a = torch.zeros(M, lda)
    a[:, :K] = torch.randn(M, K, generator=g)
    a = a.to(TDT[dtype])
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    Np = (N + 31) // 32 * 32
    wp = np.zeros((Np, K), np.float32)
    wp[:N] = w.numpy()
    n16 = int(hip_lib.fw_pack_pointwise(dt, None, Np, K, None))
    pk = np.zeros(n16, np.uint16)
    assert int(hip_lib.fw_pack_pointwise(dt, wp.ctypes.data_as(C.c_void_p), Np, K, pk.ctypes.data_as(C.c_void_p))) == n16
    pkd, ad = torch.from_numpy(pk.view(np.int16)).cuda(), a.cuda()
    want = a[:, :K].float() @ w.to(TDT[dtype]).float().T
    scale_ref = max(1.0, want.abs().max().item())
    tol = 2e-3 if dtype == "f16" else 1.6e-2
    # typed store
    out = torch.full((M, ld + 8), 7.0, dtype=TDT[dtype], device="cuda")
    _lib.check(hip_lib.fw_pointwise_nhwc(dt, P(ad), 0, lda, M, K, P(pkd), None, Np // 32, P(out), ld + 8, None, 0, None, None, _st()))
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert (got[:, :N] - want).abs().max() < tol * scale_ref
    assert (got[:, Np:] == 7.0).all()
    # residual: x += y * scale, fp32 stream in place
    x = torch.randn(M, ld, generator=g)
    sc = 0.5 + torch.rand(Np, generator=g)
    xd, sd = x.clone().cuda(), sc.cuda()
    if Np <= ld:
        _lib.check(hip_lib.fw_pointwise_nhwc(dt, P(ad), 0, lda, M, K, P(pkd), None, Np // 32, None, 0, P(xd), ld, P(xd), P(sd), _st()))
        torch.cuda.synchronize()
        got = xd.cpu()
        assert (got[:, :N] - (x[:, :N] + want * sc[:N])).abs().max() < 2e-5 * scale_ref + 1e-5   # fp32 accumulate and epilogue: only the sum order differs


def test_pixel_shuffle_and_unshuffle(hip_lib):
    g = torch.Generator().manual_seed(3)
    h, w, c = 5, 7, 6
    lo = torch.randn(h, w, 4 * c + 8, generator=g)
    hi = torch.zeros(2 * h, 2 * w, c + 2, device="cuda")
    lod = lo.cuda()
    _lib.check(hip_lib.fw_pixel_shuffle2_f32(P(lod), lo.shape[2], h, w, c, P(hi), c + 2, 1, 0, _st()))
    torch.cuda.synchronize()
    want = F.pixel_shuffle(lo[:, :, :4 * c].permute(2, 0, 1).unsqueeze(0), 2).squeeze(0).permute(1, 2, 0)
    assert torch.equal(hi.cpu()[:, :, 1:1 + c], want)
    back = torch.zeros(h, w, 4 * c, device="cuda")
    src = hi[:, :, 1:1 + c].contiguous()
    _lib.check(hip_lib.fw_pixel_shuffle2_f32(P(src), c, h, w, c, P(back), 4 * c, 0, 1, _st()))
    torch.cuda.synchronize()
    assert torch.equal(back.cpu(), lo[:, :, :4 * c])


SMALL = dict(dim=48, num_blocks=(1, 1, 1, 1), num_refinement_blocks=1, heads=(1, 2, 4, 8), ffn_expansion_factor=2.66)


@pytest.mark.parametrize("dtype,max_abs,min_psnr", [("f16", 2e-3, 55.0), ("bf16", 1.5e-2, 45.0)])
def test_restormer_vs_oracle(hip_lib, dtype, max_abs, min_psnr):
    sd = RS.synthetic_restormer_state(seed=4, **SMALL)
    H, W = 40, 56
    frame = synthetic_frames(1, H, W, seed=12)[0]
    eng = RS.RestormerEngine(dtype=dtype, **SMALL)
    eng.load_state_dict(sd)
    t = torch.from_numpy(frame).cuda()
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    u8 = torch.empty_like(t)
    eng.denoise_device(t, out=u8, out_rgb_f32=rgb)
    torch.cuda.synchronize()
    x = torch.from_numpy(frame[:, :, ::-1].astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    with torch.no_grad():
        want = ref.restormer_forward({k: torch.from_numpy(v) for k, v in sd.items()}, x, SMALL["num_blocks"],
                                     SMALL["num_refinement_blocks"], SMALL["heads"]).squeeze(0).permute(1, 2, 0).numpy()
    got = rgb.cpu().numpy()
    assert np.abs(want - x.squeeze(0).permute(1, 2, 0).numpy()).std() > 0.02      # the synthetic net does something
    assert np.abs(got - want).max() < max_abs
    want_u8 = np.clip(want * 255.0, 0, 255).astype(np.uint8)[:, :, ::-1]          # truncation, tap_denoise.py:399-415
    mse = np.mean((u8.cpu().numpy().astype(np.float64) - want_u8.astype(np.float64)) ** 2)
    assert (99.0 if mse == 0 else 10 * math.log10(255.0 ** 2 / mse)) >= min_psnr
    with pytest.raises(ValueError, match="divisible by 8"):
        eng.denoise_device(torch.zeros((30, 56, 3), dtype=torch.uint8, device="cuda"))
    eng.close()


@pytest.mark.parametrize("dw_mfma", ["0", "1"])
@pytest.mark.parametrize("H,W", [(40, 56), (136, 200)])
def test_fused_fronts_equal_the_staged_kernels(hip_lib, monkeypatch, H, W, dw_mfma):
    """LayerNorm + 1x1 + depthwise 3x3 (+ GDFN gate) of the 48- / 96-channel blocks as one kernel (pw_dw_fused.hip) against the
    three kernels it replaces (FW_REST_FUSE_FRONT=0).  Not bit-equal: the LayerNorm's affine part is folded into the 1x1 weights and
    the GELU's erf is a 1.5e-7 polynomial.  136 x 200 has ragged tiles in both directions and several tiles per workgroup row."""
    monkeypatch.setenv("FW_PW_DW_MFMA", dw_mfma)   # "1": the fused kernel with its depthwise phase on the matrix cores (pw_dw_mfma_kernel, opt-in)
    sd = RS.synthetic_restormer_state(seed=5, **SMALL)
    t = torch.from_numpy(synthetic_frames(1, H, W, seed=13)[0]).cuda()
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("FW_REST_FUSE_FRONT", mode)
        eng = RS.RestormerEngine(dtype="f16", **SMALL)
        eng.load_state_dict(sd)
        rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        u8 = torch.empty_like(t)
        eng.denoise_device(t, out=u8, out_rgb_f32=rgb)
        eng.denoise_device(t, out=u8, out_rgb_f32=rgb)      # twice: deterministic, no state left behind
        torch.cuda.synchronize()
        outs.append((rgb.cpu().numpy(), u8.cpu().numpy()))
        eng.close()
    (ra, ua), (rb, ub) = outs
    assert np.isfinite(ra).all()
    assert np.abs(ra - rb).max() < 1.5e-3 and np.abs(ra - rb).mean() < 3e-4      # measured 5.9e-4 / 1.0e-4: two f16 roundings apart
    assert np.abs(ua.astype(int) - ub.astype(int)).max() <= 1


def test_tap_denoiser_default_model_is_restormer_end_to_end(hip_lib, tmp_path, monkeypatch):
    """TAPDenoiser with the reference's default config (model = RESTORMER, tap_denoise.py:110) through the full-size
    network (4/6/6/8 + 4 blocks) with seeded weights: whole-frame and tiled paths against the oracle composed with the
    oracle of the tiling arithmetic."""
    from framewright_amd import tap_denoise as T
    from oracle import tap_ref
    monkeypatch.setenv("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS", "1")
    frames = list(synthetic_frames(2, 32, 48, seed=31))
    dn = T.TAPDenoiser(T.TAPDenoiseConfig(tile_size=0, temporal_window=1), model_dir=tmp_path / "none")
    assert dn.config.model is T.TAPModel.RESTORMER and dn.is_available()
    got = dn.denoise_clip(frames)
    sd = {k: torch.from_numpy(v) for k, v in RS.synthetic_restormer_state(**RS.RESTORMER_ARGS).items()}

    def oracle(frame):
        x = torch.from_numpy(frame[:, :, ::-1].astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
        with torch.no_grad():
            y = ref.restormer_forward(sd, x).squeeze(0).permute(1, 2, 0).numpy()
        return np.clip(y * 255.0, 0, 255).astype(np.uint8)[:, :, ::-1]

    for g, f in zip(got, frames):
        w = oracle(f)
        assert g.shape == f.shape and np.abs(g.astype(int) - w.astype(int)).max() <= 2
    # tiled: 32x32 tiles with 8 overlap on the 32x48 frame -> 2 tiles, concurrent streams, ordered blend
    dn2 = T.TAPDenoiser(T.TAPDenoiseConfig(tile_size=32, tile_overlap=8, temporal_window=1), engine=dn._engine)
    tiled = dn2.denoise_clip(frames[:1])[0]
    model = lambda t: ref.restormer_forward(sd, t)
    want = tap_ref.denoise_frame_tiled(model, frames[0], 32, 8)
    assert np.abs(tiled.astype(int) - want.astype(int)).max() <= 2
    dn.clear_cache()
