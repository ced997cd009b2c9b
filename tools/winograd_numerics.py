#!/usr/bin/env python3
"""CPU experiment (numerics only, no kernel): would Winograd F(2x2, 3x3) with f16 operands keep Real-ESRGAN x4plus inside the 1e-3
bar?  The 23-block x4 generator on a crop, three ways against the fp32 oracle:

  direct    every conv on f16-rounded inputs and weights, fp32 products and accumulation - what the MFMA kernels compute (the trunk's
            residual adds stay fp32: the hi + lo split carries 22 bits);
  winograd  the trunk convs (conv1 .. conv5 of every dense block: 90 % of the frame's MACs) as F(2x2, 3x3): input tiles transformed
            in fp32 from the f16-rounded activations and THEN rounded to f16 (V = B^T d B), weights transformed in fp32 and
            rounded to f16 (U = G g G^T), elementwise products summed over channels in fp32, output transform in fp32;
  winograd_v32  the same with V kept in fp32 (only U rounded): what a tf32-like / split operand would give.

  winograd_f4x4 / _v32  F(4x4, 3x3) the same two ways (4x fewer MACs, far larger transform constants).

Prints max-abs and PSNR of each against fp32.  Uses seeded synthetic weights (the tests' generator)."""
import sys, math
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch, torch.nn.functional as F
from framewright_amd.synth import synthetic_frames, synthetic_rrdbnet_state
from oracle import rrdbnet_ref as ref

torch.set_num_threads(8)
h16 = lambda t: t.to(torch.float16).to(torch.float32)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
# F(4x4, 3x3), Lavin & Gray's matrices: 36 products per 16 outputs (4x fewer MACs than direct), constants up to 8 and 1/24
BT6 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
                   dtype=torch.float32)
G6 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                  dtype=torch.float32)
AT6 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float32)


def conv_direct(x, w, b):
    return F.conv2d(h16(x), h16(w), b, padding=1)


def conv_winograd(x, w, b, round_v=True, m=2):
    n, c, H, W = x.shape
    assert H % m == 0 and W % m == 0
    bt, g, at = (BT, G, AT) if m == 2 else (BT6, G6, AT6)
    xp = F.pad(h16(x), (1, 1, 1, 1))
    tiles = xp.unfold(2, m + 2, m).unfold(3, m + 2, m)             # n c th tw (m+2) (m+2)
    V = torch.einsum("ij,nctujk,lk->nctuil", bt, tiles, bt)        # B^T d B
    if round_v:
        V = h16(V)
    U = h16(torch.einsum("ij,ocjk,lk->ocil", g, w, g))             # G g G^T, rounded
    Mm = torch.einsum("ocil,nctuil->notuil", U, V)                 # sum over channels, fp32
    Y = torch.einsum("ij,notujk,lk->notuil", at, Mm, at)           # n o th tw m m
    th, tw = Y.shape[2], Y.shape[3]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], th * m, tw * m)
    return y + b.view(1, -1, 1, 1)


def forward(sd, x, nb, trunk_conv):
    c = lambda k, t, f=conv_direct: f(t, sd[k + ".weight"], sd[k + ".bias"])
    lr = lambda t: F.leaky_relu(t, 0.2)
    feat = c("conv_first", x)
    body = feat
    for i in range(nb):
        rin = body
        out = body
        for r in (1, 2, 3):
            p = f"body.{i}.rdb{r}"
            x0 = out
            x1 = lr(c(p + ".conv1", x0, trunk_conv))
            x2 = lr(c(p + ".conv2", torch.cat([x0, x1], 1), trunk_conv))
            x3 = lr(c(p + ".conv3", torch.cat([x0, x1, x2], 1), trunk_conv))
            x4 = lr(c(p + ".conv4", torch.cat([x0, x1, x2, x3], 1), trunk_conv))
            x5 = c(p + ".conv5", torch.cat([x0, x1, x2, x3, x4], 1), trunk_conv)
            out = x5 * 0.2 + x0
        body = out * 0.2 + rin
    feat = feat + c("conv_body", body)
    feat = lr(c("conv_up1", F.interpolate(feat, scale_factor=2, mode="nearest")))
    feat = lr(c("conv_up2", F.interpolate(feat, scale_factor=2, mode="nearest")))
    return c("conv_last", lr(c("conv_hr", feat)))


def main():
    nb, size = 23, int(sys.argv[1]) if len(sys.argv) > 1 else 64
    sd = {k: torch.from_numpy(v) for k, v in synthetic_rrdbnet_state(nb, 4, seed=1234).items()}
    frame = synthetic_frames(1, size, size, seed=size * size)[0]
    x = torch.from_numpy(frame[:, :, ::-1].astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    with torch.no_grad():
        want = ref.rrdbnet_forward(sd, x, nb, 4)
        for name, fn in (("direct", conv_direct), ("winograd", conv_winograd), ("winograd_v32", lambda a, w, b: conv_winograd(a, w, b, False)),
                         ("winograd_f4x4", lambda a, w, b: conv_winograd(a, w, b, True, 4)), ("winograd_f4x4_v32", lambda a, w, b: conv_winograd(a, w, b, False, 4))):
            got = forward(sd, x, nb, fn)
            d = (got - want).abs()
            mse = float(((got.clamp(0, 1) - want.clamp(0, 1)) ** 2).mean())
            print(f"{name:14s} max-abs {float(d.max()):.3e}  mean-abs {float(d.mean()):.3e}  psnr {10 * math.log10(1 / max(mse, 1e-20)):.1f} dB", flush=True)


main()
