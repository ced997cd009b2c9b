"""TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing on the product path).

fp32 CPU restatement of Restormer as the reference constructs it (src/framewright/processors/tap_denoise.py:299-333):
`Restormer(inp_channels=3, out_channels=3, dim=48, num_blocks=[4,6,6,8], num_refinement_blocks=4, heads=[1,2,4,8],
ffn_expansion_factor=2.66, bias=False, LayerNorm_type='WithBias', dual_pixel_task=False)`.  The class lives in the
third-party package `basicsr.archs.restormer_arch` (swz30/Restormer fork, unpinned, absent from /root/reference), so this
follows the published architecture:  PARITY UNPINNED at the third-party boundary.

  block(x)  = x + MDTA(LN(x));  x = x + GDFN(LN(x))
  MDTA      = 1x1 qkv -> 3x3 depthwise -> per head: softmax(normalize(q) normalize(k)^T * temperature) v -> 1x1
              (q, k, v are [c/head, H*W]: the attention matrix is c/head x c/head, "transposed" attention)
  GDFN      = 1x1 (dim -> 2*hidden) -> 3x3 depthwise -> gelu(x1) * x2 -> 1x1 (hidden -> dim), hidden = int(dim * 2.66)
  down/up   = conv3x3 (C -> C/2) + PixelUnshuffle(2)  /  conv3x3 (C -> 2C) + PixelShuffle(2)
"""
import torch
import torch.nn.functional as F


def layer_norm(x, w, b):
    mu = x.mean(1, keepdim=True)
    var = x.var(1, keepdim=True, unbiased=False)
    return (x - mu) / torch.sqrt(var + 1e-5) * w[None, :, None, None] + b[None, :, None, None]


def attention(sd, p, x, heads):
    b, c, h, w = x.shape
    qkv = F.conv2d(x, sd[p + "qkv.weight"])
    qkv = F.conv2d(qkv, sd[p + "qkv_dwconv.weight"], None, 1, 1, 1, 3 * c)
    q, k, v = qkv.chunk(3, dim=1)
    q, k, v = (t.reshape(b, heads, c // heads, h * w) for t in (q, k, v))
    q = F.normalize(q, dim=-1)
    k = F.normalize(k, dim=-1)
    attn = (q @ k.transpose(-2, -1)) * sd[p + "temperature"]
    attn = attn.softmax(dim=-1)
    out = (attn @ v).reshape(b, c, h, w)
    return F.conv2d(out, sd[p + "project_out.weight"])


def feed_forward(sd, p, x):
    y = F.conv2d(x, sd[p + "project_in.weight"])
    y = F.conv2d(y, sd[p + "dwconv.weight"], None, 1, 1, 1, y.shape[1])
    x1, x2 = y.chunk(2, dim=1)
    return F.conv2d(F.gelu(x1) * x2, sd[p + "project_out.weight"])


def block(sd, p, x, heads):
    x = x + attention(sd, p + "attn.", layer_norm(x, sd[p + "norm1.body.weight"], sd[p + "norm1.body.bias"]), heads)
    return x + feed_forward(sd, p + "ffn.", layer_norm(x, sd[p + "norm2.body.weight"], sd[p + "norm2.body.bias"]))


def stage(sd, name, x, n, heads):
    for i in range(n):
        x = block(sd, f"{name}.{i}.", x, heads)
    return x


def restormer_forward(sd, inp, num_blocks=(4, 6, 6, 8), num_refinement_blocks=4, heads=(1, 2, 4, 8)):
    conv3 = lambda k, x: F.conv2d(x, sd[k], None, 1, 1)
    e1 = stage(sd, "encoder_level1", conv3("patch_embed.proj.weight", inp), num_blocks[0], heads[0])
    e2 = stage(sd, "encoder_level2", F.pixel_unshuffle(conv3("down1_2.body.0.weight", e1), 2), num_blocks[1], heads[1])
    e3 = stage(sd, "encoder_level3", F.pixel_unshuffle(conv3("down2_3.body.0.weight", e2), 2), num_blocks[2], heads[2])
    lat = stage(sd, "latent", F.pixel_unshuffle(conv3("down3_4.body.0.weight", e3), 2), num_blocks[3], heads[3])
    d3 = torch.cat([F.pixel_shuffle(conv3("up4_3.body.0.weight", lat), 2), e3], 1)
    d3 = stage(sd, "decoder_level3", F.conv2d(d3, sd["reduce_chan_level3.weight"]), num_blocks[2], heads[2])
    d2 = torch.cat([F.pixel_shuffle(conv3("up3_2.body.0.weight", d3), 2), e2], 1)
    d2 = stage(sd, "decoder_level2", F.conv2d(d2, sd["reduce_chan_level2.weight"]), num_blocks[1], heads[1])
    d1 = torch.cat([F.pixel_shuffle(conv3("up2_1.body.0.weight", d2), 2), e1], 1)
    d1 = stage(sd, "decoder_level1", d1, num_blocks[0], heads[0])
    d1 = stage(sd, "refinement", d1, num_refinement_blocks, heads[0])
    return conv3("output.weight", d1) + inp
