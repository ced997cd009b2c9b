"""CPU-side checks of the SRVGGNetCompact support: key/shape tables, checkpoint routing predicate, and the oracle against
an independent torch.nn restatement of the published module (nn.Conv2d / nn.PReLU / nn.PixelShuffle)."""
import numpy as np
import torch
import torch.nn as nn

from framewright_amd import srvgg as S
from oracle import srvgg_ref as ref


def _module(num_conv, scale, feat=64):
    body = [nn.Conv2d(3, feat, 3, 1, 1), nn.PReLU(num_parameters=feat)]
    for _ in range(num_conv):
        body += [nn.Conv2d(feat, feat, 3, 1, 1), nn.PReLU(num_parameters=feat)]
    body.append(nn.Conv2d(feat, 3 * scale * scale, 3, 1, 1))
    return nn.ModuleDict({"body": nn.Sequential(*body)}), nn.PixelShuffle(scale)


def test_oracle_matches_nn_modules():
    num_conv, scale = 3, 4
    sd = S.synthetic_srvgg_state(num_conv, scale, seed=1)
    net, shuffle = _module(num_conv, scale)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)   # same keys as the checkpoints
    x = torch.rand(1, 3, 9, 13)
    with torch.no_grad():
        want = shuffle(net["body"](x)) + nn.functional.interpolate(x, scale_factor=scale, mode="nearest")
        got = ref.srvgg_forward({k: torch.from_numpy(v) for k, v in sd.items()}, x, num_conv, scale)
    assert torch.allclose(got, want, atol=1e-6)


def test_shapes_and_routing_predicate():
    shapes = dict(S.srvgg_tensor_shapes(16, 4))
    assert shapes["body.0.weight"] == (64, 3, 3, 3) and shapes["body.1.weight"] == (64,)
    assert shapes["body.34.weight"] == (48, 64, 3, 3) and "body.35.weight" not in shapes
    assert len([k for k in shapes if k.endswith(".bias")]) == 18
    sd = S.synthetic_srvgg_state(16, 4)
    assert S.is_srvgg_state_dict(sd) and S.is_srvgg_state_dict({"params": sd}) and S.is_srvgg_state_dict({"params_ema": sd})
    assert not S.is_srvgg_state_dict({"conv_first.weight": np.zeros(1), "body.0.weight": np.zeros(1)})
    assert S.SRVGG_MODELS == {"realesr-animevideov3": (16, 4), "realesr-general-x4v3": (32, 4)}
