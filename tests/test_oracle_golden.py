"""The oracle (oracle/rrdbnet_ref.py) against outputs of the reference's own modules
(tests/golden/rrdb_reference.npz, written by oracle/gen_golden.py from
/root/reference/src/framewright/processors/aesrgan_face.py:171-269)."""
import numpy as np
import torch

from framewright_amd.synth import synthetic_rrdbnet_state
from oracle import rrdbnet_ref as ref

TOL = 2e-5  # fp32 conv re-association only


def _sd(num_block, seed):
    return {k: torch.from_numpy(v) for k, v in synthetic_rrdbnet_state(num_block, 4, seed=seed).items()}


def test_rdb_matches_reference(golden_dir):
    g = np.load(golden_dir / "rrdb_reference.npz")
    sd = _sd(1, int(g["rdb_seed"]))
    with torch.no_grad():
        y = ref.rdb_forward(sd, "body.0.rdb1", torch.from_numpy(g["rdb_in"])).numpy()
    assert np.abs(y - g["rdb_out"]).max() < TOL


def test_rrdb_matches_reference(golden_dir):
    g = np.load(golden_dir / "rrdb_reference.npz")
    sd = _sd(1, int(g["rrdb_seed"]))
    with torch.no_grad():
        y = ref.rrdb_forward(sd, "body.0", torch.from_numpy(g["rrdb_in"])).numpy()
    assert np.abs(y - g["rrdb_out"]).max() < TOL


def test_trunk_and_tail_match_reference(golden_dir):
    g = np.load(golden_dir / "rrdb_reference.npz")
    sd = _sd(2, int(g["net_seed"]))
    with torch.no_grad():
        y = ref.rrdbnet_forward(sd, torch.from_numpy(g["net_in"]), num_block=2, scale=4).numpy()
    assert y.shape == (1, 3, 64, 64)
    assert np.abs(y - g["net_out"]).max() < TOL


def test_x2_front_end_is_pixel_unshuffle():
    sd = {k: torch.from_numpy(v) for k, v in synthetic_rrdbnet_state(1, 2, seed=5).items()}
    assert sd["conv_first.weight"].shape == (64, 12, 3, 3)
    x = torch.rand(1, 3, 8, 10)
    with torch.no_grad():
        y = ref.rrdbnet_forward(sd, x, num_block=1, scale=2)
    assert y.shape == (1, 3, 16, 20)


def test_aesrgan_with_live_attention_matches_reference(golden_dir):
    """oracle.aesrgan_forward (RRDB trunk + AttentionBlocks with gamma != 0) against the reference's own AESRGAN module run in
    the build container (tests/golden/aesrgan_attention.npz, oracle/gen_golden.py aesrgan_attention)."""
    from framewright_amd.synth import aesrgan_attention_positions, synthetic_attention_state
    g = np.load(golden_dir / "aesrgan_attention.npz")
    for tag in ("s4", "s2"):
        nb, na, scale, seed = (int(v) for v in g[tag + "_cfg"])
        sd = _sd(nb, seed)
        asd = {k: torch.from_numpy(v) for k, v in synthetic_attention_state(nb, na, seed=seed + 100).items()}
        with torch.no_grad():
            y = ref.aesrgan_forward(sd, asd, torch.from_numpy(g[tag + "_in"]), nb, scale, aesrgan_attention_positions(nb, na)).numpy()
            y0 = ref.aesrgan_forward(sd, {k: (v * 0 if k.endswith("gamma") else v) for k, v in asd.items()},
                                     torch.from_numpy(g[tag + "_in"]), nb, scale, aesrgan_attention_positions(nb, na)).numpy()
        assert y.shape == g[tag + "_out"].shape
        assert np.abs(y - g[tag + "_out"]).max() < TOL
        assert np.abs(y0 - g[tag + "_out"]).max() > 50 * TOL      # the attention blocks really contribute to the vectors
