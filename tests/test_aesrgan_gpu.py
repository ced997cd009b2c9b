"""The in-tree AESRGAN (RRDB trunk + full-image self-attention, aesrgan_face.py:142-269) on the GPU against vectors the
reference's own module produced (tests/golden/aesrgan_attention.npz) and against the fp32 oracle on a larger crop."""
import numpy as np
import pytest
import torch

from framewright_amd import aesrgan as A
from framewright_amd.synth import aesrgan_attention_positions, synthetic_attention_state, synthetic_rrdbnet_state
from oracle import rrdbnet_ref as ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [("f16", 1.5e-3), ("bf16", 1.2e-2)])
@pytest.mark.parametrize("tag", ["s4", "s2"])
def test_against_reference_run_vectors(hip_lib, golden_dir, dtype, tol, tag):
    g = np.load(golden_dir / "aesrgan_attention.npz")
    nb, na, scale, seed = (int(v) for v in g[tag + "_cfg"])
    eng = A.AESRGANEngine(nb, scale, na, dtype)
    eng.load_state_dict(synthetic_rrdbnet_state(nb, 4, seed=seed), synthetic_attention_state(nb, na, seed=seed + 100))
    x = torch.from_numpy(np.ascontiguousarray(np.transpose(g[tag + "_in"][0], (1, 2, 0)))).cuda()
    y = eng.forward_rgb(x)
    torch.cuda.synchronize()
    want = np.transpose(g[tag + "_out"][0], (1, 2, 0))
    assert tuple(y.shape) == want.shape
    err = np.abs(y.cpu().numpy() - want).max()
    assert err < tol, err
    eng.close()


def test_attention_contributes_and_matches_oracle_on_a_crop(hip_lib):
    nb, na, scale = 4, 2, 2
    sd, asd = synthetic_rrdbnet_state(nb, 4, seed=3), synthetic_attention_state(nb, na, seed=4)
    eng = A.AESRGANEngine(nb, scale, na, "f16")
    eng.load_state_dict(sd, asd)
    rng = np.random.default_rng(1)
    x = rng.uniform(0, 1, (37, 45, 3)).astype(np.float32)          # 1665 pixels: not a multiple of 32
    y = eng.forward_rgb(torch.from_numpy(x).cuda()).cpu().numpy()
    t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    xin = torch.from_numpy(np.transpose(x, (2, 0, 1))[None])
    with torch.no_grad():
        want = ref.aesrgan_forward(t(sd), t(asd), xin, nb, scale, aesrgan_attention_positions(nb, na))[0].permute(1, 2, 0).numpy()
        inert = ref.aesrgan_forward(t(sd), t({k: (v * 0 if k.endswith("gamma") else v) for k, v in asd.items()}), xin, nb, scale,
                                    aesrgan_attention_positions(nb, na))[0].permute(1, 2, 0).numpy()
    assert np.abs(y - want).max() < 1.5e-3
    assert np.abs(inert - want).max() > 10 * np.abs(y - want).max()      # the blocks matter at this tolerance
    face = (x * 255).astype(np.uint8)[:, :, ::-1].copy()
    out = eng.enhance(face)
    assert out.shape == (74, 90, 3) and out.dtype == np.uint8
    eng.close()
