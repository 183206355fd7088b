"""GPU parity: Hilbert linearisation / reconstruction through the C ABI vs the oracle."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

SHAPES = [(8, 8, 1), (8, 9, 3), (15, 15, 3), (16, 16, 1), (53, 37, 3), (77, 131, 3), (300, 17, 1), (17, 300, 3),
          (255, 257, 1), (64, 64, 3), (240, 320, 3), (1080, 1920, 3), (512, 512, 1), (9, 1000, 1)]


@pytest.mark.parametrize("shape", SHAPES)
def test_linearization_matches_oracle(ctx, shape):
    import torch

    H, W, Cn = shape
    pix = orc.synth(W, H, Cn, 9, 0)
    coef, lin, _ = orc.stage_dump(pix)
    pyr = torch.from_numpy(np.ascontiguousarray(np.moveaxis(coef, 2, 0))).cuda()
    got = ctx.linearization(pyr)
    assert (got.cpu().numpy() == lin).all()
    back = ctx.reconstruction(got, W, H, Cn)
    assert torch.equal(back, pyr)


@pytest.mark.parametrize("shape", [(77, 131, 3), (240, 320, 3), (64, 64, 1), (300, 17, 1)])
def test_reconstruction_partial_levels_and_bias(ctx, shape):
    """Truncated-decode semantics: fewer levels, dequantisation bias from `missing` (decode.c:51-58)."""
    import torch

    H, W, Cn = shape
    rng = np.random.default_rng(3)
    g = orc.geometry(W, H)
    n = 2
    lin = rng.integers(-40, 40, size=(n, Cn, W * H), dtype=np.int32)
    lin[rng.random(lin.shape) < 0.5] = 0
    missing = np.zeros((n, 3, 16), dtype=np.int32)
    missing[:, :Cn, : g.levels] = rng.integers(0, 7, size=(n, Cn, g.levels))
    for levels_out in sorted({0, 1, g.levels // 2, g.levels}):
        want = np.stack([np.moveaxis(orc.reconstruct(lin[i], W, H, levels_out, missing[i].reshape(-1)), 2, 0)
                         for i in range(n)]).reshape(n * Cn, g.heights[levels_out], g.widths[levels_out])
        got = ctx.reconstruction(torch.from_numpy(lin.reshape(n * Cn, -1)).cuda(), W, H, Cn, levels_out,
                                 torch.from_numpy(missing.reshape(-1)).cuda())
        assert (got.cpu().numpy() == want).all()


def test_linearization_4096(ctx):
    import torch

    pix = orc.synth(4096, 4096, 1, 0, 0)
    coef, lin, _ = orc.stage_dump(pix)
    pyr = torch.from_numpy(np.ascontiguousarray(np.moveaxis(coef, 2, 0))).cuda()
    got = ctx.linearization(pyr)
    assert (got.cpu().numpy() == lin).all()
    assert torch.equal(ctx.reconstruction(got, 4096, 4096, 1), pyr)
