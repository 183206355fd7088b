"""GPU parity: colour transform + multi-level CDF 5/3 through the C ABI vs the oracle."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

SHAPES = [(8, 8, 1), (8, 9, 3), (9, 8, 1), (15, 15, 3), (16, 16, 1), (53, 37, 3), (77, 131, 3), (300, 17, 1),
          (17, 300, 3), (255, 257, 1), (257, 255, 3), (64, 64, 3), (240, 320, 3), (129, 1000, 1), (1080, 1920, 3),
          (540, 9, 1)]


def to_planes(a):
    """[H,W,C] interleaved -> [C,H,W]"""
    return np.ascontiguousarray(np.moveaxis(a, 2, 0))


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_matches_oracle(ctx, shape):
    import torch

    H, W, Cn = shape
    pix = orc.synth(W, H, Cn, 3, 0)
    coef, _, _ = orc.stage_dump(pix)
    t = torch.from_numpy(pix[None]).cuda()
    planes = ctx.planes_from_pixels(t)
    pyr = ctx.transformation_fwd(planes)
    got = pyr.cpu().numpy()
    assert (got == to_planes(coef)).all()
    back = ctx.transformation_inv(pyr)
    assert torch.equal(back, planes)
    out = ctx.pixels_from_planes(back, Cn)
    assert (out.cpu().numpy()[0] == pix).all()


@pytest.mark.parametrize("shape", [(8, 8), (33, 71), (128, 96), (511, 513)])
def test_random_int_planes_batch(ctx, shape):
    """Batch of planes with full-range-ish ints: forward equals the oracle, inverse restores."""
    import torch

    H, W = shape
    rng = np.random.default_rng(5)
    P = 5
    a = rng.integers(-5000, 5000, size=(P, H, W), dtype=np.int32)
    want = np.stack([orc.forward(a[p][:, :, None])[:, :, 0] for p in range(P)])
    pyr = ctx.transformation_fwd(torch.from_numpy(a).cuda())
    assert (pyr.cpu().numpy() == want).all()
    back = ctx.transformation_inv(pyr)
    assert (back.cpu().numpy() == a).all()
    # inverse of an arbitrary (non-transform) pyramid also matches the oracle
    inv_want = np.stack([orc.inverse(a[p][:, :, None])[:, :, 0] for p in range(P)])
    inv_got = ctx.transformation_inv(torch.from_numpy(a).cuda())
    assert (inv_got.cpu().numpy() == inv_want).all()


@pytest.mark.parametrize("shape", [(200, 4), (333, 8), (131, 12), (65, 256), (66, 260), (67, 516), (257, 1028), (1030, 68), (2050, 72), (3, 128), (2, 512)])
def test_wide_kernels_at_their_edges(ctx, shape):
    """The 16-byte-per-lane kernels work in batches of two row pairs with loads from clamped places (lift.hip): one quad
    per row, one lane past a full strip, odd heights (the last row pair has no odd row), strips that end inside a batch,
    heights of two and three rows.  int32 planes both ways against the oracle, and 8-bit pixels through the codec."""
    import torch

    H, W = shape
    rng = np.random.default_rng(H * 7 + W)
    a = rng.integers(-3000, 3000, size=(3, H, W), dtype=np.int32)
    want = np.stack([orc.forward(a[p][:, :, None])[:, :, 0] for p in range(3)])
    pyr = ctx.transformation_fwd(torch.from_numpy(a).cuda())
    assert (pyr.cpu().numpy() == want).all()
    inv_want = np.stack([orc.inverse(a[p][:, :, None])[:, :, 0] for p in range(3)])
    assert (ctx.transformation_inv(torch.from_numpy(a).cuda()).cpu().numpy() == inv_want).all()
    if H >= 8 and W >= 8:
        for Cn in (1, 3):
            pix = np.stack([orc.synth(W, H, Cn, 11, 1), (rng.integers(0, 2, (H, W, Cn)) * 255).astype(np.uint8)])
            streams, _ = ctx.encode(pix)
            assert streams == [orc.encode(x)[0] for x in pix]
            assert all((o == x).all() for o, x in zip(ctx.decode(streams), pix))


def test_4096_gray_roundtrip_and_checksum(ctx):
    """BASELINE config B size: round trip + checksum against the oracle's pyramid."""
    import torch

    pix = orc.synth(4096, 4096, 1, 0, 0)
    coef, _, _ = orc.stage_dump(pix)
    t = torch.from_numpy(pix[None]).cuda()
    planes = ctx.planes_from_pixels(t)
    pyr = ctx.transformation_fwd(planes)
    assert (pyr.cpu().numpy()[0] == coef[:, :, 0]).all()
    assert torch.equal(ctx.transformation_inv(pyr), planes)


def test_small_inverse_root_only(ctx):
    """decode.c quirk: a root-only decode still runs one inverse level on a <8 image."""
    import torch

    rng = np.random.default_rng(2)
    for (H, W) in [(4, 5), (5, 8), (4, 4), (7, 6)]:
        a = rng.integers(-100, 100, size=(2, H, W), dtype=np.int32)
        want = np.stack([orc.inverse(a[p][:, :, None])[:, :, 0] for p in range(2)])
        got = ctx.transformation_inv(torch.from_numpy(a).cuda())
        assert (got.cpu().numpy() == want).all()


def merge_rings(pyr, r16, mask, W, H):
    """the pyramid as one int32 array: the ring levels in `mask` come from the 16-bit planes"""
    g = orc.geometry(W, H)
    out = pyr.copy()
    for l in range(g.levels):
        if (mask >> l) & 1:
            w0, h0, w1, h1 = g.widths[l], g.heights[l], g.widths[l + 1], g.heights[l + 1]
            out[:, :h1, w0:w1] = r16[:, :h1, w0:w1]
            out[:, h0:h1, :w0] = r16[:, h0:h1, :w0]
    return out


@pytest.mark.parametrize("shape", [(128, 128, 1), (96, 260, 3), (1080, 1920, 3), (67, 516, 1), (1024, 1024, 1), (540, 72, 3), (257, 1028, 3)])
@pytest.mark.parametrize("rings16", [True, False])
def test_pixel_transforms_as_the_pipelines_run_them(ctx, shape, rings16):
    """dwtx_transformation_fwd_pixels / _inv_pixels (encode.c:155-159, decode.c:258-264 in one pass each, the finest rings
    as 16-bit values): the pyramid is the oracle's, a noise picture with the largest steps an 8-bit source can make too,
    and the inverse of the pyramid gives the pixels back."""
    import torch

    H, W, Cn = shape
    rng = np.random.default_rng(W + H)
    pix = np.stack([orc.synth(W, H, Cn, 4, 0), orc.synth(W, H, Cn, 5, 1), (rng.integers(0, 2, (H, W, Cn)) * 255).astype(np.uint8)])
    t = torch.from_numpy(pix).cuda()
    pyr, r16, mask = ctx.transformation_fwd_pixels(t, rings16=rings16)
    assert rings16 or mask == 0
    got = merge_rings(pyr.cpu().numpy(), r16.cpu().numpy() if r16 is not None else None, mask, W, H)
    want = np.concatenate([to_planes(orc.stage_dump(p)[0]) for p in pix])
    assert (got == want).all()
    back = ctx.transformation_inv_pixels(pyr, r16, mask, Cn)
    assert torch.equal(back, t)


def test_pixel_transforms_refuse_shapes_their_kernels_do_not_take(ctx):
    import torch

    for H, W in ((100, 131), (64, 64), (40, 40)):
        t = torch.zeros((1, H, W, 1), dtype=torch.uint8, device=ctx.device)
        with pytest.raises(RuntimeError):
            ctx.transformation_fwd_pixels(t)


@pytest.mark.parametrize("shape", [(68, 244), (132, 248), (260, 488), (516, 492), (72, 976), (128, 980), (388, 1220), (1080, 1920), (2048, 2048),
                                   (68, 68), (512, 4), (260, 8), (1028, 260), (68, 192), (132, 196), (260, 224), (516, 228), (1040, 448), (76, 452),
                                   (4096, 96), (644, 3588)])
def test_two_levels_per_pass_equal_one_launch_per_level(ctx, shape, opts):
    """k_fwd2_level_w / k_inv2_level_w (two levels of cdf53.h:9-61 / encode.c:16-30, decode.c:16-30 in one pass, the LL band between them never in memory):
    widths around whole numbers of 61-quad wave strips, heights that end strips inside / at the end of a row pair of the
    coarser level, the smallest shapes it takes — against the oracle, and the same bytes as one launch per level."""
    import torch

    H, W = shape
    rng = np.random.default_rng(H + 3 * W)
    a = rng.integers(-40000, 40000, size=(3, H, W), dtype=np.int32)
    want = np.stack([orc.forward(a[p][:, :, None])[:, :, 0] for p in range(3)])
    t = torch.from_numpy(a).cuda()
    fused = ctx.transformation_fwd(t)
    assert (fused.cpu().numpy() == want).all()
    assert torch.equal(ctx.transformation_inv(fused), t)                       # k_inv2_level_w
    inv_want = np.stack([orc.inverse(a[p][:, :, None])[:, :, 0] for p in range(3)])   # ... and of an arbitrary pyramid
    inv_fused = ctx.transformation_inv(t)
    assert (inv_fused.cpu().numpy() == inv_want).all()
    opts.set("no_fused_levels", 1)
    plain = ctx.transformation_fwd(t)
    assert torch.equal(plain, fused)
    assert torch.equal(ctx.transformation_inv(fused), t)
    assert torch.equal(ctx.transformation_inv(t), inv_fused)
