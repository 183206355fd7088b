"""GPU parity: the decoder's entropy stage (dwtx_decode_planes) vs the oracle."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

SHAPES = [(8, 8, 1), (8, 9, 3), (15, 15, 3), (16, 16, 1), (53, 37, 3), (77, 131, 3), (300, 17, 1), (17, 300, 3),
          (255, 257, 1), (64, 64, 3), (240, 320, 3), (512, 512, 1), (360, 640, 3)]


def check(ctx, data_list, W, H, Cn, levels_max=-1, pixels_max=-1):
    lin, infos = ctx.decode_planes(data_list, W, H, Cn, levels_max)
    got = lin.cpu().numpy().reshape(len(data_list), Cn, W * H)
    for i, data in enumerate(data_list):
        ref = orc.decode_stage(data, W, H, Cn, pixels_max)
        if ref is None:
            assert infos[i].status == 1
            continue
        rlin, level, missing, planes = ref
        if max(planes) > 16:   # the documented difference: the reference decodes on, this decoder refuses
            assert infos[i].status == 2
            continue
        assert infos[i].status == 0
        assert list(infos[i].planes)[:Cn] == planes
        assert infos[i].level == level
        assert list(infos[i].missing) == missing.tolist()
        assert (got[i] == rlin).all()


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("shape", SHAPES)
def test_full_stream(ctx, shape, kind):
    H, W, Cn = shape
    pix = orc.synth(W, H, Cn, 31, kind)
    data, _ = orc.encode(pix)
    check(ctx, [data], W, H, Cn)


def test_every_prefix_of_a_small_stream(ctx):
    """Truncation anywhere (decode.c keeps partial planes): all prefixes in one batch."""
    W, H, Cn = 37, 53, 3
    data, _ = orc.encode(orc.synth(W, H, Cn, 5, 0))
    prefixes = [data[:k] for k in range(1, len(data) + 1)]
    for i in range(0, len(prefixes), 512):
        check(ctx, prefixes[i:i + 512], W, H, Cn)


def test_prefixes_gray_noise(ctx):
    W, H, Cn = 64, 40, 1
    data, _ = orc.encode(orc.synth(W, H, Cn, 6, 1))
    check(ctx, [data[:k] for k in range(1, len(data) + 1, 3)], W, H, Cn)


@pytest.mark.parametrize("px", [0, 1, 16, 100, 1000, 5000, 20000, 100000])
def test_pixels_cap(ctx, px):
    W, H, Cn = 131, 77, 3
    data, _ = orc.encode(orc.synth(W, H, Cn, 5, 0))
    g = orc.geometry(W, H)
    lm = g.levels
    while lm > 0 and g.pixels[lm] > px:
        lm -= 1
    check(ctx, [data], W, H, Cn, levels_max=lm, pixels_max=px)


def test_flat_and_batch(ctx):
    W, H, Cn = 100, 16, 3
    imgs = [np.full((H, W, Cn), 77, dtype=np.uint8), orc.synth(W, H, Cn, 1, 0), orc.synth(W, H, Cn, 2, 1)]
    check(ctx, [orc.encode(p)[0] for p in imgs], W, H, Cn)


def test_smpte_golden_stream(ctx):
    data = open(orc.GOLDEN + "/smpte.dwt", "rb").read()
    check(ctx, [data, data[:4096], data[:100]], 320, 240, 3)


def test_bad_headers(ctx):
    data, _ = orc.encode(orc.synth(64, 64, 1, 1, 0))
    _, infos = ctx.decode_planes([b"X" + data[1:], data[:5], data[:6], data], 64, 64, 1)
    assert [i.status for i in infos] == [1, 1, 1, 0]


@pytest.mark.parametrize("shape", [(53, 37, 3), (255, 257, 1), (360, 640, 3)])
def test_both_path_families_from_the_start(ctx, shape, opts):
    """The decoder normally records one family of speculative paths and only falls back to two (even / odd start,
    unpack.hip k_spec) when its token walk gives up.  DWTX_TWO_FAMILIES starts with both: whole streams, prefixes
    and damaged streams must decode exactly as with one."""
    from test_oracle import corrupted_blobs

    opts.set("two_families", 1)
    H, W, Cn = shape
    data, _ = orc.encode(orc.synth(W, H, Cn, 31, 0))
    step = max(1, len(data) // 150)
    check(ctx, [data] + [data[:k] for k in range(1, len(data), step)], W, H, Cn)
    check(ctx, corrupted_blobs(data, 24, 11), W, H, Cn)   # (a blob that claims more than 16 planes must come back with status 2)


@pytest.mark.parametrize("case", [(40, 24, 1, [17]), (40, 24, 1, [20]), (24, 40, 3, [9, 29, 3]), (64, 64, 1, [24]), (37, 53, 3, [18, 18, 18])])
def test_streams_that_claim_more_than_16_bit_planes_are_refused(ctx, case):
    """decode.c:183-186 accepts any plane count; the reference (and the oracle, pinned on it by tests/test_oracle.py
    on these very streams) decodes such a stream to garbage.  This decoder handles at most 16 planes and says so:
    status 2 for that stream, alone and inside a batch whose other streams decode as always."""
    W, H, Cn, planes = case
    good, _ = orc.encode(orc.synth(W, H, Cn, 3, 0))
    bad = [orc.many_plane_stream(W, H, Cn, planes, seed) for seed in (1, 2, 3)]
    for b in bad:
        assert orc.decode_stage(b, W, H, Cn, -1)[3] == planes[:Cn]   # the oracle reads the claim and decodes on
    check(ctx, [bad[0]], W, H, Cn)
    check(ctx, [good, bad[0], good[:len(good) // 2], bad[1], bad[2], good], W, H, Cn)
    _, infos = ctx.decode_planes([bad[0], good], W, H, Cn)
    assert [i.status for i in infos] == [2, 0]


@pytest.mark.parametrize("case", [("damaged_order_beyond_31_47x650x1.dwt", 47, 650, 1), ("damaged_wide_root_213x18x3.dwt", 213, 18, 3)])
def test_damaged_streams_that_leave_the_range_of_the_shifts(ctx, case):
    """See tests/test_oracle.py (same name): the VLI order passes 31 / the root image claims more than 32 bits per
    coefficient.  The decoder reads such tokens bit by bit the way the reference's binary does; alone, in a batch,
    with both path families."""
    import os

    name, W, H, Cn = case
    blob = open(os.path.join(orc.GOLDEN, name), "rb").read()
    good, _ = orc.encode(orc.synth(W, H, Cn, 3, 0))
    check(ctx, [blob], W, H, Cn)
    check(ctx, [good, blob, blob[:len(blob) * 2 // 3], blob, good], W, H, Cn)
