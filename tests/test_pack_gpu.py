"""GPU parity: the encoder's entropy stage (dwtx_encode_planes) vs the oracle's .dwt bytes."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

SHAPES = [(8, 8, 1), (8, 9, 3), (15, 15, 3), (16, 16, 1), (53, 37, 3), (77, 131, 3), (300, 17, 1), (17, 300, 3),
          (255, 257, 1), (64, 64, 3), (240, 320, 3), (512, 512, 1), (360, 640, 3)]


def lin_of(pix):
    _, lin, planes = orc.stage_dump(pix)
    return lin, planes


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("shape", SHAPES)
def test_stream_matches_oracle(ctx, shape, kind):
    import torch

    H, W, Cn = shape
    pix = orc.synth(W, H, Cn, 21, kind)
    lin, planes = lin_of(pix)
    want, st = orc.encode(pix)
    streams, infos = ctx.encode_planes(torch.from_numpy(lin).cuda(), W, H, Cn)
    assert list(infos[0].planes)[:Cn] == planes
    assert infos[0].root_bits == st.root_bits
    assert infos[0].total_bits == st.total_bits
    assert len(streams[0]) == len(want)
    assert streams[0] == want


def test_smpte_stream(ctx):
    import torch

    pix = orc.read_pnm(orc.GOLDEN + "/smpte.pnm")
    lin, _ = lin_of(pix)
    streams, infos = ctx.encode_planes(torch.from_numpy(lin).cuda(), 320, 240, 3)
    assert streams[0] == open(orc.GOLDEN + "/smpte.dwt", "rb").read()
    assert (infos[0].hdr_bits, infos[0].root_bits, infos[0].total_bits) == (48 + 559 + infos[0].hdr_bits - 48 - 559, 559, 81174)


def test_batch_of_different_images(ctx):
    import torch

    W, H, Cn, n = 131, 77, 3, 6
    pixs = [orc.synth(W, H, Cn, 100 + i, i & 1) for i in range(n)]
    pixs[3] = np.full((H, W, Cn), 90, dtype=np.uint8)          # flat image quirk
    lins = np.concatenate([lin_of(p)[0] for p in pixs])
    streams, _ = ctx.encode_planes(torch.from_numpy(lins).cuda(), W, H, Cn)
    for p, s in zip(pixs, streams):
        assert s == orc.encode(p)[0]


@pytest.mark.parametrize("cap", [1, 5, 6, 7, 33, 100, 1000, 4096, 14553, 14554, 14555, 100000])
def test_capacity_is_prefix(ctx, cap):
    import torch

    pix = orc.synth(131, 77, 3, 5, 0)
    lin, _ = lin_of(pix)
    want, st = orc.encode(pix, cap)
    streams, infos = ctx.encode_planes(torch.from_numpy(lin).cuda(), 131, 77, 3, capacity=cap)
    assert streams[0] == want
    if cap >= 200:   # below that the oracle's cosmetic bit counter is inside the root coder
        assert infos[0].total_bits == st.total_bits


def test_1080p_rgb_and_4096_gray(ctx):
    import hashlib
    import json
    import os
    import torch

    G = json.load(open(os.path.join(orc.GOLDEN, "golden.json")))
    for name in ("c1920x1080", "g4096x4096"):
        rec = G[name]
        pix = orc.synth(rec["W"], rec["H"], rec["C"], rec["seed"], rec["kind"])
        lin, _ = lin_of(pix)
        streams, _ = ctx.encode_planes(torch.from_numpy(lin).cuda(), rec["W"], rec["H"], rec["C"])
        assert len(streams[0]) == rec["dwt_len"]
        assert hashlib.sha256(streams[0]).hexdigest() == rec["dwt_sha256"]


def _synthetic_planes(rng, W, H, Cn, bits, density):
    """Linearised coefficient planes no 8-bit picture produces: magnitudes up to `bits` bits, a share `density`
    of non-zero coefficients (sparse planes -> zero runs across tiles, segments and channels)."""
    n = W * H
    lin = np.zeros((Cn, n), dtype=np.int64)
    for c in range(Cn):
        nz = rng.random(n) < density
        mag = (rng.integers(1, 1 << bits, n) >> rng.integers(0, bits, n)).clip(1)   # all bit lengths occur
        lin[c] = np.where(nz, mag * rng.choice([-1, 1], n), 0)
    return lin.astype(np.int32)


@pytest.mark.parametrize("case", [(64, 64, 1, 16, 0.5), (200, 120, 3, 16, 0.02), (256, 256, 1, 13, 0.9), (333, 111, 3, 10, 0.3),
                                  (512, 384, 1, 16, 0.0005), (96, 96, 3, 9, 1.0), (1024, 512, 1, 12, 0.001)])
def test_entropy_stage_on_synthetic_coefficient_planes(ctx, case):
    """dwtx_encode_planes / dwtx_decode_planes on coefficient planes with up to 16 bit planes (the 64-bit count
    registers of k_code), very sparse ones (runs of hundreds of thousands: escaped tokens, high VLI orders, codes
    longer than 32 bits) and dense ones: bytes equal the oracle's entropy stage, both decode back."""
    import torch

    W, H, Cn, bits, density = case
    rng = np.random.default_rng(W * 7 + H + bits)
    lin = _synthetic_planes(rng, W, H, Cn, bits, density)
    want, st = orc.encode_lin(lin, W, H)
    streams, infos = ctx.encode_planes(torch.from_numpy(lin).cuda(), W, H, Cn)
    assert list(infos[0].planes)[:Cn] == list(st.planes)[:Cn]
    assert streams[0] == want
    got = orc.decode_stage(want, W, H, Cn)
    assert got is not None and (got[0] == lin).all()
    back, dinfos = ctx.decode_planes([want], W, H, Cn)
    assert dinfos[0].status == 0 and not dinfos[0].truncated
    assert (back.cpu().numpy() == lin).all()
    for cap in (len(want) // 3, len(want) - 5):
        cut, _ = ctx.encode_planes(torch.from_numpy(lin).cuda(), W, H, Cn, capacity=cap)
        assert cut[0] == want[:cap]


def _parity_locked_planes(W, H, sign, extra):
    """Coefficient planes whose finest ring is +-1 everywhere: at bit plane 0 every symbol of that ring is a new
    one with the same sign, i.e. thousands of 2-bit tokens at VLI order 0 in a row.  All tokens of such a stretch
    have even length, so a speculative parse that starts an odd number of bits off never meets the real one.
    `extra` sprinkles `extra` larger coefficients over the coarser rings: it moves the stretch by some bits."""
    import dwt_amd

    levels, lengths, pixels, _, _ = dwt_amd.compute_lengths(W, H)
    lin = np.zeros((1, W * H), dtype=np.int32)
    lin[0, pixels[levels - 1]:pixels[levels]] = sign
    rng = np.random.default_rng(extra)
    where = rng.integers(pixels[0], pixels[levels - 1], extra)
    lin[0, where] = rng.integers(2, 40, extra) * rng.choice([-1, 1], extra)
    lin[0, :pixels[0]] = rng.integers(-100, 100, pixels[0])
    return lin


@pytest.mark.parametrize("sign", [1, -1])
def test_streams_the_single_path_family_cannot_follow(ctx, sign, opts):
    """The decoder records ONE family of speculative paths per chunk and falls back to two (even and odd start) when
    its token walk gives up; a ring of equal +-1 coefficients locks the parity for ~100 000 tokens.  Either way the
    planes must come back exactly, and at least one of the variants must really have taken the second walk."""
    import torch
    import dwt_amd

    W = H = 512
    variants = [_parity_locked_planes(W, H, sign, extra) for extra in (0, 1, 2, 3, 5, 8)]
    streams = []
    for lin in variants:
        want, _ = orc.encode_lin(lin, W, H)
        got, _ = ctx.encode_planes(torch.from_numpy(lin).cuda(), W, H, 1)
        assert got[0] == want
        streams.append(want)
    for lin, s in zip(variants, streams):
        back, infos = ctx.decode_planes([s], W, H, 1)
        assert infos[0].status == 0 and not infos[0].truncated
        assert (back.cpu().numpy() == lin).all()
    back, infos = ctx.decode_planes(streams, W, H, 1)   # as one batch (parts of it walk again, others do not)
    assert (back.cpu().numpy() == np.concatenate(variants)).all()
    opts.set("no_second_walk", 1)
    gave_up = 0
    for s in streams:
        try:
            ctx.decode_planes([s, s, s], W, H, 1)   # (one or two images take both families from the start)
        except dwt_amd.DwtxError:
            gave_up += 1
    assert 0 < gave_up
    opts.set("no_second_walk", 0)
    opts.set("two_families", 1)   # both families from the start: the path the fallback takes
    back, infos = ctx.decode_planes(streams, W, H, 1)
    assert (back.cpu().numpy() == np.concatenate(variants)).all()


def test_second_walk_takes_only_the_images_that_gave_up(ctx, opts):
    """A batch of 26 streams — four parts — of which three (two of them neighbours, in different parts) lock the parity:
    only those are walked again with both families (dwtx_decode_planes_ex: run by run of consecutive images), the others
    keep their first walk; every plane must come back exactly, in both modes of running the batch."""
    import torch

    W = H = 512
    rng = np.random.default_rng(26)
    levels, lengths, pixels, _, _ = __import__("dwt_amd").compute_lengths(W, H)
    planes = []
    for i in range(26):
        if i in (5, 6, 19):
            planes.append(_parity_locked_planes(W, H, 1 if i != 6 else -1, 0))
        else:
            lin = np.zeros((1, W * H), dtype=np.int32)
            n = pixels[levels]
            lin[0, :n] = (rng.laplace(0.0, 3.0 + i, n)).astype(np.int32)
            planes.append(lin)
    streams = [orc.encode_lin(lin, W, H)[0] for lin in planes]
    want = np.concatenate(planes)
    for one_stream in (0, 1):
        opts.set("one_stream", one_stream)
        back, infos = ctx.decode_planes(streams, W, H, 1)
        assert all(i.status == 0 and not i.truncated for i in infos)
        assert (back.cpu().numpy() == want).all()
    opts.set("one_stream", 0)
    opts.set("no_second_walk", 1)
    with pytest.raises(__import__("dwt_amd").DwtxError):   # (the locked ones really do give up)
        ctx.decode_planes(streams, W, H, 1)

