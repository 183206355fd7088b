"""GPU: the sidecar index (include/dwtx.h dwtx_index) — an optional companion of a .dwt that
lets the decoder walk all segments at once.  It may speed a decode up, it must never change one."""
import ctypes

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


def _decode(ctx, streams, W, H, Cn):
    lin, infos = ctx.decode_planes(streams, W, H, Cn)
    lin = lin.cpu().numpy()
    for i, info in enumerate(infos):
        if info.status:
            lin[i * Cn:(i + 1) * Cn] = 0   # nothing is written for an unreadable stream
    return lin, infos


def _same_info(a, b):
    for f in ("status", "level", "nsegs", "truncated", "pmax", "bits_used", "zeros_left"):
        assert getattr(a, f) == getattr(b, f), f
    assert list(a.planes) == list(b.planes) and list(a.missing) == list(b.missing)


@pytest.mark.parametrize("shape", [(64, 64, 1), (53, 37, 3), (255, 257, 1), (360, 640, 3), (512, 512, 1)])
def test_a_decode_writes_an_index_and_a_decode_with_it_gives_the_same(ctx, shape, opts):
    import dwt_amd

    H, W, Cn = shape
    streams = [orc.encode(orc.synth(W, H, Cn, seed, seed & 1))[0] for seed in (3, 4, 5, 6, 7)]
    made = ctx.set_index(None, len(streams))
    want, winfos = _decode(ctx, streams, W, H, Cn)
    for i, s in enumerate(streams):
        ref = orc.decode_stage(s, W, H, Cn, -1)
        assert (want[i * Cn:(i + 1) * Cn] == ref[0]).all()
        assert made[i].magic == dwt_amd.INDEX_MAGIC and made[i].nsegs == winfos[i].nsegs > 0
        assert made[i].seg[0].bit > 48 and made[i].stream_bits == winfos[i].bits_used
    # with the index; DWTX_NO_INDEX_FALLBACK turns a rejected index into an error: these must all be accepted
    opts.set("no_index_fallback", 1)
    again = ctx.set_index(made, len(streams))
    got, ginfos = _decode(ctx, streams, W, H, Cn)
    assert (got == want).all()
    for a, b in zip(ginfos, winfos):
        _same_info(a, b)
    for i in range(len(streams)):   # the indexed walk hands the same index on
        assert again[i].nsegs == made[i].nsegs
        assert bytes(again[i])[:32 + 32 * made[i].nsegs] == bytes(made[i])[:32 + 32 * made[i].nsegs]
    # one image at a time as well (another code path: no halves)
    for i, s in enumerate(streams[:2]):
        one = (dwt_amd.Index * 1)(made[i])
        ctx.set_index(one, 0)
        got1, _ = _decode(ctx, [s], W, H, Cn)
        assert (got1 == want[i * Cn:(i + 1) * Cn]).all()
    ctx.set_index()


def test_a_wrong_index_changes_nothing(ctx, opts):
    """Stale, foreign and damaged indices: the decoder notices (the segments do not fit together) and walks the
    stream the plain way; with DWTX_NO_INDEX_FALLBACK the rejection shows as an error."""
    import dwt_amd

    W, H, Cn = 200, 117, 3
    streams = [orc.encode(orc.synth(W, H, Cn, seed, 0))[0] for seed in (11, 12, 13, 14)]
    made = ctx.set_index(None, len(streams))
    want, winfos = _decode(ctx, streams, W, H, Cn)
    rng = np.random.default_rng(5)

    def damaged(kind):
        bad = (dwt_amd.Index * len(streams))()
        for i in range(len(streams)):
            ctypes.memmove(ctypes.byref(bad[i]), ctypes.byref(made[i]), ctypes.sizeof(dwt_amd.Index))
        if kind == "foreign":      # the indices of other streams of the same geometry
            for i in range(len(streams)):
                ctypes.memmove(ctypes.byref(bad[i]), ctypes.byref(made[(i + 1) % len(streams)]), ctypes.sizeof(dwt_amd.Index))
        elif kind == "bit":
            k = int(rng.integers(1, made[0].nsegs))
            bad[0].seg[k].bit += 1
        elif kind == "order":
            bad[1].seg[made[1].nsegs // 2].order ^= 1
        elif kind == "n1":
            bad[2].seg[made[2].nsegs - 1].n1 += 1
        elif kind == "short":
            bad[3].nsegs -= 1
        elif kind == "garbage":
            raw = rng.integers(0, 256, ctypes.sizeof(dwt_amd.Index) - 32, dtype=np.uint8).tobytes()
            ctypes.memmove(ctypes.addressof(bad[0]) + 32, raw, len(raw))
        elif kind == "beyond":
            for k in range(bad[1].nsegs):
                bad[1].seg[k].bit = (1 << 40) + k
        elif kind == "symbase_wraps":      # sym_base + ring size wraps around 2^64: the bound must not be checked with a sum
            bad[0].seg[made[0].nsegs // 3].sym_base = 2 ** 64 - 40
        elif kind == "symbase_past_the_bitmap":
            bad[2].seg[made[2].nsegs // 2].sym_base = 1 << 45
        elif kind == "order_32":           # an entry order only a damaged stream's serial walk can reach
            bad[3].seg[made[3].nsegs // 2].order = 32
        return bad

    for kind in ("foreign", "bit", "order", "n1", "short", "garbage", "beyond", "symbase_wraps", "symbase_past_the_bitmap", "order_32"):
        ctx.set_index(damaged(kind), 0)
        got, ginfos = _decode(ctx, streams, W, H, Cn)
        assert (got == want).all(), kind
        for a, b in zip(ginfos, winfos):
            _same_info(a, b)
        opts.set("no_index_fallback", 1)
        with pytest.raises(dwt_amd.DwtxError):
            _decode(ctx, streams, W, H, Cn)
        opts.set("no_index_fallback", 0)
    ctx.set_index()


def test_cut_streams_have_no_index_and_take_none(ctx):
    W, H, Cn = 131, 77, 3
    data, _ = orc.encode(orc.synth(W, H, Cn, 5, 0))
    streams = [data, data[:len(data) // 2], data[:40], b"junk"]
    made = ctx.set_index(None, len(streams))
    want, winfos = _decode(ctx, streams, W, H, Cn)
    assert made[0].nsegs > 0 and [m.nsegs for m in made][1:] == [0, 0, 0]
    ctx.set_index(made, 0)        # only stream 0 has one: the part is walked the plain way
    got, ginfos = _decode(ctx, streams, W, H, Cn)
    assert (got == want).all()
    full = ctx.set_index((type(made[0]) * 4)(made[0], made[0], made[0], made[0]), 0)
    got, ginfos = _decode(ctx, streams, W, H, Cn)   # the whole stream's index offered for its prefixes
    assert (got == want).all()
    for a, b in zip(ginfos, winfos):
        _same_info(a, b)
    ctx.set_index()


def test_whole_images_through_the_host_pipeline_with_an_index(ctx, opts):
    """dwtx_decode_images decodes a batch in parts: index entries follow their images."""
    import dwt_amd

    W, H, Cn = 96, 80, 3
    pix = np.stack([orc.synth(W, H, Cn, s, 0) for s in range(7)])
    streams, _ = ctx.encode(pix)
    made = ctx.set_index(None, len(streams))
    outs = ctx.decode(streams)
    assert all((o == p).all() for o, p in zip(outs, pix))
    assert all(m.nsegs > 0 for m in made)
    opts.set("part_images", 3)
    opts.set("no_index_fallback", 1)
    ctx.set_index(made, 0)
    outs = ctx.decode(streams)
    assert all((o == p).all() for o, p in zip(outs, pix))
    ctx.set_index()
