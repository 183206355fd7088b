"""GPU parity, whole path: pixels -> .dwt -> pixels through the C ABI vs the oracle and the goldens."""
import hashlib
import json
import os

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(orc.GOLDEN, "golden.json")))


def sha(b):
    return hashlib.sha256(b).hexdigest()


def case_input(rec):
    if rec["seed"] is None:
        return orc.read_pnm(os.path.join(orc.GOLDEN, "smpte.pnm"))
    return orc.synth(rec["W"], rec["H"], rec["C"], rec["seed"], rec["kind"])


def test_config_d_16384_rgb_truncated_to_1mib(ctx):
    """BASELINE.json configs[3]: 16384x16384 RGB, CAPACITY 1 MiB, against the reference's own output hashes."""
    import torch

    rec = G["c16384x16384_cap1MiB"]
    W, H, Cn = rec["W"], rec["H"], rec["C"]
    pix = ctx.synth_pixels(1, H, W, Cn, seed0=0, kind=0)
    streams, info = ctx.encode_device(pix, capacity=rec["capacity"])
    lens = ctx.stream_lengths(info)
    assert int(lens[0]) == rec["dwt_len"]
    data = streams[0, : int(lens[0])].cpu().numpy().tobytes()
    assert sha(data) == rec["dwt_sha256"]
    del pix
    torch.cuda.empty_cache()
    out, infos = ctx.decode_device(streams, lens, W, H, Cn)
    g = orc.geometry(W, H)
    lo = infos[0].level + 1
    assert (g.widths[lo], g.heights[lo]) == (rec["dec_W"], rec["dec_H"])
    h = hashlib.sha256()
    flat = out[0, : rec["dec_W"] * rec["dec_H"] * Cn]
    for i in range(0, flat.numel(), 1 << 27):
        h.update(flat[i:i + (1 << 27)].cpu().numpy().tobytes())
    assert h.hexdigest() == rec["dec_sha256"]


@pytest.mark.parametrize("shape", [(512, 512, 3), (1000, 700, 1), (2048, 1024, 3)])
def test_capacity_stops_the_work_and_keeps_the_bytes(ctx, shape, opts):
    """encode.c:192,204,216 leave the plane loop at the first refused byte: with a CAPACITY the encoder drops every
    segment that cannot start inside it (pack.hip k_cut; dwtx_stream_info.segments_cut says how many) and the
    bytes stay the exact prefix — the same bytes as with the cut switched off, and as the oracle's."""
    import ctypes
    import dwt_amd

    W, H, Cn = shape
    pix = ctx.synth_pixels(1, H, W, Cn, seed0=3, kind=0)
    full, _ = orc.encode(pix[0].cpu().numpy())
    cuts_seen = []
    for cap in (64, 700, len(full) // 50, len(full) // 7, len(full) // 2, len(full) - 3, len(full), len(full) + 100):
        got = {}
        for off in (0, 1):
            opts.set("no_capacity_cut", off)
            streams, info = ctx.encode_device(pix, capacity=cap)
            rec = dwt_amd.StreamInfo.from_buffer_copy(info[0].cpu().numpy().tobytes())
            got[off] = (streams[0, : rec.nbytes].cpu().numpy().tobytes(), rec)
        assert got[0][0] == got[1][0] == full[:cap], cap
        assert got[1][1].segments_cut == 0
        assert got[0][1].segments + got[0][1].segments_cut == got[1][1].segments
        assert (got[0][1].total_bits, got[0][1].nbytes) == (got[1][1].total_bits, got[1][1].nbytes)
        cuts_seen.append(got[0][1].segments_cut)
    assert cuts_seen[0] > cuts_seen[3] > 0 and cuts_seen[-1] == 0 and cuts_seen[-2] == 0   # the tighter the capacity, the more is dropped


def test_config_d_16384_rgb_whole_stream(ctx):
    """The same 805 M-sample frame without a capacity: 3 Gbit of stream (bit positions beyond 2^31), its first
    MiB is the reference's CAPACITY=1 MiB output (truncation yields a prefix), and it decodes
    back to the pixels."""
    import torch

    rec = G["c16384x16384_cap1MiB"]
    W, H, Cn = rec["W"], rec["H"], rec["C"]
    pix = ctx.synth_pixels(1, H, W, Cn, seed0=0, kind=0)
    streams, info = ctx.encode_device(pix)
    lens = ctx.stream_lengths(info)
    assert int(lens[0]) * 8 > 1 << 31
    assert sha(streams[0, : rec["capacity"]].cpu().numpy().tobytes()) == rec["dwt_sha256"]
    out, infos = ctx.decode_device(streams, lens, W, H, Cn)
    assert infos[0].status == 0 and not infos[0].truncated
    assert torch.equal(out.view(1, H, W, Cn), pix)
    del out, streams, pix
    torch.cuda.empty_cache()


def test_stream_longer_than_2_to_32_bits(ctx):
    """Uniform noise at 16384x16384 RGB costs ~9 bit per sample: 7 Gbit of stream, every bit offset is 64-bit."""
    import torch

    W = H = 16384
    pix = ctx.synth_pixels(1, H, W, 3, seed0=3, kind=1)
    streams, info = ctx.encode_device(pix)
    lens = ctx.stream_lengths(info)
    assert int(lens[0]) * 8 > 1 << 32
    out, infos = ctx.decode_device(streams, lens, W, H, 3)
    assert infos[0].status == 0 and not infos[0].truncated
    assert torch.equal(out.view(1, H, W, 3), pix)
    del out, streams, pix
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name", sorted(k for k in G if not G[k].get("heavy")))
def test_goldens_from_the_real_reference(ctx, name):
    rec = G[name]
    pix = case_input(rec)
    data, st = ctx.encode(pix, rec["capacity"])
    assert len(data) == rec["dwt_len"] and sha(data) == rec["dwt_sha256"]
    lines = [f"{st.meta_bits} bits for meta data", f"{st.root_bits} bits for root image",
             f"{st.total_bits} bits ({st.kib} KiB) encoded"]
    assert lines == rec["encode_stderr"]
    px = rec["pixels_arg"]
    back = ctx.decode(data, -1 if px is None else px)
    assert back.shape[:2] == (rec["dec_H"], rec["dec_W"])
    assert sha(back.tobytes()) == rec["dec_sha256"]


@pytest.mark.parametrize("shape", [(8, 8, 1), (9, 8, 3), (77, 131, 3), (300, 17, 1), (255, 257, 3), (33, 1000, 1),
                                   (700, 1000, 3), (1025, 2047, 1), (1030, 1548, 3), (8, 4096, 1), (4096, 8, 3),
                                   (9, 3000, 3)])   # (the oracle walks the whole pow2 square: much longer sides take minutes)
def test_roundtrip_and_oracle_bytes(ctx, shape):
    H, W, Cn = shape
    for kind in (0, 1):
        pix = orc.synth(W, H, Cn, 77, kind)
        data, _ = ctx.encode(pix)
        assert data == orc.encode(pix)[0]
        assert (ctx.decode(data) == pix).all()


def test_truncated_decodes_match_oracle(ctx):
    pix = orc.synth(131, 77, 3, 5, 0)
    full, _ = ctx.encode(pix)
    for cap in (100, 300, 500, 1000, 3000, 8000, len(full) - 1):
        data, _ = ctx.encode(pix, cap)
        assert data == full[:cap]
        want = orc.decode(data)
        got = ctx.decode(data)
        if want is None:
            assert got is None
        else:
            assert got.shape == want.shape and (got == want).all()


def test_batch_mixed_truncation(ctx):
    """A batch whose streams stop at different places decodes image by image (different sizes)."""
    pix = orc.synth(131, 77, 3, 9, 0)
    full, _ = ctx.encode(pix)
    streams = [full, full[:2000], full[:400], full[:150]]
    outs = ctx.decode(streams)
    for s, o in zip(streams, outs):
        want = orc.decode(s)
        assert o.shape == want.shape and (o == want).all()


def test_batch_roundtrip(ctx):
    n, H, W, Cn = 5, 96, 160, 3
    pix = np.stack([orc.synth(W, H, Cn, 40 + i, i & 1) for i in range(n)])
    streams, _ = ctx.encode(pix)
    for i in range(n):
        assert streams[i] == orc.encode(pix[i])[0]
    outs = ctx.decode(streams)
    for i in range(n):
        assert (outs[i] == pix[i]).all()


def test_flat_image_quirk(ctx):
    pix = np.full((16, 100, 3), 77, dtype=np.uint8)
    data, st = ctx.encode(pix)
    assert data == orc.encode(pix)[0]
    back = ctx.decode(data)
    assert back.shape == (8, 50, 3) and (back == orc.decode(data)).all()


def test_pixels_argument(ctx):
    pix = orc.synth(320, 240, 3, 2, 0)
    data, _ = ctx.encode(pix)
    for px in (0, 50, 400, 5000, 30000, 1 << 30):
        want = orc.decode(data, px)
        got = ctx.decode(data, px)
        assert got.shape == want.shape and (got == want).all()


@pytest.mark.parametrize("shape", [(3, 77, 131, 3), (2, 64, 64, 1), (1, 1080, 1920, 3)])
def test_device_generator_matches_oracle(ctx, shape):
    n, H, W, Cn = shape
    for kind in (0, 1):
        got = ctx.synth_pixels(n, H, W, Cn, seed0=5, kind=kind).cpu().numpy()
        for i in range(n):
            assert (got[i] == orc.synth(W, H, Cn, 5 + i, kind)).all()


def test_device_resident_roundtrip(ctx):
    import torch

    n, H, W, Cn = 3, 240, 320, 3
    pix = ctx.synth_pixels(n, H, W, Cn, seed0=0, kind=0)
    streams, info = ctx.encode_device(pix)
    lens = ctx.stream_lengths(info)
    out, infos = ctx.decode_device(streams, lens, W, H, Cn)
    assert torch.equal(out.view(n, H, W, Cn), pix)
    host = pix.cpu().numpy()
    for i in range(n):
        assert streams[i, : int(lens[i])].cpu().numpy().tobytes() == orc.encode(host[i])[0]


@pytest.mark.parametrize("shape", [(117, 200, 1), (72, 68, 1), (64, 128, 1), (255, 512, 1), (600, 36, 1),
                                   (117, 200, 3), (72, 68, 3), (65, 132, 3), (301, 44, 3)])
def test_pixels_straight_into_the_lifting(ctx, shape):
    """W % 4 == 0 images skip the int widening / colour / clamp passes (pnm.h:69-74,108 and image.h:39-65
    fused into the finest lifting level); same bytes, full and cut short, with odd heights and levels
    that fall back, on smooth and on noise frames (the clamps matter on truncated noise)."""
    H, W, Cn = shape
    pix = orc.synth(W, H, Cn, 3, (H + Cn) & 1)
    full, _ = ctx.encode(pix)
    assert full == orc.encode(pix)[0]
    assert (ctx.decode(full) == pix).all()
    for cap in (len(full) // 7, len(full) // 2):
        want = orc.decode(full[:cap])
        got = ctx.decode(full[:cap])
        assert (want is None and got is None) or (got.shape == want.shape and (got == want).all())
    for px in (0, 300, 20000):
        want = orc.decode(full, px)
        got = ctx.decode(full, px)
        assert got.shape == want.shape and (got == want).all()


def test_gray_batch_both_halves_and_mixed_lengths(ctx):
    """n >= 4 takes the two-stream decoder whose halves queue their own inverse transform; odd n, one
    truncated member (different output size) and clamping on noise."""
    n, H, W = 7, 120, 256
    pix = np.stack([orc.synth(W, H, 1, 60 + i, i & 1)[..., 0] for i in range(n)])[..., None]
    streams, _ = ctx.encode(pix)
    for i in range(n):
        assert streams[i] == orc.encode(pix[i])[0]
    streams[5] = streams[5][: len(streams[5]) // 3]
    outs = ctx.decode(streams)
    for i in range(n):
        want = orc.decode(streams[i])
        assert outs[i].shape == want.shape and (outs[i] == want).all()


def test_random_shapes_capacities_and_cuts(ctx):
    """Seeded sweep over small random geometries (odd sizes, widths that do and do not take the fused
    pixel paths), capacities, stream cuts and PIXELS caps: bytes and pictures equal the oracle's."""
    rng = np.random.default_rng(20260101)
    for case in range(48):
        W, H = int(rng.integers(8, 161)), int(rng.integers(8, 161))
        if case % 3 == 0:
            W = (W + 3) // 4 * 4
        Cn = 1 if rng.integers(0, 2) else 3
        pix = orc.synth(W, H, Cn, 1000 + case, int(rng.integers(0, 2)))
        full, _ = ctx.encode(pix)
        assert full == orc.encode(pix)[0], (W, H, Cn)
        assert (ctx.decode(full) == pix).all(), (W, H, Cn)
        cap = int(rng.integers(40, max(41, len(full))))
        data, _ = ctx.encode(pix, cap)
        assert data == full[:cap], (W, H, Cn, cap)
        for blob, px in ((data, None), (full, int(rng.integers(0, 4 * W * H))), (full[: len(full) // 2], 200)):
            want = orc.decode(blob, -1 if px is None else px)
            got = ctx.decode(blob, -1 if px is None else px)
            assert (want is None and got is None) or (got.shape == want.shape and (got == want).all()), (W, H, Cn, cap, px)


def test_host_batches_pipeline_in_parts(ctx, opts):
    """The host-buffer entry points cut a batch into parts whose transfers overlap the neighbouring
    parts' kernels; with parts of 3 images an 11-image batch takes four of them (both staging slots
    get reused), one stream cut short."""
    opts.set("part_images", 3)
    n, H, W, Cn = 11, 72, 100, 3
    pix = np.stack([orc.synth(W, H, Cn, 500 + i, i & 1) for i in range(n)])
    streams, stats = ctx.encode(pix)
    for i in range(n):
        want, ost = orc.encode(pix[i])
        assert streams[i] == want
        assert (stats[i].root_bits, stats[i].total_bits) == (ost.root_bits, ost.total_bits)
    streams[7] = streams[7][: len(streams[7]) // 4]
    outs = ctx.decode(streams)
    for i in range(n):
        want = orc.decode(streams[i])
        assert outs[i].shape == want.shape and (outs[i] == want).all()


def test_corrupted_streams_decode_like_the_oracle(ctx):
    """Bit flips, byte garbage and junk tails after a valid header: whatever the reference's decoder makes of
    such a stream (early stop, resolution drop, rle_get_bit refusing a run, ...) the GPU path makes too."""
    from test_oracle import corrupted_blobs   # the same blobs the oracle is pinned on against the real reference

    for (W, H, Cn, seed, bseed) in ((96, 80, 3, 12, 7), (131, 77, 1, 5, 100), (64, 64, 3, 9, 101), (200, 117, 1, 3, 102)):
        pix = orc.synth(W, H, Cn, seed, 0)
        good, _ = ctx.encode(pix)
        blobs = corrupted_blobs(good, 40, bseed)
        for blob in blobs:
            want = orc.decode(blob)
            got = ctx.decode(blob)
            if want is None:
                assert got is None
            else:
                assert got is not None and got.shape == want.shape and (got == want).all()
    outs = ctx.decode(blobs[:12])   # and as one batch (both decoder halves)
    for blob, got in zip(blobs[:12], outs):
        want = orc.decode(blob)
        assert (want is None and got is None) or (got.shape == want.shape and (got == want).all())


def test_statistics_when_capacity_cuts_into_header_or_root(ctx):
    """encode.c:175-180,226-230 print the bit writer's own counters, and a refused byte makes the field being
    written give up (bits.h:58-78): with CAPACITY below the header + root image the three numbers are not the
    sizes of those parts.  Same numbers as the oracle (pinned on the reference in test_oracle.py)."""
    for (W, H, Cn, seed) in ((53, 37, 3, 4), (64, 40, 1, 9)):
        pix = orc.synth(W, H, Cn, seed, 0)
        for cap in list(range(1, 20)) + [31, 47, 48, 60, 80, 85, 86, 87, 88, 90, 100, 120, 200]:
            data, st = ctx.encode(pix, cap)
            want, ost = orc.encode(pix, cap)
            assert data == want, (W, H, Cn, cap)
            assert (st.meta_bits, st.root_bits, st.total_bits, st.kib) == (ost.meta_bits, ost.root_bits, ost.total_bits, ost.kib), (W, H, Cn, cap)


def test_length_beyond_the_stream_stride_is_refused_or_clamped(ctx):
    """A byte length that does not fit the stride cannot be real data: the host-buffer entry point refuses it,
    the device entry point reads only what the stride holds (no access beyond the buffer)."""
    import ctypes as C

    import torch

    import dwt_amd

    pix = orc.synth(200, 120, 1, 9, 0)
    data, _ = ctx.encode(pix)
    stride = (len(data) + 64 + 7) // 8 * 8
    host = np.zeros((1, stride), dtype=np.uint8)
    host[0, : len(data)] = np.frombuffer(data, dtype=np.uint8)
    lens = (C.c_size_t * 1)(stride + 4096)
    out = np.empty((1, 200 * 120), dtype=np.uint8)
    ow, oh, oc = (C.c_int * 1)(), (C.c_int * 1)(), (C.c_int * 1)()
    rc = ctx.lib.dwtx_decode_images(ctx.h, host.ctypes.data, stride, C.cast(lens, C.c_void_p), 1, -1, out.ctypes.data, 200 * 120, ow, oh, oc)
    assert rc == -3
    dev = torch.from_numpy(host).to(ctx.device)
    dl = torch.tensor([1 << 40], dtype=torch.int64, device=ctx.device)
    got, infos = ctx.decode_device(dev, dl, 200, 120, 1)
    # the zero padding after the real stream decodes as more (empty) data: same picture as the oracle on the padded bytes
    want = orc.decode(host[0].tobytes())
    lo = infos[0].level + 1
    g = orc.geometry(200, 120)
    assert (g.widths[lo], g.heights[lo]) == want.shape[1::-1]
    assert (got[0, : want.size].cpu().numpy() == want.reshape(-1)).all()


def test_image_sizes_beyond_the_references_arithmetic_are_refused(ctx):
    """Sides above 32768 (DWTX_MAX_SIDE): encode.c:45 / decode.c:47 overflow there and the reference binary's stream
    does not decode to the picture (tests/test_oracle.py pins that) — a clean argument error from every entry point,
    encode and decode, instead of bytes nobody can compare.  W*H >= 2^31 (encode.c:40) is inside that fence."""
    import torch

    import dwt_amd

    tiny = torch.zeros(64, dtype=torch.uint8, device=ctx.device)
    info = torch.zeros(256, dtype=torch.uint8, device=ctx.device)
    for W, H in ((50000, 50000), (65536, 32768), (65537, 8), (8, 7), (32776, 8), (8, 32769), (32769, 32769)):
        rc = ctx.lib.dwtx_encode_device(ctx.h, tiny.data_ptr(), W, H, 1, 1, 0, tiny.data_ptr(), 64, info.data_ptr())
        assert rc == -3, (W, H, rc)
        assert b"unsupported image size" in ctx.lib.dwtx_last_error()
    # a stream whose header claims such a size (the reference's encoder writes them): refused, nothing decoded
    head = bytes([ord("W"), ord("5"), (32776 - 1) & 255, (32776 - 1) >> 8, 7, 0])
    with pytest.raises(RuntimeError, match="unsupported image size"):
        ctx.decode(head + bytes(500))


def test_config_c_1024_frames_of_1080p_rgb_in_one_call(ctx):
    """BASELINE.json configs[2] at its real batch size: 1024 x 1920x1080 RGB (6.4 G samples, past 2^32) through
    dwtx_encode_device / dwtx_decode_device in ONE call each: every frame comes back bit for bit, sampled
    frames carry the oracle's bytes (frame 0 also the real reference's golden)."""
    import torch

    n, W, H, Cn = 1024, 1920, 1080, 3
    free, _ = torch.cuda.mem_get_info()
    if free < 230 * (1 << 30):
        pytest.skip("needs about 230 GiB of free HBM")
    pix = ctx.synth_pixels(n, H, W, Cn, seed0=0, kind=0)
    streams, info = ctx.encode_device(pix)
    lens = ctx.stream_lengths(info)
    host_lens = lens.cpu().numpy()
    assert host_lens.min() > 1 << 20 and host_lens.max() < streams.shape[1]
    rec = G["c1920x1080"]
    s0 = streams[0, : int(host_lens[0])].cpu().numpy().tobytes()
    assert len(s0) == rec["dwt_len"] and sha(s0) == rec["dwt_sha256"]
    for i in (1, 255, 256, 511, 777, 1023):
        want, _ = orc.encode(orc.synth(W, H, Cn, i, 0))
        assert streams[i, : int(host_lens[i])].cpu().numpy().tobytes() == want, i
    # a stride that fits the streams (the encoder's output stride is a worst-case bound; the decoder's
    # chunk tables are laid out per stride)
    stride = (int(host_lens.max()) + 64 + 7) // 8 * 8
    tight = streams[:, :stride].contiguous()
    del streams, info
    torch.cuda.empty_cache()
    out, infos = ctx.decode_device(tight, lens, W, H, Cn)
    assert all(i.status == 0 and not i.truncated for i in infos)
    ok = True
    for i0 in range(0, n, 64):
        ok = ok and bool(torch.equal(out[i0:i0 + 64].view(-1, H, W, Cn), pix[i0:i0 + 64]))
    assert ok
    del out, tight, pix
    torch.cuda.empty_cache()


@pytest.mark.parametrize("shape", [(64, 64, 1), (128, 128, 3), (256, 256, 1), (512, 512, 3), (270, 480, 3), (257, 256, 1), (77, 132, 3), (300, 20, 1),
                                   (1080, 1920, 3), (77, 131, 3), (257, 255, 1)])
def test_whole_squares_are_read_and_written_in_the_pyramid(ctx, shape, opts):
    """The entropy stage's tiles are the Hilbert curve's 32x32 blocks (dwtx_tiles): on images whose width is a multiple of 4
    the blocks that lie wholly inside a ring skip the linearised copy — the coder reads them from the pyramid and the
    decoder writes them there (bias of never-decoded planes included, decode.c:51-58) — and only the blocks the
    ring's edges cut (image border, LL quadrant) go through it; power-of-two squares have no such blocks at all, the
    last two shapes (width not a multiple of 4) nothing but.  Same bytes and pictures as the oracle, whole and cut at
    many lengths (cuts inside the finest level keep the full resolution, i.e. the fused path with a bias), alone
    and in a mixed batch; and the same bytes as the path that linearises everything."""
    H, W, Cn = shape
    for kind in (0, 1):
        pix = orc.synth(W, H, Cn, 31 + kind, kind)
        full, _ = ctx.encode(pix)
        assert full == orc.encode(pix)[0]
        assert (ctx.decode(full) == pix).all()
        ncuts = 23 if W * H <= 512 * 512 else 5
        cuts = sorted({len(full) * k // ncuts for k in range(1, ncuts)} | {len(full) - 1, len(full) - 9})
        for cap in cuts:
            want = orc.decode(full[:cap])
            got = ctx.decode(full[:cap])
            assert (want is None and got is None) or (got.shape == want.shape and (got == want).all()), cap
    # a batch whose members end at different places (both decoder halves, per-image finishing)
    n = 6
    pixs = np.stack([orc.synth(W, H, Cn, 90 + i, i & 1) for i in range(n)])
    streams, _ = ctx.encode(pixs)
    for i in range(n):
        assert streams[i] == orc.encode(pixs[i])[0]
    streams[1] = streams[1][: len(streams[1]) * 9 // 10]
    streams[4] = streams[4][: len(streams[4]) // 3]
    outs = ctx.decode(streams)
    for i in range(n):
        want = orc.decode(streams[i])
        assert outs[i].shape == want.shape and (outs[i] == want).all(), i
    opts.set("no_square_tiles", 1)
    plain, _ = ctx.encode(pixs)
    assert plain == [orc.encode(pixs[i])[0] for i in range(n)]
    outs2 = ctx.decode(streams)
    for i in range(n):
        assert (outs2[i] == outs[i]).all()


@pytest.mark.parametrize("shape", [(64, 64, 1), (128, 192, 3), (270, 480, 3), (256, 256, 1), (1080, 1920, 3)])
def test_the_finest_ring_in_16_bit_planes_changes_nothing(ctx, shape, opts):
    """From 8-bit pixels the codec keeps the finest ring's coefficients (|c| <= 1020, cdf53.h:13-21 on |x| <= 255) as
    16-bit values in planes of their own: the transform writes and the entropy stage reads half
    the bytes for three quarters of the coefficients.  Same bytes and pictures as the oracle (the other tests run this
    way by default), extremes included, and the same as with the ring in the int32 pyramid (no_fine16)."""
    H, W, Cn = shape
    rng = np.random.default_rng(7)
    extremes = (rng.integers(0, 2, (H, W, Cn)) * 255).astype(np.uint8)           # black / white noise: the largest details there are
    checker = (((np.arange(H)[:, None] + np.arange(W)[None, :]) & 1) * 255).astype(np.uint8)[..., None].repeat(Cn, 2)
    pixs = np.stack([orc.synth(W, H, Cn, 5, 0), orc.synth(W, H, Cn, 6, 1), extremes, np.ascontiguousarray(checker)])
    want = [orc.encode(p)[0] for p in pixs] if W * H <= 512 * 512 else None
    streams, stats = ctx.encode(pixs)
    if want:
        assert streams == want
    outs = ctx.decode(streams)
    assert all((o == p).all() for o, p in zip(outs, pixs))
    cut = [s[: len(s) * (i + 2) // 6] for i, s in enumerate(streams)]    # cut streams: some keep the full size (bias on the finest ring)
    outs_cut = ctx.decode(cut)
    opts.set("no_fine16", 1)
    streams32, stats32 = ctx.encode(pixs)
    assert streams32 == streams and [s.total_bits for s in stats32] == [s.total_bits for s in stats]
    outs32 = ctx.decode(streams)
    assert all((o == p).all() for o, p in zip(outs32, pixs))
    for a, b in zip(ctx.decode(cut), outs_cut):
        assert (a is None and b is None) or (a.shape == b.shape and (a == b).all())
    if W * H <= 512 * 512:
        for c, o in zip(cut, outs_cut):
            ref = orc.decode(c)
            assert (ref is None and o is None) or (o.shape == ref.shape and (o == ref).all())


def test_the_largest_coefficients_an_8_bit_source_can_make(ctx, opts):
    """The codec keeps the detail rings of the five finest levels — and, in the encoder, the LL bands between them — as
    16-bit values.  Pictures built to drive single coefficients as far as 8-bit samples can: the
    signs of the five-level high-pass response (the 5/3 low-pass four times, then the high-pass) as red / blue of an
    RGB picture, i.e. Co = +-255 in that pattern, at the sub-pixel shifts that line it up with a level-5 HH
    coefficient; and black / white noise at block sizes 1 .. 32.  Same bytes as the oracle (which computes in int),
    and the same with the 16-bit planes switched off."""
    g, h = np.array([-1, 2, 6, 2, -1]) / 8.0, np.array([-0.5, 1, -0.5])

    def up(f, m):
        o = np.zeros((len(f) - 1) * m + 1)
        o[::m] = f
        return o

    low = np.array([1.0])
    for k in range(4):
        low = np.convolve(low, up(g, 2 ** k))
    high5 = np.convolve(low, up(h, 16))
    N = 2048
    pics = []
    for shift in (16, 0):
        s1 = np.zeros(N)
        st = N // 2 - len(high5) // 2 + shift
        s1[st:st + len(high5)] = np.sign(high5)
        pat = np.outer(s1, s1)
        img = np.zeros((N, N, 3), dtype=np.uint8)
        img[..., 0] = np.where(pat > 0, 255, 0)
        img[..., 2] = np.where(pat < 0, 255, 0)
        img[..., 1] = 128
        pics.append(img)
    coef, _, _ = orc.stage_dump(pics[0])
    assert np.abs(coef).max() >= 2000    # (255 x the l1 norm of that response, 7.95: what this picture is for)
    rng = np.random.default_rng(3)
    blocks = np.zeros((N, N, 3), dtype=np.uint8)
    for c, b in enumerate((1, 4, 32)):
        blocks[..., c] = np.kron(rng.integers(0, 2, (N // b, N // b)), np.ones((b, b))).astype(np.uint8) * 255
    pics.append(blocks)
    pix = np.stack(pics)
    want = [orc.encode(p)[0] for p in pix]
    streams, _ = ctx.encode(pix)
    assert streams == want
    assert all((o == p).all() for o, p in zip(ctx.decode(streams), pix))
    opts.set("no_fine16", 1)
    assert ctx.encode(pix)[0] == want
    assert all((o == p).all() for o, p in zip(ctx.decode(streams), pix))


@pytest.mark.parametrize("case", [(64, 64, 1, 15), (64, 64, 1, 16), (128, 96, 3, 15), (128, 96, 3, 16), (192, 128, 1, 12), (1024, 512, 1, 15),
                                  (512, 1024, 3, 15), (1024, 512, 1, 16)])
def test_streams_with_15_or_16_bit_planes_on_the_finest_ring_decode_like_the_oracle(ctx, case):
    """No 8-bit source produces them, but a .dwt may hold them (encode.c:112-131 takes any int): streams coded by the
    oracle from arbitrary coefficient planes with magnitudes up to 2^bits - 1 on every ring.  15 bits still fit the
    16-bit planes the decoder keeps the finest ring in, 16 do not: a part of the batch with such a stream stays in
    the int32 pyramid (unpack.hip).  Whole, cut (bias on large values) and in mixed batches."""
    W, H, Cn, bits = case
    rng = np.random.default_rng(bits * 1000 + W)
    lin = np.zeros((Cn, W * H), dtype=np.int32)
    for c in range(Cn):
        k = W * H // 3
        at = rng.choice(W * H, k, replace=False)
        mag = rng.integers(1, 1 << rng.integers(1, bits + 1, k), k)
        lin[c, at] = mag * rng.choice([-1, 1], k)
        lin[c, rng.choice(W * H, 8, replace=False)] = ((1 << bits) - 1) * rng.choice([-1, 1], 8)   # the largest there is, both signs
        lin[c, W * H - 5:] = [(1 << bits) - 1, -((1 << bits) - 1), 1, -1, 0]                       # and on the finest ring for sure
    blob, _ = orc.encode_lin(lin, W, H)
    st = orc.decode_stage(blob, W, H, Cn, -1)
    assert st is not None and max(st[3]) == bits and (st[0] == lin).all()
    good, _ = orc.encode(orc.synth(W, H, Cn, 3, 0))
    cut = blob[: len(blob) * 4 // 5]
    for batch in ([blob], [cut], [good, blob, cut, good], [blob] * 5 + [good] * 4 + [cut]):
        outs = ctx.decode(batch)
        for b, o in zip(batch, outs):
            r = orc.decode(b)
            assert (r is None and o is None) or (o is not None and o.shape == r.shape and (o == r).all()), len(batch)


def test_a_batch_large_enough_for_the_encoder_to_run_it_in_parts(ctx, opts):
    """dwtx_encode_device cuts batches of 128 images and more into four staggered parts on the context's streams
    (codec.hip); smaller ones run as one.  130 small images (uneven parts): the bytes of every stream equal the
    oracle's, the pictures come back, and the one-stream run gives the same bytes."""
    import torch

    n, H, W, Cn = 130, 80, 96, 3
    pix = ctx.synth_pixels(n, H, W, Cn, seed0=7, kind=0)
    streams, info = ctx.encode_device(pix)
    lens = ctx.stream_lengths(info)
    host = pix.cpu().numpy()
    got = [streams[i, : int(lens[i])].cpu().numpy().tobytes() for i in range(n)]
    for i in (0, 1, 31, 32, 33, 64, 65, 97, 98, 129):
        assert got[i] == orc.encode(host[i])[0], i
    out, _ = ctx.decode_device(streams, lens, W, H, Cn)
    assert torch.equal(out.view(n, H, W, Cn), pix)
    opts.set("one_stream", 1)
    streams1, info1 = ctx.encode_device(pix)
    lens1 = ctx.stream_lengths(info1)
    assert torch.equal(lens1, lens)
    assert all(streams1[i, : int(lens[i])].cpu().numpy().tobytes() == got[i] for i in range(n))


@pytest.mark.parametrize("parts", ["2", "3", "4"])
def test_decode_batches_in_two_three_or_four_parts(ctx, parts, opts):
    """A decode batch runs as parts on streams of their own (unpack.hip dwtx_decode_planes_ex; four from 24 images on):
    whole, cut and damaged streams of one geometry, 29 of them, come out the same however the batch is cut."""
    from test_oracle import corrupted_blobs

    W, H, Cn = 120, 88, 3
    good = [orc.encode(orc.synth(W, H, Cn, 70 + i, i & 1))[0] for i in range(17)]
    blobs = good + [g[: len(g) * (i + 2) // 9] for i, g in enumerate(good[:6])] + corrupted_blobs(good[0], 6, 3)
    want = [orc.decode(b) for b in blobs]
    opts.set("decode_parts", int(parts))
    got = ctx.decode(blobs)
    for w, g in zip(want, got):
        assert (w is None and g is None) or (g is not None and g.shape == w.shape and (g == w).all())


@pytest.mark.parametrize("n,stride", [(1, 64), (7, 4096), (200, 1024), (33, 200008)])
def test_pack_streams_moves_a_batch_into_one_message(ctx, n, stride):
    """dwtx_pack_streams (the sender's side of the gather between GPUs): stream i lands at the sum of the 8-byte-rounded
    lengths before it, byte for byte; empty streams, streams that fill the stride, a length beyond the stride (clamped);
    nothing is written past the total."""
    import torch

    from dwt_amd.dist import packed_offsets

    rng = np.random.default_rng(n + stride)
    streams = torch.from_numpy(rng.integers(0, 256, (n, stride), dtype=np.uint8)).cuda()
    lens = rng.integers(0, stride + 1, n)
    lens[rng.integers(0, n)] = stride
    if n > 2:
        lens[0] = 0
        lens[n // 2] = stride + 100
    off = packed_offsets(lens.tolist(), stride)
    out = torch.full((off[-1] + 64,), 0xA5, dtype=torch.uint8, device=streams.device)
    offs_dev = torch.zeros(n + 1, dtype=torch.int64, device=streams.device)
    ctx.pack_streams(streams, torch.from_numpy(lens.astype(np.int64)).cuda(), out, offs_dev)
    assert offs_dev.cpu().tolist() == off
    o = out.cpu().numpy()
    s = streams.cpu().numpy()
    for i in range(n):
        v = min(int(lens[i]), stride)
        assert (o[off[i]:off[i] + v] == s[i, :v]).all(), i
    assert (o[off[-1]:] == 0xA5).all()


def test_the_widest_picture_against_the_oracle(ctx):
    """A side of exactly 32768 (DWTX_MAX_SIDE: the largest the reference's own arithmetic is defined for, DESIGN.md
    section 7) on a picture that is not thin: 32768x2048 gray — 1024 x 1024 curve blocks on the finest level, 15 levels;
    the stream is the oracle's byte for byte (which the thin reference-made goldens g32768x8 / c8x32768 pin on the
    reference binary), the round trip lossless, and the other orientation likewise."""
    import torch

    for W, H in ((32768, 2048), (1024, 32768)):
        pix = orc.synth(W, H, 1, 77, 0)
        want, st = orc.encode(pix)
        t = torch.from_numpy(pix[None]).cuda()
        streams, info = ctx.encode_device(t)
        lens = ctx.stream_lengths(info)
        assert int(lens[0]) == len(want)
        assert streams[0, : len(want)].cpu().numpy().tobytes() == want
        out, infos = ctx.decode_device(streams, lens, W, H, 1)
        assert infos[0].status == 0 and not infos[0].truncated
        assert torch.equal(out.view(1, H, W, 1), t)
        del t, streams, out
        torch.cuda.empty_cache()


def test_the_largest_picture_32768_square(ctx):
    """32768x32768 gray, one gigapixel — the largest picture the fence of DESIGN.md section 7 lets through: lossless round
    trip, CAPACITY gives the exact prefix of the unlimited stream, and that prefix decodes to the resolution the schedule
    says (size-independent properties: the oracle would take minutes and 12 GB here)."""
    import torch

    W = H = 32768
    free, _ = torch.cuda.mem_get_info()
    if free < 120 * (1 << 30):
        pytest.skip("needs about 120 GiB of free HBM")
    pix = ctx.synth_pixels(1, H, W, 1, seed0=5, kind=0)
    streams, info = ctx.encode_device(pix)
    lens = ctx.stream_lengths(info)
    n = int(lens[0])
    assert 6 < n < streams.shape[1]
    out, infos = ctx.decode_device(streams, lens, W, H, 1)
    assert infos[0].status == 0 and not infos[0].truncated and infos[0].level == orc.geometry(W, H).levels - 1
    assert torch.equal(out.view(1, H, W, 1), pix)
    del out
    cap = 3 << 20
    cut, cinfo = ctx.encode_device(pix, capacity=cap)
    clens = ctx.stream_lengths(cinfo)
    assert int(clens[0]) == cap
    assert torch.equal(cut[0, :cap], streams[0, :cap])
    small, sinfos = ctx.decode_device(cut, clens, W, H, 1)
    g = orc.geometry(W, H)
    lo = sinfos[0].level + 1
    assert sinfos[0].status == 0 and sinfos[0].truncated and 0 < lo <= g.levels
    head = streams[0, :6].cpu().numpy().tobytes()
    assert head == b"W5" + bytes([255, 127, 255, 127])


@pytest.mark.parametrize("shape", [(1024, 1024, 1), (260, 516, 1), (1080, 1920, 3), (512, 2048, 3), (68, 132, 1), (4096, 96, 1), (132, 4096, 3)])
def test_decode_with_two_levels_per_pass_and_without(ctx, shape, opts):
    """The decoder's transform (decode.c:258-264) fuses pairs of levels where their shapes allow it — k_inv2_level_w with the
    detail bands as 16-bit values, the finest pair writing a gray picture's pixels itself: same pictures as one launch per
    level, on smooth, noisy and two-level pictures (the largest steps an 8-bit source can make), whole and cut streams."""
    H, W, Cn = shape
    rng = np.random.default_rng(W * 3 + H)
    pix = np.stack([orc.synth(W, H, Cn, 9, 0), orc.synth(W, H, Cn, 10, 1), (rng.integers(0, 2, (H, W, Cn)) * 255).astype(np.uint8)])
    streams = [orc.encode(p)[0] for p in pix]
    cuts = [s[: len(s) * 2 // 3] for s in streams]
    for blobs in (streams, cuts):
        want = [orc.decode(b) for b in blobs]
        for off in (0, 1):
            opts.set("no_fused_levels", off)
            got = ctx.decode(blobs)
            assert all(g.shape == w.shape and (g == w).all() for g, w in zip(got, want)), (off, blobs is cuts)


def test_benchmark_frames_with_a_long_stretch_of_hand_parsed_chunks_take_the_first_walk(ctx, opts):
    """Frames 181 and 182 of the benchmark's synthetic sequence hold a stretch of 49..128 chunks in a row that the one-family
    token walk parses by hand.  Under the walk's earlier patience (48 in a row) they took the second, two-family walk — and
    their part of the batch with them, 40 % of a 192-frame batch's decoding time.  They must come back
    exactly, and without the second walk."""
    import torch

    W = H = 4096
    pix = ctx.synth_pixels(4, H, W, 1, seed0=180, kind=0)   # frames 180..183 (a batch of four: one family from the start)
    streams, info = ctx.encode_device(pix)
    lens = ctx.stream_lengths(info)
    opts.set("no_second_walk", 1)   # a walk that gives up is an error now
    out, infos = ctx.decode_device(streams, lens, W, H, 1)
    assert all(i.status == 0 and not i.truncated for i in infos)
    assert torch.equal(out.view(4, H, W, 1), pix)


@pytest.mark.parametrize("Cn", [1, 3])
def test_almost_empty_pictures_with_runs_of_millions_of_zeros(ctx, Cn):
    """A flat 2048x2048 picture with a handful of bright pixels: every plane is a few tokens whose runs count millions of
    zeros and cross many segments (rle.h:79-101) — the decoder's chunk tables count such symbols in saturating 32 bits, its
    walker takes the token that runs past a segment's end from the chunk it has in registers.  Bytes like the oracle's,
    pictures back exactly, alone and as a batch of four (one family of recorded paths)."""
    import torch

    W = H = 2048
    rng = np.random.default_rng(2048 + Cn)
    pics = []
    for k in range(4):
        pix = np.full((H, W, Cn), 90 + 20 * k, dtype=np.uint8)
        for _ in range(1 + 3 * k):
            pix[int(rng.integers(0, H)), int(rng.integers(0, W)), int(rng.integers(0, Cn))] = int(rng.integers(0, 256))
        pics.append(pix)
    want = [orc.encode(p)[0] for p in pics]
    t = torch.from_numpy(np.stack(pics)).cuda()
    streams, info = ctx.encode_device(t)
    lens = ctx.stream_lengths(info)
    for i, w in enumerate(want):
        assert int(lens[i]) == len(w)
        assert streams[i, : len(w)].cpu().numpy().tobytes() == w
    out, infos = ctx.decode_device(streams, lens, W, H, Cn)
    assert all(i.status == 0 and not i.truncated for i in infos)
    assert torch.equal(out.view(4, H, W, Cn), t)
    out1, infos1 = ctx.decode_device(streams[1:2].contiguous(), lens[1:2].contiguous(), W, H, Cn)
    assert torch.equal(out1.view(1, H, W, Cn), t[1:2])
