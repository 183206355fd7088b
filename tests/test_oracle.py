"""CPU tests: the oracle (our C restatement) against the committed goldens that the
real reference produced (tests/golden/make_golden.py), and — when oracle/_ref is
present — against the reference binaries directly."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import orc

G = json.load(open(os.path.join(orc.GOLDEN, "golden.json")))


def sha(b):
    return hashlib.sha256(b).hexdigest()


def case_input(rec):
    if rec["seed"] is None:
        return orc.read_pnm(os.path.join(orc.GOLDEN, "smpte.pnm"))
    return orc.synth(rec["W"], rec["H"], rec["C"], rec["seed"], rec["kind"])


SMALL = [k for k, v in G.items() if v["W"] * v["H"] <= 1024 * 1024]


@pytest.mark.parametrize("name", sorted(G))
def test_golden_encode_decode(name):
    rec = G[name]
    if rec.get("heavy"):
        pytest.skip("heavy golden (minutes of CPU): covered by the GPU test")
    if name not in SMALL and os.environ.get("DWT_FULL_GOLDEN", "1") == "0":
        pytest.skip("large golden skipped")
    if rec["W"] * rec["H"] * rec["C"] > 20_000_000:
        pytest.skip("config E frame (tens of seconds on the CPU): covered by the GPU test")
    pix = case_input(rec)
    assert sha(pix.tobytes()) == rec["input_sha256"], "synthetic generator drifted"
    data, st = orc.encode(pix, rec["capacity"])
    assert len(data) == rec["dwt_len"]
    assert sha(data) == rec["dwt_sha256"]
    if "dwt_file" in rec:
        assert data == open(os.path.join(orc.GOLDEN, rec["dwt_file"]), "rb").read()
    lines = [f"{st.meta_bits} bits for meta data", f"{st.root_bits} bits for root image",
             f"{st.total_bits} bits ({st.kib} KiB) encoded"]
    assert lines == rec["encode_stderr"]
    px = rec["pixels_arg"]
    back = orc.decode(data, -1 if px is None else px)
    assert back.shape[:2] == (rec["dec_H"], rec["dec_W"])
    assert sha(back.tobytes()) == rec["dec_sha256"]
    assert bool(back.shape == pix.shape and (back == pix).all()) == rec["lossless"]


def test_smpte_known_answers():
    # SURVEY.md §4 table
    pix = orc.read_pnm(os.path.join(orc.GOLDEN, "smpte.pnm"))
    data, st = orc.encode(pix)
    assert len(data) == 10147
    assert sha(data) == "2ac1d6b75498f2982c2fbf80edc9ea74c956affc8d45584597de530b41c9dfd0"
    assert data[:16].hex() == "57363f01ef009866940c7979c16c9221"
    assert (st.meta_bits, st.root_bits, st.total_bits) == (48, 559, 81174)
    assert list(st.planes) == [8, 9, 9] and st.levels == 6


def test_geometry():
    g = orc.geometry(4096, 4096)
    assert g.levels == 10 and g.widths[0] == 4 and g.widths[10] == 4096
    g = orc.geometry(320, 240)
    assert g.levels == 6 and (g.widths[0], g.heights[0]) == (5, 4)
    g = orc.geometry(1920, 1080)
    assert g.levels == 8 and (g.widths[0], g.heights[0]) == (8, 5)
    assert list(g.heights[:9]) == [5, 9, 17, 34, 68, 135, 270, 540, 1080]
    g = orc.geometry(8, 8)
    assert g.levels == 1 and g.lengths[1] == 8 and g.lengths[0] == 4


def test_lifting_roundtrip_and_edges():
    rng = np.random.default_rng(1)
    for (H, W, Cn) in [(8, 8, 1), (9, 13, 3), (16, 31, 1), (33, 8, 3), (100, 75, 1)]:
        img = rng.integers(-300, 300, size=(H, W, Cn), dtype=np.int32)
        pyr = orc.forward(img)
        assert (orc.inverse(pyr) == img).all()
    # odd length: last even sample is not updated (cdf53.h:19-21, M = N & ~1)
    line = np.arange(9, dtype=np.int32).reshape(1, 9, 1) ** 2
    col = np.repeat(line, 8, axis=0)
    pyr = orc.forward(col)
    # after the row pass the last low-pass sample equals the input sample x[8]=64; the
    # column pass on constant columns leaves low-pass rows unchanged
    assert pyr[0, 4, 0] == 64


def test_hilbert_is_a_curve():
    n = 16
    pts = [orc.hilbert(n, d) for d in range(n * n)]
    assert sorted(pts) == [(x, y) for x in range(n) for y in range(n)]
    for a, b in zip(pts, pts[1:]):
        assert abs(a[0] - b[0]) + abs(a[1] - b[1]) == 1
    assert pts[:4] == [(0, 0), (1, 0), (1, 1), (0, 1)]


def test_truncation_is_prefix_and_decodes():
    pix = orc.synth(131, 77, 3, 5, 0)
    full, _ = orc.encode(pix)
    for cap in (1, 6, 7, 64, 500, 5000, len(full) - 1, len(full), len(full) + 5):
        data, _ = orc.encode(pix, cap)
        assert data == full[:cap]
        if cap >= 500:   # the 9x5 root of 3 channels alone needs > 64 bytes
            assert orc.decode(data) is not None


def test_flat_image_quirk():
    # all-zero detail -> decoder emits widths[1] x heights[1]
    pix = np.full((16, 100, 3), 77, dtype=np.uint8)
    data, st = orc.encode(pix)
    assert list(st.planes) == [0, 0, 0]
    back = orc.decode(data)
    assert back.shape == (8, 50, 3)


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("shape", [(8, 8, 1), (8, 9, 3), (300, 17, 1), (53, 37, 3), (77, 131, 3), (64, 64, 1),
                                   (16, 100, 3), (15, 15, 1), (240, 320, 3)])
def test_against_reference_binaries(tmp_path, shape):
    H, W, Cn = shape
    for kind in (0, 1):
        pix = orc.synth(W, H, Cn, 11 + kind, kind)
        src, dwt, dec = (str(tmp_path / n) for n in ("i.pnm", "o.dwt", "o.pnm"))
        orc.write_pnm(src, pix)
        for cap in (0, 97, 1500):
            cmd = [os.path.join(orc.REF_DIR, "encode"), src, dwt] + ([str(cap)] if cap else [])
            subprocess.run(cmd, capture_output=True, check=True)
            ref = open(dwt, "rb").read()
            mine, _ = orc.encode(pix, cap)
            assert mine == ref
            r = subprocess.run([os.path.join(orc.REF_DIR, "decode"), dwt, dec], capture_output=True)
            back = orc.decode(ref)
            if r.returncode:
                assert back is None
            else:
                assert (orc.read_pnm(dec) == back).all() and orc.read_pnm(dec).shape == back.shape


def corrupted_blobs(good, n=40, seed=7):
    """The corruptions tests/test_codec_gpu.py feeds the GPU decoder (bit flips, garbage runs, junk tails)."""
    rng = np.random.default_rng(seed)
    blobs = []
    for case in range(n):
        b = bytearray(good)
        kind = case % 4
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                i = int(rng.integers(6, len(b)))
                b[i] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            i = int(rng.integers(6, len(b) - 40))
            b[i:i + 32] = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
        elif kind == 2:
            i = int(rng.integers(40, len(b)))
            b[i:] = bytes(rng.integers(0, 256, len(b) - i, dtype=np.uint8))
        else:
            i = int(rng.integers(40, len(b)))
            b[i:] = bytes([0 if case % 8 == 3 else 255]) * (len(b) - i)
        blobs.append(bytes(b))
    return blobs


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
def test_corrupted_streams_against_reference_binary(tmp_path):
    """Pins the restatement's error paths (rle.h:58-61,95-101, vli.h, decode.c:204-239) on damaged streams."""
    pix = orc.synth(96, 80, 3, 12, 0)
    good, _ = orc.encode(pix)
    dwt, dec = str(tmp_path / "c.dwt"), str(tmp_path / "c.pnm")
    for blob in corrupted_blobs(good):
        open(dwt, "wb").write(blob)
        if os.path.exists(dec):
            os.remove(dec)
        r = subprocess.run([os.path.join(orc.REF_DIR, "decode"), dwt, dec], capture_output=True, timeout=120)
        back = orc.decode(blob)
        if r.returncode:
            assert back is None
        else:
            ref = orc.read_pnm(dec)
            assert back is not None and ref.shape == back.shape and (ref == back).all()


MANY_PLANES = [(40, 24, 1, [17]), (40, 24, 1, [20]), (24, 40, 3, [9, 29, 3]), (64, 64, 1, [24]), (37, 53, 3, [18, 18, 18])]


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("case", MANY_PLANES)
def test_streams_that_claim_more_than_16_bit_planes_against_reference_binary(tmp_path, case):
    """decode.c:183-186 takes whatever plane count get_vli() returns; only a damaged stream can claim more than 16
    (8-bit sources stay below 12).  The reference decodes such a stream (to garbage) and so does the restatement,
    bit for bit; the GPU decoder refuses it — status 2, tests/test_unpack_gpu.py — which is the one documented
    difference."""
    W, H, Cn, planes = case
    for seed in (1, 2, 3):
        blob = orc.many_plane_stream(W, H, Cn, planes, seed)
        dwt, dec = str(tmp_path / "m.dwt"), str(tmp_path / "m.pnm")
        open(dwt, "wb").write(blob)
        r = subprocess.run([os.path.join(orc.REF_DIR, "decode"), dwt, dec], capture_output=True, timeout=120)
        assert r.returncode == 0
        st = orc.decode_stage(blob, W, H, Cn, -1)
        assert st is not None and st[3] == planes[:Cn]
        back, ref = orc.decode(blob), orc.read_pnm(dec)
        assert back is not None and ref.shape == back.shape and (ref == back).all()


DAMAGED = [("damaged_order_beyond_31_47x650x1.dwt", 47, 650, 1, b""), ("damaged_wide_root_213x18x3.dwt", 213, 18, 3, b"509 zeros not read.\n")]


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("wh", [(32776, 8), (8, 32769), (40000, 9)])
def test_sides_above_32768_what_the_reference_binary_does_and_the_oracle_refuses(tmp_path, wh):
    """The second documented difference.  encode.c:140 accepts sides up to 65536, but above 32768
    the finest level's curve square is 65536 wide and `lengths[l+1] * lengths[l+1]` (encode.c:45, decode.c:47) wraps to 0
    in int: the reference binary returns 0 in no time, leaves the finest ring out of the stream and its own decoder does
    not give the picture back.  The oracle (and the library, tests/test_codec_gpu.py / test_cli_gpu.py) refuse such sizes."""
    W, H = wh
    pix = orc.synth(W, H, 1, 5, 0)
    src, dwt, dec = (str(tmp_path / n) for n in ("i.pnm", "o.dwt", "o.pnm"))
    orc.write_pnm(src, pix)
    r = subprocess.run([os.path.join(orc.REF_DIR, "encode"), src, dwt], capture_output=True, timeout=120)
    assert r.returncode == 0
    ref = open(dwt, "rb").read()
    root_bits = int(r.stderr.decode().splitlines()[1].split()[0])
    total_bits = int(r.stderr.decode().splitlines()[2].split()[0])
    assert total_bits - root_bits < 200, "the reference coded the finest ring after all?"
    r = subprocess.run([os.path.join(orc.REF_DIR, "decode"), dwt, dec], capture_output=True, timeout=120)
    assert r.returncode == 0
    back = orc.read_pnm(dec)
    assert back.shape != pix.shape or not (back == pix).all(), "the reference round-trips this size after all?"
    with pytest.raises(ValueError):
        orc.encode(pix)
    assert orc.decode(ref) is None


@pytest.mark.parametrize("wh", [(32768, 8), (8, 32768)])
def test_the_largest_side_against_reference_made_goldens_is_quick(wh):
    """Sides of exactly 32768 are inside the range (goldens g32768x8 / c8x32768 come from the reference binary, which
    walks 2^30 curve indices for them); the oracle steps over the curve's empty squares and takes a fraction of a second."""
    import time

    W, H = wh
    pix = orc.synth(W, H, 1, 5, 0)
    t0 = time.time()
    data, _ = orc.encode(pix)
    back = orc.decode(data)
    assert time.time() - t0 < 20
    assert back.shape == pix.shape and (back == pix).all()


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("case", DAMAGED)
def test_damaged_streams_that_leave_the_range_of_the_shifts(tmp_path, case):
    """Two damaged streams a seeded sweep found (tools/fuzz_decode.py): in one the VLI order passes 31 (vli.h:90-91
    shifts by it), in the other the root image claims more than 32 bits per coefficient (bits.h:100 shifts by the
    bit index).  Both are undefined in C; what the reference's binary does (x86: shift counts modulo 32) is what
    the restatement and the GPU decoder (tests/test_unpack_gpu.py) have to do."""
    name, W, H, Cn, said = case
    blob = open(os.path.join(orc.GOLDEN, name), "rb").read()
    dwt, dec = str(tmp_path / "d.dwt"), str(tmp_path / "d.pnm")
    open(dwt, "wb").write(blob)
    r = subprocess.run([os.path.join(orc.REF_DIR, "decode"), dwt, dec], capture_output=True, timeout=120)
    assert r.returncode == 0 and r.stderr == said   # the reference decodes them to the end of its schedule, no "end of file"
    back = orc.decode(blob)
    ref = orc.read_pnm(dec)
    assert back is not None and ref.shape == back.shape and (ref == back).all()


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
def test_statistics_lines_under_tiny_capacities_against_reference_binary(tmp_path):
    """With CAPACITY below header + root image the reference's three stderr numbers are its bit writer's
    counters after fields were given up (bits.h:58-78); the restatement reproduces them."""
    for (W, H, Cn, seed) in ((53, 37, 3, 4), (64, 40, 1, 9)):
        pix = orc.synth(W, H, Cn, seed, 0)
        orc.write_pnm(str(tmp_path / "in.pnm"), pix)
        for cap in list(range(1, 20)) + [31, 47, 48, 60, 80, 85, 86, 87, 88, 90, 100, 120, 200]:
            r = subprocess.run([os.path.join(orc.REF_DIR, "encode"), "in.pnm", "o.dwt", str(cap)], cwd=tmp_path, capture_output=True)
            data, st = orc.encode(pix, cap)
            assert (tmp_path / "o.dwt").read_bytes() == data
            assert r.stderr.decode() == (f"{st.meta_bits} bits for meta data\n{st.root_bits} bits for root image\n"
                                         f"{st.total_bits} bits ({st.kib} KiB) encoded\n")


def test_entropy_stage_entry_equals_the_whole_encoder():
    """orc_encode_lin (the coder on given linearised planes, used by the GPU tests for coefficient ranges no 8-bit
    picture produces) is the same code path orc_encode runs after its transform: same bytes on pictures, and
    planes with 15 bit planes decode back."""
    for W, H, Cn in ((131, 77, 3), (64, 64, 1), (300, 17, 1)):
        pix = orc.synth(W, H, Cn, 3, 0)
        _, lin, _ = orc.stage_dump(pix)
        assert orc.encode_lin(lin, W, H)[0] == orc.encode(pix)[0]
    rng = np.random.default_rng(1)
    lin = (rng.integers(-30000, 30000, (1, 64 * 64)) * (rng.random((1, 64 * 64)) < 0.3)).astype(np.int32)
    data, st = orc.encode_lin(lin, 64, 64)
    assert list(st.planes)[:1] == [15]
    assert (orc.decode_stage(data, 64, 64, 1)[0] == lin).all()
