"""GPU: the ./encode and ./decode drop-in CLIs (argv, exit codes, stderr lines, bytes)."""
import os
import subprocess

import pytest

import orc

pytestmark = pytest.mark.gpu
ENC = os.path.join(orc.ROOT, "bin", "encode")
DEC = os.path.join(orc.ROOT, "bin", "decode")
SMPTE = os.path.join(orc.GOLDEN, "smpte.pnm")


def run(*cmd, stdin=None):
    return subprocess.run(list(cmd), input=stdin, capture_output=True, timeout=300)


def test_smpte_files_and_stderr(tmp_path):
    dwt, pnm = str(tmp_path / "a.dwt"), str(tmp_path / "a.pnm")
    r = run(ENC, SMPTE, dwt)
    assert r.returncode == 0
    assert r.stderr.decode().splitlines() == ["48 bits for meta data", "559 bits for root image",
                                              "81174 bits (10 KiB) encoded"]
    assert open(dwt, "rb").read() == open(os.path.join(orc.GOLDEN, "smpte.dwt"), "rb").read()
    r = run(DEC, dwt, pnm)
    assert r.returncode == 0
    back = orc.read_pnm(pnm)
    assert (back == orc.read_pnm(SMPTE)).all()
    assert open(pnm, "rb").read().startswith(b"P6 320 240 255\n")


def test_stdin_stdout_pipes():
    src = open(SMPTE, "rb").read()
    want = open(os.path.join(orc.GOLDEN, "smpte.dwt"), "rb").read()
    r = run(ENC, "-", "-", stdin=src)
    assert r.returncode == 0, r.stderr[-400:]
    assert len(r.stdout) == len(want), (len(r.stdout), r.stderr[-400:])
    assert r.stdout == want
    r2 = run(DEC, "-", "-", stdin=want)
    assert r2.returncode == 0, r2.stderr[-400:]
    assert r2.stdout.startswith(b"P6 320 240 255\n"), r2.stdout[:32]
    body, pix = r2.stdout[len(b"P6 320 240 255\n"):], orc.read_pnm(SMPTE).tobytes()
    assert len(body) == len(pix), (len(body), r2.stderr[-400:])
    assert body == pix, sum(a != b for a, b in zip(body, pix))


def test_capacity_and_pixels_arguments(tmp_path):
    dwt, pnm = str(tmp_path / "a.dwt"), str(tmp_path / "a.pnm")
    for cap in (100, 4096):
        r = run(ENC, SMPTE, dwt, str(cap))
        assert r.returncode == 0
        assert open(dwt, "rb").read() == open(os.path.join(orc.GOLDEN, f"smpte_cap{cap}.dwt"), "rb").read()
        assert run(DEC, dwt, pnm).returncode == 0
        want = orc.decode(open(dwt, "rb").read())
        assert (orc.read_pnm(pnm) == want).all()
    full = os.path.join(orc.GOLDEN, "smpte.dwt")
    for px in (0, 300, 5000):
        assert run(DEC, full, pnm, str(px)).returncode == 0
        assert (orc.read_pnm(pnm) == orc.decode(open(full, "rb").read(), px)).all()


def test_exit_codes(tmp_path):
    assert run(ENC).returncode == 1
    assert run(DEC, "a").returncode == 1
    assert run(ENC, str(tmp_path / "missing.pnm"), str(tmp_path / "o.dwt")).returncode == 1
    small = str(tmp_path / "small.pnm")
    orc.write_pnm(small, orc.synth(7, 20, 1, 0, 0))
    assert run(ENC, small, str(tmp_path / "o.dwt")).returncode == 1          # encode.c:145
    bad = str(tmp_path / "bad.dwt")
    open(bad, "wb").write(b"X5" + bytes(20))
    assert run(DEC, bad, str(tmp_path / "o.pnm")).returncode == 1            # decode.c:146
    open(bad, "wb").write(open(os.path.join(orc.GOLDEN, "smpte.dwt"), "rb").read()[:12])
    assert run(DEC, bad, str(tmp_path / "o.pnm")).returncode == 1            # root image cut off


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not shipped")
def test_against_reference_binaries(tmp_path):
    src = str(tmp_path / "i.pnm")
    for (W, H, Cn, cap) in [(131, 77, 3, 0), (255, 257, 1, 0), (640, 360, 3, 20000), (64, 64, 1, 333)]:
        orc.write_pnm(src, orc.synth(W, H, Cn, 3, 0))
        a, b = str(tmp_path / "a.dwt"), str(tmp_path / "b.dwt")
        extra = [str(cap)] if cap else []
        ra = run(os.path.join(orc.REF_DIR, "encode"), src, a, *extra)
        rb = run(ENC, src, b, *extra)
        assert ra.returncode == rb.returncode == 0
        assert ra.stderr == rb.stderr
        assert open(a, "rb").read() == open(b, "rb").read()
        pa, pb = str(tmp_path / "a.pnm"), str(tmp_path / "b.pnm")
        assert run(os.path.join(orc.REF_DIR, "decode"), a, pa).returncode == 0
        assert run(DEC, a, pb).returncode == 0
        assert open(pa, "rb").read() == open(pb, "rb").read()


def _stderr_cases():
    import json

    recs = json.load(open(os.path.join(orc.GOLDEN, "decode_stderr.json")))
    picked, seen = [], set()
    for r in recs:
        key = (r["fixture"], r["returncode"], r["stderr"], r["pixels_arg"] is None)
        if "zeros not read" in r["stderr"] or key not in seen:
            picked.append(r)
        seen.add(key)
    return picked


@pytest.mark.parametrize("rec", _stderr_cases(), ids=lambda r: f"{r['fixture']}-{r['cut']}-{r['pixels_arg']}")
def test_decode_diagnostics_match_the_reference(tmp_path, rec):
    """decode's exit code and stderr text (bytes.h:101 "reached end of file", rle.h:45 "zeros not
    read.") for whole, cut-off and PIXELS-capped streams, as recorded from the real reference
    (tests/golden/make_decode_stderr.py)."""
    data = open(os.path.join(orc.GOLDEN, rec["fixture"]), "rb").read()[: rec["cut"]]
    (tmp_path / "in.dwt").write_bytes(data)
    cmd = [DEC, "in.dwt", "out.pnm"] + ([str(rec["pixels_arg"])] if rec["pixels_arg"] is not None else [])
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, timeout=300)
    assert r.returncode == rec["returncode"], r.stderr[-300:]
    assert r.stderr.decode() == rec["stderr"]
    if rec["returncode"] == 0:
        want = orc.decode(data, -1 if rec["pixels_arg"] is None else rec["pixels_arg"])
        assert (orc.read_pnm(str(tmp_path / "out.pnm")) == want).all()


def _small_goldens():
    import json

    g = json.load(open(os.path.join(orc.GOLDEN, "golden.json")))
    return sorted(k for k, v in g.items() if v["seed"] is not None and v["W"] * v["H"] <= 640 * 360)


@pytest.mark.parametrize("name", _small_goldens())
def test_encode_cli_against_reference_goldens(tmp_path, name):
    """encode's three stderr lines (encode.c:176,180,230), its bytes and decode's picture for the
    synthetic goldens the real reference produced (capacity and PIXELS arguments included)."""
    import hashlib
    import json

    rec = json.load(open(os.path.join(orc.GOLDEN, "golden.json")))[name]
    src, dwt, pnm = str(tmp_path / "in.pnm"), str(tmp_path / "o.dwt"), str(tmp_path / "o.pnm")
    orc.write_pnm(src, orc.synth(rec["W"], rec["H"], rec["C"], rec["seed"], rec["kind"]))
    r = run(ENC, src, dwt, *([str(rec["capacity"])] if rec["capacity"] else []))
    assert r.returncode == 0, r.stderr[-300:]
    assert r.stderr.decode().splitlines() == rec["encode_stderr"]
    data = open(dwt, "rb").read()
    assert len(data) == rec["dwt_len"] and hashlib.sha256(data).hexdigest() == rec["dwt_sha256"]
    r = run(DEC, dwt, pnm, *([str(rec["pixels_arg"])] if rec["pixels_arg"] is not None else []))
    assert r.returncode == 0, r.stderr[-300:]
    back = orc.read_pnm(pnm)
    assert back.shape[:2] == (rec["dec_H"], rec["dec_W"])
    assert hashlib.sha256(back.tobytes()).hexdigest() == rec["dec_sha256"]


def _odd_pnms():
    px = bytes(range(256)) * 3
    body5, body6 = px[:12 * 9], px[:12 * 9 * 3]
    return {
        "plain_p5": b"P5 12 9 255\n" + body5,
        "plain_p6": b"P6\n12 9\n255\n" + body6,
        "comment_lines": b"P5\n# made by hand\n12 9\n# another\n255\n" + body5,
        "tabs_and_spaces": b"P6 \t 12   9\t255\n" + body6,
        "crlf_after_maxval": b"P5 12 9 255\r\n" + body5,
        "maxval_65535": b"P5 12 9 65535\n" + body5 * 2,
        "maxval_100": b"P5 12 9 100\n" + body5,
        "p4_bitmap": b"P4 16 9\n" + bytes(18),
        "p3_ascii": b"P3 2 2 255\n1 2 3 4 5 6 7 8 9 10 11 12\n",
        "short_pixels": b"P6 12 9 255\n" + body6[:100],
        "extra_tail": b"P5 12 9 255\n" + body5 + b"trailing junk",
        "too_small": b"P5 7 20 255\n" + bytes(140),
        "zero_width": b"P5 0 9 255\n",
        "no_maxval": b"P5 12 9\n",
        "empty": b"",
        "just_magic": b"P5",
        "negative": b"P5 -12 9 255\n" + body5,
        "huge_width": b"P5 70000 9 255\n" + body5,
    }


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name", sorted(_odd_pnms()))
def test_pnm_reading_matches_the_reference_binary(tmp_path, name):
    """read_pnm's accept/reject rules and messages (pnm.h:14-90): bin/encode and the reference's own encode
    on the same odd PNM files — exit code, stderr and bytes."""
    (tmp_path / "in.pnm").write_bytes(_odd_pnms()[name])
    mine = subprocess.run([ENC, "in.pnm", "mine.dwt"], cwd=tmp_path, capture_output=True, timeout=300)
    ref = subprocess.run([os.path.join(orc.REF_DIR, "encode"), "in.pnm", "ref.dwt"], cwd=tmp_path, capture_output=True, timeout=300)
    # on two of these ("EOF while reading" with nothing after the header) the reference prints its message and
    # then crashes with SIGSEGV; the drop-in prints the same message and exits 1
    assert mine.returncode == (1 if ref.returncode < 0 else ref.returncode), (mine.stderr, ref.stderr)
    assert mine.stderr == ref.stderr
    if ref.returncode == 0:
        assert (tmp_path / "mine.dwt").read_bytes() == (tmp_path / "ref.dwt").read_bytes()


def _odd_dwts():
    good = open(os.path.join(orc.GOLDEN, "c37x53.dwt"), "rb").read()
    return {
        "good": good,
        "bad_magic": b"X6" + good[2:],
        "bad_channel_magic": b"W7" + good[2:],
        "lowercase_magic": b"w6" + good[2:],
        "p5_magic_on_rgb_stream": b"W5" + good[2:],
        "width_7": b"W6" + bytes([6, 0]) + good[4:],
        "height_3": good[:4] + bytes([2, 0]) + good[6:],
        "other_width": good[:2] + bytes([99, 0]) + good[4:],
        "empty": b"",
        "one_byte": b"W",
        "one_foreign_byte": b"X",
        "two_bytes_bad_number": b"W7",
        "two_bytes": b"W5",
        "three_bytes": b"W5\x01",
        "five_bytes": b"W6\x24\x00\x34",
        "short_foreign": b"P6\n3",
        "header_only": good[:6],
        "zeros_after_header": good[:6] + bytes(400),
        "ones_after_header": good[:6] + b"\xff" * 400,
    }


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name", sorted(_odd_dwts()))
def test_dwt_header_handling_matches_the_reference_binary(tmp_path, name):
    """decode.c:142-186 on damaged or foreign headers: bin/decode and the reference's own decode — exit
    code, stderr and picture."""
    (tmp_path / "in.dwt").write_bytes(_odd_dwts()[name])
    mine = subprocess.run([DEC, "in.dwt", "mine.pnm"], cwd=tmp_path, capture_output=True, timeout=300)
    ref = subprocess.run([os.path.join(orc.REF_DIR, "decode"), "in.dwt", "ref.pnm"], cwd=tmp_path, capture_output=True, timeout=300)
    assert mine.returncode == (1 if ref.returncode < 0 else ref.returncode), (mine.stderr, ref.stderr)
    assert mine.stderr == ref.stderr
    if ref.returncode == 0:
        assert (tmp_path / "mine.pnm").read_bytes() == (tmp_path / "ref.pnm").read_bytes()


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("arg", ["0", "-5", "abc", "12abc", "99999999999", "1", "47", "48", "600", " 300", "+300", "3e2", ""])
def test_third_argument_parsing_matches_the_reference_binary(tmp_path, arg):
    """CAPACITY (encode.c:150-152) and PIXELS (decode.c:165-171) go through atoi in the reference: same bytes,
    exit codes and messages for odd spellings."""
    pix = orc.synth(53, 37, 3, 4, 0)
    orc.write_pnm(str(tmp_path / "in.pnm"), pix)
    res = {}
    for who, enc, dec in (("mine", ENC, DEC), ("ref", os.path.join(orc.REF_DIR, "encode"), os.path.join(orc.REF_DIR, "decode"))):
        e = subprocess.run([enc, "in.pnm", who + ".dwt", arg], cwd=tmp_path, capture_output=True, timeout=300)
        dwt = (tmp_path / (who + ".dwt")).read_bytes() if (tmp_path / (who + ".dwt")).exists() else None
        full = subprocess.run([enc, "in.pnm", who + "_full.dwt"], cwd=tmp_path, capture_output=True, timeout=300)
        assert full.returncode == 0
        d = subprocess.run([dec, who + "_full.dwt", who + ".pnm", arg], cwd=tmp_path, capture_output=True, timeout=300)
        pnm = (tmp_path / (who + ".pnm")).read_bytes() if (tmp_path / (who + ".pnm")).exists() else None
        res[who] = (e.returncode, e.stderr, dwt, d.returncode, d.stderr, pnm)
    assert res["mine"] == res["ref"]


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("args", [[], ["a"], ["a", "b", "c", "d"], ["missing.pnm", "o"], ["in.pnm", "no/such/dir/o"],
                                  ["in.pnm", "o", "10", "extra"]])
def test_argument_errors_match_the_reference_binaries(tmp_path, args):
    """Usage lines, unreadable inputs and unwritable outputs: same exit codes and messages (argv[0] aside)."""
    orc.write_pnm(str(tmp_path / "in.pnm"), orc.synth(16, 16, 1, 0, 0))
    (tmp_path / "in.dwt").write_bytes(open(os.path.join(orc.GOLDEN, "g8x8.dwt"), "rb").read())
    for mine, ref in ((ENC, os.path.join(orc.REF_DIR, "encode")), (DEC, os.path.join(orc.REF_DIR, "decode"))):
        a = [x.replace("in.pnm", "in.dwt") if mine == DEC else x for x in args]
        m = subprocess.run([mine] + a, cwd=tmp_path, capture_output=True, timeout=300)
        r = subprocess.run([ref] + a, cwd=tmp_path, capture_output=True, timeout=300)
        assert m.returncode == (1 if r.returncode < 0 else r.returncode), (m.stderr, r.stderr)
        assert m.stderr.replace(mine.encode(), b"PROG") == r.stderr.replace(ref.encode(), b"PROG")


def test_decode_leaves_and_takes_a_sidecar_index(tmp_path):
    """Beyond the reference: `decode` with DWTX_WRITE_INDEX leaves input.dwt.idx behind; a later decode finds it
    and walks the segments at once.  The output never depends on it (a damaged or foreign index is ignored)."""
    import shutil

    dwt, pnm = str(tmp_path / "a.dwt"), str(tmp_path / "a.pnm")
    shutil.copy(os.path.join(orc.GOLDEN, "smpte.dwt"), dwt)
    want = orc.read_pnm(SMPTE)

    def decode(**env):
        r = subprocess.run([DEC, dwt, pnm], capture_output=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-300:]
        assert r.stderr == b""
        assert (orc.read_pnm(pnm) == want).all()

    decode()
    assert not os.path.exists(dwt + ".idx")
    decode(DWTX_WRITE_INDEX="1")
    idx = open(dwt + ".idx", "rb").read()
    # `encode` can leave the same file behind (it walks the stream it has just written once)
    dwt2 = str(tmp_path / "b.dwt")
    r = subprocess.run([ENC, SMPTE, dwt2], capture_output=True, timeout=300, env=dict(os.environ, DWTX_WRITE_INDEX="1"))
    assert r.returncode == 0 and open(dwt2, "rb").read() == open(dwt, "rb").read()
    assert open(dwt2 + ".idx", "rb").read() == idx
    r = subprocess.run([ENC, SMPTE, dwt2, "4096"], capture_output=True, timeout=300, env=dict(os.environ, DWTX_WRITE_INDEX="1"))
    os.remove(dwt2 + ".idx")
    r = subprocess.run([ENC, SMPTE, dwt2, "4096"], capture_output=True, timeout=300, env=dict(os.environ, DWTX_WRITE_INDEX="1"))
    assert r.returncode == 0 and not os.path.exists(dwt2 + ".idx")   # a stream cut by CAPACITY has no index
    assert idx[:4] == b"DWTI" and len(idx) == 32 + 32 * int.from_bytes(idx[16:20], "little")
    decode(DWTX_NO_INDEX_FALLBACK="1")            # the index is accepted (a rejected one would be an error here)
    open(dwt + ".idx", "wb").write(idx[:200] + bytes(len(idx) - 200))
    decode()                                      # damaged: ignored
    r = subprocess.run([DEC, dwt, pnm], capture_output=True, timeout=300, env=dict(os.environ, DWTX_NO_INDEX_FALLBACK="1"))
    assert r.returncode == 1                      # ... and that it was the fallback shows here
    open(dwt + ".idx", "wb").write(idx[:20])
    decode()                                      # cut short: not even read
    open(dwt + ".idx", "wb").write(idx)
    r = run(DEC, dwt, pnm, "300")                 # a PIXELS cap: the plain walk, same picture as ever
    assert r.returncode == 0 and (orc.read_pnm(pnm) == orc.decode(open(dwt, "rb").read(), 300)).all()


@pytest.mark.parametrize("name,said", [("damaged_order_beyond_31_47x650x1.dwt", b""), ("damaged_wide_root_213x18x3.dwt", b"509 zeros not read.\n")])
def test_damaged_streams_decode_like_the_reference_binary(tmp_path, name, said):
    """tests/test_oracle.py::test_damaged_streams_that_leave_the_range_of_the_shifts, through the CLI: same exit
    code, same (empty) stderr, same picture as the oracle's."""
    src = os.path.join(orc.GOLDEN, name)
    pnm = str(tmp_path / "a.pnm")
    r = run(DEC, src, pnm)
    assert r.returncode == 0 and r.stderr == said   # what oracle/_ref/decode says (tests/test_oracle.py pins it)
    assert (orc.read_pnm(pnm) == orc.decode(open(src, "rb").read())).all()


def test_a_stream_that_claims_more_than_16_bit_planes_is_refused_with_a_message(tmp_path):
    """The one documented difference from the reference: decode.c:183-186 accepts any plane
    count and decodes such a — necessarily damaged — stream to garbage with exit code 0; this decoder handles
    at most 16 planes, says so on stderr and exits 1 without writing a picture."""
    dwt, pnm, ref_pnm = str(tmp_path / "m.dwt"), str(tmp_path / "m.pnm"), str(tmp_path / "r.pnm")
    open(dwt, "wb").write(orc.many_plane_stream(40, 24, 1, [20], 1))
    r = run(DEC, dwt, pnm)
    assert r.returncode == 1 and b"more than 16 bit planes" in r.stderr and not os.path.exists(pnm)
    if orc.have_ref():
        assert run(os.path.join(orc.REF_DIR, "decode"), dwt, ref_pnm).returncode == 0 and os.path.exists(ref_pnm)


def test_sides_above_32768_are_refused_with_a_message(tmp_path):
    """The second documented difference: encode.c:140 lets sides up to 65536 through and then overflows at encode.c:45
    (the stream the reference writes does not decode to the picture: tests/test_oracle.py); bin/encode and bin/decode
    say so and exit 1 without writing anything."""
    pix = orc.synth(32776, 8, 1, 5, 0)
    src, dwt, pnm = str(tmp_path / "i.pnm"), str(tmp_path / "o.dwt"), str(tmp_path / "o.pnm")
    orc.write_pnm(src, pix)
    r = run(ENC, src, dwt)
    assert r.returncode == 1 and b"sides above 32768 are not supported" in r.stderr and not os.path.exists(dwt)
    if orc.have_ref():
        ref = run(os.path.join(orc.REF_DIR, "encode"), src, dwt)
        assert ref.returncode == 0 and os.path.exists(dwt)      # a stream that claims 32776x8 ...
        r = run(DEC, dwt, pnm)                                    # ... which bin/decode refuses
        assert r.returncode == 1 and b"sides above 32768 are not supported" in r.stderr and not os.path.exists(pnm)
    # the largest side inside the fence works (bytes: golden g32768x8 in test_codec_gpu.py)
    orc.write_pnm(src, orc.synth(32768, 8, 1, 5, 0))
    r = run(ENC, src, dwt)
    assert r.returncode == 0
    r = run(DEC, dwt, pnm)
    assert r.returncode == 0 and (orc.read_pnm(pnm) == orc.synth(32768, 8, 1, 5, 0)).all()
