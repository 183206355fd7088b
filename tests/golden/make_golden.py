#!/usr/bin/env python3
"""Generate tests/golden/golden.json + small .dwt fixtures with the REAL reference.

Run in the build container (needs /root/reference): compiles the reference from
its own sources into oracle/_ref (oracle/Makefile target `ref`), renders the
integer-only synthetic inputs of SURVEY.md §8d, runs _ref/encode and
_ref/decode on them and records byte lengths, sha256 digests, stderr stats and
(for small cases) the .dwt bytes themselves.  Fixtures are data only: inputs
are regenerated from (W,H,C,seed,kind) by oracle/orc_synth.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402

CASES = [
    # name, W, H, C, seed, kind, capacity, pixels_arg
    ("smpte", None, None, None, None, None, 0, None),
    ("smpte_cap4096", None, None, None, None, None, 4096, None),
    ("smpte_cap100", None, None, None, None, None, 100, None),
    ("g8x8", 8, 8, 1, 1, 0, 0, None),
    ("c9x8", 9, 8, 3, 2, 1, 0, None),
    ("g17x300", 17, 300, 1, 3, 0, 0, None),
    ("c37x53", 37, 53, 3, 4, 0, 0, None),
    ("c131x77", 131, 77, 3, 5, 0, 0, None),
    ("g255x257", 255, 257, 1, 6, 1, 0, None),
    ("c64x64", 64, 64, 3, 7, 0, 0, None),
    ("c131x77_cap500", 131, 77, 3, 5, 0, 500, None),
    ("c131x77_px1000", 131, 77, 3, 5, 0, 0, 1000),
    ("g512x512", 512, 512, 1, 8, 0, 0, None),
    ("c640x360", 640, 360, 3, 9, 0, 0, None),
    ("g1024x1024_noise", 1024, 1024, 1, 10, 1, 0, None),
    ("c1920x1080", 1920, 1080, 3, 0, 0, 0, None),
    ("g4096x4096", 4096, 4096, 1, 0, 0, 0, None),
    ("c1920x1080_cap65536", 1920, 1080, 3, 0, 0, 65536, None),
    ("c4096x4096", 4096, 4096, 3, 0, 0, 0, None),          # one frame of BASELINE.json configs[4]
    ("c4096x4096_px70000", 4096, 4096, 3, 0, 0, 0, 70000),
    # sides in (16384, 32768]: the largest the reference's own arithmetic is defined for (above 32768 encode.c:45's
    # `lengths * lengths` wraps to 0 in int, see ORC_MAX_SIDE).  The reference walks 2^30 curve indices per level for these:
    # about half a minute each way.
    ("g32768x8", 32768, 8, 1, 21, 0, 0, None),
    ("c20001x9", 20001, 9, 3, 22, 0, 0, None),
    ("c8x32768", 8, 32768, 3, 23, 0, 0, None),
    ("g32768x40_cap50000", 32768, 40, 1, 24, 0, 50000, None),
    ("g16388x24_noise", 16388, 24, 1, 25, 1, 0, None),
    ("g20000x600", 20000, 600, 1, 26, 0, 0, None),
    ("c32764x12_px100000", 32764, 12, 3, 27, 0, 0, 100000),
]
# BASELINE.json configs[3]: ~3 minutes and ~10 GB of RAM with the reference; only with DWT_GOLDEN_HEAVY=1
HEAVY = [("c16384x16384_cap1MiB", 16384, 16384, 3, 0, 0, 1048576, None)]
KEEP_DWT_BELOW = 12000


def sha(b):
    return hashlib.sha256(b).hexdigest()


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all", "ref"], check=True)
    assert orc.have_ref(), "needs /root/reference to build oracle/_ref"
    out = {}
    with tempfile.TemporaryDirectory() as td:
        cases = CASES + (HEAVY if os.environ.get("DWT_GOLDEN_HEAVY") == "1" else [])
        old = {}
        if os.path.exists(os.path.join(HERE, "golden.json")):
            old = json.load(open(os.path.join(HERE, "golden.json")))
        only = set(sys.argv[1:])   # `make_golden.py name …`: (re)make only these records, keep the others as they are
        if only:
            assert only <= {c[0] for c in cases}, "unknown case name"
            out.update({k: v for k, v in old.items() if k not in only})
            cases = [c for c in cases if c[0] in only]
        for name in [h[0] for h in HEAVY]:
            if name in old and os.environ.get("DWT_GOLDEN_HEAVY") != "1":
                out[name] = old[name]   # keep the heavy record from an earlier run
        for name, W, H, C, seed, kind, cap, px in cases:
            src = os.path.join(td, "in.pnm")
            if W is None:
                pix = orc.read_pnm(os.path.join(HERE, "smpte.pnm"))
            else:
                pix = orc.synth(W, H, C, seed, kind)
            orc.write_pnm(src, pix)
            dwt, dec = os.path.join(td, "o.dwt"), os.path.join(td, "o.pnm")
            cmd = [os.path.join(orc.REF_DIR, "encode"), src, dwt] + ([str(cap)] if cap else [])
            r = subprocess.run(cmd, capture_output=True, check=True)
            data = open(dwt, "rb").read()
            cmd = [os.path.join(orc.REF_DIR, "decode"), dwt, dec] + ([str(px)] if px is not None else [])
            subprocess.run(cmd, capture_output=True, check=True)
            back = orc.read_pnm(dec)
            rec = dict(W=pix.shape[1], H=pix.shape[0], C=pix.shape[2], seed=seed, kind=kind, capacity=cap,
                       pixels_arg=px, input_sha256=sha(pix.tobytes()), dwt_len=len(data), dwt_sha256=sha(data),
                       encode_stderr=r.stderr.decode().splitlines(),
                       dec_W=back.shape[1], dec_H=back.shape[0], dec_sha256=sha(back.tobytes()),
                       lossless=bool(back.shape == pix.shape and (back == pix).all()))
            if name in [h[0] for h in HEAVY]:
                rec["heavy"] = True
            if len(data) <= KEEP_DWT_BELOW:
                open(os.path.join(HERE, name + ".dwt"), "wb").write(data)
                rec["dwt_file"] = name + ".dwt"
            out[name] = rec
            print(name, rec["dwt_len"], rec["dwt_sha256"][:16], rec["dec_W"], rec["dec_H"], rec["lossless"])
    json.dump(out, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
