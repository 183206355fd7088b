#!/usr/bin/env python3
"""Record what the REAL reference's decode prints on stderr (bytes.h:101 "reached end of file",
rle.h:45 "zeros not read") and returns, for the small .dwt fixtures: whole, cut short, and with
PIXELS arguments.  Run in the build container (needs /root/reference -> oracle/_ref); writes
tests/golden/decode_stderr.json.  The stream is always passed as the relative name "in.dwt" so
that the messages do not depend on where the test runs."""
import glob
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all", "ref"], check=True)
    assert orc.have_ref(), "needs /root/reference to build oracle/_ref"
    dec = os.path.join(orc.REF_DIR, "decode")
    out = []
    with tempfile.TemporaryDirectory() as td:
        for path in sorted(glob.glob(os.path.join(HERE, "*.dwt"))):
            data = open(path, "rb").read()
            cuts = sorted({len(data), len(data) * 2 // 3, len(data) // 3, 700, 150, 40, 7, 5} - {0})
            for cut in cuts:
                if cut > len(data):
                    continue
                for px in (None, 0, 300, 5000):
                    if cut != len(data) and px not in (None, 300):
                        continue
                    open(os.path.join(td, "in.dwt"), "wb").write(data[:cut])
                    cmd = [dec, "in.dwt", "out.pnm"] + ([str(px)] if px is not None else [])
                    r = subprocess.run(cmd, cwd=td, capture_output=True)
                    out.append(dict(fixture=os.path.basename(path), cut=cut, pixels_arg=px, returncode=r.returncode,
                                    stderr=r.stderr.decode()))
    json.dump(out, open(os.path.join(HERE, "decode_stderr.json"), "w"), indent=0)
    print(len(out), "records;", sum(1 for r in out if r["stderr"]), "with stderr text")


if __name__ == "__main__":
    main()
