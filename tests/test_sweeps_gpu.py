"""GPU: a short run of the two seeded sweeps under tools/ (whole codec on nine kinds of pictures; mixed batches of
whole, cut and damaged streams decoded plainly, with their own and with foreign sidecar indices).  The long runs
are a development aid; this keeps the scripts working and adds a few hundred cases the other tests do not have."""
import os
import subprocess
import sys

import pytest

import orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tool,args", [("fuzz_codec.py", ["7", "14"]), ("fuzz_decode.py", ["7", "14"]), ("fuzz_decode.py", ["8", "5", "big"])])
def test_seeded_sweep(tool, args):
    r = subprocess.run([sys.executable, os.path.join(orc.ROOT, "tools", tool)] + args, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr[-600:]
    assert b"equal the oracle" in r.stdout
