"""Longer seeded sweep of the decoder against the oracle (development aid, GPU box; not part of the test suite):
random geometries and contents, whole streams, cuts, damaged streams, in batches of mixed kinds (>= 3 streams, so
that the one-family walk and both halves of the batch run), with and without sidecar indices (right ones, and
the indices of other streams).   tools/fuzz_decode.py [seed] [cases] [big]   (big: batches of 30-45 streams, which the decoder
cuts into four parts instead of two)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
import dwt_amd
from test_oracle import corrupted_blobs

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
big = len(sys.argv) > 3
rng = np.random.default_rng(seed)
ctx = dwt_amd.Context(0)
t0 = time.time()
checked = 0
refused = set()
for case in range(cases):
    W, H = int(rng.integers(8, 900)), int(rng.integers(8, 900))
    if case % 4 == 0:
        W = H = int(2 ** rng.integers(3, 11))
    Cn = 1 if rng.integers(0, 2) else 3
    n = int(rng.integers(18, 30)) if big else int(rng.integers(3, 9))
    good = [orc.encode(orc.synth(W, H, Cn, int(rng.integers(0, 1 << 30)), int(rng.integers(0, 2))))[0] for _ in range(n)]
    blobs = list(good)
    for g in good[:3]:
        blobs.append(g[: int(rng.integers(1, len(g)))])
        blobs += corrupted_blobs(g, 3, int(rng.integers(0, 1 << 30)))
    order = rng.permutation(len(blobs))
    blobs = [blobs[i] for i in order]
    refs = [orc.decode_stage(b, W, H, Cn, -1) for b in blobs]

    def check(tag):
        lin, infos = ctx.decode_planes(blobs, W, H, Cn)
        got = lin.cpu().numpy().reshape(len(blobs), Cn, W * H)
        for i, ref in enumerate(refs):
            if ref is None:
                assert infos[i].status == 1, (tag, case, i)
                continue
            rlin, level, missing, planes = ref
            if max(planes) > 16:   # the documented difference: refused with status 2
                assert infos[i].status == 2, (tag, case, i)
                refused.add((case, i))
                continue
            assert infos[i].status == 0 and infos[i].level == level and list(infos[i].missing) == missing.tolist(), (tag, case, i, W, H, Cn)
            assert (got[i] == rlin).all(), (tag, case, i, W, H, Cn)
        return infos

    made = ctx.set_index(None, len(blobs))
    check("plain")
    ctx.set_index(made, 0)                       # every stream that has one is offered its own
    check("index")
    rolled = (dwt_amd.Index * len(blobs))(*[made[(i + 1) % len(blobs)] for i in range(len(blobs))])
    ctx.set_index(rolled, 0)                     # ... and then somebody else's
    check("foreign index")
    ctx.set_index()
    checked += len(blobs)
print(f"seed {seed}: {cases} cases, {checked} streams x 3 decodes equal the oracle ({len(refused)} of them claim more than 16 bit planes "
      f"and were refused with status 2), {time.time() - t0:.0f} s")
