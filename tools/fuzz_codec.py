"""Longer seeded sweep of the whole codec against the oracle (development aid, GPU box; not part of the test suite):
random geometries (incl. power-of-two squares and widths that take the fused 8-bit paths), batches, capacities,
PIXELS caps — stream bytes, statistics and decoded pictures.   tools/fuzz_codec.py [seed] [cases]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
import dwt_amd

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
ctx = dwt_amd.Context(0)
t0 = time.time()
frames = 0
for case in range(cases):
    W, H = int(rng.integers(8, 700)), int(rng.integers(8, 700))
    kind = case % 6
    if kind == 0:
        W = H = int(2 ** rng.integers(3, 11))
    elif kind == 1:
        W = (W + 3) // 4 * 4
    elif kind == 5:   # one long side, up to the largest the reference's arithmetic is defined for
        long_side, short_side = int(rng.integers(16385, 32769)), int(rng.integers(8, 48))
        W, H = (long_side, short_side) if rng.integers(0, 2) else (short_side, long_side)
    Cn = 1 if rng.integers(0, 2) else 3
    n = int(rng.integers(1, 7))
    def picture():
        what = int(rng.integers(0, 9))
        if what <= 2:
            return orc.synth(W, H, Cn, int(rng.integers(0, 1 << 30)), int(rng.integers(0, 2)))
        y, x = np.mgrid[0:H, 0:W]
        if what == 3:     # flat
            img = np.full((H, W, Cn), int(rng.integers(0, 256)))
        elif what == 4:   # checkerboard of two levels, period 1..8
            per = int(rng.integers(1, 9))
            a, b = int(rng.integers(0, 256)), int(rng.integers(0, 256))
            img = np.where((((x // per) + (y // per)) & 1)[..., None] == 0, a, b) * np.ones((1, 1, Cn), dtype=np.int64)
        elif what == 5:   # white noise over the full range
            img = rng.integers(0, 256, (H, W, Cn))
        elif what == 6:   # a few impulses on black
            img = np.zeros((H, W, Cn), dtype=np.int64)
            k = int(rng.integers(1, 30))
            img[rng.integers(0, H, k), rng.integers(0, W, k)] = rng.integers(1, 256, (k, Cn))
        elif what == 7:   # ramps
            img = ((x * int(rng.integers(1, 5)) + y * int(rng.integers(0, 5))) // int(rng.integers(1, 9)))[..., None] + np.arange(Cn) * 40
        else:             # bars with hard edges plus one noisy channel
            img = ((x * 8 // W) * 36)[..., None] + np.zeros((1, 1, Cn), dtype=np.int64)
            img[..., Cn - 1] += rng.integers(0, 3, (H, W))
        return np.ascontiguousarray(np.clip(img, 0, 255).astype(np.uint8).reshape(H, W, Cn))

    pix = np.stack([picture() for _ in range(n)])
    want = [orc.encode(p) for p in pix]
    streams, stats = ctx.encode(pix)
    for i in range(n):
        assert streams[i] == want[i][0], ("bytes", case, i, W, H, Cn)
        assert (stats[i].meta_bits, stats[i].root_bits, stats[i].total_bits) == (want[i][1].meta_bits, want[i][1].root_bits, want[i][1].total_bits), ("stats", case, i)
    outs = ctx.decode(streams)
    for i in range(n):   # (not always the input: a flat picture comes back at half size from the reference too)
        ref = orc.decode(streams[i])
        assert outs[i].shape == ref.shape and (outs[i] == ref).all(), ("whole stream", case, i, W, H, Cn)
    cap = int(rng.integers(7, max(8, len(streams[0]))))
    cut, cstats = ctx.encode(pix, cap)
    for i in range(n):
        w2, s2 = orc.encode(pix[i], cap)
        assert cut[i] == w2, ("capacity bytes", case, i, W, H, Cn, cap)
        assert (cstats[i].meta_bits, cstats[i].root_bits, cstats[i].total_bits) == (s2.meta_bits, s2.root_bits, s2.total_bits), ("capacity stats", case, i, cap)
    px = int(rng.integers(0, 3 * W * H))
    for blobs, cap_px in ((cut, -1), (streams, px), ([s[: max(6, len(s) // 3)] for s in streams], -1)):   # (6: a batch's geometry is its first stream's header)
        got = ctx.decode(list(blobs), cap_px)
        for i, b in enumerate(blobs):
            ref = orc.decode(b, cap_px)
            assert (ref is None and got[i] is None) or (got[i] is not None and got[i].shape == ref.shape and (got[i] == ref).all()), ("decode", case, i, W, H, Cn, cap, cap_px)
    frames += n
print(f"seed {seed}: {cases} cases, {frames} frames: bytes, statistics and pictures equal the oracle, {time.time() - t0:.0f} s")
