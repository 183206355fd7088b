"""BASELINE.json configs[3]: one 16384x16384 RGB frame encoded with CAPACITY = 1 MiB, timed with the capacity cut
(pack.hip k_cut: segments that start beyond the capacity are not coded) and with the cut switched off (all segments
coded, the stream clipped at the end: what round 2 did); the decode of the 1 MiB stream; golden hashes.
tools/time_capacity.py [W H C capacity]"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, dwt_amd
W, H, C, cap = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (16384, 16384, 3, 1 << 20)
ctx = dwt_amd.Context(0)
pix = ctx.synth_pixels(1, H, W, C, 0, 0)
stride = (cap + 15) // 8 * 8
streams = torch.empty((1, stride), dtype=torch.uint8, device="cuda")
info = torch.empty((1, 80), dtype=torch.uint8, device="cuda")

def timed(fn, reps=3):
    best = 1e30
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best

res = {}
for name, off in (("cut", 0), ("no_cut", 1)):
    ctx.set_option("no_capacity_cut", off)
    ctx.encode_device(pix, capacity=cap, out=streams, info=info)
    torch.cuda.synchronize()
    rec = dwt_amd.StreamInfo.from_buffer_copy(info[0].cpu().numpy().tobytes())
    data = streams[0, : rec.nbytes].cpu().numpy().tobytes()
    res[name] = {"encode_ms": round(timed(lambda: ctx.encode_device(pix, capacity=cap, out=streams, info=info)), 3), "bytes": rec.nbytes,
                 "segments_coded": rec.segments, "segments_cut": rec.segments_cut, "sha256": hashlib.sha256(data).hexdigest()}
ctx.set_option("no_capacity_cut", 0)
lens = ctx.stream_lengths(info)
out = torch.empty((1, W * H * C), dtype=torch.uint8, device="cuda")
ctx.decode_device(streams, lens, W, H, C, out=out)
res["decode_ms"] = round(timed(lambda: ctx.decode_device(streams, lens, W, H, C, out=out)), 3)
gj = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
g = next((v for v in gj.values() if (v.get("W"), v.get("H"), v.get("C"), v.get("capacity")) == (W, H, C, cap) and v.get("seed") == 0), None)
if g:
    res["matches_reference_golden"] = res["cut"]["sha256"] == g["dwt_sha256"] == res["no_cut"]["sha256"]
res["speedup_from_cut"] = round(res["no_cut"]["encode_ms"] / res["cut"]["encode_ms"], 2)
print(json.dumps(res))
