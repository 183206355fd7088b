"""Decode-only timing without checks (kernel experiments; a broken experiment may give up its walks): tools/time_decode.py W H C n"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dwt_amd
W, H, C, n = (int(v) for v in sys.argv[1:5])
ctx = dwt_amd.Context(0)
if os.environ.get("DWTX_ONE_STREAM"):
    ctx.set_option("one_stream", 1)
if os.environ.get("DWTX_DECODE_PARTS"):
    ctx.set_option("decode_parts", int(os.environ["DWTX_DECODE_PARTS"]))
pix = ctx.synth_pixels(n, H, W, C, 0, 0)
streams, info = ctx.encode_device(pix)
lens = ctx.stream_lengths(info)
out, _ = ctx.decode_device(streams, lens, W, H, C)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    ctx.decode_device(streams, lens, W, H, C, out=out)
b.record(); torch.cuda.synchronize()
print(f"{W}x{H}x{C} x{n}: decode {a.elapsed_time(b) / 5:.3f} ms")
