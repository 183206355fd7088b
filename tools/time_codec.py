"""Stage timing of encode/decode on device-resident frames (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dwt_amd
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else W
C = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = int(sys.argv[4]) if len(sys.argv) > 4 else 8
ctx = dwt_amd.Context(0)
for name in dwt_amd._lib.OPTIONS:   # the library reads no environment: DWTX_ONE_STREAM=1 etc. are this tool's switches
    if os.environ.get("DWTX_" + name.upper()):
        ctx.set_option(name, int(os.environ["DWTX_" + name.upper()]))
pix = ctx.synth_pixels(n, H, W, C, 0, 0)
streams, info = ctx.encode_device(pix)
lens = ctx.stream_lengths(info)
if os.environ.get("TIGHT_STRIDE"):   # rows as wide as the streams need (x1.5) instead of the worst-case bound: the decoder's tables follow the stride
    stride = (int(lens.max().item()) * 3 // 2 + 64 + 7) // 8 * 8
    streams = torch.empty((n, stride), dtype=torch.uint8, device=pix.device)
    streams, info = ctx.encode_device(pix, out=streams, info=info)
    lens = ctx.stream_lengths(info)
    ctx.close(); ctx = dwt_amd.Context(0)   # (scratch sized for the worst-case stride goes)
    for name in dwt_amd._lib.OPTIONS:
        if os.environ.get("DWTX_" + name.upper()):
            ctx.set_option(name, int(os.environ["DWTX_" + name.upper()]))
out, infos = ctx.decode_device(streams, lens, W, H, C)
torch.cuda.synchronize()
assert torch.equal(out.view(n, H, W, C), pix)
def timed(fn, reps=5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
te = timed(lambda: ctx.encode_device(pix, out=streams, info=info))
td = timed(lambda: ctx.decode_device(streams, lens, W, H, C, out=out))
mp = n * W * H / 1e6
if os.environ.get("TIME_INDEX"):   # the same decode with the sidecar index of a first decode (include/dwtx.h dwtx_index)
    made = ctx.set_index(None, n)
    ctx.decode_device(streams, lens, W, H, C, out=out)
    torch.cuda.synchronize()
    ctx.set_index(made, 0)
    ctx.decode_device(streams, lens, W, H, C, out=out)
    torch.cuda.synchronize()
    assert torch.equal(out.view(n, H, W, C), pix)
    ti = timed(lambda: ctx.decode_device(streams, lens, W, H, C, out=out))
    ctx.set_index()
    print(f"{W}x{H}x{C} x{n}: decode with index {ti:.2f} ms ({n * W * H / 1e6 / ti * 1e3:.0f} Mpx/s) against {td:.2f} ms without")
print(f"{W}x{H}x{C} x{n}: encode {te:.2f} ms ({mp/te*1e3:.0f} Mpx/s)  decode {td:.2f} ms ({mp/td*1e3:.0f} Mpx/s)  round trip {mp/(te+td)*1e3:.0f} Mpx/s  bytes/frame {int(lens.sum())//n}")
