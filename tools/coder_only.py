"""The entropy stage alone (what bench.py's `coder` record times): dwtx_encode_planes / dwtx_decode_planes on the
linearised coefficients of synthetic frames, `reps` times.  tools/coder_only.py [W H C frames reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dwt_amd
W, H, C, n, reps = (int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (4096, 4096, 1, 16, 2)
ctx = dwt_amd.Context(0)
if os.environ.get("DWTX_ONE_STREAM"):
    ctx.set_option("one_stream", 1)
pix = ctx.synth_pixels(n, H, W, C, 0, 0)
lin = ctx.linearization(ctx.transformation_fwd(ctx.planes_from_pixels(pix)))
stride = ctx.lib.dwtx_encode_bound(W, H, C)
streams = torch.empty((n, stride), dtype=torch.uint8, device="cuda")
info = torch.empty((n, ctypes.sizeof(dwt_amd.StreamInfo)), dtype=torch.uint8, device="cuda")
hinfo = (dwt_amd.DecodeInfo * n)()
back = torch.empty_like(lin)
for _ in range(reps):
    assert ctx.lib.dwtx_encode_planes(ctx.h, lin.data_ptr(), W, H, C, n, 0, streams.data_ptr(), stride, info.data_ptr()) == 0
    lens = ctx.stream_lengths(info)
    assert ctx.lib.dwtx_decode_planes(ctx.h, back.data_ptr(), streams.data_ptr(), stride, lens.data_ptr(), W, H, C, n, -1,
                                      ctypes.cast(hinfo, ctypes.c_void_p)) == 0
torch.cuda.synchronize()
assert torch.equal(back, lin)
print(f"{W}x{H}x{C} x{n}: {reps} x (encode_planes + decode_planes), coefficients round-trip")
