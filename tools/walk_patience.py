"""The one-family token walk's patience (k_tokenize: hand-parsed chunks in a row / in all before it gives up) against
the cost of the second walk, on a run of synthetic frames.  Development aid: needs the debug build
(make -C dwt_amd/csrc debug; copy build/libdwtx_debug.so over dwt_amd/libdwtx.so on the GPU box).
   python3 tools/walk_patience.py W H C first count"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dwt_amd
W, H, C, first, count = (int(a) for a in sys.argv[1:6])
ctx = dwt_amd.Context(0)
pix = ctx.synth_pixels(count, H, W, C, first, 0)
streams, info = ctx.encode_device(pix)
lens = ctx.stream_lengths(info)
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best, r
for streak, scans in ((48, 256), (128, 256), (512, 1024), (2048, 4096), (1 << 20, 1 << 20)):
    os.environ["DWTX_DBG_STREAK"], os.environ["DWTX_DBG_SCANS"] = str(streak), str(scans)
    ms, (out, infos) = timed(lambda: ctx.decode_device(streams, lens, W, H, C))
    assert torch.equal(out.view(count, H, W, C), pix)
    ctx.set_option("no_second_walk", 1)
    try:
        ctx.decode_device(streams, lens, W, H, C)
        second = "no"
    except dwt_amd.DwtxError:
        second = "YES"
    ctx.set_option("no_second_walk", 0)
    print(f"streak {streak} scans {scans}: decode {ms:.2f} ms, second walk {second}; hops {[i.hops for i in infos]} walked tokens {[i.walked_tokens for i in infos]}", flush=True)
