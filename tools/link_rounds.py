"""Relaxation rounds of the decoder's chunk tables (LINK_ROUNDS) against the walker's hops and the decode time.
Development aid: needs the debug build (make -C dwt_amd/csrc debug; build/libdwtx_debug.so copied over dwt_amd/libdwtx.so on the GPU box).
   python3 tools/link_rounds.py W H C n"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dwt_amd
W, H, C, n = (int(a) for a in sys.argv[1:5])
ctx = dwt_amd.Context(0)
pix = ctx.synth_pixels(n, H, W, C, 0, 0)
streams, info = ctx.encode_device(pix)
lens = ctx.stream_lengths(info)
def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best, r
for rounds in (int(a) for a in (sys.argv[5] if len(sys.argv) > 5 else "12,10,8,6,4").split(",")):
    os.environ["DWTX_DBG_ROUNDS"] = str(rounds)
    ms, (out, infos) = timed(lambda: ctx.decode_device(streams, lens, W, H, C))
    assert torch.equal(out.view(n, H, W, C), pix)
    ms1, _ = timed(lambda: ctx.decode_device(streams[:1], lens[:1], W, H, C))
    hops = [i.hops for i in infos]
    print(f"rounds {rounds}: decode {ms:.2f} ms for {n} frames, {ms1:.3f} ms for one; hops mean {sum(hops) / len(hops):.0f} max {max(hops)}; walked tokens mean {sum(i.walked_tokens for i in infos) / len(infos):.0f}", flush=True)
