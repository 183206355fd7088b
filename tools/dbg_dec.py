import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dwt_amd
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = dwt_amd.Context(0)
pix = ctx.synth_pixels(n, H, W, C, 0, 0)
streams, info = ctx.encode_device(pix)
lens = ctx.stream_lengths(info)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    out, infos = ctx.decode_device(streams, lens, W, H, C)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    i = infos[0]
    print(f"decode {1e3*(t1-t0):.1f} ms  len={int(lens[0])}  segs={i.nsegs} hops={i.hops} hopped_chunks={i.hopped_chunks} "
          f"of {int(lens[0])*8//128} walked_tokens={i.walked_tokens} lossless={torch.equal(out.view(n,H,W,C), pix)}")
