"""Walker statistics of one decode (development aid): hops, hopped chunks, tokens the walker parsed itself."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dwt_amd
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else W
C = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ctx = dwt_amd.Context(0)
pix = ctx.synth_pixels(n, H, W, C, 0, 0)
streams, info = ctx.encode_device(pix)
lens = ctx.stream_lengths(info)
out, infos = ctx.decode_device(streams, lens, W, H, C)
for i in infos[:4]:
    print("hops", i.hops, "hopped_chunks", i.hopped_chunks, "walked_tokens", i.walked_tokens, "bits", i.bits_used, "chunks", i.bits_used // 128, "nsegs", i.nsegs)
