"""Cycle counters of the token walker (development aid; needs a build with -DDWTX_DEBUG_HOOKS)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else W
C = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dbg = torch.zeros((n, 8), dtype=torch.int64, device="cuda:0")
os.environ["DWTX_DBG_PTR"] = str(dbg.data_ptr())
import dwt_amd
ctx = dwt_amd.Context(0)
pix = ctx.synth_pixels(n, H, W, C, 0, 0)
streams, info = ctx.encode_device(pix)
lens = ctx.stream_lengths(info)
out, infos = ctx.decode_device(streams, lens, W, H, C)
torch.cuda.synchronize()
d = dbg.cpu()
for k, i in enumerate(infos[:4]):
    # shader clock cycles (s_memtime)
    print("hops", i.hops, "hopped_chunks", i.hopped_chunks, "walked_tokens", i.walked_tokens, "nsegs", i.nsegs,
          "cycles all", d[k, 0].item(), "hops", d[k, 1].item(), "chunks by hand", d[k, 3].item(), ": load", d[k, 4].item(), "chunk_scan", d[k, 5].item(),
          "records", d[k, 2].item(), "; tokens read bit by bit", d[k, 7].item(), ": cycles", d[k, 6].item())
