import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
os.environ["DWTX_DBG_PTR"] = str(dbg.data_ptr())
import dwt_amd
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = dwt_amd.Context(0)
pix = ctx.synth_pixels(1, H, W, 1, 0, 0)
streams, info = ctx.encode_device(pix)
lens = ctx.stream_lengths(info)
for _ in range(3):
    out, infos = ctx.decode_device(streams, lens, W, H, 1)
torch.cuda.synchronize()
d = dbg.cpu().tolist()
i = infos[0]
print(f"walker cycles total {d[0]} (~{d[0]/100e6*1e3:.2f} ms at 100MHz refclk?)  hop-check {d[1]}  fast-parse {d[2]}  hops={i.hops} walked={i.walked_tokens} nfast={d[3]}")
