"""Which of the synthetic frames take the decoder's second (two-family) token walk?  (development aid)
   python3 tools/find_second_walk.py W H C first count"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dwt_amd
W, H, C, first, count = (int(a) for a in sys.argv[1:6])
ctx = dwt_amd.Context(0)
G = 8
found = []
for i0 in range(first, first + count, G):
    n = min(G, first + count - i0)
    pix = ctx.synth_pixels(max(n, 3), H, W, C, i0, 0)
    streams, info = ctx.encode_device(pix)
    lens = ctx.stream_lengths(info)
    out, infos = ctx.decode_device(streams, lens, W, H, C)
    assert torch.equal(out.view(pix.shape[0], H, W, C), pix)
    ctx.set_option("no_second_walk", 1)
    try:
        out, infos = ctx.decode_device(streams, lens, W, H, C)
        print(f"frames {i0}..{i0 + n - 1}: first walk; hops {[i.hops for i in infos[:n]]}", flush=True)
    except dwt_amd.DwtxError as e:
        for j in range(n):   # the frame beside two copies of another one (one or two frames alone start with both families)
            k = (j + 1) % pix.shape[0]
            sel = torch.tensor([j, k, k], device=pix.device)
            try:
                ctx.decode_device(streams[sel].contiguous(), lens[sel].contiguous(), W, H, C)
            except dwt_amd.DwtxError:
                found.append(i0 + j)
        print(f"frames {i0}..{i0 + n - 1}: second walk for {[f for f in found if f >= i0]}", flush=True)
    ctx.set_option("no_second_walk", 0)
print("frames that take the second walk:", found)
