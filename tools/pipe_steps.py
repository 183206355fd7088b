"""experiment: encode of step k+1 beside decode of step k (two contexts on two streams)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dwt_amd, ctypes
W = H = 4096; C = 1; B = 64
dev = torch.device("cuda", 0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s1):
    cE = dwt_amd.Context(0)
with torch.cuda.stream(s2):
    cD = dwt_amd.Context(0)
pix = cE.synth_pixels(B, H, W, C, 0, 0)
stride = cE.lib.dwtx_encode_bound(W, H, C)
outs = [torch.empty((B, stride), dtype=torch.uint8, device=dev) for _ in range(2)]
infos = [torch.empty((B, ctypes.sizeof(dwt_amd.StreamInfo)), dtype=torch.uint8, device=dev) for _ in range(2)]
dec = torch.empty((B, W * H * C), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
def serial(steps):
    for k in range(steps):
        with torch.cuda.stream(s1):
            st, inf = cE.encode_device(pix, out=outs[0], info=infos[0])
            lens = cE.stream_lengths(inf)
            cE.decode_device(st, lens, W, H, C, out=dec)
def piped(steps):
    ev_enc = [torch.cuda.Event() for _ in range(2)]
    ev_dec = [torch.cuda.Event() for _ in range(2)]
    for k in range(steps + 1):
        if k < steps:
            with torch.cuda.stream(s1):
                if k >= 2:
                    s1.wait_event(ev_dec[k % 2])      # the decode that read this slot is over
                cE.encode_device(pix, out=outs[k % 2], info=infos[k % 2])
                ev_enc[k % 2].record(s1)
        if k >= 1:
            j = (k - 1) % 2
            with torch.cuda.stream(s2):
                s2.wait_event(ev_enc[j])
                lens = cD.stream_lengths(infos[j])
                cD.decode_device(outs[j], lens, W, H, C, out=dec)
                ev_dec[j].record(s2)
for name, fn in (("serial", serial), ("pipelined", piped), ("serial", serial), ("pipelined", piped)):
    fn(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(20); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"{name}: {t / 20 * 1e3:.2f} ms per step, {20 * B * W * H / t / 1e6:.0f} Mpx/s", bool(torch.equal(dec.view(B, H, W, C), pix)))
