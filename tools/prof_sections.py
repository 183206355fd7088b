"""Experiment builds only (-DDWTX_PROF_SECTIONS): cycles per section of k_code.  tools/prof_sections.py W H C n"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dwt_amd
W, H, C, n = (int(v) for v in sys.argv[1:5])
ctx = dwt_amd.Context(0)
ctx.set_option("one_stream", 1)
pix = ctx.synth_pixels(n, H, W, C, 0, 0)
streams, info = ctx.encode_device(pix)
torch.cuda.synchronize()
lib = ctypes.CDLL(dwt_amd.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
lib.dwtx_debug_prof(None, 1)
ctx.encode_device(pix, out=streams, info=info)
torch.cuda.synchronize()
lib.dwtx_debug_prof(buf, 0)
tiles = n * C * (W * H // 1024)
names = {1: "pass A + scans", 2: "table", 3: "pass B", 4: "tokens in place", 5: "tokens out", 6: "rows zeroed", 7: "strings deposited", 8: "rows out"}
tot = sum(buf[1:9])
for k in range(1, 9):
    print(f"{names[k]:20s} {buf[k] / tiles:9.0f} cycles per tile  {100.0 * buf[k] / tot:5.1f} %")
print(f"{'sum':20s} {tot / tiles:9.0f}")
