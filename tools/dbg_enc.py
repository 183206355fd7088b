import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dwt_amd, ctypes
W = int(sys.argv[1]); H = int(sys.argv[2]); C = int(sys.argv[3]); n = int(sys.argv[4])
ctx = dwt_amd.Context(0)
pix = ctx.synth_pixels(n, H, W, C, 0, int(sys.argv[5]) if len(sys.argv) > 5 else 0)
streams, info = ctx.encode_device(pix)
raw = info.cpu().numpy()
infos = [dwt_amd.StreamInfo.from_buffer_copy(raw[i].tobytes()) for i in range(n)]
print("exact_orders flags:", [i.exact_orders for i in infos], "tokens:", infos[0].tokens, "segments", infos[0].segments)
