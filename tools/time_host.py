"""PCIe-inclusive timing of the host-buffer entry points (dwtx_encode_images / dwtx_decode_images) with
page-locked or pageable host buffers: python tools/time_host.py [W H C frames pinned(0/1)]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import dwt_amd

W, H, Cn, n = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (4096, 4096, 1, 64)))
pinned = int(sys.argv[5]) if len(sys.argv) > 5 else 1
ctx = dwt_amd.Context(0)
lib = ctx.lib


def host_array(nbytes):
    if not pinned:
        return np.empty(nbytes, dtype=np.uint8)
    p = lib.dwtx_host_alloc(ctx.h, nbytes)
    assert p
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))


pix_dev = ctx.synth_pixels(n, H, W, Cn, seed0=0, kind=0)
img = W * H * Cn
pix = host_array(n * img)
pix[:] = pix_dev.cpu().numpy().reshape(-1)
stride = ((img * 6) // 10 + 4096 + 7) // 8 * 8      # lossless smooth+noise frames need < 0.5 byte per sample
out = host_array(n * stride)
back = host_array(n * img)
lens = (C.c_size_t * n)()
ow, oh, oc = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)()


def enc():
    rc = lib.dwtx_encode_images(ctx.h, pix.ctypes.data, W, H, Cn, n, 0, out.ctypes.data, stride, C.cast(lens, C.c_void_p), None)
    assert rc == 0, (rc, dwt_amd._lib.last_error())


def dec():
    rc = lib.dwtx_decode_images(ctx.h, out.ctypes.data, stride, C.cast(lens, C.c_void_p), n, -1, back.ctypes.data, img, ow, oh, oc)
    assert rc == 0, (rc, dwt_amd._lib.last_error())


enc(); dec()
assert (back == pix).all(), "round trip not lossless"
t0 = time.perf_counter(); enc(); t1 = time.perf_counter(); dec(); t2 = time.perf_counter()
px = n * W * H
mb_in, mb_out = n * img / 1e6, sum(lens) / 1e6
print(f"{W}x{H}x{Cn} x{n} {'pinned' if pinned else 'pageable'} host buffers: encode {1e3*(t1-t0):.1f} ms ({px/(t1-t0)/1e6:.0f} Mpx/s, "
      f"{mb_in/(t1-t0)/1e3:.1f} GB/s in)  decode {1e3*(t2-t1):.1f} ms ({px/(t2-t1)/1e6:.0f} Mpx/s)  "
      f"round trip {px/(t2-t0)/1e6:.0f} Mpx/s  streams {mb_out/n:.2f} MB/frame")
