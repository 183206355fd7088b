"""Development aid (needs the -DDWTX_DEBUG_HOOKS library in place of dwt_amd/libdwtx.so: `make -C dwt_amd/csrc debug`, then copy
build/libdwtx_debug.so over it on the GPU box): the tiles' histograms and the pyramid that dwtx_transformation_fwd_pixels leaves
with two levels per pass against one launch per level — the entropy stage sizes its buffers from those histograms, so they
are compared here BEFORE any stream is coded with a new transform kernel."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import dwt_amd, orc

ctx = dwt_amd.Context(0)
fn = ctx.lib.dwtx_debug_hist_copy
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]


def hist(W, H, Cn, n):
    NT, NTP, tf = C.c_int(), C.c_int(), (C.c_int * 20)()
    assert fn(ctx.h, W, H, Cn, n, None, None, C.byref(NT), C.byref(NTP), tf) == 0
    cum = np.zeros((n * Cn, NT.value, 16), dtype=np.uint32)
    mx = np.zeros((n * Cn, NTP.value), dtype=np.uint32)
    assert fn(ctx.h, W, H, Cn, n, cum.ctypes.data, mx.ctypes.data, C.byref(NT), C.byref(NTP), tf) == 0
    return cum, mx, list(tf)


shapes = [(640, 360, 3), (4096, 96, 1), (512, 512, 1), (1920, 1080, 3), (260, 516, 1), (1028, 260, 3), (2048, 2048, 1), (448, 132, 1), (452, 136, 3), (224, 128, 1), (4096, 4096, 1), (3584, 512, 3)]
for W, H, Cn in shapes:
    g = orc.geometry(W, H)
    pix = torch.from_numpy(np.stack([orc.synth(W, H, Cn, 3, 0), orc.synth(W, H, Cn, 4, 1)])).cuda()
    res = []
    for off in (0, 1):
        ctx.set_option("no_fused_levels", off)
        pyr, r16, mask = ctx.transformation_fwd_pixels(pix)
        torch.cuda.synchronize()
        cum, mx, tf = hist(W, H, Cn, 2)
        res.append((pyr.cpu().numpy(), r16.cpu().numpy(), mask, cum, mx))
    ctx.set_option("no_fused_levels", 0)
    T = g.levels
    ok = True
    for l in (T - 1, T - 2):
        a, b = tf[l], tf[l + 1]
        same = (res[0][3][:, a:b] == res[1][3][:, a:b]).all() and (res[0][4][:, a:b] == res[1][4][:, a:b]).all()
        nz = int((res[1][3][:, a:b] != 0).sum())
        print(f"{W}x{H}x{Cn} level {l}: tiles {a}..{b}, histograms equal: {bool(same)} (non-zero words one launch per level: {nz})")
        ok = ok and same
        if not same:
            A, B = res[0][3][:, a:b].astype(np.int64), res[1][3][:, a:b].astype(np.int64)
            lo = lambda x: (x & 0xffff).sum(axis=(1,)), 
            print("      totals of word 7 high half (all coefficients) per plane, fused:", ((A[:, :, 7] >> 16).sum(axis=1)).tolist(), "one per level:", ((B[:, :, 7] >> 16).sum(axis=1)).tolist())
            d = np.argwhere((A != B).any(axis=2))
            print("      differing (plane, tile) pairs:", len(d), "first:", d[:8].tolist())
            for pl, t in d[:3]:
                print("      plane", pl, "tile", a + t, "fused", (A[pl, t] >> 16).tolist(), (A[pl, t] & 0xffff).tolist(), "| one per level", (B[pl, t] >> 16).tolist(), (B[pl, t] & 0xffff).tolist())
            print("      mx differs at", np.argwhere(res[0][4][:, a:b] != res[1][4][:, a:b])[:6].tolist())
    same_pyr = res[0][2] == res[1][2] and (res[0][1] == res[1][1]).all() and (res[0][0] == res[1][0]).all()
    # the int32 pyramid only matters outside the 16-bit rings; compare what both wrote: the coarse part
    w3, h3 = g.widths[max(T - 3, 0)], g.heights[max(T - 3, 0)]
    print(f"   16-bit rings equal: {bool(same_pyr)}, mask {res[0][2]:#x}")
    if not (ok and same_pyr):
        print('MISMATCH', W, H, Cn)
print("ok")
