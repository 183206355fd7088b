"""Does the lifting time depend on where its buffers lie relative to each other?  (development aid)
One big allocation; source planes at its start, the pyramid `gap` bytes after the source's end."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dwt_amd

W = H = 4096
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = dwt_amd.Context(0)
n = P * H * W
big = torch.empty(5 * n + (64 << 20), dtype=torch.int32, device="cuda")
src = big[:n].view(P, H, W)
src.copy_(torch.randint(0, 256, (P, H, W), dtype=torch.int32, device="cuda"))
for gap in (0, 4096, 1 << 20, 16 << 20, 128 << 20, 512 << 20, (512 << 20) + (3 << 20), 1 << 30, (1 << 30) + (640 << 20), 2 << 30, (3 << 30) + (5 << 20)):
    g = gap // 4
    pyr = big[n + g:2 * n + g].view(P, H, W)
    back = big[4 * n:5 * n].view(P, H, W)
    res = []
    for name, fn in (("fwd", lambda: ctx.transformation_fwd(src, pyr)), ("inv", lambda: ctx.transformation_inv(pyr, back))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10 * 1e3 / P)
    print(f"gap {gap:>9d} B: fwd {res[0]:.1f} us/plane  inv {res[1]:.1f} us/plane   base {src.data_ptr() % (1<<21):#x} {pyr.data_ptr() % (1<<21):#x}")
