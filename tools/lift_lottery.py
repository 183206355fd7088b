"""Does the forward lifting time change between allocations inside one process?  (development aid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dwt_amd

P, H, W = 64, 4096, 4096
ctx = dwt_amd.Context(0)
keep = []
for attempt in range(8):
    torch.cuda.empty_cache()
    x = torch.randint(0, 256, (P, H, W), dtype=torch.int32, device="cuda")
    pyr = torch.empty_like(x)
    back = torch.empty_like(x)
    res = []
    for fn in (lambda: ctx.transformation_fwd(x, pyr), lambda: ctx.transformation_inv(pyr, back), lambda: back.copy_(x)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10 * 1e3 / P)
    print(f"attempt {attempt}: fwd {res[0]:.1f} inv {res[1]:.1f} plain copy {res[2]:.1f} us/plane  x@{x.data_ptr():#x} pyr@{pyr.data_ptr():#x}")
    del x, pyr, back
    if attempt % 2 == 1:
        keep.append(torch.empty(int(1e9) + attempt * 12345678, dtype=torch.uint8, device="cuda"))   # shifts the next addresses
