"""in-process A/B of lifting variants on the same buffers (development aid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dwt_amd
W = H = 4096; P = 64
ctx = dwt_amd.Context(0)
x = torch.randint(0, 256, (P, H, W), dtype=torch.int32, device="cuda")
pyr = torch.empty_like(x); back = torch.empty_like(x)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3 / P
variants = [("one launch per level", dict(no_fused_levels=1)), ("two levels per pass", dict(no_fused_levels=0))]
for rnd in range(2):
    for name, o in variants:
        for k, v in o.items(): ctx.set_option(k, v)
        print(f"round {rnd} {name:22s} fwd {t(lambda: ctx.transformation_fwd(x, pyr)):6.2f} us/plane   inv {t(lambda: ctx.transformation_inv(pyr, back)):6.2f}")
