"""Quick in-process timing of forward+inverse lifting (development aid, not the bench)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dwt_amd

W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = dwt_amd.Context(0)
for name in dwt_amd._lib.OPTIONS:   # DWTX_NO_FUSED_LEVELS=1 etc. are this tool's switches
    if os.environ.get("DWTX_" + name.upper()):
        ctx.set_option(name, int(os.environ["DWTX_" + name.upper()]))
x = torch.randint(0, 256, (P, H, W), dtype=torch.int32, device="cuda")
pyr = torch.empty_like(x)
back = torch.empty_like(x)
for _ in range(3):
    ctx.transformation_fwd(x, pyr); ctx.transformation_inv(pyr, back)
torch.cuda.synchronize()
if not os.environ.get("DWTX_PART_IMAGES"): assert torch.equal(back, x)
for name, fn in (("fwd", lambda: ctx.transformation_fwd(x, pyr)), ("inv", lambda: ctx.transformation_inv(pyr, back)),
                 ("fwd+inv", lambda: (ctx.transformation_fwd(x, pyr), ctx.transformation_inv(pyr, back)))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    samples = P * H * W
    nb = (8 if name != "fwd+inv" else 16) * samples
    print(f"{name}: {ms*1e3/P:.1f} us/plane  {nb/ms/1e6:.1f} GB/s algorithmic  ({samples/ms/1e3:.1f} Mpx/s)")
