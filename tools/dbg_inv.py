import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, dwt_amd, orc
ctx = dwt_amd.Context(0)
rng = np.random.default_rng(0)
for (H, W) in [(15, 15), (16, 16), (15, 16), (16, 15), (17, 17), (31, 31), (8, 8), (9, 9), (8, 17), (17, 8)]:
    a = rng.integers(-50, 50, size=(1, H, W), dtype=np.int32)
    want = orc.inverse(a[0][:, :, None])[:, :, 0]
    got = ctx.transformation_inv(torch.from_numpy(a).cuda()).cpu().numpy()[0]
    bad = np.argwhere(got != want)
    fw = orc.forward(a[0][:, :, None])[:, :, 0]
    fg = ctx.transformation_fwd(torch.from_numpy(a).cuda()).cpu().numpy()[0]
    print((H, W), "inv mismatches:", len(bad), bad[:12].tolist(), "fwd mismatches:", int((fw != fg).sum()))
