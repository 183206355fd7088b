import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, dwt_amd, orc
ctx = dwt_amd.Context(0)
rng = np.random.default_rng(0)
for (H, W) in [(64, 512), (256, 512), (512, 1024), (1024, 1024), (2048, 256)]:
    a = rng.integers(-50, 50, size=(1, H, W), dtype=np.int32)
    want = orc.forward(a[0][:, :, None])[:, :, 0]
    got = ctx.transformation_fwd(torch.from_numpy(a).cuda()).cpu().numpy()[0]
    bad = np.argwhere(got != want)
    print((H, W), "mismatches:", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:12], "cols", sorted(set(bad[:, 1].tolist()))[:12])
