import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sd_mecha_amd as amd
torch.manual_seed(0)
for n, p in ((24, 80), (24, 64), (16, 80), (32, 80), (24, 96), (8, 24)):
    x = torch.randn(n, p, device="cuda") * 0.02
    y = torch.randn(n, p, device="cuda") * 0.02
    for cz in (1.0, 0.0):
        out = amd.rotate(x, y, alignment=1.0, alpha=0.0, centralization=cz)
        print(os.environ.get("MECHA_ROT_POLAR"), n, p, cz, bool(torch.isfinite(out).all()))
