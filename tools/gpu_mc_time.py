"""Times surs_mc_lewiner at R^3 on a smooth blob (few active cells) and on a noise-like field (a third of the cells active)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from surs_amd import native
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
ax = torch.linspace(-1, 1, R, device=dev)
z, y, x = torch.meshgrid(ax, ax, ax, indexing="ij")
blob = (1.0 / (1.0 + torch.exp(12.0 * (torch.sqrt(x * x + 1.4 * y * y + 0.8 * z * z) - 0.6)))).contiguous()
g = torch.Generator(device=dev); g.manual_seed(1)
noise = torch.rand((R, R, R), device=dev, generator=g)
noise = torch.nn.functional.avg_pool3d(noise[None, None], 3, 1, 1)[0, 0].contiguous()
noise = (noise - noise.mean()) / noise.std() * 0.25 + 0.5
for name, vol in (("blob", blob), ("noise", noise.contiguous())):
    ws = native.Workspace(dev)
    for want in (False, True):
        native.marching_cubes_lewiner(vol, 0.5, ws, want_normals=want)   # sizes the buffers
        torch.cuda.synchronize(); t = time.time()
        for _ in range(3):
            v, f, n, val = native.marching_cubes_lewiner(vol, 0.5, ws, want_normals=want)
        torch.cuda.synchronize()
        print("%s R=%d normals=%s: %.2f ms  (%d verts, %d faces)" % (name, R, want, (time.time() - t) / 3 * 1e3, len(v), len(f)))
