"""sha256 of the dense sweep's two volumes (R^3, bench field) per precision: A/B of library builds (SURS_LIB_PATH) for bit identity."""
import hashlib, os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import common, gpu_common as g, oracle
from surs_amd import native
import precision_report as pr
R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sd, Fl, Fh, _keep = pr.noise_inputs(g.dev())
mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)[:3].reshape(-1)
for dt in ("bf16", "fp16", "fp32"):
    b = native.pack_mlp({k: v for k, v in sd.items() if k.startswith("mlp_")}, dt, g.dev())[0]
    ws = native.Workspace(g.dev())
    vh = torch.empty((R, R, R), dtype=torch.float32, device=g.dev()); vl = torch.empty_like(vh)
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        native.query_grid(0, R, R, R, mat, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, b, dt, ws, vh, vl)
        torch.cuda.synchronize(); dtm = time.perf_counter() - t
    h = hashlib.sha256(vh.cpu().numpy().tobytes() + vl.cpu().numpy().tobytes()).hexdigest()[:16]
    print("%s R=%d: %.2f ms  %s" % (dt, R, dtm * 1e3, h), flush=True)
