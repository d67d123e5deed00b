"""512^3: the streamed reconstruction (marching cubes on its own streams beside the sweep) against the one-piece extraction of
the same volumes - vertices, faces bit for bit.  Run on the GPU box: python tools/gpu_stream_check.py [R]"""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from surs_amd import mesh_util, model, options, train_util, weights
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
opt = options.BaseOptions().parse(["--loadSize", "1024", "--residual", "--b_min", "-0.5", "-0.5", "-0.5", "--b_max", "0.5", "0.5", "0.5",
                                   "--resolution", str(R), "--precision", "bf16"])
net = model.SuRSNet(opt).to(device=dev)
net.load_state_dict(weights.synthetic_state_dict(opt, seed=0))
net.eval()
img = torch.from_numpy(weights.synthetic_image(512, seed=1)).to(dev)
calib = train_util.gen_calib().to(dev)
b_min, b_max = np.array([-0.5] * 3), np.array([0.5] * 3)
_, f_lr, f_hr = net.super_res(img)
net.filter_hr(f_hr); net.filter_lr(f_lr)
vh, vl, mat = mesh_util.eval_volumes(opt, net, calib, R, b_min, b_max, None)
ref = mesh_util.meshes_from_volumes(net, [vh, vl], mat, want_normals=True)   # also sizes the buffers
for rep in range(2):
    out = mesh_util.reconstruction_streamed(opt, net, calib, R, b_min, b_max, None, want_normals=True)
    assert out is not None
    names = ("verts_hr", "faces_hr", "normals_hr", "values_hr", "verts_lr", "faces_lr", "normals_lr", "values_lr")
    for n, a, b in zip(names, ref, out):
        assert a.shape == b.shape, (n, a.shape, b.shape)
        if n.startswith("normals"):
            assert np.abs(a - b).max() <= 1e-4, (n, float(np.abs(a - b).max()))   # float atomics: order-dependent last bits (more terms per vertex in a noise field)
        else:
            assert np.array_equal(a, b), n
    print("streamed == one piece (rep %d): %d / %d vertices, %d / %d faces" % (rep, len(out[0]), len(out[4]), len(out[1]), len(out[5])))
