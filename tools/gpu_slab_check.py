"""dist.reconstruction_sharded end to end on ONE GPU: `world` processes share cuda:0 (gloo with host staging - RCCL does not
allow two ranks on one device), each sweeps its x-slab and extracts its part of the meshes; rank 0 compares the assembled
meshes with the single-process reconstruction: vertices (float64 world coordinates) and faces must be bit-identical.  Runs
every precision twice: the first reconstruction of a workspace extracts each slab in one piece, the second one pipelines
the extraction into the sweep.

    python tools/gpu_slab_check.py WORLD [R] [H]

With H (an image size) the encoder runs too, twice: replicated on every rank, and sharded (dist.encode_sharded: the
super-resolution net on the rank's image strip, feature_lr all-gathered, filter_hr on the strip) - feature maps bit-identical
on what each rank holds, meshes bit-identical.
"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def encoder_leg(rank, world, R, H, dev):
    import common
    from surs_amd import dist as sdist, mesh_util, model, options, weights
    calib = torch.from_numpy(common.CALIB[None].copy())
    b_min, b_max = np.array([-0.5] * 3), np.array([0.5] * 3)
    opt = options.BaseOptions().parse(common.FLAGS + ["--precision", "fp32"])
    img = torch.from_numpy(weights.synthetic_image(H, seed=1)).to(dev)

    def make():
        net = model.SuRSNet(opt).to(device=dev)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()})
        net.eval()
        return net
    rep, shd = make(), make()
    _, f_lr, f_hr = rep.super_res(img)
    rep.filter_hr(f_hr)
    rep.filter_lr(f_lr)
    used = sdist.encode_sharded(shd, img, calib, R, b_min, b_max)
    i0, i1 = sdist.slab_range(R, rank, world)
    from surs_amd.sdf import create_grid
    lo, hi = sdist.slab_feature_columns(i0, i1, create_grid(R, R, R, b_min, b_max)[1][:3], common.CALIB, 2 * H)
    same_lr = bool(torch.equal(rep.im_feat_list_lr[-1], shd.im_feat_list_lr[-1]))
    same_hr = bool(torch.equal(rep.im_feat_list_hr[0][..., lo:hi], shd.im_feat_list_hr[0][..., lo:hi]))
    if not same_lr:   # where: the gathered feature_lr itself, or filter_lr on it
        g = getattr(shd, "_gathered_lr", None)
        print("rank %d: gathered feature_lr == replicated feature_lr: %s; max |d im_feat_lr| %.3e" %
              (rank, None if g is None else bool(torch.equal(g.permute(2, 0, 1), rep.feature_lr[0])),
               float((rep.im_feat_list_lr[-1] - shd.im_feat_list_lr[-1]).abs().max())), flush=True)
    print("rank %d: sharded encoder %s; im_feat_lr %s, im_feat_hr columns [%d, %d) %s" %
          (rank, "used" if used else "NOT used (fallback)", "identical" if same_lr else "DIFFERENT", lo, hi,
           "identical" if same_hr else "DIFFERENT"), flush=True)
    a = sdist.reconstruction_sharded(opt, rep, calib, R, b_min, b_max)
    b = sdist.reconstruction_sharded(opt, shd, calib, R, b_min, b_max)
    if rank == 0:
        ref = mesh_util.reconstruction(opt, rep, dev, calib, R, b_min, b_max, use_octree=False, want_normals=False)
        ok = all(np.array_equal(a[i], b[i]) and np.array_equal(a[i], ref[i]) for i in (0, 1, 4, 5))
        print("encoder sharded world %d R %d H %d: %s (%d / %d vertices)" % (world, R, H, "sharded == replicated == one GPU" if ok else "MISMATCH",
                                                                            len(b[0]), len(b[4])), flush=True)


def worker(rank, world, port, R, H=0):
    import common
    from surs_amd import dist as sdist, mesh_util, model, options
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    if H:
        encoder_leg(rank, world, R, H, dev)
        dist.destroy_process_group()
        return
    fl, fh = common.synth_features()
    calib = torch.from_numpy(common.CALIB[None].copy())
    b_min, b_max = np.array([-0.5] * 3), np.array([0.5] * 3)
    for prec in ("fp32", "bf16"):
        opt = options.BaseOptions().parse(common.FLAGS + ["--precision", prec])

        def make():
            net = model.SuRSNet(opt).to(device=dev)
            net.load_state_dict({k: torch.from_numpy(v) for k, v in common.state_dict().items()})
            net.eval()
            net.im_feat_list_lr = [torch.from_numpy(fl[None]).to(dev)]
            net.im_feat_list_hr = [torch.from_numpy(fh[None]).to(dev)]
            return net
        net = make()
        ref = None
        if rank == 0:
            ref = mesh_util.reconstruction(opt, make(), dev, calib, R, b_min, b_max, use_octree=False, want_normals=False)
        for it in range(2):
            os.environ["SURS_SLAB_COLUMNS"] = str(R * max(1, (R // world) // 3))   # several launches per slab: streamed extraction
            got = sdist.reconstruction_sharded(opt, net, calib, R, b_min, b_max)
            if rank == 0:
                ok = all(np.array_equal(got[i], ref[i]) and got[i].dtype == ref[i].dtype for i in (0, 1, 4, 5))
                print("%s world %d R %d pass %d: %s  (%d / %d vertices, %d / %d faces)" %
                      (prec, world, R, it, "slab == one piece" if ok else "MISMATCH", len(got[0]), len(got[4]), len(got[1]), len(got[5])),
                      flush=True)
            else:
                assert got is None
    # want_normals=True (the 8-tuple of mesh_util.reconstruction): the volumes are gathered on rank 0, which extracts them whole
    if R <= 128:
        got = sdist.reconstruction_sharded(opt, net, calib, R, b_min, b_max, want_normals=True)
        if rank == 0:
            refn = mesh_util.reconstruction(opt, make(), dev, calib, R, b_min, b_max, use_octree=False, want_normals=True)
            ok = all(np.array_equal(got[i], refn[i]) for i in (0, 1, 3, 4, 5, 7))                      # vertices, faces, values
            ok = ok and all(np.abs(got[i] - refn[i]).max() < 1e-4 for i in (2, 6))                      # normals: float atomics
            print("normals world %d R %d: %s" % (world, R, "slab == one piece" if ok else "MISMATCH"), flush=True)
        else:
            assert got is None
    # the checks of marching_cubes_lewiner, on every rank: no level crossing anywhere in the grid
    net.im_feat_list_lr = [torch.zeros_like(net.im_feat_list_lr[0])]
    net.im_feat_list_hr = [torch.zeros_like(net.im_feat_list_hr[0])]
    try:
        sdist.reconstruction_sharded(opt, net, calib, R, np.array([-0.5, -0.5, 0.2]), np.array([0.5, 0.5, 0.21]))
        print("rank %d: no error on a flat field?" % rank, flush=True)
    except (ValueError, RuntimeError) as e:
        if rank == 0:
            print("flat field raises on every rank: %s" % type(e).__name__, flush=True)
    dist.destroy_process_group()


def main():
    world = int(sys.argv[1])
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(worker, args=(world, port, R, H), nprocs=world, join=True)


if __name__ == "__main__":
    main()
