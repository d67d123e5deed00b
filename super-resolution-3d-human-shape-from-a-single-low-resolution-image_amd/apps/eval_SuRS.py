"""Test driver: same flags, flow and outputs as the reference's apps/eval_SuRS.py:27-80
(<results_path>/<name>/<subject>_HR.obj and _LR.obj per image under <dataroot>/image_final).

    python -m surs_amd.apps.eval_SuRS --residual --dataroot D --loadSize 1024 --results_path OUT --resolution 512 \
        --load_netG_checkpoint_path W/netG_epoch_12 --b_min -0.5 -0.5 -0.5 --b_max 0.5 0.5 0.5 [--precision bf16] [--no_octree]

Extensions: --precision {fp32,bf16,fp16}; --no_octree (dense sweep instead of the reference's default octree);
--synthetic (seeded image + weights when no dataset / checkpoint is at hand); --pipeline (the subjects as a pipeline:
next subject's decode, upload and encoder under the current sweep, OBJ files written in the background - same files).
"""
import os
import sys
import time

import torch

from ..data import EvalDataset, SyntheticDataset
from ..model import SuRSNet
from ..options import BaseOptions
from ..train_util import gen_mesh, gen_mesh_pipelined


def eval(opt):
    cuda = torch.device("cuda:%d" % opt.gpu_id)
    torch.cuda.set_device(cuda)
    test_dataset = SyntheticDataset(opt) if opt.synthetic else EvalDataset(opt, phase="test")
    print("test data size: ", len(test_dataset))
    netG = SuRSNet(opt, test_dataset.projection_mode).to(device=cuda)
    print("Using Network: ", netG.name)
    if opt.load_netG_checkpoint_path is not None:
        print("loading for net G ...", opt.load_netG_checkpoint_path)
        netG.load_state_dict(torch.load(opt.load_netG_checkpoint_path, map_location="cpu"))
    elif not opt.synthetic:
        raise SystemExit("--load_netG_checkpoint_path is required (or --synthetic for seeded random weights)")
    os.makedirs("%s/%s" % (opt.results_path, opt.name), exist_ok=True)
    netG.eval()
    if not opt.no_gen_mesh:
        print("generate mesh (test) ...")
        if opt.pipeline and opt.num_views == 1:
            # same files; decode / upload / encoder of the next subject and the OBJ writing of the previous one overlap the sweep
            t = time.time()
            gen_mesh_pipelined(opt, netG, cuda, test_dataset, range(len(test_dataset)),
                               lambda raw: "%s/%s/%s.obj" % (opt.results_path, opt.name, raw["name"][0]), use_octree=not opt.no_octree)
            print("%d subjects: %.3f s" % (len(test_dataset), time.time() - t))
            return
        for gen_idx in range(len(test_dataset)):
            t = time.time()
            test_data = test_dataset[gen_idx]
            save_path = "%s/%s/%s.obj" % (opt.results_path, opt.name, test_data["name"][0])
            gen_mesh(opt, netG, cuda, test_data, save_path, use_octree=not opt.no_octree)
            print("%s: %.3f s" % (test_data["name"][0], time.time() - t))


def main(argv=None):
    eval(BaseOptions().parse(argv))


if __name__ == "__main__":
    main(sys.argv[1:])
