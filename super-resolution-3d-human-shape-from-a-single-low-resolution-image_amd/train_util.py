"""gen_mesh: the per-subject driver (/root/reference/lib/train_util.py:53-85), same signature and side effects:
encoder (super_res -> filter_hr -> filter_lr), calib diag(2,-2,2,1), reconstruction, two OBJ files."""
import numpy as np
import torch

from .mesh_util import reconstruction, save_obj_mesh


def gen_calib():
    m = np.identity(4) * 2
    m[1, 1] = -2
    m[3, 3] = 1
    return torch.from_numpy(m.astype(np.float32)).unsqueeze(0)


def gen_mesh(opt, net, cuda, data, save_path, use_octree=True):
    image_tensor = data["img_LR"].to(device=cuda)
    img_sr, feature_lr, feature_hr = net.super_res(image_tensor)
    net.filter_hr(feature_hr)
    net.filter_lr(feature_lr)
    calib_tensor = gen_calib().to(device=cuda)   # the dataset's own calib is ignored, as in the reference
    verts_hr, faces_hr, _, _, verts_lr, faces_lr, _, _ = reconstruction(
        opt, net, cuda, calib_tensor, opt.resolution, data["b_min"], data["b_max"], use_octree=use_octree,
        num_samples=opt.num_samples, want_normals=False)   # normals / values are discarded here, as in the reference
    save_obj_mesh(save_path[:-4] + "_HR.obj", verts_hr, faces_hr)
    save_obj_mesh(save_path[:-4] + "_LR.obj", verts_lr, faces_lr)
    return verts_hr, faces_hr, verts_lr, faces_lr
