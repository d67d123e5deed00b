"""Input stage of the test path: the output contract of the reference's EvalDataset_LR_v2
(/root/reference/lib/data/EvalDataset_LR_v2.py:134-180,185-254,389-410):

    dataroot/image_final/<subject>.{jpg,png}   RGB image
    dataroot/mask_final/<subject>.{png,jpg}    8-bit mask
    item = {'name': (stem, ext), 'b_min', 'b_max', 'img_LR': [V,3,H,W] float32 = mask * ((rgb/255 - 0.5)/0.5),
            'calib': [V,4,4] diag(2,-2,2,1)}

Host-side file I/O (PIL) only; no resizing at eval, exactly as the reference.
"""
import os

import numpy as np
import torch


class EvalDataset:
    def __init__(self, opt, phase="test"):
        self.opt = opt
        self.projection_mode = "orthogonal"
        self.root = opt.dataroot
        self.RENDER = os.path.join(self.root, "image_final")
        self.MASK = os.path.join(self.root, "mask_final")
        self.B_MIN = np.array(opt.b_min, dtype=float)
        self.B_MAX = np.array(opt.b_max, dtype=float)
        self.is_train = phase == "train"
        self.num_views = opt.num_views
        self.subjects = sorted(os.listdir(self.RENDER))

    def __len__(self):
        return len(self.subjects)

    @staticmethod
    def _first_existing(*paths):
        for p in paths:
            if os.path.isfile(p):
                return p
        return paths[-1]

    def get_render(self, subject):
        from PIL import Image
        render_path = self._first_existing(os.path.join(self.RENDER, subject + ".jpg"), os.path.join(self.RENDER, subject + ".png"))
        mask_path = self._first_existing(os.path.join(self.MASK, subject + ".png"), os.path.join(self.MASK, subject + ".jpg"))
        mask = np.asarray(Image.open(mask_path).convert("L"), np.float32) / np.float32(255.0)        # ToTensor
        rgb = np.asarray(Image.open(render_path).convert("RGB"), np.float32) / np.float32(255.0)
        rgb = (rgb - np.float32(0.5)) / np.float32(0.5)                                               # Normalize(0.5, 0.5)
        img = np.ascontiguousarray((mask[None] * rgb.transpose(2, 0, 1)).astype(np.float32))
        calib = np.identity(4, np.float32) * 2
        calib[1, 1] = -2
        calib[3, 3] = 1
        # the reference repeats the same file for every view (get_render loops over view ids but ignores them)
        v = self.num_views
        return {"img_LR": torch.from_numpy(np.stack([img] * v, 0)), "calib": torch.from_numpy(np.stack([calib] * v, 0))}

    def get_item(self, index):
        subject = os.path.splitext(self.subjects[index])
        res = {"name": subject, "b_min": self.B_MIN, "b_max": self.B_MAX}
        res.update(self.get_render(subject[0]))
        return res

    def __getitem__(self, index):
        return self.get_item(index)


class SyntheticDataset:
    """Stand-in with the same item contract when there is no dataroot (--synthetic): seeded images."""

    def __init__(self, opt, n=1, size=None):
        from . import weights
        self.opt, self.n, self.projection_mode = opt, n, "orthogonal"
        self.size = size or opt.loadSize // 2
        self._w = weights

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        calib = np.identity(4, np.float32) * 2
        calib[1, 1] = -2
        calib[3, 3] = 1
        return {"name": ("synthetic_%04d" % i, ".png"), "b_min": np.array(self.opt.b_min, dtype=float),
                "b_max": np.array(self.opt.b_max, dtype=float),
                "img_LR": torch.from_numpy(self._w.synthetic_image(self.size, seed=1 + i)), "calib": torch.from_numpy(calib[None])}
