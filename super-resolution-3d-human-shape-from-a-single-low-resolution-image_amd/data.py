"""Input stage of the test path: the output contract of the reference's EvalDataset_LR_v2
(/root/reference/lib/data/EvalDataset_LR_v2.py:134-180,185-254,389-410):

    dataroot/image_final/<subject>.{jpg,png}   RGB image
    dataroot/mask_final/<subject>.{png,jpg}    8-bit mask
    item = {'name': (stem, ext), 'b_min', 'b_max', 'img_LR': [V,3,H,W] float32 = mask * ((rgb/255 - 0.5)/0.5),
            'calib': [V,4,4] diag(2,-2,2,1)}

File decoding (PIL) stays on the host.  `get_render` is the plain host statement of the contract (numpy); the serving path
(`get_raw` + `DeviceInputStage`, used by train_util.gen_mesh_pipelined) uploads the decoded uint8 pixels and does ToTensor +
Normalize + mask multiply in one kernel (surs_image_prepare): the same float32 operations, bit-identical img_LR, written
straight into the encoder's NHWC layout.  No resizing at eval, exactly as the reference.
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

    def get_raw(self, subject):
        """Decoded pixels only: (rgb uint8 [H,W,3], mask uint8 [H,W])."""
        from PIL import Image
        render_path = self._first_existing(os.path.join(self.RENDER, subject + ".jpg"), os.path.join(self.RENDER, subject + ".png"))
        mask_path = self._first_existing(os.path.join(self.MASK, subject + ".png"), os.path.join(self.MASK, subject + ".jpg"))
        return (np.ascontiguousarray(np.asarray(Image.open(render_path).convert("RGB"), np.uint8)),
                np.ascontiguousarray(np.asarray(Image.open(mask_path).convert("L"), np.uint8)))

    def get_raw_item(self, index):
        """What gen_mesh_pipelined consumes: the item without its tensors, plus the decoded pixels."""
        subject = os.path.splitext(self.subjects[index])
        rgb, mask = self.get_raw(subject[0])
        return {"name": subject, "b_min": self.B_MIN, "b_max": self.B_MAX, "rgb": rgb, "mask": mask}

    def get_render(self, subject):
        rgb8, mask8 = self.get_raw(subject)
        mask = mask8.astype(np.float32) / np.float32(255.0)        # ToTensor
        rgb = rgb8.astype(np.float32) / np.float32(255.0)
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


class DeviceInputStage:
    """uint8 pixels -> img_LR on the device: pinned staging buffers (reused), asynchronous upload and surs_image_prepare on the
    current stream.  Returns a [1,3,H,W] float32 tensor (channels_last strides: the encoder takes it without a copy) holding
    exactly the bytes of the reference's img_LR."""

    def __init__(self, device):
        self.device = device
        self._pin = {}

    def _staged(self, key, a):
        ent = self._pin.get((key, a.shape))
        if ent is None:
            ent = self._pin[(key, a.shape)] = [torch.empty(a.shape, dtype=torch.uint8, pin_memory=True), None]
        buf, uploaded = ent
        if uploaded is not None:
            uploaded.synchronize()   # the previous subject's upload has read the buffer: only now may the host overwrite it
        buf.numpy()[...] = a
        d = buf.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        ent[1] = ev
        return d

    def prepare(self, rgb, mask):
        from . import native
        h, w = mask.shape
        assert rgb.shape == (h, w, 3) and rgb.dtype == np.uint8 and mask.dtype == np.uint8
        d_rgb, d_mask = self._staged("rgb", rgb), self._staged("mask", mask)
        out = torch.empty((1, h, w, 3), dtype=torch.float32, device=self.device)
        native.check(native.lib().surs_image_prepare(native._ptr(d_rgb), native._ptr(d_mask), h, w, native._ptr(out), 3, native._stream()))
        return out.permute(0, 3, 1, 2)


class SyntheticDataset:
    """Stand-in with the same item contract when there is no dataroot (--synthetic): seeded images."""

    def __init__(self, opt, n=1, size=None):
        from . import weights
        self.opt, self.n, self.projection_mode = opt, n, "orthogonal"
        self.size = size or opt.loadSize // 2
        self._w = weights

    def __len__(self):
        return self.n

    def get_raw_item(self, i):
        """8-bit pixels of a seeded image + the rectangular mask (the decoded form of an input pair)."""
        from . import prng
        s = self.size
        rgb = (prng.uniform01("synthetic_rgb8", 1 + i, s * s * 3) * 256.0).astype(np.uint8).reshape(s, s, 3)
        mask = np.zeros((s, s), np.uint8)
        mask[s // 8: 7 * s // 8, s // 4: 3 * s // 4] = 255
        return {"name": ("synthetic_%04d" % i, ".png"), "b_min": np.array(self.opt.b_min, dtype=float),
                "b_max": np.array(self.opt.b_max, dtype=float), "rgb": rgb, "mask": mask}

    def __getitem__(self, i):
        calib = np.identity(4, np.float32) * 2
        calib[1, 1] = -2
        calib[3, 3] = 1
        return {"name": ("synthetic_%04d" % i, ".png"), "b_min": np.array(self.opt.b_min, dtype=float),
                "b_max": np.array(self.opt.b_max, dtype=float),
                "img_LR": torch.from_numpy(self._w.synthetic_image(self.size, seed=1 + i)), "calib": torch.from_numpy(calib[None])}
