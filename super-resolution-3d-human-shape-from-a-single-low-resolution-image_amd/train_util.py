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


def gen_mesh_pipelined(opt, net, cuda, dataset, indices, save_path_of, use_octree=True, write=True):
    """gen_mesh for a run of subjects (the loop of /root/reference/apps/eval_SuRS.py:74-80) as a pipeline: while the GPU sweeps
    subject i, (1) a host thread decodes subject i+1's image and mask, (2) its pixels are uploaded and normalised / masked
    on the device (data.DeviceInputStage) and its encoder is enqueued on a second stream - behind subject i's sweep launches,
    before the host starts driving subject i's mesh extraction, so it fills the sweep's gaps and tail - and (3) another host
    thread writes subject i-1's OBJ files.  Same files and meshes as calling gen_mesh per subject.
    dataset: get_raw_item(index) -> {'name', 'b_min', 'b_max', 'rgb' uint8 [H,W,3], 'mask' uint8 [H,W]}.
    Returns [(verts_hr, faces_hr, verts_lr, faces_lr)] in order."""
    from concurrent.futures import ThreadPoolExecutor
    from .data import DeviceInputStage
    indices = list(indices)
    if not indices:
        return []
    dev = torch.device(cuda)
    stage = DeviceInputStage(dev)
    enc_stream = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    calib_tensor = gen_calib().to(device=dev)
    decoder, writer = ThreadPoolExecutor(1), ThreadPoolExecutor(1)
    results, writes = [], []

    def encode(raw):
        # everything here is allocated, written and read on enc_stream (the caching allocator keeps per-stream pools); the
        # feature maps are handed to the main stream through `ev` + record_stream
        with torch.cuda.stream(enc_stream):
            feats = net.encode_image(stage.prepare(raw["rgb"], raw["mask"]))
            ev = torch.cuda.Event()
            ev.record(enc_stream)
        return feats, ev

    try:
        pending = decoder.submit(dataset.get_raw_item, indices[0])
        raw = pending.result()
        pending = decoder.submit(dataset.get_raw_item, indices[1]) if len(indices) > 1 else None
        feats, ready = encode(raw)
        for n, idx in enumerate(indices):
            cur_raw, cur_feats = raw, feats
            main.wait_event(ready)
            for f in cur_feats:
                f.buf.record_stream(main)
            nxt = {}

            def enqueue_next():
                if pending is not None:
                    nxt["raw"] = pending.result()
                    nxt["feats"], nxt["ready"] = encode(nxt["raw"])

            verts_hr, faces_hr, _, _, verts_lr, faces_lr, _, _ = reconstruction(
                opt, net, dev, calib_tensor, opt.resolution, cur_raw["b_min"], cur_raw["b_max"], use_octree=use_octree,
                num_samples=opt.num_samples, want_normals=False, features=cur_feats, after_enqueue=enqueue_next)
            if pending is not None:
                raw, feats, ready = nxt["raw"], nxt["feats"], nxt["ready"]
                pending = decoder.submit(dataset.get_raw_item, indices[n + 2]) if n + 2 < len(indices) else None
            results.append((verts_hr, faces_hr, verts_lr, faces_lr))
            if write:
                path = save_path_of(cur_raw)
                writes.append(writer.submit(save_obj_mesh, path[:-4] + "_HR.obj", verts_hr, faces_hr))
                writes.append(writer.submit(save_obj_mesh, path[:-4] + "_LR.obj", verts_lr, faces_lr))
        for w in writes:
            w.result()
    finally:
        decoder.shutdown(wait=True)
        writer.shutdown(wait=True)
    return results
