"""What do the sweep's per-batch preparation kernels cost on a FEW compute units, and the column kernel on 256 minus those?  (VERDICT
round 3, item 5, first option: prepare batch b + 1 on a reserved handful of CUs from a second stream while the persistent column
kernel sweeps batch b on the rest.)  Streams with a CU mask (hipExtStreamCreateWithCUMask) wrapped as torch external streams; one batch
of 32 768 columns per configuration; run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cumask -o t -- python3 tools/diag/cu_mask_probe.py
"""
import ctypes as C, os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, gpu_common as g, oracle
from surs_amd import native

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) the stream may use."""
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


R = 512
fl, fh = common.synth_features(hl=256, hh=1024)
Fl, Fh = g.upload_nhwc(fl), g.upload_nhwc(fh)
ws = native.Workspace(g.dev())
mat = oracle.coords_matrix(R, [-0.5] * 3, [0.5] * 3)[:3].reshape(-1)
b = g.blob("bf16")
vh = torch.empty((64, R, R), dtype=torch.float32, device=g.dev()); vl = torch.empty_like(vh)
f = lambda: native.query_grid(0, 64, R, R, mat, common.CALIB.reshape(-1)[:12], 512, 200.0, Fl, Fh, b, "bf16", ws, vh, vl)
f(); torch.cuda.synchronize()
spread8 = [32 * x for x in range(8)]          # one CU in every group of 32
spread16 = [16 * x for x in range(16)]
configs = [("all 256", list(range(256))), ("248 (without 8 spread)", [i for i in range(256) if i not in spread8]),
           ("240 (without 16 spread)", [i for i in range(256) if i not in spread16]), ("8 spread", spread8), ("16 spread", spread16),
           ("first 8", list(range(8))), ("first 32", list(range(32)))]
for name, bits in configs:
    st = masked_stream(bits)
    with torch.cuda.stream(st):
        f(); st.synchronize()
        t = time.perf_counter()
        f(); st.synchronize()
        dt = time.perf_counter() - t
    print("%-28s one batch of 32 768 columns (prep + column kernel): %8.2f ms" % (name, dt * 1e3), flush=True)
