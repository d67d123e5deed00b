"""BASELINE configs[3]'s path on one GPU: dist.reconstruction_sharded with 2 and 3 processes (ragged slabs) against the
single-process reconstruction - meshes bit-identical, both precisions, one-piece and streamed per-slab extraction
(tools/gpu_slab_check.py; gloo with host staging, since RCCL refuses two ranks on one device)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("world,R", [(2, 40), (3, 50)])
def test_sharded_reconstruction_equals_single_process(world, R):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_slab_check.py"), str(world), str(R)], capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("slab == one piece") == 4 and "MISMATCH" not in r.stdout, r.stdout
    assert "flat field raises on every rank" in r.stdout and "no error on a flat field" not in r.stdout, r.stdout
