"""Grid definition of the dense sweep (/root/reference/lib/sdf.py:4-29).

Only the index->world matrix is needed on the host: grid coordinates are generated inside the kernels from the
voxel index (surs_query_grid), evaluated in float64 and cast to float32 exactly as create_grid + eval_func do, so
the 3.2 GB float64 coordinate array of the reference (R = 512) is never materialised.
"""
import numpy as np


def create_grid(resX, resY, resZ, b_min=np.array([-1, -1, -1]), b_max=np.array([1, 1, 1]), transform=None):
    """Returns (None, coords_matrix): same 4x4 matrix as the reference; the coordinate array is implicit."""
    m = np.eye(4)
    length = np.asarray(b_max, np.float64) - np.asarray(b_min, np.float64)
    m[0, 0], m[1, 1], m[2, 2] = length[0] / resX, length[1] / resY, length[2] / resZ
    m[0:3, 3] = np.asarray(b_min, np.float64)
    if transform is not None:
        m = np.matmul(np.asarray(transform, np.float64), m)
    return None, m
