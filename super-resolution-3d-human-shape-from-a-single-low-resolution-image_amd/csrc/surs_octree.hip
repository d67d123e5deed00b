// Coarse-to-fine grid sweep of the reference (eval_grid_octree, /root/reference/lib/sdf.py:55-120) on the device.
//
// Per level `reso` the reference (1) evaluates every grid point on the reso-lattice that is still dirty, (2) walks
// the cells of that lattice sequentially: if the 8 corner values of a field span less than `threshold`, the whole
// reso^3 block (corner (x,y,z) included, far corners excluded) is set to (max+min)/2 and marked clean - with ONE
// dirty mask shared by the HR and LR fields (a block flat in either field is never refined for the
// other, which keeps its zero initialisation where it was not sampled).  That sequential walk is order independent:
// a cell only writes its own block, and the corners it reads are the min-corners of cells that come LATER in the
// loop order, so every cell sees the values as evaluated.  Hence two parallel kernels: decide (reads only), apply.
// Volumes are float64 like the reference's numpy arrays (the block value (max+min)/2 is not an fp32 number).
#include <hip/hip_runtime.h>

#include "surs_common.h"

namespace surs {
namespace oct {

// lattice points of stride reso that are dirty -> idx[] (unordered append), count.  One atomic per WAVE (the dirty lanes counted by
// a ballot, every lane's slot = the wave's base + its rank among them): at the last level of a 512^3 sweep five million lattice
// points are dirty, and one atomic each on the one counter took 7 ms of a 150 ms reconstruction.  32-bit index arithmetic where the
// lattice has fewer than 2^32 points.
__global__ void select_kernel(const unsigned char *__restrict__ dirty, int R, int reso, long long *__restrict__ idx,
                              int *__restrict__ count, int cap) {
    const int n = (R + reso - 1) / reso;
    const long long total = (long long)n * n * n;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    long long f = 0;
    if (t < total) {
        int i, j, k;
        if (total < (1ll << 32)) {
            const unsigned t32 = (unsigned)t, un = (unsigned)n;
            const unsigned q = t32 / un;
            k = (int)(t32 - q * un);
            i = (int)(q / un);
            j = (int)(q - (unsigned)i * un);
        } else {
            k = (int)(t % n);
            j = (int)((t / n) % n);
            i = (int)(t / ((long long)n * n));
        }
        f = ((long long)(i * reso) * R + (long long)(j * reso)) * R + (long long)(k * reso);
        hit = dirty[f] != 0;
    }
    const unsigned long long m = __ballot(hit);
    if (m == 0ull) return;
    const int lane = (int)(threadIdx.x & 63u), leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, leader);
    if (hit) {
        const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
        if (slot < cap) idx[slot] = f;
    }
}

// sdf[idx[t]] = pred[t] (float -> double), dirty[idx[t]] = 0
__global__ void scatter_kernel(const long long *__restrict__ idx, int n, const float *__restrict__ phr,
                               const float *__restrict__ plr, double *__restrict__ sdf_hr, double *__restrict__ sdf_lr,
                               unsigned char *__restrict__ dirty) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const long long f = idx[t];
    sdf_hr[f] = (double)phr[t];
    sdf_lr[f] = (double)plr[t];
    dirty[f] = 0;
}

struct Decision {
    double mid_hr, mid_lr;
    int flags;  // bit 0: fill hr, bit 1: fill lr
    int pad;
};

__device__ __forceinline__ bool flat8(const double *__restrict__ s, long long f, long long sx, long long sy, long long sz,
                                       double threshold, double &mid) {
    // corner order of the reference: (x,y,z) (x,y,z+r) (x,y+r,z) (x,y+r,z+r) (x+r,...) - min/max do not depend on it
    const double v[8] = {s[f], s[f + sz], s[f + sy], s[f + sy + sz], s[f + sx], s[f + sx + sz], s[f + sx + sy], s[f + sx + sy + sz]};
    double lo = v[0], hi = v[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) {
        lo = fmin(lo, v[q]);
        hi = fmax(hi, v[q]);
    }
    mid = (hi + lo) / 2;
    return (hi - lo) < threshold;
}

// cells x,y,z in range(0, R - reso, reso)
__global__ void decide_kernel(const double *__restrict__ sdf_hr, const double *__restrict__ sdf_lr,
                              const unsigned char *__restrict__ dirty, int R, int reso, double threshold,
                              Decision *__restrict__ dec) {
    const int n = (R - reso + reso - 1) / reso;  // len(range(0, R - reso, reso))
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * n * n) return;
    const int cz = (int)(t % n), cy = (int)((t / n) % n), cx = (int)(t / ((long long)n * n));
    const int x = cx * reso, y = cy * reso, z = cz * reso, h = reso / 2;
    Decision d;
    d.mid_hr = d.mid_lr = 0.0;
    d.flags = 0;
    d.pad = 0;
    const long long RR = (long long)R * R;
    if (dirty[(long long)(x + h) * RR + (long long)(y + h) * R + (z + h)]) {
        const long long f = (long long)x * RR + (long long)y * R + z;
        const long long sx = (long long)reso * RR, sy = (long long)reso * R, sz = reso;
        if (flat8(sdf_hr, f, sx, sy, sz, threshold, d.mid_hr)) d.flags |= 1;
        if (flat8(sdf_lr, f, sx, sy, sz, threshold, d.mid_lr)) d.flags |= 2;
    }
    dec[t] = d;
}

// one workgroup-lane group per cell would waste lanes at small reso: one thread per (cell, block voxel)
__global__ void apply_kernel(const Decision *__restrict__ dec, int R, int reso, double *__restrict__ sdf_hr,
                             double *__restrict__ sdf_lr, unsigned char *__restrict__ dirty) {
    const int n = (R - reso + reso - 1) / reso;
    const int vol = reso * reso * reso;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * n * n * vol) return;
    const long long cell = t / vol;
    const int v = (int)(t - cell * vol);
    const Decision d = dec[cell];
    if (!d.flags) return;
    const int cz = (int)(cell % n), cy = (int)((cell / n) % n), cx = (int)(cell / ((long long)n * n));
    const int dz = v % reso, dy = (v / reso) % reso, dx = v / (reso * reso);
    const long long f = ((long long)(cx * reso + dx) * R + (cy * reso + dy)) * R + (cz * reso + dz);
    if (d.flags & 1) sdf_hr[f] = d.mid_hr;
    if (d.flags & 2) sdf_lr[f] = d.mid_lr;
    dirty[f] = 0;
}

__global__ void f64_to_f32_kernel(const double *__restrict__ a, float *__restrict__ b, long long n) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) b[t] = (float)a[t];
}

}  // namespace oct
}  // namespace surs

using namespace surs;
using namespace surs::oct;

static inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }

extern "C" int surs_octree_select(const unsigned char *dirty, int R, int reso, long long *idx, int cap, int *count_dev,
                                  int *count_host, void *stream) {
    SURS_REQUIRE(dirty && idx && count_dev && count_host && R > 0 && reso > 0, "bad argument");
    hipStream_t st = as_stream(stream);
    SURS_HIP_CHECK(hipMemsetAsync(count_dev, 0, sizeof(int), st));
    const long long n = (R + reso - 1) / reso;
    hipLaunchKernelGGL(select_kernel, dim3(nblk(n * n * n)), dim3(256), 0, st, dirty, R, reso, idx, count_dev, cap);
    SURS_LAUNCH_CHECK();
    SURS_HIP_CHECK(hipMemcpyAsync(count_host, count_dev, sizeof(int), hipMemcpyDeviceToHost, st));
    SURS_HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

extern "C" int surs_octree_scatter(const long long *idx, int n, const float *pred_hr, const float *pred_lr, double *sdf_hr,
                                   double *sdf_lr, unsigned char *dirty, void *stream) {
    if (n == 0) return 0;
    SURS_REQUIRE(idx && pred_hr && pred_lr && sdf_hr && sdf_lr && dirty && n > 0, "bad argument");
    hipLaunchKernelGGL(scatter_kernel, dim3(nblk(n)), dim3(256), 0, as_stream(stream), idx, n, pred_hr, pred_lr, sdf_hr, sdf_lr, dirty);
    SURS_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t surs_octree_workspace_bytes(int R, int reso) {
    const long long n = (R - reso + reso - 1) / reso;
    return (size_t)(n * n * n) * sizeof(Decision) + 256;
}

extern "C" int surs_octree_cells(double *sdf_hr, double *sdf_lr, unsigned char *dirty, int R, int reso, double threshold,
                                 void *workspace, size_t workspace_bytes, void *stream) {
    SURS_REQUIRE(sdf_hr && sdf_lr && dirty && workspace && R > 0 && reso > 1, "bad argument");
    SURS_REQUIRE(workspace_bytes >= surs_octree_workspace_bytes(R, reso), "workspace too small");
    const long long n = (R - reso + reso - 1) / reso;
    if (n <= 0) return 0;
    hipStream_t st = as_stream(stream);
    Decision *dec = (Decision *)workspace;
    hipLaunchKernelGGL(decide_kernel, dim3(nblk(n * n * n)), dim3(256), 0, st, sdf_hr, sdf_lr, dirty, R, reso, threshold, dec);
    SURS_LAUNCH_CHECK();
    hipLaunchKernelGGL(apply_kernel, dim3(nblk(n * n * n * reso * reso * reso)), dim3(256), 0, st, dec, R, reso, sdf_hr, sdf_lr, dirty);
    SURS_LAUNCH_CHECK();
    return 0;
}

extern "C" int surs_f64_to_f32(const double *a, float *b, long long n, void *stream) {
    if (n == 0) return 0;
    SURS_REQUIRE(a && b && n > 0, "bad argument");
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3(nblk(n)), dim3(256), 0, as_stream(stream), a, b, n);
    SURS_LAUNCH_CHECK();
    return 0;
}
