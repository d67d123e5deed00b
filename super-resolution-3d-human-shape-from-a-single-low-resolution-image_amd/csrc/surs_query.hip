// Point evaluator for gfx950 (MI355X): project -> in-image mask -> bilinear gather -> z-feat ->
// SurfaceClassifier lr -> sigmoid*mask -> SurfaceClassifier hr -> sigmoid*mask.
//
// Replaces (reference, /root/reference):
//   orthogonal        lib/geometry.py:15-31        index (grid_sample)  lib/geometry.py:4-12
//   DepthNormalizer   lib/model/DepthNormalizer.py:18
//   query_mr/query_sr lib/model/SuRSNet.py:131-187   get_preds  lib/model/BaseSuRSNet.py:80-85
//   SurfaceClassifier.forward  lib/model/SurfaceClassifier.py:53-81
//   create_grid / eval_grid / eval_func   lib/sdf.py:4-52, lib/mesh_util.py:16-34   (grid entry point)
//
// Two arithmetic modes:
//   fp32  (parity mode, any points): gather kernel -> channel-major feature matrix F[336][N]; each MLP layer is
//         one launch of a 128x128x16 LDS-tiled GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains) with the
//         skip-concat done as a second K segment instead of a materialised torch.cat.
//   bf16/f16 (grid mode): all voxels of a (i,j) column share their 320 gathered features, so the feature part of
//         layers 0,2,3,4 is a per-column constant (computed in fp32 by the same GEMM, 0.1 % of the work) and
//         only the dense cores 1024->512, 512->256, 256->128 run per voxel - on v_mfma_f32_32x32x16_{bf16,f16},
//         activations never leaving registers (an accumulator tile is the next layer's B operand), weights
//         streamed L2 -> LDS by LDS-DMA in 32 KiB slabs shared by the 4 waves of the workgroup.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "surs_common.h"
#include "surs_mlp_layout.h"

namespace surs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// point generation + projection shared by the gather kernel
// ------------------------------------------------------------------------------------------------
struct PointSource {
    // mode 0: explicit points [3][n] (ld = n_total);  mode 1: grid voxels, flat index base+t, z fastest;
    // mode 2: grid columns, flat column index base+t = i*ry + j (k = 0);  mode 3: grid voxels listed in idx[t];
    // mode 4: grid columns listed in cols[t] (column index i*ry + j, k = 0): the octree levels' lattice columns
    // mode 5: the explicit points listed in cols[t] (indices into pts): the first point of every run of surs_query_points_columns
    int mode;
    const float *pts;
    const long long *idx;
    const int *cols;
    long long ld;
    long long base;
    int ry, rz;
    double mat[12];  // create_grid's coords_matrix rows 0..2
    float calib[12];
    float zmul, zdiv;
    int persp;  // 1: perspective projection (lib/geometry.py:34-48): x, y divided by the projected z
    int npts;   // mode 5: points in pts (a listed index is clamped into the array: a speculative launch of surs_query_points_columns may
                // list more runs than this call's array holds - stale list entries -, whose results nobody reads)
};

__device__ __forceinline__ void make_point(const PointSource &s, long long t, float &px, float &py, float &pz) {
    if (s.mode == 0 || s.mode == 5) {
        if (s.mode == 5) {
            t = (long long)s.cols[t];
            t = t < 0 ? 0 : (t >= s.npts ? s.npts - 1 : t);
        }
        px = s.pts[t];
        py = s.pts[s.ld + t];
        pz = s.pts[2 * s.ld + t];
    } else {
        long long f = (s.mode == 3) ? s.idx[t] : (s.mode == 4 ? (long long)s.cols[t] : s.base + t);
        double i, j, k;
        if (s.mode == 1 || s.mode == 3) {
            k = (double)(f % s.rz);
            j = (double)((f / s.rz) % s.ry);
            i = (double)(f / ((long long)s.rz * s.ry));
        } else {
            k = 0.0;
            j = (double)(f % s.ry);
            i = (double)(f / s.ry);
        }
        // np.matmul(coords_matrix[:3,:3], idx) + coords_matrix[:3,3:4] in float64, then .float()
        px = (float)(((s.mat[0] * i + s.mat[1] * j) + s.mat[2] * k) + s.mat[3]);
        py = (float)(((s.mat[4] * i + s.mat[5] * j) + s.mat[6] * k) + s.mat[7]);
        pz = (float)(((s.mat[8] * i + s.mat[9] * j) + s.mat[10] * k) + s.mat[11]);
    }
}

// ------------------------------------------------------------------------------------------------
// split-bf16 operands: x = a + b + c exactly (three bf16 parts of an fp32 value, 24 significant bits), so six bf16 MFMA
// partial products reproduce the fp32 product sum at 2.7x the fp32 matrix rate (conv_x3_kernel in surs_encoder.hip).
// A "split image" of a k-major fp32 matrix X[K][N] is [3 parts][K/16][N][16] bf16: the 16 k values of one column are
// 32 contiguous bytes, which is the MFMA operand order of v_mfma_f32_32x32x16_bf16 (lane (n, h) takes k = 8h..8h+7).
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3_bf16(float x, unsigned short &a, unsigned short &b, unsigned short &c) {
    const __bf16 ha = (__bf16)x;
    const float r1 = x - (float)ha;
    const __bf16 hb = (__bf16)r1;
    const float r2 = r1 - (float)hb;
    const __bf16 hc = (__bf16)r2;
    a = __builtin_bit_cast(unsigned short, ha);
    b = __builtin_bit_cast(unsigned short, hb);
    c = __builtin_bit_cast(unsigned short, hc);
}

#include "surs_gemm.inc"

// ------------------------------------------------------------------------------------------------
// gather: F[c][n] for c < 320 (bilinear, zeros padding, align_corners=True), F[320][n] = z_feat,
//         F[321][n] = 0 (p_lr slot), mask[n] = in_img, zproj[n] = projected Z (for the column kernel: Z at k = 0)
// One workgroup = 64 points; wave w gathers points 16w..16w+15 with lanes = channels (coalesced 256 B per
// tap), transposes through LDS and stores with lanes = points (coalesced).
// ------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(256) void gather_kernel(PointSource src, long long n, const float *__restrict__ feat_lr,
                                                     int hl, int wl, const float *__restrict__ feat_hr, int hh,
                                                     int wh, float *__restrict__ F, long long ldf,
                                                     float *__restrict__ mask, float *__restrict__ zproj,
                                                     unsigned short *__restrict__ Fs, long long fs_part) {
    // Fs (optional): the split image of F for the split-operand layer kernels, [NP][C0PAD/16][ldf][16], fs_part = part stride
    // (NP = 3: bf16 parts, NP = 2: f16 parts - SplitKind in surs_gemm.inc)
    __shared__ float tile[64][65];
    __shared__ float sx[64], sy[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long n0 = (long long)blockIdx.x * 64;
    // gridDim.y > 1: the five channel chunks of a point group are shared among gridDim.y workgroups (small launches - the ~ 100 columns of
    // a point-runs call - are a chain of load latencies: 33 us in one workgroup per 64 points); the first of them writes mask, z and the z tile
    const bool lead = blockIdx.y == 0;
    if (tid < 64) {
        long long t = n0 + tid;
        float X = 2.0f, Y = 2.0f;  // outside
        if (t < n) {
            float px, py, pz;
            make_point(src, t, px, py, pz);
            const float *c = src.calib;
            X = c[3] + ((c[0] * px + c[1] * py) + c[2] * pz);
            Y = c[7] + ((c[4] * px + c[5] * py) + c[6] * pz);
            const float Z = c[11] + ((c[8] * px + c[9] * py) + c[10] * pz);
            if (src.persp) {
                X = X / Z;
                Y = Y / Z;
            }
            const float in = (X >= -1.0f && X <= 1.0f && Y >= -1.0f && Y <= 1.0f) ? 1.0f : 0.0f;
            if (lead) {
                mask[t] = in;
                if (zproj) zproj[t] = Z;
                F[(long long)C_G * ldf + t] = Z * src.zmul / src.zdiv;
                F[(long long)(C_G + 1) * ldf + t] = 0.0f;
            }
            if (Fs && lead) {   // k-tile 20 = rows 320..335: z, the p_lr slot (mlp_last_kernel fills it), zero padding
                unsigned short zp[NP];
                SplitKind<NP>::split(Z * src.zmul / src.zdiv, zp);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    u16x8_t v = {zp[p], 0, 0, 0, 0, 0, 0, 0}, z = {0, 0, 0, 0, 0, 0, 0, 0};
                    u16x8_t *dst = reinterpret_cast<u16x8_t *>(Fs + p * fs_part + ((long long)(C_G / 16) * ldf + t) * 16);
                    dst[0] = v;
                    dst[1] = z;
                }
            }
        }
        sx[tid] = X;
        sy[tid] = Y;
    }
    __syncthreads();
    // 5 channel chunks of 64: 4 from the lr map, 1 from the hr map
    for (int chunk = blockIdx.y; chunk < 5; chunk += gridDim.y) {
        const bool is_hr = chunk == 4;
        const float *feat = is_hr ? feat_hr : feat_lr;
        const int H = is_hr ? hh : hl, W = is_hr ? wh : wl, C = is_hr ? C_HR : C_LR;
        const int c0 = is_hr ? 0 : chunk * 64;
        // eight points at a time, all 32 tap loads issued before the first use: with one workgroup per CU (the column
        // batches of the sweep) the kernel is a chain of load latencies.  Taps outside the image are loaded from a
        // clamped address and weighted with zero (zeros padding: the same sum).
        for (int q0 = 0; q0 < 16; q0 += 8) {
            float t[8][4], w[8][4];
#pragma unroll
            for (int u4 = 0; u4 < 8; ++u4) {
                const int p = wave * 16 + q0 + u4;
                const float u = sx[p], v = sy[p];
                const float ix = ((u + 1.0f) / 2.0f) * (float)(W - 1);
                const float iy = ((v + 1.0f) / 2.0f) * (float)(H - 1);
                const float fx = floorf(ix), fy = floorf(iy);
                const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
                const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
                w[u4][0] = (vy0 && vx0) ? ((float)x1 - ix) * ((float)y1 - iy) : 0.0f;
                w[u4][1] = (vy0 && vx1) ? (ix - (float)x0) * ((float)y1 - iy) : 0.0f;
                w[u4][2] = (vy1 && vx0) ? ((float)x1 - ix) * (iy - (float)y0) : 0.0f;
                w[u4][3] = (vy1 && vx1) ? (ix - (float)x0) * (iy - (float)y0) : 0.0f;
                const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
                const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
                const long long ch = c0 + lane;
                t[u4][0] = feat[((long long)cy0 * W + cx0) * C + ch];
                t[u4][1] = feat[((long long)cy0 * W + cx1) * C + ch];
                t[u4][2] = feat[((long long)cy1 * W + cx0) * C + ch];
                t[u4][3] = feat[((long long)cy1 * W + cx1) * C + ch];
            }
#pragma unroll
            for (int u4 = 0; u4 < 8; ++u4) {
                float a = 0.0f;   // nw, ne, sw, se in this order, as grid_sample accumulates them
                a += t[u4][0] * w[u4][0];
                a += t[u4][1] * w[u4][1];
                a += t[u4][2] * w[u4][2];
                a += t[u4][3] * w[u4][3];
                tile[lane][wave * 16 + q0 + u4] = a;
            }
        }
        __syncthreads();
        const int cbase = is_hr ? C_LR : chunk * 64;
        for (int q = 0; q < 16; ++q) {
            const int ch = wave * 16 + q;
            if (n0 + lane < n) F[(long long)(cbase + ch) * ldf + n0 + lane] = tile[ch][lane];
        }
        if (Fs) {   // 4 k-tiles x 64 points x 2 halves of 8 channels
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int item = tid + 256 * it, p = item & 63, half = (item >> 6) & 1, ktl = item >> 7;
                u16x8_t q[NP];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    unsigned short parts[NP];
                    SplitKind<NP>::split(tile[ktl * 16 + half * 8 + j][p], parts);
#pragma unroll
                    for (int k = 0; k < NP; ++k) q[k][j] = parts[k];
                }
                if (n0 + p < n) {
                    unsigned short *dst = Fs + ((long long)(cbase / 16 + ktl) * ldf + n0 + p) * 16 + half * 8;
#pragma unroll
                    for (int k = 0; k < NP; ++k) *reinterpret_cast<u16x8_t *>(dst + k * fs_part) = q[k];
                }
            }
        }
        __syncthreads();
    }
}

// last layer (Cout = 1) + sigmoid * mask.  One thread per point, coalesced over points.
//   logit = b4 + w4[0:128].Y3[:,n] + w4[128:128+336].F[:,n];  pred = mask * sigmoid(logit)
template <int NP>
__global__ __launch_bounds__(256) void mlp_last_kernel(const float *__restrict__ w4,
                                                       const float *__restrict__ Y3, const float *__restrict__ F,
                                                       long long ld, long long n, const float *__restrict__ mask,
                                                       float *__restrict__ pred, float *__restrict__ logit,
                                                       float *__restrict__ p_slot, unsigned short *__restrict__ Fs,
                                                       long long fs_part) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    float acc = w4[D4 + C0PAD];  // b4
    for (int k = 0; k < D4; ++k) acc = fmaf(w4[k], Y3[(long long)k * ld + t], acc);
    for (int k = 0; k < C_G + 2; ++k) acc = fmaf(w4[D4 + k], F[(long long)k * ld + t], acc);
    const float p = mask[t] * (1.0f / (1.0f + expf(-acc)));
    pred[t] = p;
    if (logit) logit[t] = acc;
    if (p_slot) p_slot[t] = p;
    if (Fs) {   // row 321 of the split image of F
        unsigned short parts[NP];
        SplitKind<NP>::split(p, parts);
        unsigned short *dst = Fs + ((long long)(C_G / 16) * ld + t) * 16 + 1;
#pragma unroll
        for (int k = 0; k < NP; ++k) dst[k * fs_part] = parts[k];
    }
}

// query_sr on points other than the preceding query_mr's (SuRSNet.py:161-187 only requires the same N): the lr occupancies that
// enter the hr classifier come from the caller instead of from this batch's lr classifier.  Writes row 321 of F and of its split image.
template <int NP>
__global__ __launch_bounds__(256) void patch_plr_kernel(const float *__restrict__ p_lr, long long ld, long long n, float *__restrict__ p_slot,
                                                        unsigned short *__restrict__ Fs, long long fs_part) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const float p = p_lr[t];
    p_slot[t] = p;
    if (Fs) {
        unsigned short parts[NP];
        SplitKind<NP>::split(p, parts);
        unsigned short *dst = Fs + ((long long)(C_G / 16) * ld + t) * 16 + 1;
#pragma unroll
        for (int k = 0; k < NP; ++k) dst[k * fs_part] = parts[k];
    }
}

// multi-view (num_views > 1, SurfaceClassifier.py:70-76): out[r][t] = (sum_v in[v][r][t]) * (1/V), views summed in order
__global__ __launch_bounds__(256) void mean_views_kernel(const float *__restrict__ in, long long view_stride, int nviews,
                                                         long long count, float inv, float *__restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    float acc = in[t];
    for (int v = 1; v < nviews; ++v) acc += in[(long long)v * view_stride + t];
    out[t] = acc * inv;
}

// fp32 k-major X[16 * ktiles][ld] -> its split image [3][ktiles][ld][16] (the view means of the multi-view path)
template <int NP>
__global__ __launch_bounds__(256) void split_rows_kernel(const float *__restrict__ X, long long ld, int ktiles,
                                                         unsigned short *__restrict__ Xs, long long part) {
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    const int kt = blockIdx.y;
    if (n >= ld) return;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        u16x8_t q[NP];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned short parts[NP];
            SplitKind<NP>::split(X[(long long)(16 * kt + 8 * half + j) * ld + n], parts);
#pragma unroll
            for (int k = 0; k < NP; ++k) q[k][j] = parts[k];
        }
        unsigned short *dst = Xs + ((long long)kt * ld + n) * 16 + half * 8;
#pragma unroll
        for (int k = 0; k < NP; ++k) *reinterpret_cast<u16x8_t *>(dst + k * part) = q[k];
    }
}

// multi-view last layer: one logit per point from the view means; every view gets it under its own in-image mask
// (SuRSNet.py:156,183: `in_img[:, None].float() * mlp(...)` broadcasts the [1,1,N] prediction over the V masks)
template <int NP>
__global__ __launch_bounds__(256) void mlp_last_views_kernel(const float *__restrict__ w4, const float *__restrict__ Y3,
                                                             const float *__restrict__ Fm, long long ld, long long n,
                                                             int nviews, const float *__restrict__ mask /*[V][ld]*/,
                                                             float *__restrict__ pred /*[V][n]*/, float *__restrict__ logit,
                                                             float *__restrict__ p_slot /*row 321 of view 0's F*/,
                                                             long long f_view_stride,
                                                             unsigned short *__restrict__ Fs /*view 0's split image*/,
                                                             long long fs_part) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    float acc = w4[D4 + C0PAD];
    for (int k = 0; k < D4; ++k) acc = fmaf(w4[k], Y3[(long long)k * ld + t], acc);
    for (int k = 0; k < C_G + 2; ++k) acc = fmaf(w4[D4 + k], Fm[(long long)k * ld + t], acc);
    const float y = 1.0f / (1.0f + expf(-acc));
    if (logit) logit[t] = acc;
    for (int v = 0; v < nviews; ++v) {
        const float p = mask[(long long)v * ld + t] * y;
        pred[(long long)v * n + t] = p;
        if (p_slot) p_slot[(long long)v * f_view_stride + t] = p;
        if (Fs) {   // a view's split image follows the previous view's parts
            unsigned short parts[NP];
            SplitKind<NP>::split(p, parts);
            unsigned short *dst = Fs + (long long)v * NP * fs_part + ((long long)(C_G / 16) * ld + t) * 16 + 1;
#pragma unroll
            for (int k = 0; k < NP; ++k) dst[k * fs_part] = parts[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host orchestration of the fp32 path
// ------------------------------------------------------------------------------------------------
struct Fp32Workspace {
    float *F, *Y0, *Y1, *Y2, *Y3, *mask;
    unsigned short *Fs, *Y0s, *Y1s, *Y2s;   // split images (split-bf16 layer kernels); they overlay Y0..Y2
    long long np;  // padded point count (multiple of 128)
};

// F, Y3 and the mask in fp32; the hidden activations either as split images (6 bytes per value, with the split image of
// F) or - SURS_GEMM_X3=0 - as fp32 in the same region
static size_t fp32_ws_bytes(long long np) {
    return (size_t)np * ((C0PAD + D4 + 1) * sizeof(float) + (size_t)(C0PAD + D1 + D2 + D3) * 6) + 4096;
}

static Fp32Workspace carve_fp32(void *ws, long long np) {
    Fp32Workspace w;
    float *p = (float *)ws;
    w.np = np;
    w.F = p; p += (size_t)C0PAD * np;
    w.Y3 = p; p += (size_t)D4 * np;
    w.mask = p; p += np;
    w.Y0 = p;
    w.Y1 = w.Y0 + (size_t)D1 * np;
    w.Y2 = w.Y1 + (size_t)D2 * np;
    unsigned short *q = (unsigned short *)p;
    w.Fs = q; q += (size_t)3 * C0PAD * np;
    w.Y0s = q; q += (size_t)3 * D1 * np;
    w.Y1s = q; q += (size_t)3 * D2 * np;
    w.Y2s = q;
    return w;
}

// SURS_GEMM_X3=0 in the environment keeps the fp32 MFMA kernel (A/B comparisons)
static bool gemm_use_x3() { return option(OPT_GEMM_X3) != 0; }

// SURS_GEMM_BIG=0 keeps the 128 x 128 layer kernel for every layer (A/B comparisons)
static bool gemm_use_big() { return option(OPT_GEMM_BIG) != 0; }

static int launch_gemm(hipStream_t st, bool transposed, const float *Wt, const void *W3, int M, const float *X1, int K1,
                       long long ld1, const float *X2, int K2, long long ld2, const float *bias, int act, float *Y,
                       long long ldy, long long np) {
    const int nblocks = (int)(np / 128);
    dim3 grid(gemm_grid(M / 128, nblocks));
    if (W3 && gemm_use_x3()) {
        const unsigned short *w3 = (const unsigned short *)W3;
        if (transposed)
            hipLaunchKernelGGL(gemm_x3_kernel<true>, grid, dim3(256), 0, st, w3, M, K1 + K2, X1, K1, ld1, X2, K2, ld2, bias, act, Y, ldy, nblocks);
        else
            hipLaunchKernelGGL(gemm_x3_kernel<false>, grid, dim3(256), 0, st, w3, M, K1 + K2, X1, K1, ld1, X2, K2, ld2, bias, act, Y, ldy, nblocks);
    } else if (transposed)
        hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, dim3(256), 0, st, Wt, M, X1, K1, ld1, X2, K2, ld2, bias, act, Y, ldy, nblocks);
    else
        hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, dim3(256), 0, st, Wt, M, X1, K1, ld1, X2, K2, ld2, bias, act, Y, ldy, nblocks);
    SURS_LAUNCH_CHECK();
    return 0;
}

// Operand split of the 256-point layer kernels (SplitKind): two f16 parts (default: three products per MAC, 4 bytes per value;
// |x| < 65504) or, with SURS_SPLIT=bf16x3, three bf16 parts (six products, 6 bytes, fp32's exponent range).  The older layer
// kernels (SURS_GEMM_BIG=0 / SURS_GEMM_X3=0) only know the bf16 form.
static int g_split_override = 0;   // surs_set_operand_split (process-wide)
static thread_local int t_split_call = 0;   // SursGridOptions::operand_parts of the surs_query_grid_opt call running on this thread
static int split_parts() {
    const int v = option(OPT_SPLIT_PARTS) == 3 ? 3 : 2;
    // (a thread's 1 = the one-product point path, run_points_fp32 only: everything else that asks here stays fp32-grade)
    const int want = t_split_call > 1 ? t_split_call : (g_split_override ? g_split_override : v);
    return (gemm_use_x3() && gemm_use_big()) ? want : 3;
}

// the 256-point layer kernels need more than 64 KB of dynamic LDS
static int g3_set_attributes() {
    static DeviceOnce attr;
    if (attr.first()) {
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_x3g_kernel<8, 256, G3_SPLIT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes(256, 2)));
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_x3g_kernel<8, 256, G3_F32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes(256, 2)));
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_x3g_kernel<8, 128, G3_F32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes(128, 2)));
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_x3g_kernel<8, 128, G3_F32_T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes(128, 2)));
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_x3g_kernel<8, 256, G3_SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes(256)));
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_x3g_kernel<16, 256, G3_SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes(256)));
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_x3g_kernel<8, 256, G3_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes(256)));
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_x3g_kernel<8, 128, G3_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes(128)));
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_x3g_kernel<8, 128, G3_F32_T>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes(128)));
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_x3g_kernel<8, 256, G3_F32_T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes(256, 2)));
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_x3g_kernel<8, 256, G3_F32_T>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes(256)));
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_x3g_kernel<8, 256, G3_F32_T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes(256, 1)));
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_x3g_kernel<8, 256, G3_SPLIT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes(256, 1)));
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_x3g_kernel<8, 256, G3_F32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes(256, 1)));
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_x3g_kernel<8, 128, G3_F32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds_bytes(128, 1)));
    }
    return 0;
}

static int launch_gemm_s(hipStream_t st, const void *W3, int M, const unsigned short *X1s, int K1, const unsigned short *X2s,
                         int K2, const float *bias, float *Y, unsigned short *Ys, long long np, int parts = 3) {
    SplitSeg s1 = {X1s, (long long)K1 * np, K1 / 16}, s2 = {X2s, (long long)K2 * np, K2 / 16};
    const int nblocks = (int)(np / 128);
    dim3 grid(gemm_grid(M / 128, nblocks));
    const unsigned short *w3 = (const unsigned short *)W3;
    if (parts == 2) {   // two f16 parts: the 256-point kernels only
        SURS_REQUIRE(np % 256 == 0 && (M % 256 == 0 || !Ys), "f16 x 2 layer kernels need 256-point tiles");
        int rca = g3_set_attributes();
        if (rca) return rca;
        const int nb256 = (int)(np / 256);
        const long long yp = (long long)M * np;
        if (Ys)
            hipLaunchKernelGGL((gemm_x3g_kernel<8, 256, G3_SPLIT, 2>), dim3(gemm_grid(M / 256, nb256)), dim3(512), g3_lds_bytes(256, 2), st,
                               w3, M, K1 + K2, s1, s2, np, bias, (float *)nullptr, 0LL, Ys, yp, nb256);
        else if (M % 256 == 0)
            hipLaunchKernelGGL((gemm_x3g_kernel<8, 256, G3_F32, 2>), dim3(gemm_grid(M / 256, nb256)), dim3(512), g3_lds_bytes(256, 2), st,
                               w3, M, K1 + K2, s1, s2, np, bias, Y, np, (unsigned short *)nullptr, 0LL, nb256);
        else
            hipLaunchKernelGGL((gemm_x3g_kernel<8, 128, G3_F32, 2>), dim3(gemm_grid(M / 128, nb256)), dim3(512), g3_lds_bytes(128, 2), st,
                               w3, M, K1 + K2, s1, s2, np, bias, Y, np, (unsigned short *)nullptr, 0LL, nb256);
        SURS_LAUNCH_CHECK();
        return 0;
    }
    if (parts == 1) {   // ONE f16 part per operand (11 significant bits: the reduced-precision point path), same kernels and tiles
        SURS_REQUIRE(np % 256 == 0 && (M % 256 == 0 || !Ys), "the one-part layer kernels need 256-point tiles");
        int rca = g3_set_attributes();
        if (rca) return rca;
        const int nb256 = (int)(np / 256);
        const long long yp = (long long)M * np;
        if (Ys)
            hipLaunchKernelGGL((gemm_x3g_kernel<8, 256, G3_SPLIT, 1>), dim3(gemm_grid(M / 256, nb256)), dim3(512), g3_lds_bytes(256, 1), st,
                               w3, M, K1 + K2, s1, s2, np, bias, (float *)nullptr, 0LL, Ys, yp, nb256);
        else if (M % 256 == 0)
            hipLaunchKernelGGL((gemm_x3g_kernel<8, 256, G3_F32, 1>), dim3(gemm_grid(M / 256, nb256)), dim3(512), g3_lds_bytes(256, 1), st,
                               w3, M, K1 + K2, s1, s2, np, bias, Y, np, (unsigned short *)nullptr, 0LL, nb256);
        else
            hipLaunchKernelGGL((gemm_x3g_kernel<8, 128, G3_F32, 1>), dim3(gemm_grid(M / 128, nb256)), dim3(512), g3_lds_bytes(128, 1), st,
                               w3, M, K1 + K2, s1, s2, np, bias, Y, np, (unsigned short *)nullptr, 0LL, nb256);
        SURS_LAUNCH_CHECK();
        return 0;
    }
    if (np % 256 == 0 && gemm_use_big() && (M % 256 == 0 || !Ys)) {
        // 256-point tiles: 256 rows per workgroup where M allows, else 128 (fp32 output only: the last hidden layer)
        int rca = g3_set_attributes();
        if (rca) return rca;
        const int nb256 = (int)(np / 256);
        const int nw = option(OPT_GEMM_WAVES);
        const long long yp = (long long)M * np;
        if (Ys && nw == 16)
            hipLaunchKernelGGL((gemm_x3g_kernel<16, 256, G3_SPLIT>), dim3(gemm_grid(M / 256, nb256)), dim3(1024), g3_lds_bytes(256), st, w3, M,
                               K1 + K2, s1, s2, np, bias, (float *)nullptr, 0LL, Ys, yp, nb256);
        else if (Ys)
            hipLaunchKernelGGL((gemm_x3g_kernel<8, 256, G3_SPLIT>), dim3(gemm_grid(M / 256, nb256)), dim3(512), g3_lds_bytes(256), st, w3, M,
                               K1 + K2, s1, s2, np, bias, (float *)nullptr, 0LL, Ys, yp, nb256);
        else if (M % 256 == 0)
            hipLaunchKernelGGL((gemm_x3g_kernel<8, 256, G3_F32>), dim3(gemm_grid(M / 256, nb256)), dim3(512), g3_lds_bytes(256), st, w3, M,
                               K1 + K2, s1, s2, np, bias, Y, np, (unsigned short *)nullptr, 0LL, nb256);
        else
            hipLaunchKernelGGL((gemm_x3g_kernel<8, 128, G3_F32>), dim3(gemm_grid(M / 128, nb256)), dim3(512), g3_lds_bytes(128), st, w3, M,
                               K1 + K2, s1, s2, np, bias, Y, np, (unsigned short *)nullptr, 0LL, nb256);
    } else if (Ys)
        hipLaunchKernelGGL(gemm_x3s_kernel<true>, grid, dim3(256), 0, st, w3, M, K1 + K2, s1, s2, np, bias, 1, (float *)nullptr,
                           0LL, Ys, (long long)M * np, nblocks);
    else
        hipLaunchKernelGGL(gemm_x3s_kernel<false>, grid, dim3(256), 0, st, w3, M, K1 + K2, s1, s2, np, bias, 1, Y, np,
                           (unsigned short *)nullptr, 0LL, nblocks);
    SURS_LAUNCH_CHECK();
#ifdef SURS_GEMM_TRACE
    if (option(OPT_GEMM_TRACE)) {
        unsigned long long t[64];
        SURS_HIP_CHECK(hipStreamSynchronize(st));
        SURS_HIP_CHECK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_gemm_trace), sizeof(t)));
        fprintf(stderr, "gemm trace M=%d K=%d np=%lld (issue, mfma, store, barrier+):", M, K1 + K2, np);
        for (int i = 0; i < 15; ++i)
            fprintf(stderr, " [%llu %llu %llu %llu]", t[4 * i + 1] - t[4 * i], t[4 * i + 2] - t[4 * i + 1], t[4 * i + 3] - t[4 * i + 2],
                    t[4 * i + 4] - t[4 * i + 3]);
        fprintf(stderr, "\n");
    }
#endif
    return 0;
}

// runs gather + both MLPs for `n` points described by src; outputs may be offset pointers
static int run_points_fp32(hipStream_t st, const PointSource &src, long long n, const float *feat_lr, int hl, int wl,
                           const float *feat_hr, int hh, int wh, const char *blob, const MlpBlobHeader &h,
                           const Fp32Workspace &w, float *pred_hr, float *pred_lr, float *logit_hr, float *logit_lr,
                           int parts = 0, const float *p_lr_in = nullptr) {
    const long long np = w.np;
    const bool x3 = gemm_use_x3();
    // (the calling thread asked for the one-product point path - surs_set_operand_split_local(1): one f16 part per operand, the
    //  first plane of the two-part weight image; three times less matrix work than the fp32-grade path, 11 significant bits)
    if (parts == 0) parts = (t_split_call == 1 && x3 && gemm_use_big()) ? 1 : split_parts();
    unsigned short *Fs = x3 ? w.Fs : nullptr;
    const long long fs_part = (long long)C0PAD * np;
    // rows 322..335 of F meet zero weights and must be finite: the caller zeroes them once (zero_pad_rows; the split
    // image gets them from gather_kernel); rows < 322 are fully written for t < n; columns n..np-1 only feed outputs
    // that are never read
    if (parts == 1)
        hipLaunchKernelGGL(gather_kernel<1>, dim3((unsigned)ceil_div(n, 64)), dim3(256), 0, st, src, n, feat_lr, hl, wl, feat_hr,
                           hh, wh, w.F, np, w.mask, (float *)nullptr, Fs, fs_part);
    else if (parts == 2)
        hipLaunchKernelGGL(gather_kernel<2>, dim3((unsigned)ceil_div(n, 64)), dim3(256), 0, st, src, n, feat_lr, hl, wl, feat_hr,
                           hh, wh, w.F, np, w.mask, (float *)nullptr, Fs, fs_part);
    else
        hipLaunchKernelGGL(gather_kernel<3>, dim3((unsigned)ceil_div(n, 64)), dim3(256), 0, st, src, n, feat_lr, hl, wl, feat_hr,
                           hh, wh, w.F, np, w.mask, (float *)nullptr, Fs, fs_part);
    SURS_LAUNCH_CHECK();
    if (p_lr_in) {   // the hr classifier alone, on the caller's lr occupancies
        if (parts == 1)
            hipLaunchKernelGGL(patch_plr_kernel<1>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, p_lr_in, np, n,
                               w.F + (size_t)(C_G + 1) * np, Fs, fs_part);
        else if (parts == 2)
            hipLaunchKernelGGL(patch_plr_kernel<2>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, p_lr_in, np, n,
                               w.F + (size_t)(C_G + 1) * np, Fs, fs_part);
        else
            hipLaunchKernelGGL(patch_plr_kernel<3>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, p_lr_in, np, n,
                               w.F + (size_t)(C_G + 1) * np, Fs, fs_part);
        SURS_LAUNCH_CHECK();
    }
    for (int m = p_lr_in ? 1 : 0; m < 2; ++m) {
        auto WT = [&](int l) { return (const float *)(blob + h.wt[m][l]); };
        auto W3 = [&](int l) { return (const void *)(blob + (parts <= 2 ? h.wt2[m][l] : h.wt3[m][l])); };   // (one part: the image's first plane, f16(w))
        auto BI = [&](int l) { return (const float *)(blob + h.bias[m][l]); };
        int rc;
        if (x3) {
            if ((rc = launch_gemm_s(st, W3(0), D1, Fs, C0PAD, nullptr, 0, BI(0), nullptr, w.Y0s, np, parts))) return rc;
            if ((rc = launch_gemm_s(st, W3(1), D2, w.Y0s, D1, nullptr, 0, BI(1), nullptr, w.Y1s, np, parts))) return rc;
            if ((rc = launch_gemm_s(st, W3(2), D3, w.Y1s, D2, Fs, C0PAD, BI(2), nullptr, w.Y2s, np, parts))) return rc;
            if ((rc = launch_gemm_s(st, W3(3), D4, w.Y2s, D3, Fs, C0PAD, BI(3), w.Y3, nullptr, np, parts))) return rc;
        } else {
            if ((rc = launch_gemm(st, false, WT(0), nullptr, D1, w.F, C0PAD, np, nullptr, 0, 0, BI(0), 1, w.Y0, np, np))) return rc;
            if ((rc = launch_gemm(st, false, WT(1), nullptr, D2, w.Y0, D1, np, nullptr, 0, 0, BI(1), 1, w.Y1, np, np))) return rc;
            if ((rc = launch_gemm(st, false, WT(2), nullptr, D3, w.Y1, D2, np, w.F, C0PAD, np, BI(2), 1, w.Y2, np, np))) return rc;
            if ((rc = launch_gemm(st, false, WT(3), nullptr, D4, w.Y2, D3, np, w.F, C0PAD, np, BI(3), 1, w.Y3, np, np))) return rc;
        }
        if (parts == 1)
            hipLaunchKernelGGL(mlp_last_kernel<1>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st,
                               (const float *)(blob + h.w4[m]), w.Y3, w.F, np, n, w.mask, m == 0 ? pred_lr : pred_hr,
                               m == 0 ? logit_lr : logit_hr, m == 0 ? w.F + (size_t)(C_G + 1) * np : (float *)nullptr,
                               m == 0 ? Fs : (unsigned short *)nullptr, fs_part);
        else if (parts == 2)
            hipLaunchKernelGGL(mlp_last_kernel<2>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st,
                               (const float *)(blob + h.w4[m]), w.Y3, w.F, np, n, w.mask, m == 0 ? pred_lr : pred_hr,
                               m == 0 ? logit_lr : logit_hr, m == 0 ? w.F + (size_t)(C_G + 1) * np : (float *)nullptr,
                               m == 0 ? Fs : (unsigned short *)nullptr, fs_part);
        else
            hipLaunchKernelGGL(mlp_last_kernel<3>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st,
                               (const float *)(blob + h.w4[m]), w.Y3, w.F, np, n, w.mask, m == 0 ? pred_lr : pred_hr,
                               m == 0 ? logit_lr : logit_hr, m == 0 ? w.F + (size_t)(C_G + 1) * np : (float *)nullptr,
                               m == 0 ? Fs : (unsigned short *)nullptr, fs_part);
        SURS_LAUNCH_CHECK();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// grid column kernel (bf16 / f16 MFMA)
// ------------------------------------------------------------------------------------------------
template <int DT> struct HalfT;
template <> struct HalfT<SURS_BF16> {
    typedef __bf16 elem;
    typedef __bf16 vec8 __attribute__((ext_vector_type(8)));
    typedef __bf16 vec2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct HalfT<SURS_F16> {
    typedef _Float16 elem;
    typedef _Float16 vec8 __attribute__((ext_vector_type(8)));
    typedef _Float16 vec2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

struct GridArgs {
    const float *cc;       // [ncols][CC_PAD] column constants
    const float *colmask;  // [ncols]
    const float *zvec;     // [ZV_N]
    const char *core;      // SLABS_TOTAL slabs
    const char *corex;     // the same cores as two f16 parts per weight (kernel v5, fp32-grade)
    const char *b1frag;    // layer-1 biases as A fragments (kernel v3)
    const char *w1t;       // layer-1 weights channel-major (kernel v7)
    const char *w1tx;      // the same as two f16 parts (kernel v8)
    const float *rvec_lr, *rvec_hr;   // restated kernels: R = W1 (g . [a0 | w0z | w0p]) of the batch, fp32 [column][2 | 3 vectors][512 rows]
    long long rld_lr, rld_hr;         // (unused)
    float zmid;            // zf at mid column: where kernel v7's per-column LeakyReLU branch g_c is taken
    unsigned *colctr;      // kernels v7 / v8: next column to hand out (zeroed before the launch); kernel v11: one counter per pass
    int phase;             // kernel v11: 0 = both classifiers per tile, 1 = the lr classifier only, 2 = the hr classifier on vol_lr
    unsigned long long *kstat;   // profiling only: sum of kernel v7's residual k-steps (null otherwise)
    float b1_inv_scale;    // what B_ones holds: 1 / B1FRAG_SCALE of the blob's dtype
    float *vol_hr, *vol_lr;  // [ncols][rz]
    int ncols, rz;
    // lattice sweeps (the octree levels, surs_octree_level_columns): column c has kcount[c] <= rz items, the voxels
    // klist[c * rz + e] * zstride (ascending), e < kcount[c]; outputs at vol[c * rz + e].  Dense sweeps: zstride 1, both null
    // (item e = voxel e, rz items).
    int zstride;
    const int *kcount;
    const unsigned short *klist;
    // point runs (surs_query_points_columns; kernels v10 / v11): column c's items are the POINTS colstart[c] + e, e < kcount[c], of a
    // caller's point array - world z from zpts[colstart[c] + e] instead of the grid's (z0, dz), outputs at vol[colstart[c] + e]
    // (the callers' prediction arrays); klist null.  Dense and lattice sweeps: both null.
    const int *colstart;
    const float *zpts;
    // kernel v12 -> kernel v10: the (column, z tile) pairs v12 did not evaluate (ovf_list[2 t], [2 t + 1], t < *ovf_count); kernel v10
    // with tile_list set walks those instead of whole columns
    unsigned *ovf_count;
    int *ovf_list;
    const int *tile_list;
    const unsigned *tile_count;
    double z0, dz;  // world z of voxel k = (float)(dz*k + z0)
    float c22, c23;  // Z(k) = c23 + c22 * z(k)   (calib[2][0] = calib[2][1] = 0 in column mode)
    float zmul, zdiv;
};

__device__ __forceinline__ float lrelu01(float x) { return fmaxf(x, 0.01f * x); }

// All LDS addresses in the hot loops are (one per-lane base register) + (compile-time immediate): the per-lane
// parts are made opaque so that hipcc neither re-associates them into hundreds of distinct hoisted address
// registers (which it then spills) nor loses the ds_read `offset:` folding.
__device__ __forceinline__ unsigned opaque(unsigned v) {
    asm volatile("" : "+v"(v));
    return v;
}
// The lane index made on the spot (two VALU instructions, no input register): for per-lane values that are needed at the END of a long
// item - derived from the copy made at its start they stay live across it and are spilled, with a scratch store per item (round 6:
// the output pointer and `lane & 31` of kernel v12 were 0.4 GB of HBM writes per launch).
__device__ __forceinline__ unsigned fresh_lane() {
    unsigned v;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(v));
    return v;
}

#include "surs_grid_v3.inc"
#include "surs_grid_v5.inc"
#include "surs_grid_restated.inc"
#include "surs_grid_v10.inc"
#include "surs_grid_v12.inc"
#include "surs_grid_v11.inc"

// Column kernel v7's per-column affine part, step 1: the vectors g . a0, g . w0z (lr) and g . a0, g . w0z, g . w0p (hr) of a
// column batch as the split image [parts][1024 / 16][nvec * ncp][16] that the layer GEMM kernel reads as its point operand
// (point index = column * nvec + vector).  g_c = 1 or 0.01: the LeakyReLU branch of channel c at mid column (the same
// expression as in grid_mlp_v7).  Thread = (column, group of 16 channels).
// Until round 6 a thread took (column, 16 channels) with the LANES along the columns: every lane read its own 11.8 KB-strided row
// of the column constants - 64 cache lines per load instruction - and the kernel ran at 3 TB/s (197 us per batch of 32 768
// columns, a quarter of a batch's preparation).  Now a WAVE takes one column and its lanes the 64 groups of 16 channels: the two
// rows of column constants are read as whole 4 KB lines; the 32-byte results of a workgroup's four columns go through LDS, part by
// part, and leave as 256-byte (lr: 4 columns x 2 vectors) and 384-byte (hr: x 3) runs of the image.  The same arithmetic per value:
// the same bits.
template <int NP>
__global__ __launch_bounds__(256) void colsum_prepare_kernel(const float *__restrict__ cc, const float *__restrict__ zvec, int ncp,
                                                             float zmid, unsigned short *__restrict__ g_lr, long long part_lr,
                                                             unsigned short *__restrict__ g_hr, long long part_hr) {
    typedef SplitKind<NP> SK;
    __shared__ __attribute__((aligned(16))) unsigned stage[64][20][8];   // [channel group][column of four x vector: 4 x 2 lr, then 4 x 3 hr][32 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col0 = blockIdx.x * 4, col = col0 + wave;   // (ncp is a multiple of 256)
    const int c16 = lane;
    const float *row = cc + (size_t)col * CC_PAD + 16 * c16;
    unsigned o[5][NP][8];
    {
        float a0l[16], a0h[16], wzl[16], wzh[16], wph[16];
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
            const f32x4 t0 = *reinterpret_cast<const f32x4 *>(row + j), t1 = *reinterpret_cast<const f32x4 *>(row + CC_A0_HR + j);
            const f32x4 t2 = *reinterpret_cast<const f32x4 *>(zvec + ZV_W0Z_LR + 16 * c16 + j);
            const f32x4 t3 = *reinterpret_cast<const f32x4 *>(zvec + ZV_W0Z_HR + 16 * c16 + j);
            const f32x4 t4 = *reinterpret_cast<const f32x4 *>(zvec + ZV_W0P_HR + 16 * c16 + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0l[j + e] = t0[e]; a0h[j + e] = t1[e]; wzl[j + e] = t2[e]; wzh[j + e] = t3[e]; wph[j + e] = t4[e];
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float gl = fmaf(zmid, wzl[j], a0l[j]) > 0.0f ? 1.0f : 0.01f;
            const float gh = fmaf(0.5f, wph[j], fmaf(zmid, wzh[j], a0h[j])) > 0.0f ? 1.0f : 0.01f;
            const float val[5] = {gl * a0l[j], gl * wzl[j], gh * a0h[j], gh * wzh[j], gh * wph[j]};
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                unsigned short p[NP];
                SK::split(val[v], p);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    if (j & 1) o[v][q][j >> 1] |= (unsigned)p[q] << 16;
                    else o[v][q][j >> 1] = p[q];
                }
            }
        }
    }
    const long long np_lr = 2LL * ncp, np_hr = 3LL * ncp;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        if (q) __syncthreads();   // (the previous part has left)
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const int slot = v < 2 ? 2 * wave + v : 8 + 3 * wave + (v - 2);
            *reinterpret_cast<u32x4 *>(&stage[c16][slot][0]) = u32x4{o[v][q][0], o[v][q][1], o[v][q][2], o[v][q][3]};
            *reinterpret_cast<u32x4 *>(&stage[c16][slot][4]) = u32x4{o[v][q][4], o[v][q][5], o[v][q][6], o[v][q][7]};
        }
        __syncthreads();
        // lr: 64 rows of 256 B = 16 pieces of 16 B; hr: 64 rows of 384 B = 24 pieces
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int item = tid + 256 * k, r = item >> 4, pc = item & 15;
            const u32x4 w = *reinterpret_cast<const u32x4 *>(&stage[r][0][0] + 4 * pc);
            *reinterpret_cast<u32x4 *>(g_lr + q * part_lr + ((long long)r * np_lr + 2LL * col0) * 16 + 8 * pc) = w;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int item = tid + 256 * k, r = item / 24, pc = item - 24 * r;
            const u32x4 w = *reinterpret_cast<const u32x4 *>(&stage[r][8][0] + 4 * pc);
            *reinterpret_cast<u32x4 *>(g_hr + q * part_hr + ((long long)r * np_hr + 3LL * col0) * 16 + 8 * pc) = w;
        }
    }
}

}  // namespace surs

using namespace surs;

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static int zero_pad_rows(hipStream_t st, const Fp32Workspace &w) {
    SURS_HIP_CHECK(hipMemsetAsync(w.F + (size_t)(C_G + 2) * w.np, 0, (size_t)(C0PAD - C_G - 2) * w.np * sizeof(float), st));
    return 0;
}

#ifndef SURS_DEFAULT_GRID_KERNEL
#define SURS_DEFAULT_GRID_KERNEL 12
#endif
#ifndef SURS_DEFAULT_GRID_F32_KERNEL
#define SURS_DEFAULT_GRID_F32_KERNEL 11
#endif
static int g_grid_kernel_override = 0;   // surs_set_grid_kernel
static bool grid_kernel_known(int v) { return v == 0 || v == 3 || v == 10 || v == 12 || v == 5 || v == 11; }
extern "C" int surs_set_grid_kernel(int version) {
    SURS_REQUIRE(grid_kernel_known(version),
                 "column kernel: 0 (default / SURS_GRID_KERNEL), reduced precision 3, 10 or 12, fp32-grade 5 or 11");
    g_grid_kernel_override = version;
    return 0;
}

// The same choice for the calling host thread only (0 = back to the process setting): what a retry after an f16 overflow uses,
// without touching what other threads compute with.
extern "C" int surs_set_operand_split_local(int parts) {
    SURS_REQUIRE(parts >= 0 && parts <= 3, "parts: 0 (process setting), 1 (one f16 part: the point path of the reduced precisions), 2 (f16 x 2) or 3 (bf16 x 3)");
    t_split_call = parts;
    return 0;
}

extern "C" int surs_set_operand_split(int parts) {
    SURS_REQUIRE(parts == 0 || parts == 2 || parts == 3, "parts: 0 (default / SURS_SPLIT), 2 (f16 x 2) or 3 (bf16 x 3)");
    g_split_override = parts;
    return 0;
}

extern "C" size_t surs_query_workspace_bytes(int max_points) {
    long long np = (long long)ceil_div(max_points, 256) * 256;
    return fp32_ws_bytes(np);
}

static void fill_calib(PointSource &s, const float *calib, float zmul, float zdiv) {
    for (int i = 0; i < 12; ++i) s.calib[i] = calib[i];
    s.zmul = zmul;
    s.zdiv = zdiv;
}

extern "C" int surs_query_points(const float *points, int n, const float *calib, float zmul, float zdiv,
                                 const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh,
                                 const void *mlp_blob, void *workspace, size_t workspace_bytes, float *pred_hr,
                                 float *pred_lr, float *logit_hr, float *logit_lr, void *stream) {
    SURS_REQUIRE(n >= 0, "negative point count");
    if (n == 0) return 0;  // empty batch: nothing to do (pointers may be null)
    SURS_REQUIRE(points && calib && feat_lr && feat_hr && mlp_blob && workspace && pred_hr && pred_lr, "null argument");
    SURS_REQUIRE(hl > 0 && wl > 0 && hh > 0 && wh > 0, "bad sizes");
    hipStream_t st = as_stream(stream);
    const long long np = (long long)ceil_div(n, 256) * 256;
    SURS_REQUIRE(workspace_bytes >= fp32_ws_bytes(np), "workspace too small: need %zu bytes", fp32_ws_bytes(np));
    const MlpBlobHeader h = blob_layout(SURS_BF16);
    PointSource src;
    memset(&src, 0, sizeof(src));
    src.mode = 0;
    src.pts = points;
    src.ld = n;
    fill_calib(src, calib, zmul, zdiv);
    Fp32Workspace w = carve_fp32(workspace, np);
    int rc = zero_pad_rows(st, w);
    if (rc) return rc;
    return run_points_fp32(st, src, n, feat_lr, hl, wl, feat_hr, hh, wh, (const char *)mlp_blob, h, w, pred_hr, pred_lr,
                           logit_hr, logit_lr);
}

extern "C" int surs_query_points_hr(const float *points, int n, const float *calib, float zmul, float zdiv,
                                    const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh,
                                    const void *mlp_blob, void *workspace, size_t workspace_bytes, const float *p_lr,
                                    float *pred_hr, float *logit_hr, void *stream) {
    SURS_REQUIRE(n >= 0, "negative point count");
    if (n == 0) return 0;
    SURS_REQUIRE(points && calib && feat_lr && feat_hr && mlp_blob && workspace && p_lr && pred_hr, "null argument");
    SURS_REQUIRE(hl > 0 && wl > 0 && hh > 0 && wh > 0, "bad sizes");
    hipStream_t st = as_stream(stream);
    const long long np = (long long)ceil_div(n, 256) * 256;
    SURS_REQUIRE(workspace_bytes >= fp32_ws_bytes(np), "workspace too small: need %zu bytes", fp32_ws_bytes(np));
    const MlpBlobHeader h = blob_layout(SURS_BF16);
    PointSource src;
    memset(&src, 0, sizeof(src));
    src.mode = 0;
    src.pts = points;
    src.ld = n;
    fill_calib(src, calib, zmul, zdiv);
    Fp32Workspace w = carve_fp32(workspace, np);
    int rc = zero_pad_rows(st, w);
    if (rc) return rc;
    return run_points_fp32(st, src, n, feat_lr, hl, wl, feat_hr, hh, wh, (const char *)mlp_blob, h, w, pred_hr, nullptr, logit_hr,
                           nullptr, 0, p_lr);
}

// ------------------------------------------------------------------------------------------------
// multi-view / perspective queries: SurfaceClassifier.py:70-76, train_util.py:40-51, geometry.py:34-48
// ------------------------------------------------------------------------------------------------
static size_t views_ws_bytes(long long np, int nviews) {
    // per view: F (C0PAD rows), Y2 (D3 rows), mask; shared: Y0, Y1, Y3, mean F, mean Y2
    // + the split images of the split-bf16 layer kernels: every view's F, Y0, Y1, the two means
    return ((size_t)nviews * (C0PAD + D3 + 1) + (D1 + D2 + D4 + C0PAD + D3)) * (size_t)np * sizeof(float) +
           ((size_t)nviews * C0PAD + D1 + D2 + D3 + C0PAD) * (size_t)np * 6 + 4096;
}

extern "C" size_t surs_query_views_workspace_bytes(int max_points, int num_views) {
    return views_ws_bytes((long long)ceil_div(max_points, 256) * 256, num_views < 1 ? 1 : num_views);
}

// The n points every srcs[v] (v < num_views) describes - the same samples seen by each view: explicit points repeated per view, or grid
// voxels - through both classifiers with the view mean after layer 2; pred_* [num_views][n].
static int run_points_views(hipStream_t st, const PointSource *srcs, int n, int num_views, const float *feat_lr, int hl, int wl,
                            const float *feat_hr, int hh, int wh, const void *mlp_blob, void *workspace, float *pred_hr, float *pred_lr,
                            float *logit_hr, float *logit_lr) {
    const long long np = (long long)ceil_div(n, 256) * 256;
    const MlpBlobHeader h = blob_layout(SURS_BF16);
    const char *blob = (const char *)mlp_blob;
    const int V = num_views;
    float *p = (float *)workspace;
    float *F = p;      p += (size_t)V * C0PAD * np;  // [V][C0PAD][np]
    float *Y2 = p;     p += (size_t)V * D3 * np;     // [V][D3][np]
    float *mask = p;   p += (size_t)V * np;          // [V][np]
    float *Y0 = p;     p += (size_t)D1 * np;
    float *Y1 = p;     p += (size_t)D2 * np;
    float *Y3 = p;     p += (size_t)D4 * np;
    float *Fm = p;     p += (size_t)C0PAD * np;
    float *Y2m = p;    p += (size_t)D3 * np;
    const long long fstride = (long long)C0PAD * np;
    const bool x3 = gemm_use_x3();
    const int parts = split_parts();
    unsigned short *q = (unsigned short *)p;
    unsigned short *Fs = q;   q += (size_t)V * 3 * fstride;   // [V][parts][C0PAD/16][np][16] (sized for three parts)
    unsigned short *Y0s = q;  q += (size_t)3 * D1 * np;
    unsigned short *Y1s = q;  q += (size_t)3 * D2 * np;
    unsigned short *Y2ms = q; q += (size_t)3 * D3 * np;
    unsigned short *Fms = q;
    // gather every view with its own calibration and feature maps; zero the padding rows of every F
    for (int v = 0; v < V; ++v) {
        float *Fv = F + (size_t)v * fstride;
        SURS_HIP_CHECK(hipMemsetAsync(Fv + (size_t)(C_G + 2) * np, 0, (size_t)(C0PAD - C_G - 2) * np * sizeof(float), st));
        const PointSource &src = srcs[v];
        unsigned short *Fsv = x3 ? Fs + (size_t)v * parts * fstride : (unsigned short *)nullptr;
        if (parts == 2)
            hipLaunchKernelGGL(gather_kernel<2>, dim3((unsigned)ceil_div(n, 64)), dim3(256), 0, st, src, (long long)n,
                               feat_lr + (size_t)v * hl * wl * C_LR, hl, wl, feat_hr + (size_t)v * hh * wh * C_HR, hh, wh, Fv, np,
                               mask + (size_t)v * np, (float *)nullptr, Fsv, fstride);
        else
            hipLaunchKernelGGL(gather_kernel<3>, dim3((unsigned)ceil_div(n, 64)), dim3(256), 0, st, src, (long long)n,
                               feat_lr + (size_t)v * hl * wl * C_LR, hl, wl, feat_hr + (size_t)v * hh * wh * C_HR, hh, wh, Fv, np,
                               mask + (size_t)v * np, (float *)nullptr, Fsv, fstride);
        SURS_LAUNCH_CHECK();
    }
    for (int m = 0; m < 2; ++m) {
        auto WT = [&](int l) { return (const float *)(blob + h.wt[m][l]); };
        auto W3 = [&](int l) { return (const void *)(blob + (parts == 2 ? h.wt2[m][l] : h.wt3[m][l])); };
        auto BI = [&](int l) { return (const float *)(blob + h.bias[m][l]); };
        int rc;
        for (int v = 0; v < V && x3; ++v) {   // split images between the layers, fp32 out of layer 2 for the mean
            const unsigned short *Fsv = Fs + (size_t)v * parts * fstride;
            if ((rc = launch_gemm_s(st, W3(0), D1, Fsv, C0PAD, nullptr, 0, BI(0), nullptr, Y0s, np, parts))) return rc;
            if ((rc = launch_gemm_s(st, W3(1), D2, Y0s, D1, nullptr, 0, BI(1), nullptr, Y1s, np, parts))) return rc;
            if ((rc = launch_gemm_s(st, W3(2), D3, Y1s, D2, Fsv, C0PAD, BI(2), Y2 + (size_t)v * D3 * np, nullptr, np, parts))) return rc;
        }
        for (int v = 0; v < V && !x3; ++v) {
            const float *Fv = F + (size_t)v * fstride;
            if ((rc = launch_gemm(st, false, WT(0), W3(0), D1, Fv, C0PAD, np, nullptr, 0, 0, BI(0), 1, Y0, np, np))) return rc;
            if ((rc = launch_gemm(st, false, WT(1), W3(1), D2, Y0, D1, np, nullptr, 0, 0, BI(1), 1, Y1, np, np))) return rc;
            if ((rc = launch_gemm(st, false, WT(2), W3(2), D3, Y1, D2, np, Fv, C0PAD, np, BI(2), 1, Y2 + (size_t)v * D3 * np, np, np)))
                return rc;
        }
        // the view mean after layer 2 (index len(filters) // 2) of both the activations and the input features
        const float inv = 1.0f / (float)V;
        hipLaunchKernelGGL(mean_views_kernel, dim3((unsigned)ceil_div((long long)D3 * np, 256)), dim3(256), 0, st, Y2,
                           (long long)D3 * np, V, (long long)D3 * np, inv, Y2m);
        SURS_LAUNCH_CHECK();
        hipLaunchKernelGGL(mean_views_kernel, dim3((unsigned)ceil_div(fstride, 256)), dim3(256), 0, st, F, fstride, V, fstride,
                           inv, Fm);
        SURS_LAUNCH_CHECK();
        if (x3) {
            if (parts == 2) {
                hipLaunchKernelGGL(split_rows_kernel<2>, dim3((unsigned)ceil_div(np, 256), D3 / 16), dim3(256), 0, st, Y2m, np, D3 / 16,
                                   Y2ms, (long long)D3 * np);
                hipLaunchKernelGGL(split_rows_kernel<2>, dim3((unsigned)ceil_div(np, 256), C0PAD / 16), dim3(256), 0, st, Fm, np,
                                   C0PAD / 16, Fms, fstride);
            } else {
                hipLaunchKernelGGL(split_rows_kernel<3>, dim3((unsigned)ceil_div(np, 256), D3 / 16), dim3(256), 0, st, Y2m, np, D3 / 16,
                                   Y2ms, (long long)D3 * np);
                hipLaunchKernelGGL(split_rows_kernel<3>, dim3((unsigned)ceil_div(np, 256), C0PAD / 16), dim3(256), 0, st, Fm, np,
                                   C0PAD / 16, Fms, fstride);
            }
            SURS_LAUNCH_CHECK();
            if ((rc = launch_gemm_s(st, W3(3), D4, Y2ms, D3, Fms, C0PAD, BI(3), Y3, nullptr, np, parts))) return rc;
        } else if ((rc = launch_gemm(st, false, WT(3), W3(3), D4, Y2m, D3, np, Fm, C0PAD, np, BI(3), 1, Y3, np, np)))
            return rc;
        if (parts == 2)
            hipLaunchKernelGGL(mlp_last_views_kernel<2>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st,
                               (const float *)(blob + h.w4[m]), Y3, Fm, np, (long long)n, V, mask, m == 0 ? pred_lr : pred_hr,
                               m == 0 ? logit_lr : logit_hr, m == 0 ? F + (size_t)(C_G + 1) * np : (float *)nullptr, fstride,
                               (m == 0 && x3) ? Fs : (unsigned short *)nullptr, fstride);
        else
            hipLaunchKernelGGL(mlp_last_views_kernel<3>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st,
                               (const float *)(blob + h.w4[m]), Y3, Fm, np, (long long)n, V, mask, m == 0 ? pred_lr : pred_hr,
                               m == 0 ? logit_lr : logit_hr, m == 0 ? F + (size_t)(C_G + 1) * np : (float *)nullptr, fstride,
                               (m == 0 && x3) ? Fs : (unsigned short *)nullptr, fstride);
        SURS_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int surs_query_points_views(const float *points, int n, int num_views, int projection, const float *calibs,
                                       float zmul, float zdiv, const float *feat_lr, int hl, int wl, const float *feat_hr,
                                       int hh, int wh, const void *mlp_blob, void *workspace, size_t workspace_bytes,
                                       float *pred_hr, float *pred_lr, float *logit_hr, float *logit_lr, void *stream) {
    SURS_REQUIRE(n >= 0, "negative point count");
    SURS_REQUIRE(num_views >= 1 && num_views <= 64, "num_views must be in [1, 64]");
    SURS_REQUIRE(projection == 0 || projection == 1, "projection: 0 = orthogonal, 1 = perspective");
    if (n == 0) return 0;
    SURS_REQUIRE(points && calibs && feat_lr && feat_hr && mlp_blob && workspace && pred_hr && pred_lr, "null argument");
    SURS_REQUIRE(hl > 0 && wl > 0 && hh > 0 && wh > 0, "bad sizes");
    const long long np = (long long)ceil_div(n, 256) * 256;
    SURS_REQUIRE(workspace_bytes >= views_ws_bytes(np, num_views), "workspace too small: need %zu bytes",
                 views_ws_bytes(np, num_views));
    std::vector<PointSource> srcs((size_t)num_views);
    for (int v = 0; v < num_views; ++v) {
        PointSource &src = srcs[v];
        memset(&src, 0, sizeof(src));
        src.mode = 0;
        src.pts = points + (size_t)v * 3 * n;
        src.ld = n;
        src.persp = projection;
        fill_calib(src, calibs + 12 * v, zmul, zdiv);
    }
    return run_points_views(as_stream(stream), srcs.data(), n, num_views, feat_lr, hl, wl, feat_hr, hh, wh, mlp_blob, workspace, pred_hr,
                            pred_lr, logit_hr, logit_lr);
}

// The dense sweep of a multi-view / perspective model (eval_grid over eval_func, lib/sdf.py:32-52, lib/mesh_util.py:20-28: every batch
// of grid points repeated per view, query_mr + query_sr, VIEW 0's predictions kept) as one call: the voxels of the slab [i0, i1) are
// generated in the gather (create_grid's float64 arithmetic) for every view's calibration, VIEWS_GRID_BATCH at a time.
static const int VIEWS_GRID_BATCH = 262144;

extern "C" size_t surs_query_grid_views_workspace_bytes(int num_views) {
    const int V = num_views < 1 ? 1 : num_views;
    return views_ws_bytes(VIEWS_GRID_BATCH, V) + (size_t)2 * V * VIEWS_GRID_BATCH * sizeof(float);
}

extern "C" int surs_query_grid_views(int i0, int i1, int ry, int rz, const double *mat, int num_views, int projection, const float *calibs,
                                     float zmul, float zdiv, const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh,
                                     const void *mlp_blob, void *workspace, size_t workspace_bytes, float *vol_hr, float *vol_lr,
                                     void *stream) {
    SURS_REQUIRE(mat && calibs && feat_lr && feat_hr && mlp_blob && workspace && vol_hr && vol_lr, "null argument");
    SURS_REQUIRE(i1 >= i0 && ry > 0 && rz > 0 && hl > 0 && wl > 0 && hh > 0 && wh > 0, "bad sizes");
    SURS_REQUIRE(num_views >= 1 && num_views <= 64, "num_views must be in [1, 64]");
    SURS_REQUIRE(projection == 0 || projection == 1, "projection: 0 = orthogonal, 1 = perspective");
    SURS_REQUIRE(workspace_bytes >= surs_query_grid_views_workspace_bytes(num_views), "workspace too small");
    hipStream_t st = as_stream(stream);
    const int V = num_views;
    float *ph = (float *)((char *)workspace + views_ws_bytes(VIEWS_GRID_BATCH, V)), *pl = ph + (size_t)V * VIEWS_GRID_BATCH;
    const long long total = (long long)(i1 - i0) * ry * rz;
    std::vector<PointSource> srcs((size_t)V);
    for (long long b0 = 0; b0 < total; b0 += VIEWS_GRID_BATCH) {
        const int nb = (int)((total - b0 < VIEWS_GRID_BATCH) ? total - b0 : VIEWS_GRID_BATCH);
        for (int v = 0; v < V; ++v) {
            PointSource &src = srcs[v];
            memset(&src, 0, sizeof(src));
            src.mode = 1;
            src.base = (long long)i0 * ry * rz + b0;
            src.ry = ry;
            src.rz = rz;
            src.persp = projection;
            for (int i = 0; i < 12; ++i) src.mat[i] = mat[i];
            fill_calib(src, calibs + 12 * v, zmul, zdiv);
        }
        const int rc = run_points_views(st, srcs.data(), nb, V, feat_lr, hl, wl, feat_hr, hh, wh, mlp_blob, workspace, ph, pl, nullptr,
                                        nullptr);
        if (rc) return rc;
        // view 0's row of the [V][nb] predictions
        SURS_HIP_CHECK(hipMemcpyAsync(vol_hr + b0, ph, (size_t)nb * sizeof(float), hipMemcpyDeviceToDevice, st));
        SURS_HIP_CHECK(hipMemcpyAsync(vol_lr + b0, pl, (size_t)nb * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// optional per-launch timing of the dominant kernel (grid_mlp_kernel) with HIP events on the launch stream:
// what bench.py's roofline figure is computed from
// ------------------------------------------------------------------------------------------------
namespace {
struct GridProf {
    std::mutex mu;
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    std::vector<double> pts;
    std::map<int, unsigned long long *> kstat;   // per device: counter of the residual k-steps of column kernels v7 / v8 (executed-FLOP statistic)
    double tiles = 0;                      // (z tile, MLP) pairs the timed v7 launches processed
} g_prof;
}  // namespace

static void prof_reset_locked(bool on) {
    for (auto &e : g_prof.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    g_prof.ev.clear();
    g_prof.pts.clear();
    g_prof.tiles = 0;
    for (auto &k : g_prof.kstat) (void)hipMemset(k.second, 0, sizeof(unsigned long long));   // (unified addressing: any current device)
    g_prof.on = on;
}

extern "C" int surs_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_prof.mu);
    prof_reset_locked(on != 0);
    return 0;
}

// launches: number of timed column-kernel launches since enable; total_ms: sum of their durations;
// points: voxels they evaluated.  Synchronises on the recorded events and resets the counters.
extern "C" int surs_profile_read(double *launches, double *total_ms, double *points) {
    std::lock_guard<std::mutex> lock(g_prof.mu);
    double ms = 0, p = 0;
    for (size_t i = 0; i < g_prof.ev.size(); ++i) {
        SURS_HIP_CHECK(hipEventSynchronize(g_prof.ev[i].second));
        float t = 0;
        SURS_HIP_CHECK(hipEventElapsedTime(&t, g_prof.ev[i].first, g_prof.ev[i].second));
        ms += t;
        p += g_prof.pts[i];
    }
    if (launches) *launches = (double)g_prof.ev.size();
    if (total_ms) *total_ms = ms;
    if (points) *points = p;
    prof_reset_locked(g_prof.on);
    return 0;
}

// Column kernel v7 runs a data-dependent number of layer-1 k-steps; while profiling is enabled it adds them up.  Returns the
// number of (z tile, MLP) pairs the timed launches processed and the residual k-steps (16 listed channels each) they ran -
// the affine k-step every pair runs is not counted.  Call before surs_profile_read (which resets the counters).
extern "C" int surs_profile_read_ksteps(double *tile_mlps, double *ksteps) {
    std::lock_guard<std::mutex> lock(g_prof.mu);
    for (auto &e : g_prof.ev) SURS_HIP_CHECK(hipEventSynchronize(e.second));
    unsigned long long v = 0;
    for (auto &k : g_prof.kstat) {
        unsigned long long one = 0;
        SURS_HIP_CHECK(hipMemcpy(&one, k.second, sizeof(one), hipMemcpyDeviceToHost));
        v += one;
    }
    if (tile_mlps) *tile_mlps = g_prof.tiles;
    if (ksteps) *ksteps = (double)v;
    return 0;
}

// grid batches: the general fp32 mode evaluates GRID_BATCH voxels per pass; column mode COL_BATCH columns per pass
static const long long GRID_BATCH = 65536;
// (32 768 columns = 64 planes of a 512^3 grid per launch of the column kernel: half as many gaps - per-batch gather, GEMMs, fragment
//  packing - as with 16 384, 1 - 1.4 % of a reconstruction, when the streamed extraction ends on a taper of small slabs:
//  mesh_util.reconstruction_streamed; 65 536 gains nothing more)
#ifndef SURS_COL_BATCH
#define SURS_COL_BATCH 32768
#endif
static const long long COL_BATCH = SURS_COL_BATCH;

static size_t col_base_bytes(long long ncb) {
    // F rows 0..335 for the column gather + CC + mask
    return (size_t)ncb * (C0PAD + CC_PAD + 1) * sizeof(float) + (size_t)ncb * C0PAD * 6 + 4096;   // + split image of F
}
// restated column kernels: split images of the five g-scaled vectors per column (up to three parts), the five R vectors, a zero bias
static size_t col_v7_image_bytes(long long ncb) { return (size_t)ncb * 5 * D1 * 6; }
static size_t col_v7_bytes(long long ncb) { return col_v7_image_bytes(ncb) + (size_t)ncb * 5 * D2 * 4 + D2 * 4 + 256; }
static size_t col_ws_bytes(long long ncb) { return col_base_bytes(ncb) + col_v7_bytes(ncb); }

extern "C" size_t surs_query_grid_workspace_bytes(int ry, int rz, int dtype) {
    (void)ry; (void)rz;
    const size_t col = col_ws_bytes(COL_BATCH);
    if (dtype == SURS_F32 || dtype == SURS_F32_GEMM) {   // column kernel v5 where the sweep allows it, the layer kernels otherwise
        const size_t gen = fp32_ws_bytes(GRID_BATCH);
        return gen > col ? gen : col;
    }
    return col;
}

// SURS_GRID_F32=gemm keeps the fp32 sweep on the per-point layer kernels (A/B comparisons; the path of general calibrations)
static bool grid_f32_use_columns() { return option(OPT_GRID_F32_COLUMNS) != 0; }

static int grid_set_attributes() {
    static DeviceOnce attr;
    if (!attr.first()) return 0;
    SURS_HIP_CHECK(hipFuncSetAttribute((const void *)grid_mlp_kernel_v3<SURS_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, GRID3_LDS_BYTES));
    SURS_HIP_CHECK(hipFuncSetAttribute((const void *)grid_mlp_kernel_v3<SURS_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, GRID3_LDS_BYTES));
    SURS_HIP_CHECK(hipFuncSetAttribute((const void *)grid_mlp_kernel_v5, hipFuncAttributeMaxDynamicSharedMemorySize, GRID5_LDS_BYTES));
    SURS_HIP_CHECK(hipFuncSetAttribute((const void *)grid_mlp_kernel_v11, hipFuncAttributeMaxDynamicSharedMemorySize, GRID11_LDS_BYTES));
    SURS_HIP_CHECK(hipFuncSetAttribute((const void *)grid_mlp_kernel_v10<SURS_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, GRID10_LDS_BYTES));
    SURS_HIP_CHECK(hipFuncSetAttribute((const void *)grid_mlp_kernel_v10<SURS_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, GRID10_LDS_BYTES));
    SURS_HIP_CHECK(hipFuncSetAttribute((const void *)grid_mlp_kernel_v12<SURS_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, GRID12_LDS_BYTES));
    SURS_HIP_CHECK(hipFuncSetAttribute((const void *)grid_mlp_kernel_v12<SURS_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, GRID12_LDS_BYTES));
    return 0;
}

// Column constants of a batch of columns (src.mode = 2, src.base = first column): F = gathered features, cmask, and
// CC[col][2944] = Wc^T F[:, col] + bc - the split-operand layer kernel in its transposed-output form on the split image of F
// (SURS_GEMM_X3=0 / SURS_GEMM_BIG=0: the older kernels on the fp32 F).  Workspace carved by the caller (col_base_bytes).
static int column_constants(hipStream_t st, const PointSource &src, long long nc, long long ncp, const float *feat_lr, int hl, int wl,
                            const float *feat_hr, int hh, int wh, const char *blob, const MlpBlobHeader &h, float *F, float *CC,
                            float *cmask) {
    int rc = 0;
    const bool split = gemm_use_x3() && gemm_use_big();
    const int parts = split_parts();
    unsigned short *Fs = (unsigned short *)(cmask + COL_BATCH);
    const long long fs_part = (long long)C0PAD * COL_BATCH;
    const dim3 gg((unsigned)ceil_div(nc, 64), nc <= 4096 ? 5u : 1u);
    if (parts == 2)
        hipLaunchKernelGGL(gather_kernel<2>, gg, dim3(256), 0, st, src, nc, feat_lr, hl, wl,
                           feat_hr, hh, wh, F, COL_BATCH, cmask, (float *)nullptr, split ? Fs : (unsigned short *)nullptr, fs_part);
    else
        hipLaunchKernelGGL(gather_kernel<3>, gg, dim3(256), 0, st, src, nc, feat_lr, hl, wl,
                           feat_hr, hh, wh, F, COL_BATCH, cmask, (float *)nullptr, split ? Fs : (unsigned short *)nullptr, fs_part);
    SURS_LAUNCH_CHECK();
    if (split) {
        if ((rc = g3_set_attributes())) return rc;
        SplitSeg s1 = {Fs, fs_part, C_G / 16}, s2 = {nullptr, 0, 0};
        const int nb256 = (int)(ncp / 256);
        if (parts == 2)
            hipLaunchKernelGGL((gemm_x3g_kernel<8, 128, G3_F32_T, 2>), dim3(gemm_grid(CC_PAD / 128, nb256)), dim3(512), g3_lds_bytes(128, 2),
                               st, (const unsigned short *)(blob + h.wc2), CC_PAD, C_G, s1, s2, COL_BATCH,
                               (const float *)(blob + h.bc), CC, (long long)CC_PAD, (unsigned short *)nullptr, 0LL, nb256);
        else
            hipLaunchKernelGGL((gemm_x3g_kernel<8, 128, G3_F32_T>), dim3(gemm_grid(CC_PAD / 128, nb256)), dim3(512), g3_lds_bytes(128),
                               st, (const unsigned short *)(blob + h.wc3), CC_PAD, C_G, s1, s2, COL_BATCH,
                               (const float *)(blob + h.bc), CC, (long long)CC_PAD, (unsigned short *)nullptr, 0LL, nb256);
        SURS_LAUNCH_CHECK();
    } else {
        rc = launch_gemm(st, true, (const float *)(blob + h.wc), blob + h.wc3, CC_PAD, F, C_G, COL_BATCH, nullptr, 0, 0,
                         (const float *)(blob + h.bc), 0, CC, CC_PAD, ncp);
        if (rc) return rc;
    }
    return 0;
}

// One workgroup per column: how many layer-0 channels the restated column kernels (v7 / v8) would list per z tile - the
// classification of grid_mlp_v7 with the hr classifier's p range taken as [0, 1] (an upper bound).  out[0] += lr, out[1] += hr.
__global__ __launch_bounds__(256) void probe_count_kernel(const float *__restrict__ cc, const float *__restrict__ zvec, int rz, int tile,
                                                          double z0, double dz, float c22, float c23, float zmul, float zdiv, float zmid,
                                                          unsigned long long *__restrict__ out) {
    const int tid = threadIdx.x;
    const float *row = cc + (size_t)blockIdx.x * CC_PAD;
    const f32x4 al = *reinterpret_cast<const f32x4 *>(row + CC_A0_LR + 4 * tid), ah = *reinterpret_cast<const f32x4 *>(row + CC_A0_HR + 4 * tid);
    const f32x4 zl = *reinterpret_cast<const f32x4 *>(zvec + ZV_W0Z_LR + 4 * tid), zh = *reinterpret_cast<const f32x4 *>(zvec + ZV_W0Z_HR + 4 * tid);
    const f32x4 ph = *reinterpret_cast<const f32x4 *>(zvec + ZV_W0P_HR + 4 * tid);
    auto zf_of = [&](int k) {
        const double zt = dz * (double)k;
        const float zw = (float)(zt + z0);
        const float Z = c23 + c22 * zw;
        return Z * zmul / zdiv;
    };
    unsigned nl = 0, nh = 0;
    for (int k0 = 0; k0 < rz; k0 += tile) {
        const float e0 = zf_of(k0), e1 = zf_of(k0 + tile - 1);
        const float zlo = fminf(e0, e1), zhi = fmaxf(e0, e1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            {
                const float xm = fmaf(zmid, zl[i], al[i]);
                const float t0 = zl[i] * zlo, t1 = zl[i] * zhi;
                const float mn = al[i] + fminf(t0, t1), mx = al[i] + fmaxf(t0, t1);
                nl += ((xm > 0.0f) ? (mn > 0.0f) : (mx <= 0.0f)) ? 0u : 1u;
            }
            {
                const float xm = fmaf(0.5f, ph[i], fmaf(zmid, zh[i], ah[i]));
                const float t0 = zh[i] * zlo, t1 = zh[i] * zhi;
                const float mn = ah[i] + fminf(t0, t1) + fminf(0.0f, ph[i]), mx = ah[i] + fmaxf(t0, t1) + fmaxf(0.0f, ph[i]);
                nh += ((xm > 0.0f) ? (mn > 0.0f) : (mx <= 0.0f)) ? 0u : 1u;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        nl += __shfl_xor(nl, d);
        nh += __shfl_xor(nh, d);
    }
    if ((tid & 63) == 0) {
        atomicAdd(out, (unsigned long long)nl);
        atomicAdd(out + 1, (unsigned long long)nh);
    }
}

static int query_grid_impl(int i0, int i1, int ry, int rz, const double *mat, const float *calib, float zmul,
                           float zdiv, const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh,
                           const void *mlp_blob, int dtype, void *workspace, size_t workspace_bytes, float *vol_hr,
                           float *vol_lr, int kernel_call, void *stream);

extern "C" int surs_query_grid_opt(int i0, int i1, int ry, int rz, const double *mat, const float *calib, float zmul,
                                   float zdiv, const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh,
                                   const void *mlp_blob, int dtype, void *workspace, size_t workspace_bytes, float *vol_hr,
                                   float *vol_lr, const SursGridOptions *opt, void *stream) {
    int kernel = 0, parts = 0;
    if (opt) {
        kernel = opt->kernel;
        parts = opt->operand_parts;
        SURS_REQUIRE(grid_kernel_known(kernel), "SursGridOptions.kernel: 0, 3, 10, 12 (reduced precision), 5, 11 (fp32-grade)");
        SURS_REQUIRE(parts == 0 || parts == 2 || parts == 3, "SursGridOptions.operand_parts: 0, 2 or 3");
    }
    // the operand split is read deep inside the launch helpers: scoped to this call and this thread, the process setting
    // (surs_set_operand_split) is neither read nor changed when the caller gave one
    const int saved = t_split_call;
    if (parts) t_split_call = parts;
    const int rc = query_grid_impl(i0, i1, ry, rz, mat, calib, zmul, zdiv, feat_lr, hl, wl, feat_hr, hh, wh, mlp_blob, dtype, workspace,
                                   workspace_bytes, vol_hr, vol_lr, kernel, stream);
    t_split_call = saved;
    return rc;
}

extern "C" int surs_query_grid(int i0, int i1, int ry, int rz, const double *mat, const float *calib, float zmul,
                               float zdiv, const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh,
                               const void *mlp_blob, int dtype, void *workspace, size_t workspace_bytes, float *vol_hr,
                               float *vol_lr, void *stream) {
    return query_grid_impl(i0, i1, ry, rz, mat, calib, zmul, zdiv, feat_lr, hl, wl, feat_hr, hh, wh, mlp_blob, dtype, workspace,
                           workspace_bytes, vol_hr, vol_lr, 0, stream);
}

// What every column batch of one sweep shares (dense sweeps: query_grid_impl; lattice sweeps: surs_octree_level_columns).
namespace {
struct ColumnSweep {
    hipStream_t st;
    const char *blob;
    MlpBlobHeader h;
    int dtype;            // SURS_F32 / SURS_BF16 / SURS_F16
    int kver, kver32;     // column kernel of the reduced precisions / of SURS_F32
    bool restated;
    const float *feat_lr; int hl, wl;
    const float *feat_hr; int hh, wh;
    const double *mat; const float *calib; float zmul, zdiv;
    void *workspace;
    int cus;
    int kmid;             // axis-2 voxel index at which the restated kernels take the per-column LeakyReLU branches g_c
    // point runs (surs_query_points_columns): the columns are runs of a caller's point array, the work items (column, z tile) pairs
    const int *run_colstart = nullptr;
    const float *run_z = nullptr;
    const int *run_tiles = nullptr;
    const unsigned *run_ntiles = nullptr;
    long long run_ntiles_host = 0;
};
}  // namespace

// Column kernel of a sweep.  Reduced precision: 12 (default) and 10 = layer 1 restated along the column as a per-column affine part +
// the residuals of the channels whose LeakyReLU branch changes inside the z tile - 10 on eight waves per workgroup with y1 whole in
// LDS, 12 as two workgroups of four waves per CU with layer 1 streamed into layer 2 (same bits; tiles that list more than it stages
// go to 10's tile mode); 3 = dense layer 1 (what the host asks for on fields that list most channels; differs from 10 / 12 by a few
// 16-bit roundings of layer 0).  fp32-grade
// (SURS_F32): 11 (default: restated, eight waves) or 5 (dense).  Precedence: the call's SursGridOptions.kernel, then
// surs_set_grid_kernel, then SURS_GRID_KERNEL / SURS_GRID_F32_KERNEL, then the default.
static void resolve_column_kernels(int kernel_call, int &kver, int &kver32) {
    const int ve = option(OPT_GRID_KERNEL), ve32 = option(OPT_GRID_F32_KERNEL);
    const int kver_env = (ve == 3 || ve == 10 || ve == 12) ? ve : SURS_DEFAULT_GRID_KERNEL;
    const int kver32_env = (ve32 == 5 || ve32 == 11) ? ve32 : SURS_DEFAULT_GRID_F32_KERNEL;
    const auto is32 = [](int v) { return v == 5 || v == 11; };
    kver = kver_env;
    kver32 = kver32_env;
    if (g_grid_kernel_override) (is32(g_grid_kernel_override) ? kver32 : kver) = g_grid_kernel_override;
    if (kernel_call) (is32(kernel_call) ? kver32 : kver) = kernel_call;
}

static int device_cus() {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) return prop.multiProcessorCount;
    return 256;
}

// One batch of nc <= COL_BATCH columns described by src (mode 2: consecutive columns from src.base; mode 4: listed columns):
// gather + column constants, the restated kernels' per-column affine part, then the column kernel over `items` z items per column
// (dense: the voxels 0 .. items - 1; lattice sweeps: the kcount[column] listed voxels klist[column][.] * zstride).  vol_*: [nc][items].
static int run_column_batch(const ColumnSweep &cs, const PointSource &src, long long nc, int items, int zstride,
                            const int *kcount, const unsigned short *klist, float *vol_hr, float *vol_lr, bool trace_this) {
    hipStream_t st = cs.st;
    const char *blob = cs.blob;
    const MlpBlobHeader &h = cs.h;
    const int dtype = cs.dtype, kver = cs.kver, kver32 = cs.kver32;
    const bool restated = cs.restated;
    const double *mat = cs.mat;
    const float *calib = cs.calib;
    void *workspace = cs.workspace;
    float *F = (float *)workspace;
    float *CC = F + (size_t)C0PAD * COL_BATCH;
    float *cmask = CC + (size_t)CC_PAD * COL_BATCH;
    int rc = 0;
    const long long ncp = (long long)ceil_div(nc, 256) * 256;   // <= COL_BATCH; rows nc.. of CC are never read
    const int parts = split_parts();
    if ((rc = column_constants(st, src, nc, ncp, cs.feat_lr, cs.hl, cs.wl, cs.feat_hr, cs.hh, cs.wh, blob, h, F, CC, cmask))) return rc;
    GridArgs a;
    a.cc = CC;
    a.colmask = cmask;
    a.zvec = (const float *)(blob + h.zvec);
    a.core = blob + h.core;
    a.corex = blob + h.corex;
    a.b1frag = blob + h.b1frag;
    a.b1_inv_scale = 1.0f / (dtype == SURS_F16 ? B1FRAG_SCALE_F16 : B1FRAG_SCALE_BF16);
    a.w1t = blob + h.w1t;
    a.w1tx = blob + h.w1tx;
    a.rvec_lr = a.rvec_hr = nullptr;
    a.rld_lr = a.rld_hr = 0;
    a.colctr = nullptr;
    a.phase = 0;
    a.zmid = 0.0f;
    a.zstride = zstride;
    a.kcount = kcount;
    a.klist = klist;
    a.ovf_count = nullptr;
    a.ovf_list = nullptr;
    a.tile_list = cs.run_tiles;
    a.tile_count = cs.run_ntiles;
    a.colstart = cs.run_colstart;
    a.zpts = cs.run_z;
    if (restated) {
        // the affine part of layer 1: R = W1 (g . [a0 | w0z | w0p]) for the batch's columns, on the layer GEMM kernel
        if ((rc = g3_set_attributes())) return rc;
        char *v7 = (char *)workspace + col_base_bytes(COL_BATCH);
        unsigned short *g_lr = (unsigned short *)v7, *g_hr = g_lr + (size_t)COL_BATCH * 2 * D1 * 3;
        float *r_lr = (float *)(v7 + col_v7_image_bytes(COL_BATCH)), *r_hr = r_lr + (size_t)COL_BATCH * 2 * D2;
        float *zero_bias = r_hr + (size_t)COL_BATCH * 3 * D2;
        SURS_HIP_CHECK(hipMemsetAsync(zero_bias, 0, D2 * 4 + 256, st));   // + the column counter behind it
        a.colctr = (unsigned *)(zero_bias + D2);
        {
            // (any depth serves - the restated layer 1 is an identity for every choice -; point runs have no grid: world z = 0)
            const float zw = cs.run_z ? 0.0f : (float)(mat[10] * (double)cs.kmid + mat[11]);
            a.zmid = (calib[11] + calib[10] * zw) * cs.zmul / cs.zdiv;
        }
        const long long part_lr = 2LL * ncp * D1, part_hr = 3LL * ncp * D1;
        const dim3 pg((unsigned)(ncp / 4));   // four columns per workgroup, a wave per column
        // operand parts of this GEMM: the sweep's split (two f16 / three bf16 parts: fp32 grade) for the fp32-grade and the f16
        // kernel; ONE f16 part for the bf16 kernel - 11 significant bits, 8x what bf16 gives the rest of the classifier, a third
        // of the MFMA work, no measurable change of the bf16 sweep's error (SURS_R_PARTS=0: the split's parts there too)
        const int r_parts_env = option(OPT_R_PARTS);
        const int rparts = (dtype == SURS_BF16 && r_parts_env == 1) ? 1 : parts;
        if (rparts == 1)
            hipLaunchKernelGGL(colsum_prepare_kernel<1>, pg, dim3(256), 0, st, CC, (const float *)(blob + h.zvec), (int)ncp, a.zmid,
                               g_lr, part_lr, g_hr, part_hr);
        else if (rparts == 2)
            hipLaunchKernelGGL(colsum_prepare_kernel<2>, pg, dim3(256), 0, st, CC, (const float *)(blob + h.zvec), (int)ncp, a.zmid,
                               g_lr, part_lr, g_hr, part_hr);
        else
            hipLaunchKernelGGL(colsum_prepare_kernel<3>, pg, dim3(256), 0, st, CC, (const float *)(blob + h.zvec), (int)ncp, a.zmid,
                               g_lr, part_lr, g_hr, part_hr);
        SURS_LAUNCH_CHECK();
        if (ncp <= 1024 && option(OPT_RVEC_SMALL)) {
            // small batches (the ~ 100 runs of a point-runs call, small slabs, coarse octree levels): both classifiers in one launch of
            // 64-point x 128-row workgroups (rvec_small_kernel: the same bits as the 256 x 256-tile GEMM below, 4 - 6 workgroups of which
            // walk K = 1024 in 34 - 62 us each here)
            RvecSmallArgs ra;
            for (int m = 0; m < 2; ++m) {
                const int nvec = m ? 3 : 2;
                ra.w[m] = (const unsigned short *)(blob + (rparts == 3 ? h.wt3[m][1] : h.wt2[m][1]));
                ra.g[m] = m ? g_hr : g_lr;
                ra.g_part[m] = m ? part_hr : part_lr;
                ra.npm[m] = (long long)nvec * ncp;
                ra.r[m] = m ? r_hr : r_lr;
            }
            ra.wgs0 = (int)(ra.npm[0] / 64) * 4;
            const unsigned wgs = (unsigned)(ra.wgs0 + (ra.npm[1] / 64) * 4);
            if (rparts == 1) hipLaunchKernelGGL(rvec_small_kernel<1>, dim3(wgs), dim3(256), 0, st, ra);
            else if (rparts == 2) hipLaunchKernelGGL(rvec_small_kernel<2>, dim3(wgs), dim3(256), 0, st, ra);
            else hipLaunchKernelGGL(rvec_small_kernel<3>, dim3(wgs), dim3(256), 0, st, ra);
            SURS_LAUNCH_CHECK();
        } else
        for (int m = 0; m < 2; ++m) {
            const int nvec = m ? 3 : 2;
            const long long npm = (long long)nvec * ncp;
            SplitSeg s1 = {m ? g_hr : g_lr, m ? part_hr : part_lr, D1 / 16}, s2 = {nullptr, 0, 0};
            const int nb256 = (int)(npm / 256);
            float *R = m ? r_hr : r_lr;
            // output R[column][vector][512] (transposed; the kernel swaps the MFMA operands so that its stores run along the rows)
            if (rparts == 1)   // (the first part of the two-part weight image is f16(w))
                hipLaunchKernelGGL((gemm_x3g_kernel<8, 256, G3_F32_T, 1>), dim3(gemm_grid(D2 / 256, nb256)), dim3(512), g3_lds_bytes(256, 1),
                                   st, (const unsigned short *)(blob + h.wt2[m][1]), D2, D1, s1, s2, npm, (const float *)zero_bias, R,
                                   (long long)D2, (unsigned short *)nullptr, 0LL, nb256);
            else if (rparts == 2)
                hipLaunchKernelGGL((gemm_x3g_kernel<8, 256, G3_F32_T, 2>), dim3(gemm_grid(D2 / 256, nb256)), dim3(512), g3_lds_bytes(256, 2),
                                   st, (const unsigned short *)(blob + h.wt2[m][1]), D2, D1, s1, s2, npm, (const float *)zero_bias, R,
                                   (long long)D2, (unsigned short *)nullptr, 0LL, nb256);
            else
                hipLaunchKernelGGL((gemm_x3g_kernel<8, 256, G3_F32_T>), dim3(gemm_grid(D2 / 256, nb256)), dim3(512), g3_lds_bytes(256),
                                   st, (const unsigned short *)(blob + h.wt3[m][1]), D2, D1, s1, s2, npm, (const float *)zero_bias, R,
                                   (long long)D2, (unsigned short *)nullptr, 0LL, nb256);
            SURS_LAUNCH_CHECK();
        }
        a.rvec_lr = r_lr;
        a.rvec_hr = r_hr;
        a.rld_lr = a.rld_hr = 0;
    }
    a.vol_hr = vol_hr;
    a.vol_lr = vol_lr;
    a.ncols = (int)nc;
    a.rz = items;
    a.z0 = mat ? mat[11] : 0.0;
    a.dz = mat ? mat[10] : 0.0;
    a.c22 = calib[10];
    a.c23 = calib[11];
    a.zmul = cs.zmul;
    a.zdiv = cs.zdiv;
    const long long nwork = cs.run_tiles ? cs.run_ntiles_host : nc;   // work items the column counter hands out
    const unsigned grid = (unsigned)((nwork < cs.cus) ? nwork : cs.cus);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool prof;
    a.kstat = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_prof.mu);
        prof = g_prof.on && !kcount;   // (the bench's per-launch figures are those of dense sweeps)
        if (prof && restated) {
            int dev = 0;
            SURS_HIP_CHECK(hipGetDevice(&dev));
            unsigned long long *&ctr = g_prof.kstat[dev];
            if (!ctr) {
                SURS_HIP_CHECK(hipMalloc((void **)&ctr, sizeof(unsigned long long)));
                SURS_HIP_CHECK(hipMemset(ctr, 0, sizeof(unsigned long long)));
            }
            a.kstat = ctr;
            g_prof.tiles += 2.0 * (double)nc * (dtype == SURS_F32 ? (items + 63) / 64 : (items + 127) / 128);
        }
    }
    if (prof) {
        SURS_HIP_CHECK(hipEventCreate(&e0));
        SURS_HIP_CHECK(hipEventCreate(&e1));
        SURS_HIP_CHECK(hipEventRecord(e0, st));
    }
    if (dtype == SURS_F32) {
        if (kver32 == 11) {
            // two passes over the batch (lr, then hr on the lr occupancies just written): each pass streams ONE classifier's
            // split weights (2.6 MiB), which an XCD's 4 MiB L2 holds; both together do not fit and 8 % of the stream came from
            // HBM.  Only with whole 64-voxel tiles (the hr pass reads its tile's lr occupancies back from vol_lr, which has no
            // room for the voxels beyond the column's end that the one-pass form computes and classifies on); same bits.
            const int passes_env = option(OPT_GRID_F32_PASSES);
            const bool two = passes_env == 2 && items % 64 == 0 && !cs.run_tiles;
            for (int ph = two ? 1 : 0; ph <= (two ? 2 : 0); ++ph) {
                a.phase = ph;
                hipLaunchKernelGGL(grid_mlp_kernel_v11, dim3(grid), dim3(V11_THREADS), GRID11_LDS_BYTES, st, a);
            }
        } else
            hipLaunchKernelGGL(grid_mlp_kernel_v5, dim3(grid), dim3(256), GRID5_LDS_BYTES, st, a);
    } else if (kver == 12) {
        // two workgroups of four waves per compute unit
        // (its overflow list lies over the g-scaled vectors' split image, which the R GEMMs above were the last to read)
#ifdef SURS_V12_ONE_WG
        const unsigned grid12 = (unsigned)((nc < cs.cus) ? nc : cs.cus);
#else
        const unsigned grid12 = (unsigned)((nc < 2 * cs.cus) ? nc : 2 * cs.cus);
#endif
        a.ovf_count = a.colctr + 1;
        a.ovf_list = (int *)((char *)workspace + col_base_bytes(COL_BATCH));
        const int lds12 = GRID12_LDS_BYTES;
        if (dtype == SURS_BF16)
            hipLaunchKernelGGL(grid_mlp_kernel_v12<SURS_BF16>, dim3(grid12), dim3(V12_THREADS), lds12, st, a);
        else
            hipLaunchKernelGGL(grid_mlp_kernel_v12<SURS_F16>, dim3(grid12), dim3(V12_THREADS), lds12, st, a);
        SURS_LAUNCH_CHECK();
        // the tiles it left: kernel v10 over the list (an empty list costs one launch of workgroups that leave at once)
        GridArgs b = a;
        b.tile_list = a.ovf_list;
        b.tile_count = a.ovf_count;
        b.colctr = a.colctr + 2;
        if (dtype == SURS_BF16)
            hipLaunchKernelGGL(grid_mlp_kernel_v10<SURS_BF16>, dim3(grid), dim3(V10_THREADS), GRID10_LDS_BYTES, st, b);
        else
            hipLaunchKernelGGL(grid_mlp_kernel_v10<SURS_F16>, dim3(grid), dim3(V10_THREADS), GRID10_LDS_BYTES, st, b);
    } else if (kver == 10) {
        if (dtype == SURS_BF16)
            hipLaunchKernelGGL(grid_mlp_kernel_v10<SURS_BF16>, dim3(grid), dim3(V10_THREADS), GRID10_LDS_BYTES, st, a);
        else
            hipLaunchKernelGGL(grid_mlp_kernel_v10<SURS_F16>, dim3(grid), dim3(V10_THREADS), GRID10_LDS_BYTES, st, a);
    } else {
        if (dtype == SURS_BF16)
            hipLaunchKernelGGL(grid_mlp_kernel_v3<SURS_BF16>, dim3(grid), dim3(256), GRID3_LDS_BYTES, st, a);
        else
            hipLaunchKernelGGL(grid_mlp_kernel_v3<SURS_F16>, dim3(grid), dim3(256), GRID3_LDS_BYTES, st, a);
    }
    SURS_LAUNCH_CHECK();
#ifdef SURS_V12_HIST
    if (kver == 12 && dtype != SURS_F32) {
        static unsigned long long *hist = nullptr;
        if (!hist) {
            SURS_HIP_CHECK(hipMalloc((void **)&hist, 65 * 8));
            SURS_HIP_CHECK(hipMemset(hist, 0, 65 * 8));
            SURS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_v12_hist), &hist, sizeof(hist)));
            static unsigned long long *h2 = hist;
            atexit([] {
                unsigned long long v[65];
                if (hipMemcpy(v, h2, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return;
                unsigned long long tot = 0;
                for (int i = 0; i < 65; ++i) tot += v[i];
                fprintf(stderr, "v12 tiles by residual k-steps (total %llu):", tot);
                for (int i = 0; i < 65; ++i) if (v[i]) fprintf(stderr, " %d:%.4f", i, (double)v[i] / (double)tot);
                fprintf(stderr, "\n");
            });
        }
    }
#endif
#ifdef SURS_V3_TRACE
    if (trace_this && option(OPT_V3_TRACE)) {
        unsigned long long t[64];
        SURS_HIP_CHECK(hipStreamSynchronize(st));
        SURS_HIP_CHECK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_v3_trace), sizeof(t)));
        for (int m = 0; m < 2; ++m) {
            fprintf(stderr, "v3 trace MLP %d:", m);
            for (int i = 1; i < 10; ++i) fprintf(stderr, " %llu", t[16 * m + i] - t[16 * m + i - 1]);
            fprintf(stderr, "  total %llu\n", t[16 * m + 9] - t[16 * m]);
            if (restated) {
                fprintf(stderr, "   v7 layer 1 (since start): init issued %llu, list %llu, chunk 0: gathered %llu, residuals %llu, barrier %llu, mfma+barrier %llu; listed %llu\n",
                        t[16 * m + 10] - t[16 * m], t[16 * m + 1] - t[16 * m], t[16 * m + 11] - t[16 * m], t[16 * m + 12] - t[16 * m],
                        t[16 * m + 13] - t[16 * m], t[16 * m + 14] - t[16 * m], t[48 + m]);
            }
        }
        fprintf(stderr, "v3 trace between MLPs: %llu\n", t[16] - t[9]);
        if (kver == 12 && dtype != SURS_F32) {
            unsigned long long acc[2][16];
            SURS_HIP_CHECK(hipMemcpyFromSymbol(acc, HIP_SYMBOL(g_v12_acc), sizeof(acc)));
            for (int m = 0; m < 2; ++m) {
                const double n = (double)(acc[m][0] ? acc[m][0] : 1);
                double tot = 0;
                fprintf(stderr, "v12 mean cycles per item, MLP %d (%llu items, %.2f residual k-steps): ", m, acc[m][0], (double)acc[m][15] / n);
                for (int i = 1; i < 10; ++i) {
                    fprintf(stderr, " %.0f", (double)acc[m][i] / n);
                    tot += (double)acc[m][i] / n;
                }
                fprintf(stderr, "  total %.0f  [list | residuals + slices | acc2 init + barrier | stages 0-8 | stages 9-16 | y2 publish | barrier | layer 3 | layer 4]\n", tot);
            }
            fprintf(stderr, "v12 traced workgroup: %.0f cycles in %.1f us = %.3f GHz\n", (double)acc[0][13], (double)acc[0][14] / 100.0,
                    (double)acc[0][13] / ((double)acc[0][14] / 100.0) * 1e-3);
            unsigned long long zero[2][16] = {};
            SURS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_v12_acc), zero, sizeof(zero)));
        }
        const double cyc = (double)(t[42] - t[40]), us = (double)(t[43] - t[41]) / 100.0;
        const int tile = dtype == SURS_F32 ? 64 : 128;
        fprintf(stderr, "workgroup 0: %.0f cycles in %.1f us = %.3f GHz; %.0f cycles per %d-point tile\n", cyc, us,
                cyc / us * 1e-3, cyc / ((double)((nc + grid - 1) / grid) * ((items + tile - 1) / tile)), tile);
    }
#else
    (void)trace_this;
#endif
    if (prof) {
        SURS_HIP_CHECK(hipEventRecord(e1, st));
        std::lock_guard<std::mutex> lock(g_prof.mu);
        g_prof.ev.emplace_back(e0, e1);
        g_prof.pts.push_back((double)nc * items);
    }
    return 0;
}

// column mode needs: projected X,Y independent of k; world z a function of k only
static bool sweep_has_columns(const double *mat, const float *calib) {
    const float cX = (float)(calib[0] * mat[2] + calib[1] * mat[6] + calib[2] * mat[10]);
    const float cY = (float)(calib[4] * mat[2] + calib[5] * mat[6] + calib[6] * mat[10]);
    return !(cX != 0.0f || cY != 0.0f || mat[8] != 0.0 || mat[9] != 0.0 || calib[8] != 0.0f || calib[9] != 0.0f);
}

static int query_grid_impl(int i0, int i1, int ry, int rz, const double *mat, const float *calib, float zmul,
                           float zdiv, const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh,
                           const void *mlp_blob, int dtype, void *workspace, size_t workspace_bytes, float *vol_hr,
                           float *vol_lr, int kernel_call, void *stream) {
    SURS_REQUIRE(mat && calib && feat_lr && feat_hr && mlp_blob && workspace && vol_hr && vol_lr, "null argument");
    SURS_REQUIRE(i1 >= i0 && ry > 0 && rz > 0, "bad grid range");
    SURS_REQUIRE(dtype == SURS_F32 || dtype == SURS_BF16 || dtype == SURS_F16 || dtype == SURS_F32_GEMM, "unknown dtype %d", dtype);
    const bool force_gemm = dtype == SURS_F32_GEMM;
    if (force_gemm) dtype = SURS_F32;
    SURS_REQUIRE(workspace_bytes >= surs_query_grid_workspace_bytes(ry, rz, dtype), "workspace too small");
    if (i1 == i0) return 0;
    hipStream_t st = as_stream(stream);
    const MlpBlobHeader h = blob_layout((uint32_t)dtype);
    int rc = 0;
    PointSource src;
    memset(&src, 0, sizeof(src));
    src.ry = ry;
    src.rz = rz;
    for (int i = 0; i < 12; ++i) src.mat[i] = mat[i];
    fill_calib(src, calib, zmul, zdiv);
    const char *blob = (const char *)mlp_blob;
    const bool columns = sweep_has_columns(mat, calib);

    if (dtype == SURS_F32 && !(columns && grid_f32_use_columns() && !force_gemm)) {
        // general calibration: every voxel is its own point, the five layers are GEMMs on the split-bf16 layer kernels
        const long long total = (long long)(i1 - i0) * ry * rz;
        Fp32Workspace w = carve_fp32(workspace, GRID_BATCH);
        if ((rc = zero_pad_rows(st, w))) return rc;
        src.mode = 1;
        for (long long b0 = 0; b0 < total; b0 += GRID_BATCH) {
            const long long nb = (total - b0 < GRID_BATCH) ? total - b0 : GRID_BATCH;
            src.base = (long long)i0 * ry * rz + b0;
            // SURS_F32_GEMM is asked for when the f16 range does not suffice: three bf16 parts
            rc = run_points_fp32(st, src, nb, feat_lr, hl, wl, feat_hr, hh, wh, blob, h, w, vol_hr + b0, vol_lr + b0,
                                 nullptr, nullptr, force_gemm ? 3 : 0);
            if (rc) return rc;
        }
        return 0;
    }
    if (!columns)
        return fail(SURS_E_UNSUPPORTED,
                    "column kernel needs an axis-aligned orthographic sweep (X,Y independent of k); use SURS_F32");
    const long long ncols = (long long)(i1 - i0) * ry;
    ColumnSweep cs;
    cs.st = st;
    cs.blob = blob;
    cs.h = h;
    cs.dtype = dtype;
    resolve_column_kernels(kernel_call, cs.kver, cs.kver32);
    cs.restated = dtype == SURS_F32 ? cs.kver32 == 11 : (cs.kver == 10 || cs.kver == 12);
    cs.feat_lr = feat_lr; cs.hl = hl; cs.wl = wl;
    cs.feat_hr = feat_hr; cs.hh = hh; cs.wh = wh;
    cs.mat = mat; cs.calib = calib; cs.zmul = zmul; cs.zdiv = zdiv;
    cs.workspace = workspace;
    cs.cus = device_cus();
    cs.kmid = rz / 2;
    if ((rc = grid_set_attributes())) return rc;
    src.mode = 2;
    for (long long c0 = 0; c0 < ncols; c0 += COL_BATCH) {
        const long long nc = (ncols - c0 < COL_BATCH) ? ncols - c0 : COL_BATCH;
        src.base = (long long)i0 * ry + c0;
        if ((rc = run_column_batch(cs, src, nc, rz, 1, nullptr, nullptr, vol_hr + (size_t)c0 * rz, vol_lr + (size_t)c0 * rz, c0 == 0))) return rc;
    }
    return 0;
}


// ------------------------------------------------------------------------------------------------
// How many of the 1024 layer-0 channels would the restated column kernels list per z tile on this sweep?  Evaluated on the
// columns of ONE axis-0 plane of the grid (any slab of the same grid gives the same answer, so the ranks of a sharded sweep
// agree): mean listed channels per (column, tile) for the lr classifier and an upper bound (p_lr range taken as [0, 1]) for
// the hr classifier.  listed[0] = listed[1] = -1 where the column kernels do not apply (general calibration).  Synchronises
// the stream.  The host uses it to fall back to the dense column kernels on fields that list most channels.
// ------------------------------------------------------------------------------------------------
extern "C" int surs_query_grid_probe(int i_plane, int ry, int rz, int tile, const double *mat, const float *calib, float zmul,
                                     float zdiv, const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh,
                                     const void *mlp_blob, void *workspace, size_t workspace_bytes, float *listed, void *stream) {
    SURS_REQUIRE(mat && calib && feat_lr && feat_hr && mlp_blob && workspace && listed, "null argument");
    SURS_REQUIRE(ry > 0 && rz > 0 && i_plane >= 0 && (tile == 64 || tile == 128), "bad probe arguments");
    SURS_REQUIRE(ry <= COL_BATCH, "probe plane wider than a column batch");
    SURS_REQUIRE(workspace_bytes >= col_ws_bytes(COL_BATCH), "workspace too small");
    listed[0] = listed[1] = -1.0f;
    const float cX = (float)(calib[0] * mat[2] + calib[1] * mat[6] + calib[2] * mat[10]);
    const float cY = (float)(calib[4] * mat[2] + calib[5] * mat[6] + calib[6] * mat[10]);
    if (cX != 0.0f || cY != 0.0f || mat[8] != 0.0 || mat[9] != 0.0 || calib[8] != 0.0f || calib[9] != 0.0f) return 0;
    hipStream_t st = as_stream(stream);
    const MlpBlobHeader h = blob_layout(SURS_BF16);
    const char *blob = (const char *)mlp_blob;
    PointSource src;
    memset(&src, 0, sizeof(src));
    src.ry = ry;
    src.rz = rz;
    for (int i = 0; i < 12; ++i) src.mat[i] = mat[i];
    fill_calib(src, calib, zmul, zdiv);
    src.mode = 2;
    src.base = (long long)i_plane * ry;
    float *F = (float *)workspace;
    float *CC = F + (size_t)C0PAD * COL_BATCH;
    float *cmask = CC + (size_t)CC_PAD * COL_BATCH;
    const long long nc = ry, ncp = (long long)ceil_div(nc, 256) * 256;
    int rc = column_constants(st, src, nc, ncp, feat_lr, hl, wl, feat_hr, hh, wh, blob, h, F, CC, cmask);
    if (rc) return rc;
    unsigned long long *ctr = (unsigned long long *)((char *)workspace + col_base_bytes(COL_BATCH));
    SURS_HIP_CHECK(hipMemsetAsync(ctr, 0, 16, st));
    const float zw = (float)(mat[10] * (double)(rz / 2) + mat[11]);
    const float zmid = (calib[11] + calib[10] * zw) * zmul / zdiv;
    hipLaunchKernelGGL(probe_count_kernel, dim3((unsigned)nc), dim3(256), 0, st, CC, (const float *)(blob + h.zvec), rz, tile, mat[11], mat[10],
                       calib[10], calib[11], zmul, zdiv, zmid, ctr);
    SURS_LAUNCH_CHECK();
    unsigned long long v[2] = {0, 0};
    SURS_HIP_CHECK(hipMemcpyAsync(v, ctr, 16, hipMemcpyDeviceToHost, st));
    SURS_HIP_CHECK(hipStreamSynchronize(st));
    const double denom = (double)nc * (double)((rz + tile - 1) / tile);
    listed[0] = (float)((double)v[0] / denom);
    listed[1] = (float)((double)v[1] / denom);
    return 0;
}


// ------------------------------------------------------------------------------------------------
// octree support (lib/sdf.py:55-120): evaluate the grid voxels listed in idx[] (fp32 arithmetic)
// ------------------------------------------------------------------------------------------------
extern "C" int surs_query_grid_indexed(const long long *idx, int n, int ry, int rz, const double *mat, const float *calib,
                                       float zmul, float zdiv, const float *feat_lr, int hl, int wl, const float *feat_hr,
                                       int hh, int wh, const void *mlp_blob, void *workspace, size_t workspace_bytes,
                                       float *pred_hr, float *pred_lr, void *stream) {
    SURS_REQUIRE(n >= 0, "negative point count");
    if (n == 0) return 0;
    SURS_REQUIRE(idx && mat && calib && feat_lr && feat_hr && mlp_blob && workspace && pred_hr && pred_lr, "null argument");
    hipStream_t st = as_stream(stream);
    const long long np = (long long)ceil_div(n, 256) * 256;
    SURS_REQUIRE(workspace_bytes >= fp32_ws_bytes(np), "workspace too small: need %zu bytes", fp32_ws_bytes(np));
    const MlpBlobHeader h = blob_layout(SURS_BF16);
    PointSource src;
    memset(&src, 0, sizeof(src));
    src.mode = 3;
    src.idx = idx;
    src.ry = ry;
    src.rz = rz;
    for (int i = 0; i < 12; ++i) src.mat[i] = mat[i];
    fill_calib(src, calib, zmul, zdiv);
    Fp32Workspace w = carve_fp32(workspace, np);
    int rc = zero_pad_rows(st, w);
    if (rc) return rc;
    return run_points_fp32(st, src, n, feat_lr, hl, wl, feat_hr, hh, wh, (const char *)mlp_blob, h, w, pred_hr, pred_lr, nullptr,
                           nullptr);
}

// ------------------------------------------------------------------------------------------------
// One level of the octree sweep on the COLUMN kernel (lib/sdf.py:68-74 for an axis-aligned sweep).  The lattice points of stride
// `reso` form columns along axis 2 that share their image position, exactly like the dense sweep's columns: the per-column
// constants and the restated layer 1 apply unchanged.  A column's work items are its DIRTY lattice points, compacted in ascending
// k (a surface crosses a column in a few short runs: whole z tiles would be a tenth full), 64 per tile; the reference evaluates
// exactly these points (lib/sdf.py:68-73).
//   select : one wave per lattice column counts its dirty lattice points; columns with any are appended to the list (any order:
//            a column's values do not depend on its place in the list)
//   klist  : per batch of listed columns, the lattice indices of their dirty points
//   sweep  : run_column_batch over the listed columns (mode 4 gather), items = klist, zstride = reso, compact outputs [column][nl]
//   scatter: sdf[voxel] = value, dirty[voxel] = 0 for the listed points
// ------------------------------------------------------------------------------------------------
// One workgroup takes 64 consecutive lattice columns (a wave 16 of them, one after the other) and appends the ones that hold dirty
// points with ONE atomic per counter (an atomic per listed column serialised on the three counters: 3 ms per level at 512^3).
__global__ __launch_bounds__(256) void lattice_select_kernel(const unsigned char *__restrict__ dirty, int R, int reso, int nl,
                                                             int *__restrict__ cols, int *__restrict__ counts,
                                                             unsigned long long *__restrict__ counters /* [0] columns, [1] dirty points, [2] tiles */) {
    __shared__ int cnt[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long ncolumns = (long long)nl * nl;
    const long long lc0 = (long long)blockIdx.x * 64;
    for (int q = 0; q < 16; ++q) {
        const long long lc = lc0 + wave * 16 + q;   // lattice column = li * nl + lj
        unsigned npts = 0;
        if (lc < ncolumns) {
            const int li = (int)(lc / nl), lj = (int)(lc - (long long)li * nl);
            const unsigned char *row = dirty + ((long long)(li * reso) * R + (long long)(lj * reso)) * R;
            if (reso == 1 && (R & 7) == 0) {
                // the finest level reads every byte of the mask (134 MB at 512^3): 8 bytes per lane (the mask holds 0 / 1: the
                // word's population count is the number of dirty voxels)
                unsigned c = 0;
                for (int k = lane * 8; k < nl; k += 512) c += (unsigned)__popcll(*reinterpret_cast<const unsigned long long *>(row + k));
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
                npts = c;
            } else {
                for (int t = 0; t * 64 < nl; ++t) {
                    const int k = t * 64 + lane;
                    npts += (unsigned)__popcll(__ballot(k < nl && row[(long long)k * reso] != 0));
                }
            }
        }
        if (lane == 0) cnt[wave * 16 + q] = (int)npts;
    }
    __syncthreads();
    if (wave == 0) {
        const int n = cnt[lane];
        const unsigned long long b = __ballot(n > 0);
        unsigned pts = (unsigned)n, tiles = (unsigned)((n + 63) / 64);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            pts += __shfl_xor(pts, d);
            tiles += __shfl_xor(tiles, d);
        }
        unsigned long long mine = 0;
        if (lane == 0 && b) {
            mine = atomicAdd(counters, (unsigned long long)__popcll(b));
            atomicAdd(counters + 1, (unsigned long long)pts);
            atomicAdd(counters + 2, (unsigned long long)tiles);
        }
        const unsigned long long base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(mine >> 32)) << 32) |
                                        (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(mine & 0xffffffffull));
        if (n > 0) {
            const long long lc = lc0 + lane;
            const int li = (int)(lc / nl), lj = (int)(lc - (long long)li * nl);
            const unsigned long long slot = base + (unsigned long long)__popcll(b & ((1ull << lane) - 1ull));
            cols[slot] = (int)((long long)(li * reso) * R + (long long)(lj * reso));   // full-resolution column index i * R + j
            counts[slot] = n;
        }
    }
}

// the listed lattice points of a batch of listed columns, ascending: klist[c][e] = lattice index k of the e-th dirty point
__global__ __launch_bounds__(256) void lattice_klist_kernel(const unsigned char *__restrict__ dirty, const int *__restrict__ cols, long long ncols,
                                                            int R, int reso, int nl, unsigned short *__restrict__ klist) {
    const int lane = threadIdx.x & 63;
    const long long c = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= ncols) return;
    const unsigned char *row = dirty + (long long)cols[c] * R;
    unsigned short *out = klist + c * nl;
    int base = 0;
    for (int t = 0; t * 64 < nl; ++t) {
        const int k = t * 64 + lane;
        const bool hit = k < nl && row[(long long)k * reso] != 0;
        const unsigned long long b = __ballot(hit);
        if (hit) out[base + __popcll(b & ((1ull << lane) - 1ull))] = (unsigned short)k;
        base += __popcll(b);
    }
}

__global__ __launch_bounds__(256) void lattice_scatter_kernel(const int *__restrict__ cols, const int *__restrict__ counts,
                                                              const unsigned short *__restrict__ klist, long long ncols, int R, int reso, int nl,
                                                              const float *__restrict__ vh, const float *__restrict__ vl,
                                                              double *__restrict__ sdf_hr, double *__restrict__ sdf_lr,
                                                              unsigned char *__restrict__ dirty) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= ncols * nl) return;
    const long long c = t / nl;
    const int e = (int)(t - c * nl);
    if (e >= counts[c]) return;
    const long long f = (long long)cols[c] * R + (long long)klist[t] * reso;
    sdf_hr[f] = (double)vh[t];
    sdf_lr[f] = (double)vl[t];
    dirty[f] = 0;
}

static size_t lattice_list_bytes(int R) { return align_up((size_t)R * R * 8 + 64, 256); }

extern "C" size_t surs_octree_columns_workspace_bytes(int R) {
    return col_ws_bytes(COL_BATCH) + lattice_list_bytes(R) + (size_t)COL_BATCH * R * (2 * sizeof(float) + sizeof(unsigned short)) + 512;
}

extern "C" int surs_octree_level_columns_dt(double *sdf_hr, double *sdf_lr, unsigned char *dirty, int R, int reso, int kmid,
                                            const double *mat, const float *calib, float zmul, float zdiv, const float *feat_lr, int hl,
                                            int wl, const float *feat_hr, int hh, int wh, const void *mlp_blob, int dtype, void *workspace,
                                            size_t workspace_bytes, long long *counts, void *stream);

extern "C" int surs_octree_level_columns(double *sdf_hr, double *sdf_lr, unsigned char *dirty, int R, int reso, int kmid,
                                         const double *mat, const float *calib, float zmul, float zdiv, const float *feat_lr, int hl,
                                         int wl, const float *feat_hr, int hh, int wh, const void *mlp_blob, void *workspace,
                                         size_t workspace_bytes, long long *counts, void *stream) {
    return surs_octree_level_columns_dt(sdf_hr, sdf_lr, dirty, R, reso, kmid, mat, calib, zmul, zdiv, feat_lr, hl, wl, feat_hr, hh, wh,
                                        mlp_blob, SURS_F32, workspace, workspace_bytes, counts, stream);
}

// ... with the arithmetic of the level's evaluator chosen: SURS_F32 = kernel v11 (fp32-grade: the default of surs_octree_level_columns);
// SURS_BF16 / SURS_F16 = kernel v10 on the blob's 16-bit cores (tiles of 128 listed points) - the octree sweep of `--precision bf16 |
// fp16`, whose dense sweep runs the same arithmetic.
extern "C" int surs_octree_level_columns_dt(double *sdf_hr, double *sdf_lr, unsigned char *dirty, int R, int reso, int kmid,
                                            const double *mat, const float *calib, float zmul, float zdiv, const float *feat_lr, int hl,
                                            int wl, const float *feat_hr, int hh, int wh, const void *mlp_blob, int dtype, void *workspace,
                                            size_t workspace_bytes, long long *counts, void *stream) {
    SURS_REQUIRE(dtype == SURS_F32 || dtype == SURS_BF16 || dtype == SURS_F16, "unknown dtype %d", dtype);
    SURS_REQUIRE(sdf_hr && sdf_lr && dirty && mat && calib && feat_lr && feat_hr && mlp_blob && workspace, "null argument");
    SURS_REQUIRE(R > 0 && reso > 0 && R <= 32768, "bad grid");
    const int nl = (R + reso - 1) / reso;
    SURS_REQUIRE(nl <= 65535, "lattice indices are 16-bit");
    SURS_REQUIRE(workspace_bytes >= surs_octree_columns_workspace_bytes(R), "workspace too small");
    if (!sweep_has_columns(mat, calib))
        return fail(SURS_E_UNSUPPORTED, "the column kernel needs an axis-aligned orthographic sweep: use surs_octree_select + "
                                        "surs_query_grid_indexed + surs_octree_scatter");
    hipStream_t st = as_stream(stream);
    int rc = 0;
    char *base = (char *)workspace + col_ws_bytes(COL_BATCH);
    int *cols = (int *)base;
    int *kcnt = cols + (size_t)nl * nl;
    unsigned long long *ctr = (unsigned long long *)(base + lattice_list_bytes(R) - 64);
    float *vh = (float *)(base + lattice_list_bytes(R)), *vl = vh + (size_t)COL_BATCH * nl;
    unsigned short *klist = (unsigned short *)(vl + (size_t)COL_BATCH * nl);
    SURS_HIP_CHECK(hipMemsetAsync(ctr, 0, 32, st));
    hipLaunchKernelGGL(lattice_select_kernel, dim3((unsigned)ceil_div((long long)nl * nl, 64)), dim3(256), 0, st, dirty, R, reso, nl, cols,
                       kcnt, ctr);
    SURS_LAUNCH_CHECK();
    unsigned long long host[3] = {0, 0, 0};
    SURS_HIP_CHECK(hipMemcpyAsync(host, ctr, 24, hipMemcpyDeviceToHost, st));
    SURS_HIP_CHECK(hipStreamSynchronize(st));
    if (counts) {
        counts[0] = (long long)host[1];   // dirty lattice points: what the reference evaluates at this level
        counts[1] = (long long)host[0];   // lattice columns that hold them
        counts[2] = (long long)host[2];   // 64-point tiles the column kernel evaluates for them
    }
    const long long ncols = (long long)host[0];
    if (ncols == 0) return 0;
    ColumnSweep cs;
    cs.st = st;
    cs.blob = (const char *)mlp_blob;
    cs.h = blob_layout((uint32_t)dtype);
    cs.dtype = dtype;
    cs.kver = 10;     // (the eight-wave kernel: the streamed kernel v12 and its overflow hand-over have not been run on lattice lists)
    cs.kver32 = 11;   // (the dense-layer-1 kernels v3 / v5 have no strided / masked form)
    cs.restated = true;
    cs.feat_lr = feat_lr; cs.hl = hl; cs.wl = wl;
    cs.feat_hr = feat_hr; cs.hh = hh; cs.wh = wh;
    cs.mat = mat; cs.calib = calib; cs.zmul = zmul; cs.zdiv = zdiv;
    cs.workspace = workspace;
    cs.cus = device_cus();
    cs.kmid = kmid;
    if ((rc = grid_set_attributes())) return rc;
    PointSource src;
    memset(&src, 0, sizeof(src));
    src.mode = 4;
    src.ry = R;
    src.rz = R;
    for (int i = 0; i < 12; ++i) src.mat[i] = mat[i];
    fill_calib(src, calib, zmul, zdiv);
    for (long long c0 = 0; c0 < ncols; c0 += COL_BATCH) {
        const long long nc = (ncols - c0 < COL_BATCH) ? ncols - c0 : COL_BATCH;
        src.cols = cols + c0;
        hipLaunchKernelGGL(lattice_klist_kernel, dim3((unsigned)ceil_div(nc, 4)), dim3(256), 0, st, (const unsigned char *)dirty, (const int *)(cols + c0),
                           nc, R, reso, nl, klist);
        SURS_LAUNCH_CHECK();
        if ((rc = run_column_batch(cs, src, nc, nl, reso, kcnt + c0, klist, vh, vl, false))) return rc;
        hipLaunchKernelGGL(lattice_scatter_kernel, dim3((unsigned)ceil_div(nc * nl, 256)), dim3(256), 0, st, (const int *)(cols + c0),
                           (const int *)(kcnt + c0), (const unsigned short *)klist, nc, R, reso, nl, (const float *)vh, (const float *)vl, sdf_hr,
                           sdf_lr, dirty);
        SURS_LAUNCH_CHECK();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Point runs: the column kernels behind surs_query_points' signature.
//
// The reference's dense sweep (lib/sdf.py:32-45 batch_eval -> eval_func -> query_mr / query_sr, lib/mesh_util.py:20-28) hands the
// facade consecutive pieces of the flattened grid: 50 000 points per call, z fastest, i.e. ~ 98 runs of up to 512 points that share
// their (x, y).  A run is a column of the sweep - same projected image position, same 320 gathered features - so the restated
// column kernels (v10 / v11) serve it: per-column constants from ONE gather + GEMM per run instead of per point, layer 1 as the
// affine part + the listed channels' residuals.  Nothing about a grid is assumed: the runs are found in the data (maximal sequences
// of points with bit-equal x and y, cut at PR_CAP points, z monotonic inside every run - the kernels take a tile's z range from its
// ends), every point's own z is read from the caller's array, results go straight to the caller's prediction arrays.  Arrays
// without such runs (random samples: fewer than a quarter tile of points per run on average) are refused (*columns = 0, nothing written) and
// the caller takes surs_query_points.
// ------------------------------------------------------------------------------------------------
static const int PR_CHUNK = 262144;   // points per call
static const int PR_CAP = 4096;       // points per column: a longer run is cut into several (one more gather row each)
// Average points per run below which the layer kernels are the faster evaluator: a quarter of the column kernel's tile - 16 for v11
// (64 slots), 32 for v10 (128).  Measured on 50 000 points as runs of 8 - 24 / 12 - 36 / 16 - 48 / 24 - 72 points (the dirty lattice points
// of the reference's octree levels come as ~ 25 per column; tools/dev/short_runs_time.py), column kernels against layer kernels:
// fp32-grade 0.87 / 0.70 / 0.63 / 0.55 against 0.94 ms, bf16 0.81 / 0.64 / 0.51 / 0.41 against 0.55 ms.
static inline __host__ __device__ int pr_min_run(int tile) { return tile / 4; }
// Do two prediction arrays hold a non-finite value?  One workgroup, one word written (no zeroing in front of it): what the facade asks
// after every query (an activation beyond the f16 range of the two-part split shows up as NaN: model.SuRSNet._finite_or_wide) - until
// round 6 two torch reductions and an addition per call.
__device__ __forceinline__ int nonfinite_bits(unsigned u) { return (u & 0x7f800000u) == 0x7f800000u; }
__device__ __forceinline__ int nonfinite_bits(const u32x4 &v) {
    return nonfinite_bits(v[0]) | nonfinite_bits(v[1]) | nonfinite_bits(v[2]) | nonfinite_bits(v[3]);
}
// (one workgroup is a latency chain: VEC = 16-byte loads, eight of them in flight per lane - 19 -> 6 us for two arrays of 50 000)
template <bool VEC>
__global__ __launch_bounds__(1024) void nonfinite_kernel(const float *__restrict__ a, const float *__restrict__ b, long long n, int *__restrict__ flag) {
    __shared__ int any;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    int bad = 0;
    const unsigned *ua = reinterpret_cast<const unsigned *>(a), *ub = reinterpret_cast<const unsigned *>(b);
    long long done = 0;
    if (VEC) {
        const long long n4 = n >> 2;
        const u32x4 *a4 = reinterpret_cast<const u32x4 *>(a), *b4 = reinterpret_cast<const u32x4 *>(b);
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (long long i = threadIdx.x; i < n4; i += 4096) {
            u32x4 va[4], vb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long long j = i + 1024 * k;
                va[k] = j < n4 ? a4[j] : zero;
                vb[k] = (b && j < n4) ? b4[j] : zero;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) bad |= nonfinite_bits(va[k]) | nonfinite_bits(vb[k]);
        }
        done = n4 << 2;
    }
    for (long long i = done + threadIdx.x; i < n; i += 1024) bad |= nonfinite_bits(ua[i]) | (b ? nonfinite_bits(ub[i]) : 0);
    if (__ballot(bad) && (threadIdx.x & 63) == 0) any = 1;   // (benign race: every writer writes 1)
    __syncthreads();
    if (threadIdx.x == 0) *flag = any;
}

extern "C" int surs_nonfinite(const float *a, const float *b, long long n, int *flag, void *stream) {
    SURS_REQUIRE(a && flag && n >= 0, "bad argument");
    const bool vec = ((reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(b)) & 15) == 0;
    if (vec) hipLaunchKernelGGL(nonfinite_kernel<true>, dim3(1), dim3(1024), 0, as_stream(stream), a, b, n, flag);
    else hipLaunchKernelGGL(nonfinite_kernel<false>, dim3(1), dim3(1024), 0, as_stream(stream), a, b, n, flag);
    SURS_LAUNCH_CHECK();
    return 0;
}

static const int PR_THREADS = 1024;

__device__ __forceinline__ int pr_wave_scan_sum(int v, int lane) {   // inclusive, 64 lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ int pr_wave_scan_max(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d);
        if (lane >= d) v = max(v, o);
    }
    return v;
}

// ONE workgroup walks the array in blocks of 4096 points: wave w takes 256 consecutive points of a block as four steps of 64 (lane =
// point: coalesced loads, and a wave's scans are ballots and bit counts - no cross-lane traffic); what crosses the waves goes
// through 16 LDS words twice per block (the last natural head, then the number of heads), what crosses the blocks is carried in
// registers.  colstart[c] / kcount[c]: first point and length of column c; tiles[2 j], tiles[2 j + 1]: column and z tile (of `tile`
// points) of work item j; meta = {columns, tiles, ascending violated, descending violated}.  An array with more than one column
// per quarter tile of points is not worth the column kernels: lengths and work items are skipped.
__global__ __launch_bounds__(PR_THREADS) void point_runs_kernel(const float *__restrict__ px, const float *__restrict__ py,
                                                                const float *__restrict__ pz, int n, int tile,
                                                                int *__restrict__ colstart, int *__restrict__ kcount,
                                                                int *__restrict__ tiles, int *__restrict__ meta) {
    constexpr int V = 4, NW = PR_THREADS / 64, BS = PR_THREADS * V;
    static_assert((PR_CAP & (PR_CAP - 1)) == 0, "the cut is a bit mask");
    __shared__ int tmpm[NW], tmps[NW];
    __shared__ int viol[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long lt = (1ull << lane) - 1ull, le = lt | (1ull << lane);
    if (tid < 2) viol[tid] = 0;
    int carry_r = -1, carry_c = 0, va = 0, vd = 0;
    // The next block's loads fly while this block is scanned (the blocks are a serial chain of memory round trips otherwise).
    // __syncthreads() would wait for them - its release fence drains the vector-memory counter -, so the barriers inside the loop
    // are bare: what crosses the waves there lives in LDS only.
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    float xn[V], yn[V], zn[V], xq[V], yq[V], zq[V];   // the points of the wave's four steps and their predecessors
    auto fetch = [&](int b0) {
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const int i = b0 + wave * 256 + q * 64 + lane;
            const bool ok = i < n, okp = i >= 1 && i < n;
            xn[q] = ok ? px[i] : 0.0f;
            yn[q] = ok ? py[i] : 0.0f;
            zn[q] = ok ? pz[i] : 0.0f;
            xq[q] = okp ? px[i - 1] : 0.0f;
            yq[q] = okp ? py[i - 1] : 0.0f;
            zq[q] = okp ? pz[i - 1] : 0.0f;
        }
    };
    fetch(0);
    __syncthreads();   // viol zeroed
    for (int b0 = 0; b0 < n; b0 += BS) {
        const int w0 = b0 + wave * 256;
        // natural heads: the point's (x, y) differs from its predecessor's
        unsigned long long mnh[V];
        bool zup[V], zdn[V];
        int lastnh = -1;
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const int i = w0 + q * 64 + lane;
            const bool nh = i < n && (i == 0 || xn[q] != xq[q] || yn[q] != yq[q]);
            mnh[q] = __ballot(nh);
            zup[q] = zn[q] < zq[q] || zn[q] != zn[q];   // (NaN: neither order holds)
            zdn[q] = zn[q] > zq[q] || zn[q] != zn[q];
            if (mnh[q]) lastnh = w0 + q * 64 + 63 - __clzll(mnh[q]);
        }
        if (b0 + BS < n) fetch(b0 + BS);
        if (lane == 0) tmpm[wave] = lastnh;
        lds_barrier();
        int r = carry_r, blockmax = -1;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int t = tmpm[w];
            if (w < wave) r = max(r, t);
            blockmax = max(blockmax, t);
        }
        // heads: natural, or PR_CAP points into a run; z monotonic inside the columns
        unsigned long long mh[V];
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const int i = w0 + q * 64 + lane;
            const unsigned long long mine = mnh[q] & le;
            const int rl = mine ? w0 + q * 64 + 63 - __clzll(mine) : r;     // the run that holds this lane's point starts here
            const bool head = i < n && (((i - rl) & (PR_CAP - 1)) == 0);   // (a natural head: i == rl)
            mh[q] = __ballot(head);
            cnt += __popcll(mh[q]);
            if (i < n && !head) {
                va |= zup[q] ? 1 : 0;
                vd |= zdn[q] ? 1 : 0;
            }
            if (mnh[q]) r = w0 + q * 64 + 63 - __clzll(mnh[q]);
        }
        if (lane == 0) tmps[wave] = cnt;
        lds_barrier();
        int c = carry_c, total = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int t = tmps[w];
            if (w < wave) c += t;
            total += t;
        }
#pragma unroll
        for (int q = 0; q < V; ++q) {
            if ((mh[q] >> lane) & 1ull) colstart[c + __popcll(mh[q] & lt)] = w0 + q * 64 + lane;
            c += __popcll(mh[q]);
        }
        carry_r = max(carry_r, blockmax);
        carry_c += total;
        lds_barrier();   // tmpm / tmps are written again in the next block
    }
    if (va) atomicOr(&viol[0], 1);
    if (vd) atomicOr(&viol[1], 1);
    const int ncols = carry_c;
    __threadfence_block();
    __syncthreads();   // colstart (global) and viol visible to the workgroup
    int ntiles = 0;
    if ((long long)ncols * pr_min_run(tile) <= n) {
        int carry_t = 0;
        for (int c0 = 0; c0 < ncols; c0 += PR_THREADS) {
            const int c = c0 + tid;
            int k = 0, nt = 0;
            if (c < ncols) {
                k = (c + 1 < ncols ? colstart[c + 1] : n) - colstart[c];
                kcount[c] = k;
                nt = (k + tile - 1) / tile;
            }
            const int vs = pr_wave_scan_sum(nt, lane);
            if (lane == 63) tmps[wave] = vs;
            __syncthreads();
            int addt = 0, total = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int t = tmps[w];
                if (w < wave) addt += t;
                total += t;
            }
            int j = carry_t + addt + vs - nt;
            for (int zc = 0; zc < nt; ++zc) {
                tiles[2 * j] = c;
                tiles[2 * j + 1] = zc;
                ++j;
            }
            carry_t += total;
            __syncthreads();
        }
        ntiles = carry_t;
    }
    if (tid == 0) {
        meta[0] = ncols;
        meta[1] = ntiles;
        meta[2] = viol[0];
        meta[3] = viol[1];
    }
}

static size_t point_runs_list_bytes() { return align_up((size_t)PR_CHUNK * 4 * sizeof(int) + 256, 256); }

// The run finder alone (tests, diagnostics): colstart / kcount [n] ints, tiles [2 n] ints, meta [4] ints = {columns, work items (0 when
// the array holds more than one column per tile / 4 points: lengths and work items are then not written), ascending violated, descending
// violated}; tile = 64 | 128 points per work item.  All device pointers; no synchronisation.
extern "C" int surs_point_runs(const float *points, long long ld, int n, int tile, int *colstart, int *kcount, int *tiles, int *meta,
                               void *stream) {
    SURS_REQUIRE(points && colstart && kcount && tiles && meta, "null argument");
    SURS_REQUIRE(n > 0 && n <= PR_CHUNK && ld >= n && (tile == 64 || tile == 128), "1 .. %d points, row pitch ld >= n, tile 64 or 128", PR_CHUNK);
    hipLaunchKernelGGL(point_runs_kernel, dim3(1), dim3(PR_THREADS), 0, as_stream(stream), points, points + ld, points + 2 * ld, n, tile,
                       colstart, kcount, tiles, meta);
    SURS_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t surs_query_points_columns_workspace_bytes(void) { return col_ws_bytes(COL_BATCH) + point_runs_list_bytes(); }

extern "C" int surs_query_points_columns(const float *points, long long ld, int n, const float *calib, float zmul, float zdiv,
                                         const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh,
                                         const void *mlp_blob, int dtype, void *workspace, size_t workspace_bytes, float *pred_hr,
                                         float *pred_lr, int *columns, void *stream) {
    SURS_REQUIRE(columns, "null argument");
    *columns = 0;
    SURS_REQUIRE(n >= 0 && n <= PR_CHUNK && ld >= n, "at most %d points per call, [3][n] with row pitch ld >= n", PR_CHUNK);
    if (n == 0) return 0;
    SURS_REQUIRE(points && calib && feat_lr && feat_hr && mlp_blob && workspace && pred_hr && pred_lr, "null argument");
    SURS_REQUIRE(dtype == SURS_F32 || dtype == SURS_BF16 || dtype == SURS_F16, "unknown dtype %d", dtype);
    SURS_REQUIRE(workspace_bytes >= surs_query_points_columns_workspace_bytes(), "workspace too small");
    // a run is a column only if its image position does not depend on z and its depth only on z
    if (calib[2] != 0.0f || calib[6] != 0.0f || calib[8] != 0.0f || calib[9] != 0.0f) return 0;
    if (n < 2048) return 0;   // (the per-call preparation outweighs the point kernels' work below that)
    hipStream_t st = as_stream(stream);
    int rc = 0;
    char *lists = (char *)workspace + col_ws_bytes(COL_BATCH);
    int *colstart = (int *)lists, *kcount = colstart + PR_CHUNK, *tiles = kcount + PR_CHUNK;
    int *meta = (int *)(lists + point_runs_list_bytes() - 256);
    const int tile = dtype == SURS_F32 ? 64 : 128;
    hipLaunchKernelGGL(point_runs_kernel, dim3(1), dim3(PR_THREADS), 0, st, points, points + ld, points + 2 * ld, n, tile, colstart,
                       kcount, tiles, meta);
    SURS_LAUNCH_CHECK();
    // The host needs the run count (how many columns the preparation launches cover, whether the array holds runs at all).  Reading
    // it back in the MIDDLE of the call left the GPU idle for the round trip plus the host's eight launches, every call (rounds 5's
    // form).  A caller that sends arrays of one length - the reference's sweep loop: 50 000 consecutive grid
    // points per call - gets the same counts (+- 1 column) call after call, so the launches are sized from the PREVIOUS call's counts
    // (a quarter more) and enqueued at once behind the run finder; the four words - copied to pinned memory behind the run finder -
    // are read AFTER the enqueue, when they have long arrived: the wait is on an event recorded right behind the copy, not on the
    // stream.  The kernels take the true counts from device memory where it matters (the work-item count of the column kernel); rows
    // of the preparation beyond the true run count are computed from stale list entries (clamped into the array) and never read.
    // If the array holds more runs than guessed the batch runs again with the true counts (same stream: it overwrites); if it holds
    // no runs the call reports 0 columns as before and the caller's layer kernels overwrite the outputs.
    struct Speculation {
        int *host = nullptr;
        hipEvent_t ev = nullptr;
        int n = -1, tile = 0;
        long long ncols = 0, ntiles = 0;
    };
    static thread_local Speculation sp;
    if (!sp.host) SURS_HIP_CHECK(hipHostMalloc((void **)&sp.host, 4 * sizeof(int), hipHostMallocDefault));
    if (!sp.ev) SURS_HIP_CHECK(hipEventCreateWithFlags(&sp.ev, hipEventDisableTiming));
    int *host = sp.host;
    host[0] = host[1] = host[2] = host[3] = 0;
    SURS_HIP_CHECK(hipMemcpyAsync(host, meta, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    SURS_HIP_CHECK(hipEventRecord(sp.ev, st));
    ColumnSweep cs;
    cs.st = st;
    cs.blob = (const char *)mlp_blob;
    cs.h = blob_layout((uint32_t)dtype);
    cs.dtype = dtype;
    cs.kver = 10;    // (tile mode: the work items are (column, z tile) pairs - a call holds ~ 100 columns, fewer than the chip has CUs)
    cs.kver32 = 11;
    cs.restated = true;
    cs.feat_lr = feat_lr; cs.hl = hl; cs.wl = wl;
    cs.feat_hr = feat_hr; cs.hh = hh; cs.wh = wh;
    cs.mat = nullptr; cs.calib = calib; cs.zmul = zmul; cs.zdiv = zdiv;
    cs.workspace = workspace;
    cs.cus = device_cus();
    cs.kmid = 0;
    cs.run_colstart = colstart;
    cs.run_z = points + 2 * ld;
    cs.run_tiles = tiles;
    cs.run_ntiles = (const unsigned *)(meta + 1);
    if ((rc = grid_set_attributes())) return rc;
    PointSource src;
    memset(&src, 0, sizeof(src));
    src.mode = 5;
    src.pts = points;
    src.ld = ld;
    src.cols = colstart;
    src.npts = n;
    fill_calib(src, calib, zmul, zdiv);
    const bool speculate = option(OPT_POINT_RUNS_SPECULATE) != 0;
    long long guessed = 0;
    if (speculate && sp.n == n && sp.tile == tile && sp.ncols > 0) {
        guessed = sp.ncols + sp.ncols / 4 + 8;
        if (guessed > COL_BATCH) guessed = COL_BATCH;
        cs.run_ntiles_host = sp.ntiles + sp.ntiles / 4 + 8;
        if ((rc = run_column_batch(cs, src, guessed, PR_CAP, 1, kcount, nullptr, pred_hr, pred_lr, false))) return rc;
    }
    SURS_HIP_CHECK(hipEventSynchronize(sp.ev));
    const long long ncols = host[0];
    sp.n = -1;
    if (ncols <= 0 || ncols * pr_min_run(tile) > n || ncols > COL_BATCH || (host[2] && host[3])) return 0;
    if (ncols > guessed) {   // no guess, or more runs than guessed: the batch with the true counts
        cs.run_ntiles_host = host[1];
        if ((rc = run_column_batch(cs, src, ncols, PR_CAP, 1, kcount, nullptr, pred_hr, pred_lr, false))) return rc;
    }
    sp.n = n;
    sp.tile = tile;
    sp.ncols = ncols;
    sp.ntiles = host[1];
    *columns = (int)ncols;
    return 0;
}
