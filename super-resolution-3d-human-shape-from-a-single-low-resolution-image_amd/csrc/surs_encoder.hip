// Image-encoder primitives for gfx950, NHWC fp32 with explicit channel pitch.
//
// Replaces the ATen ops on the encoder path of the reference (/root/reference):
//   conv3x3 / conv1x1 (+bias, +LeakyReLU/ReLU, +residual)  lib/net_util.py:94-97, lib/model/SuRSSR_v3.py:46-138,
//                                                          lib/model/common.py:14-33, lib/model/HGFilters.py:60-64,153-174
//   GroupNorm(32, C) -> per-channel scale/shift, applied (+ReLU) in the conv's staging prologue
//                                                          lib/model/HGFilters.py:41-45,57-64
//   avg_pool2d(2,2)  lib/model/HGFilters.py:101      bicubic x2  lib/model/HGFilters.py:115, lib/model/SuRSSR_v3.py:140
//   PixelShuffle(2)+LeakyReLU  lib/model/SuRSSR_v3.py:111-115     adds  lib/model/HGFilters.py:66-74,117,203-206
//
// The convolution is an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32): A = input patch (pixels x cin),
// B = weights (cin x cout), so a lane holds one output channel and the NHWC store is 128 B coalesced per pixel.
// A workgroup computes TR rows x 32 columns of pixels x 64 output channels; per 16-channel input chunk it stages
// the (TR-1)*S+K by 31*S+K input patch (GroupNorm-apply + ReLU fused into the staging, zero padding applied
// after it) and the K*K x 16 x 64 weight slice in LDS.  Pixel stride 17 floats keeps the A reads conflict-free.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "surs_common.h"

namespace surs {
namespace enc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CK = 16;    // input channels per chunk
constexpr int PS = 17;    // LDS pixel stride in floats (odd: conflict-free ds_read_b32 across 32 pixels)
// (output channels per workgroup: template parameter NT = 64, or 32 for maps too small to fill 256 CUs otherwise)
constexpr int TC = 32;    // tile columns (pixels) = one MFMA tile

struct ConvArgs {
    const float *x; int h, w, cin, x_ld;
    const float *wp; int cin_pad, cout_pad;
    const float *bias;
    float *y; int ho, wo, cout, y_ld;
    const float *in_scale, *in_shift;
    int act; float slope;
    const float *res; int res_ld;
    // GroupNorm(32) carried between kernels (conv_x3_kernel only; null = not used).  gn_in: the statistics of x as partial sums
    // [32 groups][gn_in_slots][2] (sum, sum of squares: doubles) left by the kernel that produced x; every workgroup folds them, in a
    // fixed order, into the scale / shift of GroupNorm(gamma, beta, eps) + ReLU that its staging applies - instead of in_scale /
    // in_shift from surs_groupnorm_coeffs' two launches.  gn_out: the same partial sums of THIS kernel's output, one slot per
    // pixel tile (blockIdx.y * gridDim.x + blockIdx.x), for the GroupNorm(32, cout) that follows.
    const double *gn_in; int gn_in_slots; const float *gamma, *beta; float eps;
    double *gn_out;
    // A map written by SEVERAL kernels, each with its own tiling (a ConvBlock's cat(o1, o2, o3) + x, each slice by its convolution):
    // the rows of gn_in are gn_in_pitch doubles-pairs apart (0: gn_in_slots) and the groups [gn_in_g1, gn_in_g2) / [gn_in_g2, 32) hold
    // gn_in_slots1 / gn_in_slots2 slots (gn_in_g1 = 0: one count for all).
    int gn_in_pitch, gn_in_g1, gn_in_g2, gn_in_slots1, gn_in_slots2;
    // Second output (conv_x3_kernel only; null = not used): y2 = value + res (res required) while the value itself goes to y (if y is
    // not null) - the ConvBlock's closing sum out = cat(o1, o2, o3) + x (lib/model/HGFilters.py:66-73) leaves with the convolution
    // that makes the slice instead of in a pass of its own.  gn_out2: statistics of the y2 values in the numbering of the WHOLE sum -
    // group gn2_g0 + (channel / gn2_cg), rows gn2_pitch slots apart, one slot per pixel tile.
    float *y2; int y2_ld;
    double *gn_out2; int gn2_pitch, gn2_cg, gn2_g0;
    int fast_offsets;   // every output / residual element offset fits 32 bits (set by launch_conv_x3_cfg)
};

template <int KS, int STRIDE, int TR, int NT>
__global__ __launch_bounds__(256) void conv_kernel(ConvArgs a) {
    constexpr int NJ = NT / 32;  // MFMA column tiles (32 output channels each) per wave
    constexpr int PAD = KS / 2;
    constexpr int PR = (TR - 1) * STRIDE + KS, PC = (TC - 1) * STRIDE + KS;  // patch rows / cols
    constexpr int RPW = TR / 4;                                             // rows per wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *xs = lds;                         // [PR*PC][PS]
    float *ws = lds + PR * PC * PS;          // [KS*KS][CK][NT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ox0 = blockIdx.x * TC, oy0 = blockIdx.y * TR, n0 = blockIdx.z * NT;
    const int ix0 = ox0 * STRIDE - PAD, iy0 = oy0 * STRIDE - PAD;

    f32x16 acc[RPW][NJ];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[r][j][q] = 0.0f;

    const bool vec_ok = (a.x_ld % 4 == 0) && ((reinterpret_cast<size_t>(a.x) & 15) == 0);
    for (int c0 = 0; c0 < a.cin_pad; c0 += CK) {
        // ---- stage the input patch: PR*PC pixels x 16 channels, 4 channels per thread-item
        for (int item = tid; item < PR * PC * (CK / 4); item += 256) {
            const int pix = item >> 2, cq = (item & 3) * 4;
            const int pr = pix / PC, pc = pix - pr * PC;
            const int iy = iy0 + pr, ix = ix0 + pc;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (iy >= 0 && iy < a.h && ix >= 0 && ix < a.w) {
                const float *src = a.x + ((size_t)iy * a.w + ix) * a.x_ld + c0 + cq;
                if (vec_ok && c0 + cq + 4 <= a.cin) {
                    const f32x4 t = *reinterpret_cast<const f32x4 *>(src);
                    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (c0 + cq + q < a.cin) v[q] = src[q];
                }
                if (a.in_scale) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (c0 + cq + q < a.cin) {
                            const float t = v[q] * a.in_scale[c0 + cq + q] + a.in_shift[c0 + cq + q];
                            v[q] = t > 0.f ? t : 0.f;
                        }
                }
            }
            float *dst = xs + pix * PS + cq;
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
        // ---- stage the weights: [tap][16][64]
        for (int item = tid; item < KS * KS * CK * (NT / 4); item += 256) {
            const int co4 = (item % (NT / 4)) * 4, row = item / (NT / 4);  // row = tap*CK + c
            const int tap = row / CK, c = row - tap * CK;
            const f32x4 t = *reinterpret_cast<const f32x4 *>(a.wp + ((size_t)tap * a.cin_pad + c0 + c) * a.cout_pad + n0 + co4);
            *reinterpret_cast<f32x4 *>(ws + row * NT + co4) = t;
        }
        __syncthreads();
        const int kh = lane >> 5, li = lane & 31;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const float *wt = ws + (ky * KS + kx) * CK * NT;
#pragma unroll
                for (int s = 0; s < CK / 2; ++s) {
                    float av[RPW], bv[NJ];
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        const int prow = (wave * RPW + r) * STRIDE + ky, pcol = li * STRIDE + kx;
                        av[r] = xs[(prow * PC + pcol) * PS + 2 * s + kh];
                    }
#pragma unroll
                    for (int j = 0; j < NJ; ++j) bv[j] = wt[(2 * s + kh) * NT + j * 32 + li];
#pragma unroll
                    for (int r = 0; r < RPW; ++r)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r], bv[j], acc[r][j], 0, 0, 0);
                }
            }
        __syncthreads();
    }
    // ---- epilogue: register q of a tile is pixel column (q&3) + 8*(q>>2) + 4*(lane>>5), lane&31 is the channel
    const int kh = lane >> 5, li = lane & 31;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int oy = oy0 + wave * RPW + r;
        if (oy >= a.ho) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int co = n0 + j * 32 + li;
            if (co >= a.cout) continue;
            const float b = a.bias ? a.bias[co] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int ox = ox0 + (q & 3) + 8 * (q >> 2) + 4 * kh;
                if (ox >= a.wo) continue;
                float t = acc[r][j][q] + b;
                if (a.act == 1) t = t > 0.f ? t : a.slope * t;
                const size_t pix = (size_t)oy * a.wo + ox;
                if (a.res) t += a.res[pix * a.res_ld + co];
                a.y[pix * a.y_ld + co] = t;
            }
        }
    }
}

template <int KS, int STRIDE, int TR, int NT>
static int launch_conv_cfg(const ConvArgs &a, hipStream_t st) {
    constexpr int PR = (TR - 1) * STRIDE + KS, PC = (TC - 1) * STRIDE + KS;
    const size_t lds = (size_t)(PR * PC * PS + KS * KS * CK * NT) * sizeof(float);
    static DeviceOnce attr;
    if (attr.first())
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)conv_kernel<KS, STRIDE, TR, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(ceil_div(a.wo, TC), ceil_div(a.ho, TR), ceil_div(a.cout_pad, NT));
    hipLaunchKernelGGL((conv_kernel<KS, STRIDE, TR, NT>), grid, dim3(256), lds, st, a);
    SURS_LAUNCH_CHECK();
    return 0;
}

// Tile choice: TR rows x 32 columns x NT channels per workgroup.  The big tile (TR_BIG x 64) re-uses a staged patch most;
// maps that would give fewer than two workgroups per CU (the hourglass's 128^2 / 64^2 levels: 32-128 workgroups)
// take 4 rows and, if still short, 32 channels.  Every output element is the same sum in the same order in all of them.
template <int KS, int STRIDE, int TR_BIG>
static int launch_conv(const ConvArgs &a, hipStream_t st) {
    const long long cols = ceil_div(a.wo, TC);
    const long long wg_big = cols * ceil_div(a.ho, TR_BIG) * (a.cout_pad / 64);
    const long long wg_r4 = cols * ceil_div(a.ho, 4) * (a.cout_pad / 64);
    // two workgroups per CU (LDS: 60 KB each) let one stage while the other multiplies: ask for >= 512 workgroups
    constexpr long long WANT = 512;
    if (wg_big >= WANT || TR_BIG == 4) {
        if (TR_BIG == 4 && wg_big < WANT) return launch_conv_cfg<KS, STRIDE, 4, 32>(a, st);
        return launch_conv_cfg<KS, STRIDE, TR_BIG, 64>(a, st);
    }
    if (wg_r4 >= WANT) return launch_conv_cfg<KS, STRIDE, 4, 64>(a, st);
    return launch_conv_cfg<KS, STRIDE, 4, 32>(a, st);
}

// ---------------------------------------------------------------- 3x3 convolution on the bf16 matrix pipe, fp32-exact
// The fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate.  Every fp32 operand is instead split into three
// bf16 parts, x = x1 + x2 + x3 exactly (8 + 8 + 8 significant bits), and the product is accumulated from the six
// partial products that matter, x1 w1 + (x1 w2 + x2 w1) + (x1 w3 + x2 w2 + x3 w1): each is exact in the fp32
// accumulator's input, what is dropped is below 2^-24 of the product - fp32 accuracy (measured 3e-7 of the output range
// on a 2304-term convolution against 5e-7 for the fp32 MFMA) at 6 bf16 MFMAs of 8 passes for 16 channels against
// 8 fp32 MFMAs of 16 passes: 2.7x.  Weights are split by the packer (surs_conv_pack_weights_x3), activations while they
// are staged (after the fused GroupNorm-apply + ReLU).  Tile: TR rows x 32 columns x NT3 output channels.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
constexpr int XS = 24;    // LDS row pitch in halfwords (48 B): conflict-free ds_read_b128 across consecutive pixels / channels

__device__ __forceinline__ void split3(float x, unsigned short &a, unsigned short &b, unsigned short &c) {
    const __bf16 ha = (__bf16)x;
    const float r1 = x - (float)ha;
    const __bf16 hb = (__bf16)r1;
    const float r2 = r1 - (float)hb;
    const __bf16 hc = (__bf16)r2;
    a = __builtin_bit_cast(unsigned short, ha);
    b = __builtin_bit_cast(unsigned short, hb);
    c = __builtin_bit_cast(unsigned short, hc);
}

// Operand split of the 3x3 kernel: NP = 3 bf16 parts (six products per MAC, fp32's exponent range) or NP = 2 f16 parts
// (hi = f16(x), lo = f16(x - hi); products hi*hi + hi*lo + lo*hi: half the MFMA work, 22 significant bits, |x| < 65504 - the
// split of the fp32-grade column kernel).  Both pass the encoder's 1e-4 parity tests; two parts is the default.
template <int NP> struct ConvSplit;
template <> struct ConvSplit<3> {
    typedef bf16x8 vec8;
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ void split(float x, unsigned short (&p)[3]) { split3(x, p[0], p[1], p[2]); }
};
template <> struct ConvSplit<2> {
    typedef _Float16 vec8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ void split(float x, unsigned short (&p)[2]) {
        asm volatile("" : "+v"(x));   // (one value feeds the conversion and the remainder: see split2_f16 in surs_grid_v5.inc)
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)__builtin_fmaf((float)hi, -1.0f, x);
        p[0] = __builtin_bit_cast(unsigned short, hi);
        p[1] = __builtin_bit_cast(unsigned short, lo);
    }
};

// NP = 1: ONE f16 part per operand, one product per MAC - 11 significant bits, NOT fp32-grade: the encoder of the reduced-precision
// modes (--precision bf16 / fp16), whose consumer - the 16-bit column kernel - rounds the features' contributions to 8 / 11 bits
// anyway.  Reads part 0 of the two-part weight image (hi = f16(w)).
template <> struct ConvSplit<1> {
    typedef _Float16 vec8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ void split(float x, unsigned short (&p)[1]) { p[0] = __builtin_bit_cast(unsigned short, (_Float16)x); }
};

// GroupNorm(32) statistics handed over by the producer of x (ConvArgs::gn_in) -> per-channel scale at gn[c], shift at
// gn[shift_off + c], by all 256 threads of a workgroup: eight threads per group, thread `sub` adds the slots sub, sub + 8, ... in
// order (eight loads in flight), then a butterfly over the eight - a fixed summation tree, the same in every workgroup of every
// launch - then gn_finish_kernel's arithmetic.  cin is a multiple of 32: no pad channels.  The caller synchronises.
__device__ __forceinline__ void gn_fold_to_lds(const ConvArgs &a, float *gn, int shift_off, int tid) {
    const int g = tid >> 3, sub = tid & 7, cgi = a.cin / 32;
    const int slots = (a.gn_in_g1 <= 0 || g < a.gn_in_g1) ? a.gn_in_slots : (g < a.gn_in_g2 ? a.gn_in_slots1 : a.gn_in_slots2);
    const double2 *pg = reinterpret_cast<const double2 *>(a.gn_in) + (size_t)g * (a.gn_in_pitch > 0 ? a.gn_in_pitch : a.gn_in_slots);
    double S = 0, SS = 0;
    for (int base = 0; base < slots; base += 64) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sl = base + sub + 8 * u;
            v[u] = sl < slots ? pg[sl] : double2{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            S += v[u].x;
            SS += v[u].y;
        }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        S += __shfl_xor(S, o);
        SS += __shfl_xor(SS, o);
    }
    const double n = (double)a.h * (double)a.w * cgi;
    const double mean = S / n;
    double var = SS / n - mean * mean;
    if (var < 0) var = 0;
    const double rstd = 1.0 / sqrt(var + (double)a.eps);
    for (int k = sub; k < cgi; k += 8) {
        const int ch = g * cgi + k;
        gn[ch] = (float)(rstd * a.gamma[ch]);
        gn[shift_off + ch] = (float)(a.beta[ch] - mean * rstd * a.gamma[ch]);
    }
}

// The statistics of the values a workgroup stored (ConvArgs::gn_out), from the MFMA epilogue layout of both convolution kernels:
// lane & 31 = channel co0 + (lane & 31) of a 32-channel tile, this lane's sums S, SS over its pixels.  Folds the two pixel halves
// (lanes l, l + 32) and the cg = cout / 32 (a power of two <= 32) channels of a group (neighbouring lanes); lanes < 32 with
// (lane & (cg - 1)) == 0 then hold the sums of group (co0 + lane) / cg.
__device__ __forceinline__ void gn_fold_lanes(double &S, double &SS, int cg) {
    S += __shfl_xor(S, 32);
    SS += __shfl_xor(SS, 32);
    for (int o = 1; o < cg; o <<= 1) {
        S += __shfl_xor(S, o);
        SS += __shfl_xor(SS, o);
    }
}

#ifdef SURS_CONV_TRACE
__device__ unsigned long long g_conv_trace[8];
#define CSTAMP(i) do { if (blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 0 && tid == 0) { unsigned long long t_ = __builtin_readcyclecounter(); g_conv_trace[i] += t_ - tprev; tprev = t_; } } while (0)
#else
#define CSTAMP(i) do { } while (0)
#endif
template <int KS, int STRIDE, int TR, int NT3, int NP>
__global__ __launch_bounds__(256, (TR == 8 && STRIDE == 1) ? 2 : 1) void conv_x3_kernel(ConvArgs a, const unsigned short *__restrict__ wsplit) {
    typedef ConvSplit<NP> CS;
    typedef typename CS::vec8 vec8;
    constexpr int NJ = NT3 / 32;   // 32-channel MFMA column tiles per wave
    constexpr int PAD = KS / 2, T = KS * KS;
    constexpr int PR = (TR - 1) * STRIDE + KS, PC = (TC - 1) * STRIDE + KS;
    constexpr int RPW = TR / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];
    unsigned short *xs = lds16;                         // [NP][PR*PC][XS]
    // Weights: TWO buffers of [NP][T][NJ] fragments of 1 KiB, each in MFMA B-fragment order (lane (kh, li) = halfwords [8 kh, + 8) of
    // output channel 32 j + li: read back conflict-free at lane * 16, no padded rows).  They arrive by LDS-DMA (global_load_lds, 16 B
    // per lane: a fragment per instruction, no staging registers, no vector work); chunk c + 1's are issued behind the barrier that
    // opens chunk c's multiply loop and land under it.  (Until round 6: 9 register loads + 9 LDS stores per thread and chunk, 36
    // registers of prefetch, 48-byte rows.)
    constexpr int WFR = NP * T * NJ;                    // fragments per chunk
    // (64-channel tile: ONE weight buffer - two would push the workgroup past half a CU's LDS -, filled behind the barrier that ends the
    //  previous chunk's multiply loop, landing under this chunk's patch staging)
    constexpr int WB = NT3 >= 64 ? 1 : 2;
    unsigned short *wsm = lds16 + NP * PR * PC * XS;    // [2][WFR][512 halfwords]
    float *gn = reinterpret_cast<float *>(wsm + WB * WFR * 512);   // [2][cin_pad]: GroupNorm scale, shift (if any)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef SURS_CONV_TRACE
    const unsigned long long t_kernel_start = __builtin_readcyclecounter();
#endif
    const int ox0 = blockIdx.x * TC, oy0 = blockIdx.y * TR, n0 = blockIdx.z * NT3;
    const int ix0 = ox0 * STRIDE - PAD, iy0 = oy0 * STRIDE - PAD;
    const int nch = a.cin_pad / CK;
    const size_t per_part = (size_t)T * nch * a.cout_pad * 16;

    // A dependent MFMA (same accumulator as the one before it) waits for that one's result, ~2x its issue interval:
    // consecutive MFMAs always go to different accumulators.  The big tile has 4 (rows x channel tiles); the small tile
    // has one, so its six partial products are spread over three accumulators that are added at the end.
    constexpr int NA = (RPW * NJ >= 4) ? 1 : 3;
    f32x16 acc[NA][RPW][NJ];
#pragma unroll
    for (int s_ = 0; s_ < NA; ++s_)
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[s_][r][j][q] = 0.0f;

    const bool norm = a.in_scale != nullptr || a.gn_in != nullptr;   // GroupNorm-apply + ReLU in the staging
    // A chunk's MFMAs take ~1.7 k cycles, less than a global-memory round trip: the next chunk's patch and weight
    // slices are fetched into registers while the current chunk multiplies, and split / stored to LDS afterwards.
    constexpr int NPI = (PR * PC * (CK / 4) + 255) / 256;   // patch items per thread (4 channels of one pixel each)
    // chunks in flight: the small tile's MFMAs (1.7 k cycles) are shorter than a memory round trip -> two; the big tile's
    // (7 k cycles) cover it, and a second buffer (80 more registers) would spill
    constexpr int PD = (RPW * NJ >= 2) ? 1 : 2;   // (the 8 x 32 tile: 1.7 k cycles of MFMAs per chunk, and 256 registers for two workgroups per CU)
    f32x4 pre_x[PD][NPI];
    // per-thread item descriptors, computed once (the index arithmetic would otherwise cost more than the MFMAs).
    // Loads are unconditional - items outside the image or beyond the item count read element 0 and are masked /
    // dropped afterwards - so that the compiler emits them back to back instead of one branch per item.
    unsigned px_src[NPI];    // element offset of the item's 4 channels in x for chunk 0
    bool px_ok[NPI];         // inside the image
    int px_dst[NPI];         // halfword offset in one part's LDS image, or -1: no such item
    const int cq = (tid & 3) * 4;   // the same for every patch item of a thread (256 is a multiple of 4)
#pragma unroll
    for (int k = 0; k < NPI; ++k) {
        const int item = tid + k * 256;
        px_src[k] = 0;
        px_ok[k] = false;
        px_dst[k] = -1;
        if (item < PR * PC * (CK / 4)) {
            const int pix = item >> 2;
            const int pr = pix / PC, pc = pix - pr * PC;
            const int iy = iy0 + pr, ix = ix0 + pc;
            px_dst[k] = pix * XS + cq;
            if (iy >= 0 && iy < a.h && ix >= 0 && ix < a.w) {
                px_ok[k] = true;
                px_src[k] = (unsigned)((iy * a.w + ix) * a.x_ld + cq);
            }
        }
    }
    const unsigned w_step = (unsigned)a.cout_pad * 16;   // halfwords per chunk in wsplit
    // wave w issues the fragments w, w + 4, ..: fragment f = (part * T + tap) * NJ + j
    const unsigned w_lane = (unsigned)((n0 + (lane & 31)) * 16 + (lane >> 5) * 8);
    auto weights_dma = [&](int ch, int buf) {
        typedef const __attribute__((address_space(1))) void gptr_t;
        typedef __attribute__((address_space(3))) void lptr_t;
#pragma unroll
        for (int q = 0; q < (WFR + 3) / 4; ++q) {
            const int f = wave + 4 * q;
            if (f < WFR) {
                const int j = f % NJ, pt = f / NJ, part = pt / T, tap = pt - part * T;
                const unsigned short *src = wsplit + ((size_t)part * per_part + (size_t)tap * nch * a.cout_pad * 16 + (size_t)ch * w_step +
                                                      (size_t)j * 512 + w_lane);
                __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)(wsm + (buf * WFR + f) * 512), 16, 0, 0);
            }
        }
    };
    auto fetch = [&](int ch, int pb) {
#pragma unroll
        for (int k = 0; k < NPI; ++k) pre_x[pb][k] = *reinterpret_cast<const f32x4 *>(a.x + (px_src[k] + (unsigned)(ch * CK)));
    };
    // the same loads in slices, one per tap of the multiply loop: a wave's 15 loads take ~ 1 000 cycles to issue (1 KiB each through
    // the one texture-address unit the four waves share, all four at the same point of the chunk) - a twelfth of the kernel when
    // they were issued in one burst in front of the MFMAs; between the taps' MFMAs they cost nothing
    constexpr int FT = T > 4 ? T - 3 : T;   // ... of the first taps, so that the last loads have the rest of the loop to arrive
    constexpr int NFI = NPI, FPT = (NFI + FT - 1) / FT;
    auto fetch_slice = [&](int ch, int pb, int tap) {
#pragma unroll
        for (int q = 0; q < FPT; ++q) {
            const int k = tap * FPT + q;
            if (k < NPI)
                pre_x[pb][k < NPI ? k : 0] = *reinterpret_cast<const f32x4 *>(a.x + (px_src[k < NPI ? k : 0] + (unsigned)(ch * CK)));
        }
    };
    auto stage = [&](int ch, int pb) {
        const int c0 = ch * CK;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (norm) {
            sc = *reinterpret_cast<const f32x4 *>(gn + c0 + cq);
            sh = *reinterpret_cast<const f32x4 *>(gn + a.cin_pad + c0 + cq);
        }
#pragma unroll
        for (int k = 0; k < NPI; ++k) {
            if (px_dst[k] < 0) continue;
            u16x4 pp[NP];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = pre_x[pb][k][q];
                if (norm) v = fmaxf(v * sc[q] + sh[q], 0.f);
                v = px_ok[k] ? v : 0.f;   // zero padding is applied after the norm
                unsigned short parts[NP];
                CS::split(v, parts);
#pragma unroll
                for (int e = 0; e < NP; ++e) pp[e][q] = parts[e];
            }
#pragma unroll
            for (int e = 0; e < NP; ++e) *reinterpret_cast<u16x4 *>(xs + e * PR * PC * XS + px_dst[k]) = pp[e];
        }
    };
    weights_dma(0, 0);
    fetch(0, 0);
    if (PD == 2 && nch > 1) fetch(1, 1);
    // (behind the first chunks' loads: their latency covers the fold)
    if (a.in_scale) {   // once per workgroup: read per element from global memory they stall every chunk's staging
        for (int c = tid; c < a.cin_pad; c += 256) {
            gn[c] = c < a.cin ? a.in_scale[c] : 0.f;
            gn[a.cin_pad + c] = c < a.cin ? a.in_shift[c] : 0.f;
        }
        __syncthreads();
    } else if (a.gn_in) {
        gn_fold_to_lds(a, gn, a.cin_pad, tid);
        __syncthreads();
    }
#ifdef SURS_CONV_TRACE
    unsigned long long tprev = t_kernel_start;
    if (blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 0 && tid == 0) for (int i = 0; i < 8; ++i) g_conv_trace[i] = 0;
    CSTAMP(5);
#endif
    auto chunk = [&](int ch, int pb) {
        CSTAMP(4);
        stage(ch, pb);
        CSTAMP(0);
        __syncthreads();   // (also waits for this chunk's weight fragments: the barrier's fence drains the DMA counter)
        if (WB == 2 && ch + 1 < nch) weights_dma(ch + 1, (ch + 1) & 1);   // the other buffer: chunk ch - 1's multiply loop ended at the barrier behind it
        CSTAMP(1);
        const bool more = ch + PD < nch;   // (uniform) the next fetch goes into the buffer this chunk has just been staged from
        CSTAMP(2);
        const int kh = lane >> 5, li = lane & 31;
        // operands of tap t + 1 are read from LDS while tap t multiplies (register double buffer): without it every tap
        // starts with an exposed LDS round trip
        vec8 bw[2][NJ][NP], ax[2][RPW][NP];
        auto ldtap = [&](int tap, int buf) {
            const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    bw[buf][j][p] = *reinterpret_cast<const vec8 *>(wsm + (((WB == 2 ? (ch & 1) : 0) * WFR + (p * T + tap) * NJ + j) * 512 + lane * 8));
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int prow = (wave * RPW + r) * STRIDE + ky, pcol = li * STRIDE + kx;
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    ax[buf][r][p] = *reinterpret_cast<const vec8 *>(xs + ((size_t)p * PR * PC + prow * PC + pcol) * XS + 8 * kh);
            }
        };
        ldtap(0, 0);
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
            const int cur = tap & 1;
            if (tap + 1 < T) ldtap(tap + 1, cur ^ 1);
            if (more) fetch_slice(ch + PD, pb, tap);
            __builtin_amdgcn_sched_barrier(0);   // the next tap's reads and this tap's slice of the prefetch are in flight before its MFMAs start
            // the partial products that matter, smallest first: (x part, w part); six of three parts, three of two
            constexpr int NPROD = NP == 3 ? 6 : (NP == 2 ? 3 : 1);
            constexpr int PA[6] = {NP == 3 ? 2 : (NP == 2 ? 1 : 0), NP == 3 ? 1 : 0, 0, 1, 0, 0}, PB[6] = {0, NP == 1 ? 0 : 1, NP == 3 ? 2 : 0, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < NPROD; ++t)
#pragma unroll
                for (int r = 0; r < RPW; ++r)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int sl = (NP == 1 ? tap : t) % NA;   // (one product per tap: consecutive taps on different accumulators)
                        acc[sl][r][j] = CS::mfma(ax[cur][r][PA[t]], bw[cur][j][PB[t]], acc[sl][r][j]);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        CSTAMP(3);
        __syncthreads();
        if (WB == 1 && ch + 1 < nch) weights_dma(ch + 1, 0);
    };
    for (int ch = 0; ch < nch; ch += 2) {
        chunk(ch, 0);
        if (ch + 1 < nch) chunk(ch + 1, PD - 1);
    }
    CSTAMP(4);   // (the interval since the last chunk's second barrier)
    // ---- epilogue: register q of a tile is pixel column (q&3) + 8*(q>>2) + 4*(lane>>5), lane&31 is the channel
    const int kh = lane >> 5, li = lane & 31;
    const bool stats = a.gn_out != nullptr, dual = a.y2 != nullptr, stats2 = a.gn_out2 != nullptr;
    double st_s[NJ], st_ss[NJ];   // this lane's sums of the values it stores, per channel tile
    double s2_s[NJ], s2_ss[NJ];   // ... and of the values of the second output
#pragma unroll
    for (int j = 0; j < NJ; ++j) st_s[j] = st_ss[j] = s2_s[j] = s2_ss[j] = 0.0;
    // A tile that lies inside the map and the channel range (every tile of the encoder's maps: their sizes are multiples of the tile)
    // takes the straight form: one 32-bit element offset per output row, the 16 pixel columns of a register tile at scalar multiples
    // of the pitch, no per-element bounds.  (Until round 6 every tile ran the guarded form below: 8 000 instructions - 64-bit
    // multiplies and two branches per element - against 2 100 for the whole multiply loop, 15 - 38 % of a workgroup's cycles.)
    // The same operations per element in the same order: the same bits.
    const bool full = ox0 + TC <= a.wo && oy0 + TR <= a.ho && n0 + NT3 <= a.cout;   // (uniform)
    if (full && a.fast_offsets) {
        const float sl = a.act == 1 ? a.slope : 1.0f;   // (x > 0 ? x : 1 * x: the identity, bit for bit, where there is no activation)
        // (one instantiation per combination of outputs: the flags are compile-time inside, so that the 64 elements of a wave's
        //  tiles are straight-line code)
        auto emit = [&](auto dual_c, auto res_c, auto st_c, auto st2_c) {
            constexpr bool DUAL = decltype(dual_c)::value, RES = decltype(res_c)::value, ST = decltype(st_c)::value, ST2 = decltype(st2_c)::value;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const unsigned pix0 = (unsigned)((oy0 + wave * RPW + r) * a.wo + ox0 + 4 * kh);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned co = (unsigned)(n0 + j * 32 + li);
                    const float b = a.bias ? a.bias[co] : 0.f;
                    float rv[16];
                    if (RES) {
                        const float *rp = a.res + (pix0 * (unsigned)a.res_ld + co);
#pragma unroll
                        for (int q = 0; q < 16; ++q) rv[q] = rp[((q & 3) + 8 * (q >> 2)) * a.res_ld];
                    }
                    float *yp = a.y + (pix0 * (unsigned)a.y_ld + co);
                    float *y2p = DUAL ? a.y2 + (pix0 * (unsigned)a.y2_ld + co) : nullptr;
                    const bool have_y = !DUAL || a.y != nullptr;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int pc = (q & 3) + 8 * (q >> 2);
                        float t = acc[0][r][j][q];
                        if (NA == 3) t = (acc[0][r][j][q] + acc[1][r][j][q]) + acc[2][r][j][q];
                        t += b;
                        t = t > 0.f ? t : sl * t;
                        if (DUAL) {   // the value to y (the next convolution's input), value + res to y2 (the block's output slice)
                            if (have_y) yp[pc * a.y_ld] = t;
                            if (ST) {
                                const double d = (double)t;
                                st_s[j] += d;
                                st_ss[j] += d * d;
                            }
                            const float u = t + rv[q];
                            y2p[pc * a.y2_ld] = u;
                            if (ST2) {
                                const double d = (double)u;
                                s2_s[j] += d;
                                s2_ss[j] += d * d;
                            }
                        } else {
                            if (RES) t += rv[q];
                            yp[pc * a.y_ld] = t;
                            if (ST) {
                                const double d = (double)t;
                                st_s[j] += d;
                                st_ss[j] += d * d;
                            }
                        }
                    }
                }
            }
        };
        using T = std::true_type;
        using F = std::false_type;
        if (dual) {
            if (stats) { if (stats2) emit(T(), T(), T(), T()); else emit(T(), T(), T(), F()); }
            else       { if (stats2) emit(T(), T(), F(), T()); else emit(T(), T(), F(), F()); }
        } else if (a.res) {
            if (stats) emit(F(), T(), T(), F()); else emit(F(), T(), F(), F());
        } else {
            if (stats) emit(F(), F(), T(), F()); else emit(F(), F(), F(), F());
        }
    } else {
        // the guarded form (ragged tiles; the second output is not made here: its callers hand over whole tiles only)
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int oy = oy0 + wave * RPW + r;
            if (oy >= a.ho) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int co = n0 + j * 32 + li;
                if (co >= a.cout) continue;
                const float b = a.bias ? a.bias[co] : 0.f;
                // the tile's 16 residual values first, all in flight together: read one by one between the stores (which the compiler
                // must keep in order: res and y may alias) every element cost a memory round trip
                float rv[16];
                if (a.res) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int ox = ox0 + (q & 3) + 8 * (q >> 2) + 4 * kh;
                        rv[q] = ox < a.wo ? a.res[((size_t)oy * a.wo + ox) * a.res_ld + co] : 0.f;
                    }
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int ox = ox0 + (q & 3) + 8 * (q >> 2) + 4 * kh;
                    if (ox >= a.wo) continue;
                    float t = acc[0][r][j][q];
                    if (NA == 3) t = (acc[0][r][j][q] + acc[1][r][j][q]) + acc[2][r][j][q];
                    t += b;
                    if (a.act == 1) t = t > 0.f ? t : a.slope * t;
                    const size_t pix = (size_t)oy * a.wo + ox;
                    if (a.res) t += rv[q];
                    a.y[pix * a.y_ld + co] = t;
                    if (stats) {
                        const double d = (double)t;
                        st_s[j] += d;
                        st_ss[j] += d * d;
                    }
                }
            }
        }
    }
    // fold: the two pixel halves of a channel (lanes l, l + 32), the cg channels of a group (neighbouring lanes), the four waves
    // (rows) through LDS - every step in a fixed order; one slot per pixel tile
    auto publish = [&](double (&vs)[NJ], double (&vss)[NJ], double *dst, int cg, int g0, size_t pitch) {
        const int gpt = 32 / cg;
        double *red = reinterpret_cast<double *>(lds16);   // [4 waves][NJ][32][2] (the chunk loop ended with a barrier)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            double S = vs[j], SS = vss[j];
            gn_fold_lanes(S, SS, cg);
            if (lane < 32 && (li & (cg - 1)) == 0) {
                red[((wave * NJ + j) * 32 + li / cg) * 2] = S;
                red[((wave * NJ + j) * 32 + li / cg) * 2 + 1] = SS;
            }
        }
        __syncthreads();
        if (tid < NJ * gpt) {
            const int j = tid / gpt, k = tid - j * gpt;
            const int g = g0 + (n0 + j * 32) / cg + k;
            if (g < 32) {
                double S = 0, SS = 0;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) {
                    S += red[((wv * NJ + j) * 32 + k) * 2];
                    SS += red[((wv * NJ + j) * 32 + k) * 2 + 1];
                }
                const size_t slot = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
                dst[((size_t)g * pitch + slot) * 2] = S;
                dst[((size_t)g * pitch + slot) * 2 + 1] = SS;
            }
        }
    };
    if (stats) publish(st_s, st_ss, a.gn_out, a.cout / 32, 0, (size_t)gridDim.x * gridDim.y);
    if (stats && stats2) __syncthreads();   // (the reduction buffer is used again)
    if (stats2) publish(s2_s, s2_ss, a.gn_out2, a.gn2_cg, a.gn2_g0, (size_t)a.gn2_pitch);
    CSTAMP(6);
}

template <int KS, int STRIDE, int TR, int NT3, int NP>
static int launch_conv_x3_cfg(const ConvArgs &a, const unsigned short *wsplit, hipStream_t st) {
    constexpr int PR = (TR - 1) * STRIDE + KS, PC = (TC - 1) * STRIDE + KS;
    SURS_REQUIRE(a.cin_pad <= 1024, "split-operand convolution: at most 1024 input channels");
    const size_t lds = (size_t)(NP * PR * PC * XS + (NT3 >= 64 ? 1 : 2) * NP * KS * KS * (NT3 / 32) * 512) * sizeof(unsigned short) + 2 * 1024 * sizeof(float);
    static DeviceOnce attr;
    if (attr.first())
        SURS_HIP_CHECK(hipFuncSetAttribute((const void *)conv_x3_kernel<KS, STRIDE, TR, NT3, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(ceil_div(a.wo, TC), ceil_div(a.ho, TR), ceil_div(a.cout_pad, NT3));
    ConvArgs b = a;
    {
        const long long npix = (long long)a.ho * a.wo, lim = 1ll << 31;
        b.fast_offsets = npix * (a.y ? a.y_ld : 1) < lim && npix * (a.res ? a.res_ld : 1) < lim && npix * (a.y2 ? a.y2_ld : 1) < lim;
    }
    hipLaunchKernelGGL((conv_x3_kernel<KS, STRIDE, TR, NT3, NP>), grid, dim3(256), lds, st, b, wsplit);
    SURS_LAUNCH_CHECK();
#ifdef SURS_CONV_TRACE
    if (option(OPT_CONV_TRACE)) {
        unsigned long long t[8];
        SURS_HIP_CHECK(hipStreamSynchronize(st));
        SURS_HIP_CHECK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_conv_trace), sizeof(t)));
        fprintf(stderr, "conv_x3 <%d,%d,np %d> %dx%d cin %d cout %d: cycles of workgroup (1,1,0) over %d chunks: stage %llu, barrier %llu, fetch issue %llu, taps %llu, barrier2 %llu; prologue %llu, epilogue %llu\n",
                TR, NT3, NP, a.h, a.w, a.cin, a.cout, a.cin_pad / CK, t[0], t[1], t[2], t[3], t[4], t[5], t[6]);
    }
#endif
    return 0;
}

// 8 rows x 64 channels (140 KB of LDS, one workgroup per CU, 216 MFMAs per wave and chunk: long enough to cover the
// prefetch, and the input is re-read least) where that still gives every CU two workgroups' worth of tiles; 4 rows x 32
// channels (79 KB, two workgroups per CU) for the small maps.
// surs_conv_tile_scale: the tile is chosen as if the map were num / den times as wide (calling thread only).  The two tiles sum
// their partial products in different orders, so a column strip of a map reproduces the bits of the full map's columns only if it
// runs the full map's tile (encoder.super_res_strip: one rank's share of a sharded encoder).
static thread_local int t_tile_num = 1, t_tile_den = 1;
extern "C" int surs_conv_tile_scale(int num, int den) {
    SURS_REQUIRE(num >= 1 && den >= 1, "scale must be positive");
    t_tile_num = num;
    t_tile_den = den;
    return 0;
}

// The tile of a stride-1 launch, all three two workgroups per CU (one's staging and barriers run under the other's MFMAs):
//   8 rows x 32 pixels x 64 channels where that still gives 512 workgroups (the maps of 256^2 and more with >= 128 output channels, the
//       super-resolution net's large maps): the staged patch - GroupNorm, ReLU, split: the loop's vector work - serves twelve MFMAs
//       per tap and wave; ONE weight buffer (two would not leave room for a second workgroup), filled by DMA behind the barrier that
//       ends a chunk's multiply loop and landing under the next chunk's patch staging;
//   8 rows x 32 x 32 from 256 workgroups: six MFMAs per tap and wave; two weight buffers, the next chunk's landing under this one's MFMAs;
//   4 rows x 32 x 32 below (the hourglass's 64^2 maps): three per tap - its multiply loop waits for its LDS operands (175 cycles per
//       tap against 96 of MFMAs).
// The 32-channel tiles add their three partial products in one order, the 64-channel tile in another: the choice is part of the bits
// (and reproduces round 5's, whose 8 x 64 tile - 140 KB of LDS, padded weight rows staged through registers, one workgroup per CU - ran
// the same launches: 7.26 -> 5.8 ms per 512^2 image with the epilogues of R6.3).  One place: the statistics' slot count follows it.
// Three bf16 parts (the wide-operand retry) would spill at 256 registers with 8 rows: the 4-row tile there.
struct ConvTile { int rows, chans; };
static ConvTile conv_x3_tile(const ConvArgs &a, int stride, int np = 2) {
    const long long wo_eff = ((long long)a.wo * t_tile_num + t_tile_den - 1) / t_tile_den;
    const long long px = (long long)ceil_div(wo_eff, TC) * ceil_div(a.ho, 8);
    if (stride != 1 || np > 2) return {4, 32};
    if (np == 2 && option(OPT_CONV_WIDE_MIN_WG) > 0 && px * (a.cout_pad / 64) >= option(OPT_CONV_WIDE_MIN_WG)) return {8, 64};
    if (option(OPT_CONV_TALL_MIN_WG) > 0 && px * (a.cout_pad / 32) >= option(OPT_CONV_TALL_MIN_WG)) return {8, 32};
    return {4, 32};
}
static int conv_x3_tile_rows(const ConvArgs &a, int stride, int np = 2) { return conv_x3_tile(a, stride, np).rows; }

template <int KS, int STRIDE, int NP>
static int launch_conv_x3(const ConvArgs &a, const unsigned short *wsplit, hipStream_t st) {
    // (A 16-row tile - 501 registers - was measured in round 5: wrong values when four processes share the GPU; NOTES R5.7.)
    const ConvTile t = conv_x3_tile(a, STRIDE, NP);
    if (t.rows == 8 && t.chans == 64)
        return launch_conv_x3_cfg<KS, STRIDE, (STRIDE == 1 && NP == 2) ? 8 : 4, (STRIDE == 1 && NP == 2) ? 64 : 32, NP>(a, wsplit, st);
    if (t.rows == 8) return launch_conv_x3_cfg<KS, STRIDE, (STRIDE == 1 && NP <= 2) ? 8 : 4, 32, NP>(a, wsplit, st);
    return launch_conv_x3_cfg<KS, STRIDE, 4, 32, NP>(a, wsplit, st);
}

// ---------------------------------------------------------------- 1x1 convolution (pointwise), split-f16 operands
// Y[pix][co] = bias[co] + sum_c X'[pix][c] W[co][c]: HBM-bound (one read of the input, one write of the output), so the whole
// output row of a pixel is produced from ONE read of its input row: a workgroup takes 128 pixels x CT * 64 output channels
// (all 256 of the hourglass's 1x1 convolutions), wave (ph, ch) = pixels [64 ph, + 64) x channels [32 CT ch, + 32 CT).
// Activations go from global memory straight into MFMA A fragments (lane = pixel, 8 consecutive channels = 32 contiguous
// bytes of the NHWC row; the fused GroupNorm-apply + ReLU and the hi / lo split happen in registers: no LDS staging, no
// barrier in the loop), the weights come as B fragments from the packer's split image (surs_conv_pack_weights_x2 with one
// tap: L2 resident).  Products as in conv_x3_kernel<..., 2>: x_lo w_hi + x_hi w_lo + x_hi w_hi in fp32.  The fp32-MFMA kernel
// it replaces re-read the input once per 64 output channels and ran at 0.7 TB/s (185 us per 256^2 x 256 -> 256 layer).
template <int CT>
__global__ __launch_bounds__(256, 2) void conv1x1_x2_kernel(ConvArgs a, const unsigned short *__restrict__ wsplit) {
    typedef ConvSplit<2> CS;
    typedef typename CS::vec8 vec8;
    __shared__ __attribute__((aligned(16))) float gn[2 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ph = wave & 1, ch = wave >> 1, li = lane & 31, kh = lane >> 5;
    const long long npix = (long long)a.h * a.w;
    const long long p0 = (long long)blockIdx.x * 128 + 64 * ph;
    const int n0 = blockIdx.y * (64 * CT) + 32 * CT * ch;
    const int nch = a.cin_pad / CK;
    const size_t per_part = (size_t)nch * a.cout_pad * 16;
    const bool norm = a.in_scale != nullptr || a.gn_in != nullptr;
    if (a.in_scale) {
        for (int c = tid; c < a.cin_pad; c += 256) {
            gn[c] = c < a.cin ? a.in_scale[c] : 0.f;
            gn[1024 + c] = c < a.cin ? a.in_shift[c] : 0.f;
        }
        __syncthreads();
    } else if (a.gn_in) {
        gn_fold_to_lds(a, gn, 1024, tid);
        __syncthreads();
    }
    f32x16 acc[2][CT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][j][q] = 0.0f;
    const float *xp[2];
    bool ok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const long long pix = p0 + 32 * t + li;
        ok[t] = pix < npix;
        xp[t] = a.x + (size_t)(ok[t] ? pix : 0) * a.x_ld + 8 * kh;
    }
    const unsigned short *wq = wsplit + ((size_t)n0 + li) * 16 + 8 * kh;
    f32x4 xa[2][2], xn[2][2];
    vec8 wb[CT][2];
    auto fetch_x = [&](int s, f32x4 (&xx)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            xx[t][0] = *reinterpret_cast<const f32x4 *>(xp[t] + s * CK);
            xx[t][1] = *reinterpret_cast<const f32x4 *>(xp[t] + s * CK + 4);
        }
    };
    fetch_x(0, xa);
    for (int s = 0; s < nch; ++s) {
        // this k-step's weight fragments (L2) and the next k-step's activations (HBM) are in flight while the activations of
        // this one are normalised and split
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                wb[j][p] = *reinterpret_cast<const vec8 *>(wq + p * per_part + ((size_t)s * a.cout_pad + 32 * j) * 16);
        if (s + 1 < nch) fetch_x(s + 1, xn);
        vec8 ah[2], al[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            unsigned short hi[8], lo[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float v = xa[t][q >> 2][q & 3];
                if (norm) v = fmaxf(v * gn[s * CK + 8 * kh + q] + gn[1024 + s * CK + 8 * kh + q], 0.f);
                v = ok[t] ? v : 0.f;
                unsigned short parts[2];
                CS::split(v, parts);
                hi[q] = parts[0];
                lo[q] = parts[1];
            }
            typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
            const u16x8 h8 = {hi[0], hi[1], hi[2], hi[3], hi[4], hi[5], hi[6], hi[7]}, l8 = {lo[0], lo[1], lo[2], lo[3], lo[4], lo[5], lo[6], lo[7]};
            ah[t] = __builtin_bit_cast(vec8, h8);
            al[t] = __builtin_bit_cast(vec8, l8);
        }
        // the partial products that matter, smallest first
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[t][j] = CS::mfma(al[t], wb[j][0], acc[t][j]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[t][j] = CS::mfma(ah[t], wb[j][1], acc[t][j]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[t][j] = CS::mfma(ah[t], wb[j][0], acc[t][j]);
        if (s + 1 < nch) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                xa[t][0] = xn[t][0];
                xa[t][1] = xn[t][1];
            }
        }
    }
    // ---- epilogue: register q of a tile is pixel (q & 3) + 8 (q >> 2) + 4 (lane >> 5) of the tile, lane & 31 is the channel
    const bool stats = a.gn_out != nullptr;
    double st_s[CT], st_ss[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) st_s[j] = st_ss[j] = 0.0;
    // a wave whose 64 pixels and 32 CT channels all exist (every wave of the encoder's maps) takes the straight form: 32-bit element
    // offsets, no per-element bounds, the output flags compile-time (as in conv_x3_kernel: the guarded form below was 5 600 of the
    // kernel's 6 800 instructions)
    const bool full = p0 + 64 <= npix && n0 + 32 * CT <= a.cout && a.fast_offsets;   // (wave-uniform)
    if (full) {
        const float sl = a.act == 1 ? a.slope : 1.0f;
        auto emit = [&](auto res_c, auto st_c) {
            constexpr bool RES = decltype(res_c)::value, ST = decltype(st_c)::value;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned pix0 = (unsigned)(p0 + 32 * t + 4 * kh);
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    const unsigned co = (unsigned)(n0 + 32 * j + li);
                    const float b = a.bias ? a.bias[co] : 0.f;
                    float rv[16];
                    if (RES) {
                        const float *rp = a.res + (pix0 * (unsigned)a.res_ld + co);
#pragma unroll
                        for (int q = 0; q < 16; ++q) rv[q] = rp[((q & 3) + 8 * (q >> 2)) * a.res_ld];
                    }
                    float *yp = a.y + (pix0 * (unsigned)a.y_ld + co);
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        float v = acc[t][j][q] + b;
                        v = v > 0.f ? v : sl * v;
                        if (RES) v += rv[q];
                        yp[((q & 3) + 8 * (q >> 2)) * a.y_ld] = v;
                        if (ST) {
                            const double d = (double)v;
                            st_s[j] += d;
                            st_ss[j] += d * d;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);   // (one tile's residual loads in flight at a time: registers)
                }
            }
        };
        using T = std::true_type;
        using F = std::false_type;
        if (a.res) { if (stats) emit(T(), T()); else emit(T(), F()); }
        else       { if (stats) emit(F(), T()); else emit(F(), F()); }
    } else {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int co = n0 + 32 * j + li;
            if (co >= a.cout) continue;
            const float b = a.bias ? a.bias[co] : 0.f;
            float rv[16];   // (all 16 residual loads in flight together: see conv_x3_kernel)
            if (a.res) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const long long pix = p0 + 32 * t + (q & 3) + 8 * (q >> 2) + 4 * kh;
                    rv[q] = pix < npix ? a.res[(size_t)pix * a.res_ld + co] : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const long long pix = p0 + 32 * t + (q & 3) + 8 * (q >> 2) + 4 * kh;
                if (pix >= npix) continue;
                float v = acc[t][j][q] + b;
                if (a.act == 1) v = v > 0.f ? v : a.slope * v;
                if (a.res) v += rv[q];
                a.y[(size_t)pix * a.y_ld + co] = v;
                if (stats) {
                    const double d = (double)v;
                    st_s[j] += d;
                    st_ss[j] += d * d;
                }
            }
        }
    }
    if (stats) {
        // as in conv_x3_kernel: lanes, then the two pixel halves of the workgroup (waves ph = 0, 1) through LDS; one slot per
        // 128-pixel block (blockIdx.x)
        const int cg = a.cout / 32, gpt = 32 / cg;
        __syncthreads();   // (everybody is done with gn[])
        double *red = reinterpret_cast<double *>(gn);   // [4 waves][CT][32][2] doubles = 8 KB at CT = 4: the 8 KB of gn[]
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            double S = st_s[j], SS = st_ss[j];
            gn_fold_lanes(S, SS, cg);
            if (lane < 32 && (li & (cg - 1)) == 0) {
                red[((wave * CT + j) * 32 + li / cg) * 2] = S;
                red[((wave * CT + j) * 32 + li / cg) * 2 + 1] = SS;
            }
        }
        __syncthreads();
        if (tid < 2 * CT * gpt) {   // (channel half chh, tile j, group k of the tile)
            const int chh = tid / (CT * gpt), j = (tid / gpt) % CT, k = tid % gpt;
            const int g = (blockIdx.y * (64 * CT) + 32 * CT * chh + 32 * j) / cg + k;
            if (g < 32) {
                const int w0 = 2 * chh, w1 = 2 * chh + 1;   // the waves (ph = 0, 1) of this channel half
                const double S = red[((w0 * CT + j) * 32 + k) * 2] + red[((w1 * CT + j) * 32 + k) * 2];
                const double SS = red[((w0 * CT + j) * 32 + k) * 2 + 1] + red[((w1 * CT + j) * 32 + k) * 2 + 1];
                a.gn_out[((size_t)g * gridDim.x + blockIdx.x) * 2] = S;
                a.gn_out[((size_t)g * gridDim.x + blockIdx.x) * 2 + 1] = SS;
            }
        }
    }
}

static int launch_conv1x1_x2(const ConvArgs &a, const unsigned short *wsplit, hipStream_t st) {
    SURS_REQUIRE(a.cin_pad <= 1024, "split-operand convolution: at most 1024 input channels");
    const long long npix = (long long)a.h * a.w;
    const unsigned gx = (unsigned)((npix + 127) / 128);
    ConvArgs b = a;
    b.fast_offsets = npix * a.y_ld < (1ll << 31) && npix * (a.res ? a.res_ld : 1) < (1ll << 31);
    if (a.cout_pad % 256 == 0)
        hipLaunchKernelGGL(conv1x1_x2_kernel<4>, dim3(gx, a.cout_pad / 256), dim3(256), 0, st, b, wsplit);
    else
        hipLaunchKernelGGL(conv1x1_x2_kernel<1>, dim3(gx, a.cout_pad / 64), dim3(256), 0, st, b, wsplit);
    SURS_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- 3x3 convolutions with 3 input or 3 output channels (direct, fp32 FMA)
// The first and the last convolution of the super-resolution net at 1024^2 (3 -> 32 and 32 -> 3): 864 multiply-adds per pixel, far
// too thin for a 32 x 32 x k matrix tile (the MFMA kernel pads 3 channels to 16 / 64 and took 230 us each: 24 TFLOP/s).  EIGHT lanes
// share a pixel, each with 4 of the 32 wide channels, so that the 128-byte side of every pixel is one coalesced access (one lane per
// pixel ran at the rate of its 128-byte-strided accesses: 160 - 550 us).  Weights (the packed [tap][cin_pad][cout_pad] image of
// surs_conv_pack_weights) in LDS; fp32 FMAs, taps in order.  Zero padding; bias, LeakyReLU, residual as in conv_kernel; no fused
// GroupNorm (not on the path).
__global__ __launch_bounds__(256) void conv3x3_3to32_kernel(ConvArgs a) {   // cin == 3, cout == 32
    __shared__ __attribute__((aligned(16))) float ws[27 * 32];   // [tap * 3 + c][32]
    for (int i = threadIdx.x; i < 27 * 32; i += 256) {
        const int j = i & 31, tc = i >> 5;
        ws[i] = a.wp[((size_t)(tc / 3) * a.cin_pad + tc % 3) * a.cout_pad + j];
    }
    __syncthreads();
    const unsigned t = blockIdx.x * 256u + threadIdx.x, pix = t >> 3;
    const int g = (int)(t & 7u);   // output channels 4 g .. 4 g + 3
    if (pix >= (unsigned)a.ho * a.wo) return;
    const int oy = (int)(pix / (unsigned)a.wo), ox = (int)(pix - (unsigned)oy * a.wo);
    f32x4 acc = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
        const bool ok = iy >= 0 && iy < a.h && ix >= 0 && ix < a.w;
        const float *px = a.x + ((size_t)(ok ? iy : 0) * a.w + (ok ? ix : 0)) * a.x_ld;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = ok ? px[c] : 0.f;
            const f32x4 wr = *reinterpret_cast<const f32x4 *>(&ws[(tap * 3 + c) * 32 + 4 * g]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(v, wr[j], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (a.act == 1) acc[j] = acc[j] > 0.f ? acc[j] : a.slope * acc[j];
        if (a.res) acc[j] += a.res[(size_t)pix * a.res_ld + 4 * g + j];
    }
    *reinterpret_cast<f32x4 *>(a.y + (size_t)pix * a.y_ld + 4 * g) = acc;
}

// The same convolution for whole tiles of 16 rows x 32 pixels (the head of the super-resolution net at 1024^2: 8.4 M lanes of the
// kernel above each fetched 27 weight vectors from LDS and 27 input values through the texture path for 108 fmas - 120 us for a
// 134 MB map).  A lane keeps its pixel column and its four output channels for the tile's 16 rows: the 27 weight vectors live in
// registers, the 18 x 34-pixel input patch is staged in LDS once and a lane reads three new values per row (the 3 x 3 window
// slides down in registers).  The same fmas in the same order: the same bits.
constexpr int H3_ROWS = 16;
__global__ __launch_bounds__(256) void conv3x3_3to32_rows_kernel(ConvArgs a) {
    __shared__ float patch[(H3_ROWS + 2) * 34 * 3];
    const int tid = threadIdx.x, g = tid & 7, px = tid >> 3;
    const int tiles_x = a.wo / 32;
    const int ox0 = ((int)blockIdx.x % tiles_x) * 32, oy0 = ((int)blockIdx.x / tiles_x) * H3_ROWS;
    for (int i = tid; i < (H3_ROWS + 2) * 34 * 3; i += 256) {
        const int c = i % 3, p = i / 3, ix = ox0 + p % 34 - 1, iy = oy0 + p / 34 - 1;
        const bool ok = iy >= 0 && iy < a.h && ix >= 0 && ix < a.w;
        patch[i] = ok ? a.x[((size_t)iy * a.w + ix) * a.x_ld + c] : 0.f;
    }
    f32x4 wr[27];
#pragma unroll
    for (int tc = 0; tc < 27; ++tc)
        wr[tc] = *reinterpret_cast<const f32x4 *>(a.wp + ((size_t)(tc / 3) * a.cin_pad + tc % 3) * a.cout_pad + 4 * g);
    const f32x4 b4 = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    float win[3][9];   // [patch row of the window][dx * 3 + c]
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 9; ++k) win[r + 1][k] = patch[(r * 34 + px) * 3 + k];
#pragma unroll
    for (int r = 0; r < H3_ROWS; ++r) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            win[0][k] = win[1][k];
            win[1][k] = win[2][k];
            win[2][k] = patch[((r + 2) * 34 + px) * 3 + k];
        }
        f32x4 acc = b4;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = win[tap / 3][(tap % 3) * 3 + c];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(v, wr[tap * 3 + c][j], acc[j]);
            }
        const size_t pix = (size_t)(oy0 + r) * a.wo + ox0 + px;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (a.act == 1) acc[j] = acc[j] > 0.f ? acc[j] : a.slope * acc[j];
            if (a.res) acc[j] += a.res[pix * a.res_ld + 4 * g + j];
        }
        *reinterpret_cast<f32x4 *>(a.y + pix * a.y_ld + 4 * g) = acc;
    }
}

__global__ __launch_bounds__(256) void conv3x3_32to3_kernel(ConvArgs a) {   // cin == 32, cout <= 4
    __shared__ __attribute__((aligned(16))) float ws[9 * 32 * 4];   // [tap * 32 + c][4]
    for (int i = threadIdx.x; i < 9 * 32 * 4; i += 256) {
        const int j = i & 3, tc = i >> 2;
        ws[i] = a.wp[((size_t)(tc >> 5) * a.cin_pad + (tc & 31)) * a.cout_pad + j];   // (cout_pad >= 64: columns >= cout are zero)
    }
    __syncthreads();
    const unsigned t = blockIdx.x * 256u + threadIdx.x, pix = t >> 3;
    const int g = (int)(t & 7u);   // input channels 4 g .. 4 g + 3
    const bool live = pix < (unsigned)a.ho * a.wo;   // (the lanes of a pixel stay together: they exchange partial sums below)
    const unsigned pc = live ? pix : 0u;
    const int oy = (int)(pc / (unsigned)a.wo), ox = (int)(pc - (unsigned)oy * a.wo);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
        const bool ok = iy >= 0 && iy < a.h && ix >= 0 && ix < a.w;
        f32x4 v = *reinterpret_cast<const f32x4 *>(a.x + ((size_t)(ok ? iy : 0) * a.w + (ok ? ix : 0)) * a.x_ld + 4 * g);
        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 wr = *reinterpret_cast<const f32x4 *>(&ws[(tap * 32 + 4 * g + k) * 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(v[k], wr[j], acc[j]);
        }
    }
    // the eight partial sums of a pixel: a butterfly over the lane bits 0..2 (every lane ends with the same total)
#pragma unroll
    for (int o = 1; o < 8; o <<= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += __shfl_xor(acc[j], o);
    if (live && g < a.cout) {
        float r = acc[0];
        if (g == 1) r = acc[1];
        if (g == 2) r = acc[2];
        if (g == 3) r = acc[3];
        r += a.bias ? a.bias[g] : 0.f;
        if (a.act == 1) r = r > 0.f ? r : a.slope * r;
        if (a.res) r += a.res[(size_t)pix * a.res_ld + g];
        a.y[(size_t)pix * a.y_ld + g] = r;
    }
}

// ---------------------------------------------------------------- GroupNorm coefficients
// two launches: (1) GN_SPLIT workgroups reduce a pixel slice each - ALL channels of it, so that the NHWC rows are read whole and
// coalesced (16 bytes per thread): double sums of x and x^2 per thread and channel quad, folded per group in a fixed order; (2)
// one small launch folds the partials with the affine parameters.  (Round 1: one workgroup per group took 15 of the encoder's
// 28 ms; round 2: one workgroup per (group, slice) read 32-byte pieces with a division per element: 1.4 TB/s.)
constexpr int GN_SPLIT = 512;

__global__ __launch_bounds__(256) void gn_partial_kernel(const float *__restrict__ x, int hw, int c, int x_ld, int groups,
                                                         double *__restrict__ partial /* [groups][GN_SPLIT][2] */) {
    __shared__ double red[256][8];
    const int sp = blockIdx.x, cg = c / groups, tid = threadIdx.x;
    const int p0 = (int)((long long)hw * sp / GN_SPLIT), p1 = (int)((long long)hw * (sp + 1) / GN_SPLIT);
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    const int c4 = c / 4;                 // channel quads per pixel (the host checks c % 4 == 0, c <= 1024, 16-byte aligned rows)
    const int ppl = 256 / c4;             // pixels in flight per iteration (>= 1)
    const int q = tid % c4, pl = tid / c4;
    if (pl < ppl)
        for (int p = p0 + pl; p < p1; p += ppl) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(x + (size_t)p * x_ld + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double d = v[k];
                s[k] += d;
                ss[k] += d * d;
            }
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[tid][k] = s[k];
        red[tid][4 + k] = ss[k];
    }
    __syncthreads();
    if (tid < groups) {
        // channels [g cg, (g + 1) cg): quad ch / 4, element ch % 4, of every pixel lane, in a fixed order
        double S = 0, SS = 0;
        for (int l = 0; l < ppl; ++l)
            for (int ch = tid * cg; ch < (tid + 1) * cg; ++ch) {
                S += red[l * c4 + ch / 4][ch % 4];
                SS += red[l * c4 + ch / 4][4 + ch % 4];
            }
        partial[((size_t)tid * GN_SPLIT + sp) * 2 + 0] = S;
        partial[((size_t)tid * GN_SPLIT + sp) * 2 + 1] = SS;
    }
}

__global__ __launch_bounds__(64) void gn_finish_kernel(const double *__restrict__ partial, int hw, int c, int groups, float eps,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       float *__restrict__ scale, float *__restrict__ shift) {
    // one wave per group: lane l adds partials l, l + 64, ... in order, then a butterfly over the lanes - a fixed summation tree
    const int g = blockIdx.x, lane = threadIdx.x, cg = c / groups;
    double S = 0, SS = 0;
    for (int sp = lane; sp < GN_SPLIT; sp += 64) {
        S += partial[((size_t)g * GN_SPLIT + sp) * 2 + 0];
        SS += partial[((size_t)g * GN_SPLIT + sp) * 2 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        S += __shfl_xor(S, o);
        SS += __shfl_xor(SS, o);
    }
    const double n = (double)hw * cg;
    const double mean = S / n;
    double var = SS / n - mean * mean;
    if (var < 0) var = 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    for (int k = lane; k < cg; k += 64) {
        const int ch = g * cg + k;
        scale[ch] = (float)(rstd * gamma[ch]);
        shift[ch] = (float)(beta[ch] - mean * rstd * gamma[ch]);
    }
}

// ---------------------------------------------------------------- small HBM-bound kernels (one thread = one pixel x 4 channels)
__global__ void scale_shift_kernel(const float *__restrict__ x, size_t hw, int c, int x_ld, const float *__restrict__ scale,
                                   const float *__restrict__ shift, int relu, float *__restrict__ y, int y_ld) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = c / 4;
    if (i >= hw * c4) return;
    const size_t pix = i / c4;
    const int ch = (int)(i - pix * c4) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float t = x[pix * x_ld + ch + q] * scale[ch + q] + shift[ch + q];
        if (relu) t = t > 0.f ? t : 0.f;
        y[pix * y_ld + ch + q] = t;
    }
}

__global__ void avgpool2_kernel(const float *__restrict__ x, int h, int w, int c, int x_ld, float *__restrict__ y, int y_ld) {
    const int ho = h / 2, wo = w / 2;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ho * wo * c) return;
    const int ch = (int)(i % c);
    const size_t pix = i / c;
    const int ox = (int)(pix % wo), oy = (int)(pix / wo);
    const float *p = x + ((size_t)(2 * oy) * w + 2 * ox) * x_ld + ch;
    y[pix * y_ld + ch] = (p[0] + p[x_ld] + p[(size_t)w * x_ld] + p[(size_t)w * x_ld + x_ld]) * 0.25f;
}

__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
    const float A = -0.75f;
    float x = t + 1.0f;
    c[0] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
    x = t;
    c[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 1.0f - t;
    c[2] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 2.0f - t;
    c[3] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
}

__global__ void bicubic_up2_kernel(const float *__restrict__ x, int h, int w, int c, int x_ld, int align_corners,
                                   const float *__restrict__ addend, int add_ld, float *__restrict__ y, int y_ld) {
    const int ho = 2 * h, wo = 2 * w;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ho * wo * c) return;
    const int ch = (int)(i % c);
    const size_t pix = i / c;
    const int ox = (int)(pix % wo), oy = (int)(pix / wo);
    const float sy = align_corners ? (ho > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.f) : 0.5f;
    const float sx = align_corners ? (wo > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.f) : 0.5f;
    const float ry = align_corners ? sy * (float)oy : sy * ((float)oy + 0.5f) - 0.5f;
    const float rx = align_corners ? sx * (float)ox : sx * ((float)ox + 0.5f) - 0.5f;
    const int iy = (int)floorf(ry), ix = (int)floorf(rx);
    float cy[4], cx[4];
    cubic_coeffs(ry - (float)iy, cy);
    cubic_coeffs(rx - (float)ix, cx);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), h - 1);
        float r = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int xx = min(max(ix - 1 + b, 0), w - 1);
            r += cx[b] * x[((size_t)yy * w + xx) * x_ld + ch];
        }
        acc += cy[a] * r;
    }
    if (addend) acc = addend[pix * add_ld + ch] + acc;
    y[pix * y_ld + ch] = acc;
}

__global__ void pixel_shuffle2_kernel(const float *__restrict__ x, int h, int w, int c4, int x_ld, float slope,
                                      float *__restrict__ y, int y_ld) {
    const int c = c4 / 4, ho = 2 * h, wo = 2 * w;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ho * wo * c) return;
    const int ch = (int)(i % c);
    const size_t pix = i / c;
    const int ox = (int)(pix % wo), oy = (int)(pix / wo);
    const float v = x[((size_t)(oy >> 1) * w + (ox >> 1)) * x_ld + 4 * ch + 2 * (oy & 1) + (ox & 1)];
    y[pix * y_ld + ch] = v > 0.f ? v : slope * v;
}

__global__ void add3_kernel(const float *__restrict__ a, int a_ld, const float *__restrict__ b, int b_ld,
                            const float *__restrict__ c, int c_ld, size_t hw, int ch, float *__restrict__ y, int y_ld) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hw * ch) return;
    const size_t pix = i / ch;
    const int k = (int)(i - pix * ch);
    float t = a[pix * a_ld + k] + b[pix * b_ld + k];
    if (c) t += c[pix * c_ld + k];
    y[pix * y_ld + k] = t;
}

// ---- 16-byte forms of the same kernels (one thread = one pixel x 4 consecutive channels; the same operations per element in the same
// order, so the same bits): used when the channel count, every row pitch and every base address are multiples of 4 floats.  The scalar
// forms above ran at 0.1 - 1.4 TB/s (one 4-byte access per thread, 64-bit divisions per element).

// (each of the three kernels below as a functor: item i = 4 consecutive channels of one output pixel, computed, stored and returned -
//  the plain kernel and the form that also leaves GroupNorm statistics of the output run the same instructions per element)
struct AvgPool2Op {
    const float *x; int h, w, c, x_ld; float *y; int y_ld;
    __device__ __forceinline__ unsigned items() const { return (unsigned)(h / 2) * (w / 2) * (c / 4); }
    __device__ __forceinline__ void store(unsigned i, f32x4 r) const { *reinterpret_cast<f32x4 *>(y + (size_t)i * 4 + (size_t)(i / (c / 4)) * (y_ld - c)) = r; }
    __device__ __forceinline__ f32x4 compute(unsigned i) const {
    const int wo = w / 2, c4 = c / 4;
    const unsigned pix = i / c4, q = i - pix * c4, oy = pix / wo, ox = pix - oy * wo;
    const float *p = x + ((size_t)(2 * oy) * w + 2 * ox) * x_ld + 4 * q;
    const f32x4 a = *reinterpret_cast<const f32x4 *>(p), b = *reinterpret_cast<const f32x4 *>(p + x_ld);
    const f32x4 cc = *reinterpret_cast<const f32x4 *>(p + (size_t)w * x_ld), dd = *reinterpret_cast<const f32x4 *>(p + (size_t)w * x_ld + x_ld);
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = (a[k] + b[k] + cc[k] + dd[k]) * 0.25f;
    return r;
    }
};

struct BicubicUp2Op {
    const float *x; int h, w, c, x_ld, align_corners; const float *addend; int add_ld; float *y; int y_ld;
    __device__ __forceinline__ unsigned items() const { return (unsigned)(2 * h) * (2 * w) * (c / 4); }
    __device__ __forceinline__ void store(unsigned i, f32x4 r) const { *reinterpret_cast<f32x4 *>(y + (size_t)i * 4 + (size_t)(i / (c / 4)) * (y_ld - c)) = r; }
    __device__ __forceinline__ f32x4 compute(unsigned i) const {
    const int ho = 2 * h, wo = 2 * w, c4 = c / 4;
    const unsigned pix = i / c4, q = i - pix * c4;
    const int oy = (int)(pix / wo), ox = (int)(pix - (unsigned)oy * wo);
    const float sy = align_corners ? (ho > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.f) : 0.5f;
    const float sx = align_corners ? (wo > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.f) : 0.5f;
    const float ry = align_corners ? sy * (float)oy : sy * ((float)oy + 0.5f) - 0.5f;
    const float rx = align_corners ? sx * (float)ox : sx * ((float)ox + 0.5f) - 0.5f;
    const int iy = (int)floorf(ry), ix = (int)floorf(rx);
    float cy[4], cx[4];
    cubic_coeffs(ry - (float)iy, cy);
    cubic_coeffs(rx - (float)ix, cx);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), h - 1);
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int xx = min(max(ix - 1 + b, 0), w - 1);
            const f32x4 v = *reinterpret_cast<const f32x4 *>(x + ((size_t)yy * w + xx) * x_ld + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] += cx[b] * v[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += cy[a] * r[k];
    }
    if (addend) {
        const f32x4 ad = *reinterpret_cast<const f32x4 *>(addend + (size_t)pix * add_ld + 4 * q);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = ad[k] + acc[k];
    }
    return acc;
    }
};

// one thread = one INPUT pixel x 16 consecutive input channels = 4 output channels of its 2 x 2 output pixels
__global__ __launch_bounds__(256) void pixel_shuffle2_vec4_kernel(const float *__restrict__ x, int h, int w, int c4, int x_ld, float slope,
                                                                  float *__restrict__ y, int y_ld) {
    const int g = c4 / 16, wo = 2 * w;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (unsigned)h * w * g) return;
    const unsigned pix = i / g, q = i - pix * g, iy = pix / w, ix = pix - iy * w;
    const float *p = x + (size_t)pix * x_ld + 16 * q;
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4 *>(p + 4 * k);   // v[k][2 dy + dx]: output channel 4 q + k
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            f32x4 r;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = v[k][2 * dy + dx];
                r[k] = t > 0.f ? t : slope * t;
            }
            *reinterpret_cast<f32x4 *>(y + ((size_t)(2 * iy + dy) * wo + 2 * ix + dx) * y_ld + 4 * q) = r;
        }
}

struct Add3Op {
    const float *a; int a_ld; const float *b; int b_ld; const float *c; int c_ld; unsigned hw; int ch; float *y; int y_ld;
    __device__ __forceinline__ unsigned items() const { return hw * (unsigned)(ch / 4); }
    __device__ __forceinline__ void store(unsigned i, f32x4 r) const { *reinterpret_cast<f32x4 *>(y + (size_t)i * 4 + (size_t)(i / (ch / 4)) * (y_ld - ch)) = r; }
    __device__ __forceinline__ f32x4 compute(unsigned i) const {
    const int c4 = ch / 4;
    const unsigned pix = i / c4, q = i - pix * c4;
    const f32x4 va = *reinterpret_cast<const f32x4 *>(a + (size_t)pix * a_ld + 4 * q);
    const f32x4 vb = *reinterpret_cast<const f32x4 *>(b + (size_t)pix * b_ld + 4 * q);
    f32x4 t;
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = va[k] + vb[k];
    if (c) {
        const f32x4 vc = *reinterpret_cast<const f32x4 *>(c + (size_t)pix * c_ld + 4 * q);
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] += vc[k];
    }
    return t;
    }
};

template <class Op>
__global__ __launch_bounds__(256) void vec4_kernel(Op op) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < op.items()) op.store(i, op.compute(i));
}

// The same, and the GroupNorm(32, c) statistics of the output as partial sums [32][gridDim.x][2] (ConvArgs::gn_out): workgroups of
// 1024 threads (few slots for the consumer's fold, still enough threads in flight), a grid-stride loop over the items; c / 4 divides
// 1024, so a thread keeps its channel quad.  Fold: eight threads per group, thread `sub` adds the pixel lanes sub, sub + 8, ... of the
// group's channels in order, then a butterfly - fixed order.
constexpr int VS_THREADS = 1024;
template <class Op>
__global__ __launch_bounds__(VS_THREADS) void vec4_stats_kernel(Op op, int c, double *__restrict__ partial) {
    __shared__ double red[VS_THREADS][8];
    const int tid = threadIdx.x, c4 = c / 4, cg = c / 32, ppl = VS_THREADS / c4;
    const unsigned n = op.items();
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    // two items per trip: the loads of both are in flight before either is stored (x, the addend and y may alias for all the
    // compiler knows: item by item, every trip was a full memory round trip)
    const unsigned stride = gridDim.x * (unsigned)VS_THREADS;
    for (unsigned i = blockIdx.x * (unsigned)VS_THREADS + tid; i < n; i += 2 * stride) {
        const bool two = i + stride < n;
        const f32x4 v0 = op.compute(i);
        f32x4 v1 = {0.f, 0.f, 0.f, 0.f};
        if (two) v1 = op.compute(i + stride);
        op.store(i, v0);
        if (two) op.store(i + stride, v1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double d = v0[k], e = v1[k];
            s[k] += d;
            ss[k] += d * d;
            s[k] += e;       // (an item that does not exist adds zeros)
            ss[k] += e * e;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[tid][k] = s[k];
        red[tid][4 + k] = ss[k];
    }
    __syncthreads();
    if (tid < 256) {
        const int g = tid >> 3, sub = tid & 7;
        double S = 0, SS = 0;
        for (int l = sub; l < ppl; l += 8)
            for (int chn = g * cg; chn < (g + 1) * cg; ++chn) {
                S += red[l * c4 + chn / 4][chn % 4];
                SS += red[l * c4 + chn / 4][4 + chn % 4];
            }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            S += __shfl_xor(S, o);
            SS += __shfl_xor(SS, o);
        }
        if (sub == 0) {
            partial[((size_t)g * gridDim.x + blockIdx.x) * 2 + 0] = S;
            partial[((size_t)g * gridDim.x + blockIdx.x) * 2 + 1] = SS;
        }
    }
}

// Bicubic x2 with the statistics of its output, a 2 x 2 block of output pixels (x 4 channels) per item: the four outputs' 4 x 4
// neighbourhoods lie in ONE 5 x 5 window of the input (floor(s (2 i + 1)) - floor(s 2 i) is 0 or 1 for a scale s <= 1/2), read once -
// 25 loads for four outputs instead of 64 (the one-output form is bound by its loads: 62 - 79 us per 256^2 x 256 map, a twelfth of a
// stack of the hourglass).  Every output is the SAME expression as BicubicUp2Op::compute - the horizontal sums r = sum_b cx[b] v[b] in
// order, then sum_a cy[a] r[a] in order, then addend + acc - on the same (clamped) input pixels: the same bits.  Statistics as in
// vec4_stats_kernel (a thread keeps its channel quad; the fold is the same fixed tree).
constexpr int BB_THREADS = 512;   // (256 registers per lane: the window's ten horizontal sums stay in registers)
__global__ __launch_bounds__(BB_THREADS) void bicubic_block_stats_kernel(BicubicUp2Op op, double *__restrict__ partial) {
    __shared__ double red[BB_THREADS][8];
    const int tid = threadIdx.x, c = op.c, c4 = c / 4, cg = c / 32, ppl = BB_THREADS / c4;
    const int h = op.h, w = op.w, ho = 2 * h, wo = 2 * w;
    const unsigned n = (unsigned)h * w * c4;   // one item per input pixel position = one 2 x 2 output block
    const float sy = op.align_corners ? (ho > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.f) : 0.5f;
    const float sx = op.align_corners ? (wo > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.f) : 0.5f;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    const unsigned stride = gridDim.x * (unsigned)BB_THREADS;
    for (unsigned i = blockIdx.x * (unsigned)BB_THREADS + tid; i < n; i += stride) {
        const unsigned blk = i / c4, q = i - blk * c4;
        const int by = (int)(blk / w), bx = (int)(blk - (unsigned)by * w);
        int iy[2], ix[2];
        float cy[2][4], cx[2][4];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int oy = 2 * by + d, ox = 2 * bx + d;
            const float ry = op.align_corners ? sy * (float)oy : sy * ((float)oy + 0.5f) - 0.5f;
            const float rx = op.align_corners ? sx * (float)ox : sx * ((float)ox + 0.5f) - 0.5f;
            iy[d] = (int)floorf(ry);
            ix[d] = (int)floorf(rx);
            cubic_coeffs(ry - (float)iy[d], cy[d]);
            cubic_coeffs(rx - (float)ix[d], cx[d]);
        }
        const int dy = iy[1] - iy[0], dx = ix[1] - ix[0];   // 0 or 1
        // horizontal sums of the window's five rows for both output columns
        f32x4 r[5][2];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int yy = min(max(iy[0] - 1 + k, 0), h - 1);
            const float *rowp = op.x + (size_t)yy * w * op.x_ld + 4 * q;
            f32x4 v[5];
#pragma unroll
            for (int m = 0; m < 5; ++m) v[m] = *reinterpret_cast<const f32x4 *>(rowp + (size_t)min(max(ix[0] - 1 + m, 0), w - 1) * op.x_ld);
            f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f32x4 u = dx ? v[b + 1] : v[b];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r0[e] += cx[0][b] * v[b][e];
                    r1[e] += cx[1][b] * u[e];
                }
            }
            r[k][0] = r0;
            r[k][1] = r1;
        }
#pragma unroll
        for (int d = 0; d < 2; ++d)        // output row 2 by + d: window rows a + (d ? dy : 0)
#pragma unroll
            for (int g = 0; g < 2; ++g) {  // output column 2 bx + g
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f32x4 rr = (d && dy) ? r[a + 1][g] : r[a][g];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += cy[d][a] * rr[e];
                }
                const size_t pix = (size_t)(2 * by + d) * wo + (2 * bx + g);
                if (op.addend) {
                    const f32x4 ad = *reinterpret_cast<const f32x4 *>(op.addend + pix * op.add_ld + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = ad[e] + acc[e];
                }
                *reinterpret_cast<f32x4 *>(op.y + pix * op.y_ld + 4 * q) = acc;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double v = acc[e];
                    s[e] += v;
                    ss[e] += v * v;
                }
            }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[tid][k] = s[k];
        red[tid][4 + k] = ss[k];
    }
    __syncthreads();
    if (tid < 256) {
        const int g = tid >> 3, sub = tid & 7;
        double S = 0, SS = 0;
        for (int l = sub; l < ppl; l += 8)
            for (int chn = g * cg; chn < (g + 1) * cg; ++chn) {
                S += red[l * c4 + chn / 4][chn % 4];
                SS += red[l * c4 + chn / 4][4 + chn % 4];
            }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            S += __shfl_xor(S, o);
            SS += __shfl_xor(SS, o);
        }
        if (sub == 0) {
            partial[((size_t)g * gridDim.x + blockIdx.x) * 2 + 0] = S;
            partial[((size_t)g * gridDim.x + blockIdx.x) * 2 + 1] = SS;
        }
    }
}

__global__ void nchw_to_nhwc_kernel(const float *__restrict__ x, int c, size_t hw, float *__restrict__ y, int y_ld) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hw * c) return;
    const size_t pix = i / c;
    const int k = (int)(i - pix * c);
    y[pix * y_ld + k] = x[(size_t)k * hw + pix];
}

__global__ void nhwc_to_nchw_kernel(const float *__restrict__ x, int c, size_t hw, int x_ld, float *__restrict__ y) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hw * c) return;
    const int k = (int)(i / hw);
    const size_t pix = i - (size_t)k * hw;
    y[i] = x[pix * x_ld + k];
}

}  // namespace enc
}  // namespace surs

using namespace surs;
using namespace surs::enc;

static inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }
static inline bool vec4_fits(const void *p, int ld) { return (reinterpret_cast<size_t>(p) & 15) == 0 && (ld & 3) == 0; }

extern "C" int surs_conv2d_nhwc(const float *x, int h, int w, int cin, int x_ld, const float *wpacked, const float *bias,
                                float *y, int cout, int y_ld, int ksize, int stride, const float *in_scale,
                                const float *in_shift, int act, float slope, const float *residual, int res_ld,
                                void *stream) {
    SURS_REQUIRE(x && wpacked && y, "null argument");
    SURS_REQUIRE(h > 0 && w > 0 && cin > 0 && cout > 0 && x_ld >= cin && y_ld >= cout, "bad sizes");
    SURS_REQUIRE((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), "ksize must be 1 or 3, stride 1 or 2");
    SURS_REQUIRE(!(ksize == 1 && stride != 1), "1x1 convolution with stride 2 is not on the path");
    SURS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "in_scale / in_shift must come together");
    ConvArgs a{};
    a.x = x; a.h = h; a.w = w; a.cin = cin; a.x_ld = x_ld;
    a.wp = wpacked; a.cin_pad = (cin + 15) / 16 * 16; a.cout_pad = (cout + 63) / 64 * 64;
    a.bias = bias;
    const int pad = ksize / 2;
    a.y = y; a.ho = (h + 2 * pad - ksize) / stride + 1; a.wo = (w + 2 * pad - ksize) / stride + 1; a.cout = cout; a.y_ld = y_ld;
    a.in_scale = in_scale; a.in_shift = in_shift;
    a.act = act; a.slope = slope;
    a.res = residual; a.res_ld = res_ld;
    hipStream_t st = as_stream(stream);
    if (ksize == 3 && stride == 1 && !in_scale && (long long)a.ho * a.wo < (1ll << 28)) {
        const unsigned gx8 = (unsigned)(((long long)a.ho * a.wo * 8 + 255) / 256);   // eight lanes per pixel
        if (cin == 3 && cout == 32 && y_ld % 4 == 0 && (reinterpret_cast<size_t>(y) & 15) == 0 &&
            (!bias || (reinterpret_cast<size_t>(bias) & 15) == 0)) {
            if (a.wo % 32 == 0 && a.ho % H3_ROWS == 0)
                hipLaunchKernelGGL(conv3x3_3to32_rows_kernel, dim3((unsigned)((a.wo / 32) * (a.ho / H3_ROWS))), dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL(conv3x3_3to32_kernel, dim3(gx8), dim3(256), 0, st, a);
            SURS_LAUNCH_CHECK();
            return 0;
        }
        if (cout <= 4 && cin == 32 && x_ld % 4 == 0 && (reinterpret_cast<size_t>(x) & 15) == 0) {
            hipLaunchKernelGGL(conv3x3_32to3_kernel, dim3(gx8), dim3(256), 0, st, a);
            SURS_LAUNCH_CHECK();
            return 0;
        }
    }
    if (ksize == 1) return launch_conv<1, 1, 8>(a, st);
    if (stride == 1) return launch_conv<3, 1, 8>(a, st);
    return launch_conv<3, 2, 4>(a, st);
}

extern "C" int surs_conv2d_nhwc_x3(const float *x, int h, int w, int cin, int x_ld, const void *wsplit, const float *bias,
                                   float *y, int cout, int y_ld, int ksize, int stride, const float *in_scale,
                                   const float *in_shift, int act, float slope, const float *residual, int res_ld,
                                   void *stream) {
    SURS_REQUIRE(x && wsplit && y, "null argument");
    SURS_REQUIRE(h > 0 && w > 0 && cin > 0 && cout > 0 && x_ld >= cin && y_ld >= cout, "bad sizes");
    SURS_REQUIRE(ksize == 3 && stride == 1, "the split-bf16 kernel is built for 3x3, stride 1");
    SURS_REQUIRE(cin % 16 == 0 && x_ld % 4 == 0 && (reinterpret_cast<size_t>(x) & 15) == 0,
                 "the split-bf16 kernel needs cin %% 16 == 0 and 16-byte aligned pixels");
    SURS_REQUIRE((long long)h * w * x_ld < (1ll << 31), "input too large for 32-bit element offsets");
    SURS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "in_scale / in_shift must come together");
    ConvArgs a{};
    a.x = x; a.h = h; a.w = w; a.cin = cin; a.x_ld = x_ld;
    a.wp = nullptr; a.cin_pad = (cin + 15) / 16 * 16; a.cout_pad = (cout + 63) / 64 * 64;
    a.bias = bias;
    a.y = y; a.ho = h; a.wo = w; a.cout = cout; a.y_ld = y_ld;
    a.in_scale = in_scale; a.in_shift = in_shift;
    a.act = act; a.slope = slope;
    a.res = residual; a.res_ld = res_ld;
    return launch_conv_x3<3, 1, 3>(a, (const unsigned short *)wsplit, as_stream(stream));
}

struct GnLink {   // GroupNorm(32) statistics handed from kernel to kernel (ConvArgs)
    const double *in; int in_slots; const float *gamma, *beta; float eps;
    double *out; int out_capacity; int *out_slots;
};
static int conv2d_split_f16(int parts, const float *x, int h, int w, int cin, int x_ld, const void *wsplit, const float *bias,
                            float *y, int cout, int y_ld, int ksize, int stride, const float *in_scale, const float *in_shift, int act,
                            float slope, const float *residual, int res_ld, void *stream, const GnLink *gn = nullptr);

extern "C" int surs_conv2d_nhwc_x2(const float *x, int h, int w, int cin, int x_ld, const void *wsplit, const float *bias,
                                   float *y, int cout, int y_ld, int ksize, int stride, const float *in_scale,
                                   const float *in_shift, int act, float slope, const float *residual, int res_ld,
                                   void *stream) {
    return conv2d_split_f16(2, x, h, w, cin, x_ld, wsplit, bias, y, cout, y_ld, ksize, stride, in_scale, in_shift, act, slope, residual,
                            res_ld, stream);
}

extern "C" int surs_conv2d_nhwc_x1(const float *x, int h, int w, int cin, int x_ld, const void *wsplit, const float *bias,
                                   float *y, int cout, int y_ld, int ksize, int stride, const float *in_scale,
                                   const float *in_shift, int act, float slope, const float *residual, int res_ld,
                                   void *stream) {
    return conv2d_split_f16(1, x, h, w, cin, x_ld, wsplit, bias, y, cout, y_ld, ksize, stride, in_scale, in_shift, act, slope, residual,
                            res_ld, stream);
}

extern "C" int surs_conv2d_nhwc_gn(int parts, const float *x, int h, int w, int cin, int x_ld, const void *wsplit, const float *bias,
                                   float *y, int cout, int y_ld, int ksize, int stride, const double *gn_in, int gn_in_slots,
                                   const float *gamma, const float *beta, float eps, int act, float slope, const float *residual,
                                   int res_ld, double *gn_out, int gn_out_capacity, int *gn_out_slots, void *stream) {
    SURS_REQUIRE(parts == 1 || parts == 2, "one or two f16 parts");
    GnLink gn = {gn_in, gn_in_slots, gamma, beta, eps, gn_out, gn_out_capacity, gn_out_slots};
    return conv2d_split_f16(parts, x, h, w, cin, x_ld, wsplit, bias, y, cout, y_ld, ksize, stride, nullptr, nullptr, act, slope, residual,
                            res_ld, stream, &gn);
}

static int conv2d_split_f16(int parts, const float *x, int h, int w, int cin, int x_ld, const void *wsplit, const float *bias,
                                   float *y, int cout, int y_ld, int ksize, int stride, const float *in_scale,
                                   const float *in_shift, int act, float slope, const float *residual, int res_ld,
                                   void *stream, const GnLink *gn) {
    SURS_REQUIRE(x && wsplit && y, "null argument");
    SURS_REQUIRE(h > 0 && w > 0 && cin > 0 && cout > 0 && x_ld >= cin && y_ld >= cout, "bad sizes");
    SURS_REQUIRE((ksize == 3 && (stride == 1 || stride == 2)) || (ksize == 1 && stride == 1),
                 "the split-f16 kernels are built for 3x3 (stride 1, 2) and 1x1 (stride 1)");
    SURS_REQUIRE(cin % 16 == 0 && x_ld % 4 == 0 && (reinterpret_cast<size_t>(x) & 15) == 0,
                 "the split-f16 kernels need cin %% 16 == 0 and 16-byte aligned pixels");
    SURS_REQUIRE((long long)h * w * x_ld < (1ll << 31), "input too large for 32-bit element offsets");
    SURS_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "in_scale / in_shift must come together");
    ConvArgs a{};
    a.x = x; a.h = h; a.w = w; a.cin = cin; a.x_ld = x_ld;
    a.wp = nullptr; a.cin_pad = (cin + 15) / 16 * 16; a.cout_pad = (cout + 63) / 64 * 64;
    a.bias = bias;
    a.y = y; a.ho = (h + 2 * (ksize / 2) - ksize) / stride + 1; a.wo = (w + 2 * (ksize / 2) - ksize) / stride + 1; a.cout = cout; a.y_ld = y_ld;
    a.in_scale = in_scale; a.in_shift = in_shift;
    a.act = act; a.slope = slope;
    a.res = residual; a.res_ld = res_ld;
    if (gn) {
        if (gn->in) {
            SURS_REQUIRE(!in_scale, "GroupNorm input: either coefficients or statistics");
            SURS_REQUIRE(gn->gamma && gn->beta && gn->in_slots > 0 && cin % 32 == 0 && cin <= 1024, "bad GroupNorm(32) input statistics");
            a.gn_in = gn->in; a.gn_in_slots = gn->in_slots; a.gamma = gn->gamma; a.beta = gn->beta; a.eps = gn->eps;
        }
        if (gn->out) {
            const int cg = cout / 32;
            SURS_REQUIRE(gn->out_slots && cout % 32 == 0 && cg >= 1 && cg <= 32 && (cg & (cg - 1)) == 0,
                         "GroupNorm(32) output statistics: cout / 32 must be a power of two <= 32");
            // one slot per pixel tile of the launch below (launch_conv_x3's choice, restated)
            const int slots = ksize == 1 ? (int)(((long long)a.ho * a.wo + 127) / 128) : ceil_div(a.wo, TC) * ceil_div(a.ho, conv_x3_tile_rows(a, stride, parts));
            SURS_REQUIRE(slots <= gn->out_capacity, "GroupNorm statistics buffer too small: %d slots needed", slots);
            *gn->out_slots = slots;
            a.gn_out = gn->out;
        }
    }
    if (ksize == 1) return launch_conv1x1_x2(a, (const unsigned short *)wsplit, as_stream(stream));   // (HBM-bound: two parts always)
    // stride 2 (the three down-sampling convolutions of the super-resolution net): the 4-row x 32-channel tile, whose 9 x 65 pixel
    // patch fits the LDS
    if (parts == 1) {
        if (stride == 2) return launch_conv_x3_cfg<3, 2, 4, 32, 1>(a, (const unsigned short *)wsplit, as_stream(stream));
        return launch_conv_x3<3, 1, 1>(a, (const unsigned short *)wsplit, as_stream(stream));
    }
    if (stride == 2) return launch_conv_x3_cfg<3, 2, 4, 32, 2>(a, (const unsigned short *)wsplit, as_stream(stream));
    return launch_conv_x3<3, 1, 2>(a, (const unsigned short *)wsplit, as_stream(stream));
}

// One convolution of a ConvBlock (lib/model/HGFilters.py:57-73) with the block's closing sum in its epilogue: see include/surs.h.
extern "C" int surs_conv2d_nhwc_gn_sum(int parts, const float *x, int h, int w, int cin, int x_ld, const void *wsplit, const float *bias,
                                       const SursGnStats *gn_in, const float *in_scale, const float *in_shift, const float *gamma,
                                       const float *beta, float eps, float *y, int cout, int y_ld, SursGnStats *gn_out,
                                       const float *residual, int res_ld, float *y2, int y2_ld, double *gn_out2, int gn2_pitch, int gn2_g0,
                                       int gn2_cg, int *gn2_slots, void *stream) {
    SURS_REQUIRE(parts == 1 || parts == 2, "one or two f16 parts");
    SURS_REQUIRE(x && wsplit && y2 && residual, "null argument (the sum needs its second operand and its output)");
    SURS_REQUIRE(h > 0 && w > 0 && cin > 0 && cout > 0 && x_ld >= cin && y2_ld >= cout && (!y || y_ld >= cout), "bad sizes");
    SURS_REQUIRE(cin % 16 == 0 && x_ld % 4 == 0 && (reinterpret_cast<size_t>(x) & 15) == 0,
                 "the split-f16 kernels need cin %% 16 == 0 and 16-byte aligned pixels");
    SURS_REQUIRE((long long)h * w * x_ld < (1ll << 31), "input too large for 32-bit element offsets");
    SURS_REQUIRE(!(gn_in && in_scale) && (in_scale == nullptr) == (in_shift == nullptr), "GroupNorm input: either coefficients or statistics");
    ConvArgs a{};
    a.x = x; a.h = h; a.w = w; a.cin = cin; a.x_ld = x_ld;
    a.cin_pad = (cin + 15) / 16 * 16; a.cout_pad = (cout + 63) / 64 * 64;
    a.bias = bias;
    a.y = y; a.ho = h; a.wo = w; a.cout = cout; a.y_ld = y_ld;
    a.in_scale = in_scale; a.in_shift = in_shift;
    a.res = residual; a.res_ld = res_ld;
    a.y2 = y2; a.y2_ld = y2_ld;
    if (gn_in) {
        SURS_REQUIRE(gn_in->sums && gamma && beta && gn_in->slots[0] > 0 && cin % 32 == 0 && cin <= 1024, "bad GroupNorm(32) input statistics");
        a.gn_in = gn_in->sums; a.gn_in_slots = gn_in->slots[0]; a.gamma = gamma; a.beta = beta; a.eps = eps;
        a.gn_in_pitch = gn_in->pitch; a.gn_in_g1 = gn_in->g1; a.gn_in_g2 = gn_in->g2;
        a.gn_in_slots1 = gn_in->slots[1]; a.gn_in_slots2 = gn_in->slots[2];
    }
    const ConvTile tile = conv_x3_tile(a, 1, parts);
    const int trows = tile.rows;
    const int slots = ceil_div(a.wo, TC) * ceil_div(a.ho, trows);
    // (the kernel makes the second output in its whole-tile epilogue only)
    SURS_REQUIRE(a.wo % TC == 0 && a.ho % trows == 0 && cout % tile.chans == 0 &&
                 (long long)h * w * (y2_ld > res_ld ? y2_ld : res_ld) < (1ll << 31) && (!y || (long long)h * w * y_ld < (1ll << 31)),
                 "the sum in the epilogue needs whole tiles: width %% 32, height %% %d, cout %% %d (use surs_conv2d_nhwc_gn + surs_add3_gn)",
                 trows, tile.chans);
    if (gn_out) {
        const int cg = cout / 32;
        SURS_REQUIRE(y && gn_out->sums && cout % 32 == 0 && cg >= 1 && cg <= 32 && (cg & (cg - 1)) == 0 && slots <= gn_out->pitch,
                     "GroupNorm(32) output statistics: cout / 32 must be a power of two <= 32, %d slots needed", slots);
        a.gn_out = gn_out->sums;
        gn_out->pitch = slots;   // (rows of a single-kernel map lie slot count apart)
        gn_out->g1 = gn_out->g2 = 0;
        gn_out->slots[0] = gn_out->slots[1] = gn_out->slots[2] = slots;
    }
    if (gn_out2) {
        SURS_REQUIRE(gn2_slots && gn2_cg >= 1 && gn2_cg <= 32 && (gn2_cg & (gn2_cg - 1)) == 0 && cout % gn2_cg == 0 && gn2_g0 >= 0 &&
                     gn2_g0 + cout / gn2_cg <= 32 && slots <= gn2_pitch, "bad statistics of the sum: %d slots needed", slots);
        a.gn_out2 = gn_out2; a.gn2_pitch = gn2_pitch; a.gn2_cg = gn2_cg; a.gn2_g0 = gn2_g0;
        *gn2_slots = slots;
    }
    if (parts == 1) return launch_conv_x3<3, 1, 1>(a, (const unsigned short *)wsplit, as_stream(stream));
    return launch_conv_x3<3, 1, 2>(a, (const unsigned short *)wsplit, as_stream(stream));
}

extern "C" int surs_groupnorm_coeffs_ws(const float *x, int hw, int c, int x_ld, int groups, float eps, const float *gamma,
                                        const float *beta, float *scale, float *shift, void *scratch, void *stream);

extern "C" int surs_groupnorm_coeffs(const float *x, int hw, int c, int x_ld, int groups, float eps, const float *gamma,
                                     const float *beta, float *scale, float *shift, void *stream) {
    SURS_REQUIRE(x && gamma && beta && scale && shift, "null argument");
    SURS_REQUIRE(groups > 0 && groups <= 64 && c % groups == 0 && hw > 0, "bad GroupNorm shape");
    // the partial sums live in a small scratch buffer owned by the library, one per (device, stream): calls on one stream
    // are ordered, calls on different streams (two subjects' encoders side by side) get different buffers.  A few KB each,
    // kept for the life of the process.
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, double *> scratch;
    int dev = 0;
    SURS_HIP_CHECK(hipGetDevice(&dev));
    hipStream_t st = as_stream(stream);
    double *buf = nullptr;
    {
        std::lock_guard<std::mutex> lock(mu);
        double *&slot = scratch[std::make_pair(dev, st)];
        if (!slot) SURS_HIP_CHECK(hipMalloc((void **)&slot, sizeof(double) * 64 * GN_SPLIT * 2));
        buf = slot;
    }
    return surs_groupnorm_coeffs_ws(x, hw, c, x_ld, groups, eps, gamma, beta, scale, shift, buf, stream);
}

// The same with the scratch for the partial sums supplied by the caller (surs_groupnorm_scratch_bytes() bytes of device memory):
// no allocation inside - what a caller that captures its launches into a HIP graph needs (hipMalloc is not permitted on a
// capturing stream) - and no state shared between streams.
extern "C" size_t surs_groupnorm_scratch_bytes(void) { return sizeof(double) * 64 * GN_SPLIT * 2; }

extern "C" int surs_groupnorm_coeffs_ws(const float *x, int hw, int c, int x_ld, int groups, float eps, const float *gamma,
                                        const float *beta, float *scale, float *shift, void *scratch, void *stream) {
    SURS_REQUIRE(x && gamma && beta && scale && shift && scratch, "null argument");
    SURS_REQUIRE(groups > 0 && groups <= 64 && c % groups == 0 && hw > 0, "bad GroupNorm shape");
    SURS_REQUIRE(c % 4 == 0 && x_ld % 4 == 0 && (reinterpret_cast<size_t>(x) & 15) == 0 && c <= 1024,
                 "GroupNorm statistics: channels and pitch must be multiples of 4 (16-byte aligned rows), at most 1024 channels");
    hipStream_t st = as_stream(stream);
    double *buf = (double *)scratch;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(GN_SPLIT), dim3(256), 0, st, x, hw, c, x_ld, groups, buf);
    SURS_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finish_kernel, dim3(groups), dim3(64), 0, st, buf, hw, c, groups, eps, gamma, beta,
                       scale, shift);
    SURS_LAUNCH_CHECK();
    return 0;
}

// Input stage (lib/data/EvalDataset_LR_v2.py:227-243): ToTensor (uint8 / 255), Normalize(0.5, 0.5), mask multiply - the same
// three float32 operations in the same order, so the bytes equal the reference's tensor; output NHWC for the encoder.
namespace surs {
__global__ void image_prepare_kernel(const unsigned char *__restrict__ rgb, const unsigned char *__restrict__ mask, int npix,
                                     float *__restrict__ y, int y_ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float m = (float)mask[i] / 255.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = (float)rgb[3 * (size_t)i + c] / 255.0f;
        v = (v - 0.5f) / 0.5f;
        y[(size_t)i * y_ld + c] = m * v;
    }
}
}  // namespace surs

extern "C" int surs_image_prepare(const unsigned char *rgb, const unsigned char *mask, int h, int w, float *y, int y_ld, void *stream) {
    SURS_REQUIRE(rgb && mask && y && h > 0 && w > 0 && y_ld >= 3, "bad argument");
    hipLaunchKernelGGL(surs::image_prepare_kernel, dim3(blocks_for((size_t)h * w)), dim3(256), 0, as_stream(stream), rgb, mask, h * w, y, y_ld);
    SURS_LAUNCH_CHECK();
    return 0;
}

extern "C" int surs_scale_shift_act(const float *x, int hw, int c, int x_ld, const float *scale, const float *shift, int relu,
                                    float *y, int y_ld, void *stream) {
    SURS_REQUIRE(x && scale && shift && y && c % 4 == 0, "bad argument");
    hipLaunchKernelGGL(scale_shift_kernel, dim3(blocks_for((size_t)hw * (c / 4))), dim3(256), 0, as_stream(stream), x, (size_t)hw, c,
                       x_ld, scale, shift, relu, y, y_ld);
    SURS_LAUNCH_CHECK();
    return 0;
}

extern "C" int surs_avgpool2(const float *x, int h, int w, int c, int x_ld, float *y, int y_ld, void *stream) {
    SURS_REQUIRE(x && y && h >= 2 && w >= 2, "bad argument");
    if (c % 4 == 0 && vec4_fits(x, x_ld) && vec4_fits(y, y_ld) && (size_t)(h / 2) * (w / 2) * (c / 4) < (1ull << 32))
        hipLaunchKernelGGL(vec4_kernel<AvgPool2Op>, dim3(blocks_for((size_t)(h / 2) * (w / 2) * (c / 4))), dim3(256), 0, as_stream(stream),
                           AvgPool2Op{x, h, w, c, x_ld, y, y_ld});
    else
        hipLaunchKernelGGL(avgpool2_kernel, dim3(blocks_for((size_t)(h / 2) * (w / 2) * c)), dim3(256), 0, as_stream(stream), x, h, w, c,
                           x_ld, y, y_ld);
    SURS_LAUNCH_CHECK();
    return 0;
}

extern "C" int surs_bicubic_up2(const float *x, int h, int w, int c, int x_ld, int align_corners, const float *addend,
                                int add_ld, float *y, int y_ld, void *stream) {
    SURS_REQUIRE(x && y && h > 0 && w > 0, "bad argument");
    if (c % 4 == 0 && vec4_fits(x, x_ld) && vec4_fits(y, y_ld) && (!addend || vec4_fits(addend, add_ld)) &&
        (size_t)4 * h * w * (c / 4) < (1ull << 32))
        hipLaunchKernelGGL(vec4_kernel<BicubicUp2Op>, dim3(blocks_for((size_t)4 * h * w * (c / 4))), dim3(256), 0, as_stream(stream),
                           BicubicUp2Op{x, h, w, c, x_ld, align_corners, addend, add_ld, y, y_ld});
    else
        hipLaunchKernelGGL(bicubic_up2_kernel, dim3(blocks_for((size_t)4 * h * w * c)), dim3(256), 0, as_stream(stream), x, h, w, c, x_ld,
                           align_corners, addend, add_ld, y, y_ld);
    SURS_LAUNCH_CHECK();
    return 0;
}

extern "C" int surs_pixel_shuffle2(const float *x, int h, int w, int c4, int x_ld, float slope, float *y, int y_ld,
                                   void *stream) {
    SURS_REQUIRE(x && y && c4 % 4 == 0, "bad argument");
    if (c4 % 16 == 0 && vec4_fits(x, x_ld) && vec4_fits(y, y_ld) && (size_t)h * w * (c4 / 16) < (1ull << 32))
        hipLaunchKernelGGL(pixel_shuffle2_vec4_kernel, dim3(blocks_for((size_t)h * w * (c4 / 16))), dim3(256), 0, as_stream(stream), x, h,
                           w, c4, x_ld, slope, y, y_ld);
    else
        hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3(blocks_for((size_t)4 * h * w * (c4 / 4))), dim3(256), 0, as_stream(stream), x, h, w,
                           c4, x_ld, slope, y, y_ld);
    SURS_LAUNCH_CHECK();
    return 0;
}

extern "C" int surs_add3(const float *a, int a_ld, const float *b, int b_ld, const float *c, int c_ld, int hw, int ch,
                         float *y, int y_ld, void *stream) {
    SURS_REQUIRE(a && b && y, "bad argument");
    if (ch % 4 == 0 && vec4_fits(a, a_ld) && vec4_fits(b, b_ld) && (!c || vec4_fits(c, c_ld)) && vec4_fits(y, y_ld) &&
        (size_t)hw * (ch / 4) < (1ull << 32))
        hipLaunchKernelGGL(vec4_kernel<Add3Op>, dim3(blocks_for((size_t)hw * (ch / 4))), dim3(256), 0, as_stream(stream),
                           Add3Op{a, a_ld, b, b_ld, c, c_ld, (unsigned)hw, ch, y, y_ld});
    else
        hipLaunchKernelGGL(add3_kernel, dim3(blocks_for((size_t)hw * ch)), dim3(256), 0, as_stream(stream), a, a_ld, b, b_ld, c, c_ld,
                           (size_t)hw, ch, y, y_ld);
    SURS_LAUNCH_CHECK();
    return 0;
}

// ---- the three map-sized elementwise kernels of the hourglass, leaving the GroupNorm(32, c) statistics of their output for the
// ConvBlock that follows (surs_conv2d_nhwc_gn's gn_in): partial sums [32][*slots][2], *slots <= capacity written to the host
template <class Op>
static int launch_vec4_stats(const Op &op, size_t items, int c, double *gn_out, int capacity, int *slots, void *stream) {
    SURS_REQUIRE(gn_out && slots && capacity >= 1, "null statistics buffer");
    SURS_REQUIRE(c % 32 == 0 && c <= 1024 && (c & (c - 1)) == 0, "GroupNorm(32) statistics: the channel count must be a power of two in [32, 1024]");
    SURS_REQUIRE(items < (1ull << 32), "map too large");
    const size_t want = (items + VS_THREADS - 1) / VS_THREADS;
    size_t nn = want;
    if (nn > (size_t)capacity) nn = (size_t)capacity;
    if (nn > (size_t)GN_SPLIT) nn = (size_t)GN_SPLIT;
    const int n = (int)nn;
    *slots = n;
    hipLaunchKernelGGL(vec4_stats_kernel<Op>, dim3(n), dim3(VS_THREADS), 0, as_stream(stream), op, c, gn_out);
    SURS_LAUNCH_CHECK();
    return 0;
}

extern "C" int surs_avgpool2_gn(const float *x, int h, int w, int c, int x_ld, float *y, int y_ld, double *gn_out, int gn_out_capacity,
                                int *gn_out_slots, void *stream) {
    SURS_REQUIRE(x && y && h >= 2 && w >= 2, "bad argument");
    SURS_REQUIRE(c % 4 == 0 && vec4_fits(x, x_ld) && vec4_fits(y, y_ld), "the statistics form needs 16-byte aligned rows");
    return launch_vec4_stats(AvgPool2Op{x, h, w, c, x_ld, y, y_ld}, (size_t)(h / 2) * (w / 2) * (c / 4), c, gn_out, gn_out_capacity,
                             gn_out_slots, stream);
}

extern "C" int surs_bicubic_up2_gn(const float *x, int h, int w, int c, int x_ld, int align_corners, const float *addend, int add_ld,
                                   float *y, int y_ld, double *gn_out, int gn_out_capacity, int *gn_out_slots, void *stream) {
    SURS_REQUIRE(x && y && h > 0 && w > 0, "bad argument");
    SURS_REQUIRE(c % 4 == 0 && vec4_fits(x, x_ld) && vec4_fits(y, y_ld) && (!addend || vec4_fits(addend, add_ld)),
                 "the statistics form needs 16-byte aligned rows");
    // the 2 x 2-block form (a quarter of the items, 25 loads for four outputs): the same bits per output (option bicubic_block = 0:
    // the one-output form, for A/B timing)
    const bool block = option(OPT_BICUBIC_BLOCK) != 0;
    if (block && (long long)h * w * (c / 4) < (1ll << 32)) {
        SURS_REQUIRE(gn_out && gn_out_slots && gn_out_capacity >= 1, "null statistics buffer");
        SURS_REQUIRE(c % 32 == 0 && c <= 1024 && (c & (c - 1)) == 0, "GroupNorm(32) statistics: the channel count must be a power of two in [32, 1024]");
        const size_t items = (size_t)h * w * (c / 4), want = (items + BB_THREADS - 1) / BB_THREADS;
        size_t nn = want;
        if (nn > (size_t)gn_out_capacity) nn = (size_t)gn_out_capacity;
        if (nn > (size_t)GN_SPLIT) nn = (size_t)GN_SPLIT;
        *gn_out_slots = (int)nn;
        hipLaunchKernelGGL(bicubic_block_stats_kernel, dim3((unsigned)nn), dim3(BB_THREADS), 0, as_stream(stream),
                           BicubicUp2Op{x, h, w, c, x_ld, align_corners, addend, add_ld, y, y_ld}, gn_out);
        SURS_LAUNCH_CHECK();
        return 0;
    }
    return launch_vec4_stats(BicubicUp2Op{x, h, w, c, x_ld, align_corners, addend, add_ld, y, y_ld}, (size_t)4 * h * w * (c / 4), c, gn_out,
                             gn_out_capacity, gn_out_slots, stream);
}

extern "C" int surs_add3_gn(const float *a, int a_ld, const float *b, int b_ld, const float *c, int c_ld, int hw, int ch, float *y, int y_ld,
                            double *gn_out, int gn_out_capacity, int *gn_out_slots, void *stream) {
    SURS_REQUIRE(a && b && y && hw > 0, "bad argument");
    SURS_REQUIRE(ch % 4 == 0 && vec4_fits(a, a_ld) && vec4_fits(b, b_ld) && (!c || vec4_fits(c, c_ld)) && vec4_fits(y, y_ld),
                 "the statistics form needs 16-byte aligned rows");
    return launch_vec4_stats(Add3Op{a, a_ld, b, b_ld, c, c_ld, (unsigned)hw, ch, y, y_ld}, (size_t)hw * (ch / 4), ch, gn_out,
                             gn_out_capacity, gn_out_slots, stream);
}

extern "C" int surs_nchw_to_nhwc(const float *x, int c, int h, int w, float *y, int y_ld, void *stream) {
    SURS_REQUIRE(x && y, "bad argument");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(blocks_for((size_t)h * w * c)), dim3(256), 0, as_stream(stream), x, c, (size_t)h * w,
                       y, y_ld);
    SURS_LAUNCH_CHECK();
    return 0;
}

extern "C" int surs_nhwc_to_nchw(const float *x, int c, int h, int w, int x_ld, float *y, void *stream) {
    SURS_REQUIRE(x && y, "bad argument");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(blocks_for((size_t)h * w * c)), dim3(256), 0, as_stream(stream), x, c, (size_t)h * w,
                       x_ld, y);
    SURS_LAUNCH_CHECK();
    return 0;
}
