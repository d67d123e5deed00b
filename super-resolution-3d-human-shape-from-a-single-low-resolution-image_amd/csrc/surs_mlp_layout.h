// Layout of the packed SurfaceClassifier blob shared by the host packer and the kernels.
//
// Architecture (fixed; /root/reference/lib/options.py:88-97, lib/model/SurfaceClassifier.py:30-43):
//   lr: 321 -> 1024 -> 512 -> (512+321) 256 -> (256+321) 128 -> (128+321) 1
//   hr: 322 -> ...  (channel 321 = masked sigmoid output of lr)
// Feature channel order: [ lr map 256 | hr map 64 | z_feat | p_lr ]   (SuRSNet.py:151-154,176-181)
#pragma once
#include <cstddef>
#include <cstdint>

namespace surs {

constexpr int C_LR = 256, C_HR = 64, C_G = 320;  // gathered channels
constexpr int C0PAD = 336;                       // feature rows incl. z, p_lr, zero padding (multiple of 16)
constexpr int D1 = 1024, D2 = 512, D3 = 256, D4 = 128;

// column-constant vector of the grid path (fp32), one per (i,j) column, in this order:
constexpr int CC_A0_LR = 0, CC_A0_HR = 1024, CC_A2_LR = 2048, CC_A2_HR = 2304, CC_A3_LR = 2560, CC_A3_HR = 2688,
              CC_A4_LR = 2816, CC_A4_HR = 2817, CC_N = 2818, CC_PAD = 2944;  // 23 * 128

// z-vector block (fp32 offsets in floats), column independent
constexpr int ZV_W0Z_LR = 0, ZV_W0Z_HR = 1024, ZV_W0P_HR = 2048, ZV_B1_LR = 3072, ZV_B1_HR = 3584,
              ZV_W2Z_LR = 4096, ZV_W2Z_HR = 4352, ZV_W2P_HR = 4608, ZV_W3Z_LR = 4864, ZV_W3Z_HR = 4992,
              ZV_W3P_HR = 5120, ZV_W4C_LR = 5248, ZV_W4C_HR = 5376, ZV_W4Z_LR = 5504, ZV_W4Z_HR = 5505,
              ZV_W4P_HR = 5506, ZV_N = 5632;

// dense cores in MFMA A-fragment order, streamed as fixed 32 KiB slabs (16-bit elements):
//   per MLP: L1 (512x1024): 32 slabs of [2 k-steps][16 row tiles][64 lanes][8]
//            L2 core (256x512): 8 slabs of [4 k-steps][8 row tiles][64][8]
//            L3 core (128x256): 2 slabs of [8 k-steps][4 row tiles][64][8]
// The second image (core16) holds the same matrices as 1 KiB fragments [64 lanes][8] of the 16x16x32 shape, lane =
// (row n = lane & 15, k block q = lane >> 4):
//   per MLP: L1: [32 k-steps][32 row tiles]  element j = W1[16T + n][32s + 8q + j]
//            L2: [16 k-steps][16 row tiles]  element j = W2[16T + n][32s + 16(j>>2) + 4q + (j&3)]   (k order of a result
//            L3: [ 8 k-steps][ 8 row tiles]  likewise with W3                                          tile pair reused as B)
constexpr int SLAB_BYTES = 32768, SLABS_L1 = 32, SLABS_L2 = 8, SLABS_L3 = 2, SLABS_PER_MLP = 42, SLABS_TOTAL = 84;

// Layer-1 bias through the matrix pipe (column kernel v3): the accumulators start as A_bias x B_ones instead of 256
// v_accvgpr_write per lane (8 cycles each).  The bias is split into three 16-bit parts in k-slots 0..2 (hi + mid + lo
// = the fp32 value exactly in bf16; every partial sum is exact, so the accumulator starts bit-identical to the fp32 bias),
// B_ones holds 1 / B1FRAG_SCALE there.  f16 has too little range for the low parts: they are scaled by 2^12 (exact).
constexpr float B1FRAG_SCALE_BF16 = 1.0f, B1FRAG_SCALE_F16 = 4096.0f;

// split-f16 image of the dense cores (column kernel v5): fragment index (1 KiB units) of each layer inside one MLP's image
constexpr int X_F_L1 = 0, X_F_L2 = (D1 / 16) * (D2 / 32) * 2, X_F_L3 = X_F_L2 + (D2 / 16) * (D3 / 32) * 2,
              X_F_MLP = X_F_L3 + (D3 / 16) * (D4 / 32) * 2;   // 2048 + 512 + 128 = 2688 KiB per MLP

struct MlpBlobHeader {
    uint32_t magic;  // 'SURS'
    uint32_t dtype;  // SURS_BF16 / SURS_F16 of the core slabs
    // generic fp32 path; per MLP m (0 lr, 1 hr): k-major weights Wt[Kpad][M] of layers 0..3, biases, last layer
    uint32_t wt[2][4];
    uint32_t bias[2][4];
    uint32_t w4[2];  // [128 + 336 + 1] : [y3 part | feature part (zero padded) | bias b4]
    uint32_t reserved[2];
    // grid path
    uint32_t wc;    // fp32 k-major [320][CC_PAD]
    uint32_t bc;    // fp32 [CC_PAD]
    uint32_t zvec;  // fp32 [ZV_N]
    uint32_t core;  // SLABS_TOTAL * SLAB_BYTES: A fragments of the 32x32x16 MFMA shape (column kernels v1-v3)
    uint32_t total_bytes;
    uint32_t core16;  // the same cores as A fragments of the 16x16x32 shape (column kernel v4), same size
    uint32_t b1frag;  // layer-1 biases as 32x32x16 A fragments [2 MLPs][16 row tiles][64 lanes][8] (see B1FRAG_SCALE)
    // fp32 path on the bf16 matrix pipe (gemm_x3_kernel): the k-major matrices wt[m][l] and wc again, every element as
    // three bf16 parts (hi + mid + lo = the fp32 value exactly), [3 parts][Kpad / 16][M][16] uint16
    uint32_t wt3[2][4];
    uint32_t wc3;
    // fp32-grade column kernel (v5): the dense cores as TWO f16 parts per weight (hi = f16(w), lo = f16(w - hi)), 1 KiB
    // A fragments of the 32x32x16 shape in the order the waves stream them: per MLP, per layer, [k-step][row tile][part][64 lanes][8]
    uint32_t corex;
    // fp32 path with TWO f16 parts per operand (hi + lo, three products per MAC): the k-major matrices wt[m][l] and wc once
    // more, [2 parts][Kpad / 16][M / 32][2][32][8] uint16 (f16) in MFMA A-fragment order
    uint32_t wt2[2][4];
    uint32_t wc2;
    // layer-1 weights channel-major, W1t[m][c][r] = W1[r][c] in the blob's 16-bit dtype, [2 MLPs][1024][512]: column kernel v7
    // gathers the rows of the channels whose LeakyReLU branch is not constant over a z tile
    uint32_t w1t;
    // the same as two f16 parts per weight, [2 MLPs][1024][2 parts][512]: fp32-grade column kernel v8
    uint32_t w1tx;
    uint32_t pad[2];
};
static_assert(sizeof(MlpBlobHeader) % 16 == 0, "header must keep 16-byte alignment");
constexpr uint32_t MLP_MAGIC = 0x53525553u;

// The offsets depend on nothing but the constants above, so host packer and launch code both derive them here
// (no device->host read of the header is ever needed).
inline MlpBlobHeader blob_layout(uint32_t dtype) {
    MlpBlobHeader h = {};
    h.magic = MLP_MAGIC;
    h.dtype = dtype;
    size_t off = 256;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = (off + bytes + 255) / 256 * 256;
        return (uint32_t)o;
    };
    const int mout[4] = {D1, D2, D3, D4};
    const int kpad[4] = {C0PAD, D1, D2 + C0PAD, D3 + C0PAD};
    for (int m = 0; m < 2; ++m) {
        for (int l = 0; l < 4; ++l) {
            h.wt[m][l] = take((size_t)kpad[l] * mout[l] * 4);
            h.bias[m][l] = take((size_t)mout[l] * 4);
        }
        h.w4[m] = take((size_t)(D4 + C0PAD + 1) * 4);
    }
    h.wc = take((size_t)C_G * CC_PAD * 4);
    h.bc = take((size_t)CC_PAD * 4);
    h.zvec = take((size_t)ZV_N * 4);
    h.core = take((size_t)SLABS_TOTAL * SLAB_BYTES);
    h.core16 = take((size_t)SLABS_TOTAL * SLAB_BYTES);
    h.b1frag = take((size_t)2 * (D2 / 32) * 1024);
    for (int m = 0; m < 2; ++m)
        for (int l = 0; l < 4; ++l) h.wt3[m][l] = take((size_t)kpad[l] * mout[l] * 6);
    h.wc3 = take((size_t)C_G * CC_PAD * 6);
    h.corex = take((size_t)2 * X_F_MLP * 1024);
    for (int m = 0; m < 2; ++m)
        for (int l = 0; l < 4; ++l) h.wt2[m][l] = take((size_t)kpad[l] * mout[l] * 4);
    h.wc2 = take((size_t)C_G * CC_PAD * 4);
    h.w1t = take((size_t)2 * D1 * D2 * 2);
    h.w1tx = take((size_t)2 * D1 * 2 * D2 * 2);
    h.total_bytes = (uint32_t)off;
    return h;
}

}  // namespace surs
