// HOST-side repacking of reference state-dict tensors into kernel layouts (no device code).
//   surs_mlp_pack           SurfaceClassifier x2  (lib/model/SurfaceClassifier.py:30-43)
//   surs_conv_pack_weights  nn.Conv2d weights     (lib/net_util.py:94-97)
#include <cmath>
#include <cstdlib>
#include <vector>

#include "surs_common.h"
#include "surs_mlp_layout.h"

namespace surs {

static uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static uint16_t f32_to_f16(float f) {
    _Float16 h = (_Float16)f;  // round to nearest even
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

static float bf16_to_f32(uint16_t u) {
    const uint32_t v = (uint32_t)u << 16;
    float f;
    memcpy(&f, &v, 4);
    return f;
}

static float f16_to_f32(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}

static const int kDims[2][6] = {{321, D1, D2, D3, D4, 1}, {322, D1, D2, D3, D4, 1}};

}  // namespace surs

using namespace surs;

extern "C" size_t surs_mlp_pack(const float *const w_lr[5], const float *const b_lr[5], const float *const w_hr[5],
                                const float *const b_hr[5], int dtype, void *blob) {
    if (dtype != SURS_BF16 && dtype != SURS_F16) dtype = SURS_BF16;
    MlpBlobHeader h = blob_layout((uint32_t)dtype);
    const size_t off = h.total_bytes;
    const int kin_main[4] = {0, D1, D2, D3};  // width of the y part for layers 0..3 (layer 0 has none)
    if (!blob) return off;

    const float *const *W[2] = {w_lr, w_hr};
    const float *const *B[2] = {b_lr, b_hr};
    char *base = (char *)blob;
    memset(base, 0, off);
    memcpy(base, &h, sizeof(h));

    for (int m = 0; m < 2; ++m) {
        const int c0 = kDims[m][0];
        for (int l = 0; l < 4; ++l) {
            const int mo = kDims[m][l + 1];
            const int ymain = kin_main[l];
            const int kin = (l == 0) ? c0 : (l == 1 ? D1 : ymain + c0);  // Conv1d in_channels
            float *wt = (float *)(base + h.wt[m][l]);
            // reference weight row o: [ y part (ymain) | feature part (c0) ] for l >= 2; all features for l = 0
            for (int o = 0; o < mo; ++o)
                for (int k = 0; k < kin; ++k) wt[(size_t)k * mo + o] = W[m][l][(size_t)o * kin + k];
            memcpy(base + h.bias[m][l], B[m][l], (size_t)mo * 4);
        }
        float *w4 = (float *)(base + h.w4[m]);
        memcpy(w4, W[m][4], (size_t)(D4 + c0) * 4);
        w4[D4 + C0PAD] = B[m][4][0];
    }
    // ---- grid path: column-constant matrix, k-major [320][CC_PAD]
    {
        float *wc = (float *)(base + h.wc);
        float *bc = (float *)(base + h.bc);
        auto put_rows = [&](int cc0, const float *w, int rows, int kin, int col0, const float *b) {
            for (int o = 0; o < rows; ++o) {
                for (int k = 0; k < C_G; ++k) wc[(size_t)k * CC_PAD + cc0 + o] = w[(size_t)o * kin + col0 + k];
                bc[cc0 + o] = b[o];
            }
        };
        put_rows(CC_A0_LR, W[0][0], D1, 321, 0, B[0][0]);
        put_rows(CC_A0_HR, W[1][0], D1, 322, 0, B[1][0]);
        put_rows(CC_A2_LR, W[0][2], D3, D2 + 321, D2, B[0][2]);
        put_rows(CC_A2_HR, W[1][2], D3, D2 + 322, D2, B[1][2]);
        put_rows(CC_A3_LR, W[0][3], D4, D3 + 321, D3, B[0][3]);
        put_rows(CC_A3_HR, W[1][3], D4, D3 + 322, D3, B[1][3]);
        put_rows(CC_A4_LR, W[0][4], 1, D4 + 321, D4, B[0][4]);
        put_rows(CC_A4_HR, W[1][4], 1, D4 + 322, D4, B[1][4]);
    }
    // ---- z-vectors
    {
        float *zv = (float *)(base + h.zvec);
        for (int o = 0; o < D1; ++o) {
            zv[ZV_W0Z_LR + o] = W[0][0][(size_t)o * 321 + 320];
            zv[ZV_W0Z_HR + o] = W[1][0][(size_t)o * 322 + 320];
            zv[ZV_W0P_HR + o] = W[1][0][(size_t)o * 322 + 321];
        }
        memcpy(zv + ZV_B1_LR, B[0][1], D2 * 4);
        memcpy(zv + ZV_B1_HR, B[1][1], D2 * 4);
        for (int o = 0; o < D3; ++o) {
            zv[ZV_W2Z_LR + o] = W[0][2][(size_t)o * (D2 + 321) + D2 + 320];
            zv[ZV_W2Z_HR + o] = W[1][2][(size_t)o * (D2 + 322) + D2 + 320];
            zv[ZV_W2P_HR + o] = W[1][2][(size_t)o * (D2 + 322) + D2 + 321];
        }
        for (int o = 0; o < D4; ++o) {
            zv[ZV_W3Z_LR + o] = W[0][3][(size_t)o * (D3 + 321) + D3 + 320];
            zv[ZV_W3Z_HR + o] = W[1][3][(size_t)o * (D3 + 322) + D3 + 320];
            zv[ZV_W3P_HR + o] = W[1][3][(size_t)o * (D3 + 322) + D3 + 321];
            zv[ZV_W4C_LR + o] = W[0][4][o];
            zv[ZV_W4C_HR + o] = W[1][4][o];
        }
        zv[ZV_W4Z_LR] = W[0][4][D4 + 320];
        zv[ZV_W4Z_HR] = W[1][4][D4 + 320];
        zv[ZV_W4P_HR] = W[1][4][D4 + 321];
    }
    // ---- dense cores in MFMA A-fragment order
    {
        uint16_t *core = (uint16_t *)(base + h.core);
        auto cvt = [&](float f) { return dtype == SURS_F16 ? f32_to_f16(f) : f32_to_bf16(f); };
        for (int m = 0; m < 2; ++m) {
            const int c0 = kDims[m][0];
            uint16_t *p = core + (size_t)m * SLABS_PER_MLP * (SLAB_BYTES / 2);
            // L1: natural k order inside a k-step
            for (int s = 0; s < D1 / 16; ++s)
                for (int T = 0; T < D2 / 32; ++T)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int r = lane & 31, hh = lane >> 5;
                            p[(((size_t)s * 16 + T) * 64 + lane) * 8 + j] =
                                cvt(W[m][1][(size_t)(32 * T + r) * D1 + 16 * s + 8 * hh + j]);
                        }
            p += (size_t)SLABS_L1 * (SLAB_BYTES / 2);
            // L2 / L3 cores: k order of an accumulator tile reused as B operand
            for (int s = 0; s < D2 / 16; ++s)
                for (int T = 0; T < D3 / 32; ++T)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int r = lane & 31, hh = lane >> 5;
                            const int kin = 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
                            p[(((size_t)s * 8 + T) * 64 + lane) * 8 + j] =
                                cvt(W[m][2][(size_t)(32 * T + r) * (D2 + c0) + kin]);
                        }
            p += (size_t)SLABS_L2 * (SLAB_BYTES / 2);
            for (int s = 0; s < D3 / 16; ++s)
                for (int T = 0; T < D4 / 32; ++T)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int r = lane & 31, hh = lane >> 5;
                            const int kin = 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
                            p[(((size_t)s * 4 + T) * 64 + lane) * 8 + j] =
                                cvt(W[m][3][(size_t)(32 * T + r) * (D3 + c0) + kin]);
                        }
        }
    }
    // ---- the same cores as A fragments of the 16x16x32 MFMA shape
    {
        uint16_t *core = (uint16_t *)(base + h.core16);
        auto cvt = [&](float f) { return dtype == SURS_F16 ? f32_to_f16(f) : f32_to_bf16(f); };
        for (int m = 0; m < 2; ++m) {
            const int c0 = kDims[m][0];
            uint16_t *p = core + (size_t)m * SLABS_PER_MLP * (SLAB_BYTES / 2);
            const int rows[3] = {D2, D3, D4}, kin[3] = {D1, D2, D3}, ld[3] = {D1, D2 + c0, D3 + c0};
            for (int l = 0; l < 3; ++l) {
                const int nT = rows[l] / 16, nS = kin[l] / 32;
                for (int s = 0; s < nS; ++s)
                    for (int T = 0; T < nT; ++T)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int n = lane & 15, q = lane >> 4;
                                const int k = l == 0 ? 32 * s + 8 * q + j : 32 * s + 16 * (j >> 2) + 4 * q + (j & 3);
                                p[(((size_t)s * nT + T) * 64 + lane) * 8 + j] = cvt(W[m][l + 1][(size_t)(16 * T + n) * ld[l] + k]);
                            }
                p += (size_t)nS * nT * 512;
            }
        }
    }
    // ---- the k-major fp32 matrices again as three bf16 parts in MFMA A-operand order: [3][Kpad / 16][M / 32][2][32][8]
    //      (16 k x 32 rows = 1 KB, lane 32h + r of a wave holds k = 8h..8h+7 of row r: a wave-wide 16-byte load of the
    //      block is contiguous AND already the v_mfma_f32_32x32x16_bf16 fragment)
    {
        auto split_kmajor = [&](const float *wt, int kpad, int M, uint16_t *out) {
            const size_t per_part = (size_t)kpad * M;
            for (int k = 0; k < kpad; ++k)
                for (int o = 0; o < M; ++o) {
                    float rest = wt[(size_t)k * M + o];
                    const size_t idx = ((((size_t)(k / 16) * (M / 32) + o / 32) * 2 + ((k >> 3) & 1)) * 32 + (o & 31)) * 8 + (k & 7);
                    for (int part = 0; part < 3; ++part) {
                        const uint16_t u = f32_to_bf16(rest);
                        out[part * per_part + idx] = u;
                        rest -= bf16_to_f32(u);
                    }
                }
        };
        const int mout[4] = {D1, D2, D3, D4};
        const int kpad[4] = {C0PAD, D1, D2 + C0PAD, D3 + C0PAD};
        for (int m = 0; m < 2; ++m)
            for (int l = 0; l < 4; ++l)
                split_kmajor((const float *)(base + h.wt[m][l]), kpad[l], mout[l], (uint16_t *)(base + h.wt3[m][l]));
        split_kmajor((const float *)(base + h.wc), C_G, CC_PAD, (uint16_t *)(base + h.wc3));
    }
    // ---- the k-major fp32 matrices as TWO f16 parts in the same A-operand order (the default split of the fp32 point path)
    {
        auto split_kmajor2 = [&](const float *wt, int kpad, int M, uint16_t *out) {
            const size_t per_part = (size_t)kpad * M;
            for (int k = 0; k < kpad; ++k)
                for (int o = 0; o < M; ++o) {
                    const float w = wt[(size_t)k * M + o];
                    const size_t idx = ((((size_t)(k / 16) * (M / 32) + o / 32) * 2 + ((k >> 3) & 1)) * 32 + (o & 31)) * 8 + (k & 7);
                    const uint16_t hi = f32_to_f16(w);
                    out[idx] = hi;
                    out[per_part + idx] = f32_to_f16(w - f16_to_f32(hi));
                }
        };
        const int mout[4] = {D1, D2, D3, D4};
        const int kpad[4] = {C0PAD, D1, D2 + C0PAD, D3 + C0PAD};
        for (int m = 0; m < 2; ++m)
            for (int l = 0; l < 4; ++l)
                split_kmajor2((const float *)(base + h.wt[m][l]), kpad[l], mout[l], (uint16_t *)(base + h.wt2[m][l]));
        split_kmajor2((const float *)(base + h.wc), C_G, CC_PAD, (uint16_t *)(base + h.wc2));
    }
    // ---- the dense cores as two f16 parts per weight (hi + lo), A fragments [k-step][row tile][part][lane][8]: the
    //      fp32-grade column kernel (v5) multiplies hi*hi + hi*lo + lo*hi.  Same k order inside a k-step as `core`.
    {
        uint16_t *cx = (uint16_t *)(base + h.corex);
        for (int m = 0; m < 2; ++m) {
            const int c0 = kDims[m][0];
            uint16_t *p = cx + (size_t)m * X_F_MLP * 512;
            const int rows[3] = {D2, D3, D4}, kin[3] = {D1, D2, D3}, ld[3] = {D1, D2 + c0, D3 + c0};
            for (int l = 0; l < 3; ++l) {
                const int nT = rows[l] / 32, nS = kin[l] / 16;
                for (int s = 0; s < nS; ++s)
                    for (int T = 0; T < nT; ++T)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int r = lane & 31, hh = lane >> 5;
                                const int k = l == 0 ? 16 * s + 8 * hh + j : 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
                                const float w = W[m][l + 1][(size_t)(32 * T + r) * ld[l] + k];
                                const uint16_t hi = f32_to_f16(w);
                                const uint16_t lo = f32_to_f16(w - f16_to_f32(hi));
                                const size_t f = ((size_t)s * nT + T) * 2;
                                p[(f * 64 + lane) * 8 + j] = hi;
                                p[((f + 1) * 64 + lane) * 8 + j] = lo;
                            }
                p += (size_t)nS * nT * 2 * 512;
            }
        }
    }
    // ---- layer-1 weights channel-major (column kernel v7)
    {
        uint16_t *wt = (uint16_t *)(base + h.w1t);
        auto cvt = [&](float f) { return dtype == SURS_F16 ? f32_to_f16(f) : f32_to_bf16(f); };
        for (int m = 0; m < 2; ++m)
            for (int c = 0; c < D1; ++c)
                for (int r = 0; r < D2; ++r) wt[((size_t)m * D1 + c) * D2 + r] = cvt(W[m][1][(size_t)r * D1 + c]);
        uint16_t *wx = (uint16_t *)(base + h.w1tx);
        for (int m = 0; m < 2; ++m)
            for (int c = 0; c < D1; ++c)
                for (int r = 0; r < D2; ++r) {
                    const float w = W[m][1][(size_t)r * D1 + c];
                    const uint16_t hi = f32_to_f16(w);
                    wx[(((size_t)m * D1 + c) * 2) * D2 + r] = hi;
                    wx[(((size_t)m * D1 + c) * 2 + 1) * D2 + r] = f32_to_f16(w - f16_to_f32(hi));
                }
    }
    // ---- layer-1 biases as A fragments (three exact 16-bit parts in k-slots 0..2 of lanes 0..31)
    {
        uint16_t *bf = (uint16_t *)(base + h.b1frag);
        memset(bf, 0, (size_t)2 * (D2 / 32) * 1024);
        const float scale = dtype == SURS_F16 ? B1FRAG_SCALE_F16 : B1FRAG_SCALE_BF16;
        auto cvt = [&](float f) { return dtype == SURS_F16 ? f32_to_f16(f) : f32_to_bf16(f); };
        auto back = [&](uint16_t u) { return dtype == SURS_F16 ? f16_to_f32(u) : bf16_to_f32(u); };
        for (int m = 0; m < 2; ++m)
            for (int T = 0; T < D2 / 32; ++T)
                for (int r = 0; r < 32; ++r) {
                    float rest = B[m][1][32 * T + r] * scale;
                    for (int part = 0; part < 3; ++part) {
                        const uint16_t u = cvt(rest);
                        bf[(((size_t)m * (D2 / 32) + T) * 64 + r) * 8 + part] = u;
                        rest -= back(u);
                    }
                }
    }
    return off;
}

extern "C" size_t surs_conv_pack_weights(const float *w, int cout, int cin, int ksize, float *out) {
    // kernel layout [tap][cin_pad][cout_pad], cin_pad multiple of 16, cout_pad multiple of 64
    const int cin_pad = (cin + 15) / 16 * 16, cout_pad = (cout + 63) / 64 * 64, taps = ksize * ksize;
    const size_t n = (size_t)taps * cin_pad * cout_pad;
    if (!out) return n;
    memset(out, 0, n * sizeof(float));
    for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < taps; ++t)
                out[((size_t)t * cin_pad + c) * cout_pad + o] = w[((size_t)o * cin + c) * taps + t];
    return n;
}

// Split-bf16 layout of a conv weight for surs_conv2d_nhwc_x3: every weight as three bf16 parts (hi + mid + lo = the fp32
// value exactly), [3 parts][k*k taps][cin_pad / 16 chunks][cout_pad][16 channels of the chunk] uint16 - the order the
// kernel's B fragments are read in (a lane reads 8 consecutive channels of one output channel).
extern "C" size_t surs_conv_pack_weights_x3(const float *w, int cout, int cin, int ksize, void *out) {
    const int cin_pad = (cin + 15) / 16 * 16, cout_pad = (cout + 63) / 64 * 64, taps = ksize * ksize, nch = cin_pad / 16;
    const size_t per_part = (size_t)taps * nch * cout_pad * 16;
    const size_t bytes = 3 * per_part * sizeof(uint16_t);
    if (!out) return bytes;
    uint16_t *o = (uint16_t *)out;
    memset(o, 0, bytes);
    for (int oc = 0; oc < cout; ++oc)
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < taps; ++t) {
                float rest = w[((size_t)oc * cin + c) * taps + t];
                const size_t idx = (((size_t)t * nch + c / 16) * cout_pad + oc) * 16 + (c & 15);
                for (int part = 0; part < 3; ++part) {
                    const uint16_t u = f32_to_bf16(rest);
                    o[part * per_part + idx] = u;
                    rest -= bf16_to_f32(u);
                }
            }
    return bytes;
}

// The same layout with TWO f16 parts per weight (hi = f16(w), lo = f16(w - hi)) for surs_conv2d_nhwc_x2:
// [2 parts][k*k taps][cin_pad / 16 chunks][cout_pad][16 channels of the chunk] uint16.
extern "C" size_t surs_conv_pack_weights_x2(const float *w, int cout, int cin, int ksize, void *out) {
    const int cin_pad = (cin + 15) / 16 * 16, cout_pad = (cout + 63) / 64 * 64, taps = ksize * ksize, nch = cin_pad / 16;
    const size_t per_part = (size_t)taps * nch * cout_pad * 16;
    const size_t bytes = 2 * per_part * sizeof(uint16_t);
    if (!out) return bytes;
    uint16_t *o = (uint16_t *)out;
    memset(o, 0, bytes);
    for (int oc = 0; oc < cout; ++oc)
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < taps; ++t) {
                const float v = w[((size_t)oc * cin + c) * taps + t];
                const size_t idx = (((size_t)t * nch + c / 16) * cout_pad + oc) * 16 + (c & 15);
                const uint16_t hi = f32_to_f16(v);
                o[idx] = hi;
                o[per_part + idx] = f32_to_f16(v - f16_to_f32(hi));
            }
    return bytes;
}
