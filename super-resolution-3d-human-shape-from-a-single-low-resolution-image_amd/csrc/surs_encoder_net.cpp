// The image encoder sequenced in C: surs_encoder_super_res / _filter_lr / _filter_hr / _forward (include/surs.h).
//
// Replaces, as ONE call each, what the reference runs as nn.Module graphs (/root/reference):
//   SuRSSR_v3.forward          lib/model/SuRSSR_v3.py:143-181   (ResBlock: lib/model/common.py:14-33)
//   HGFilter.forward low_res   lib/model/HGFilters.py:183-206   (ConvBlock :57-74, HourGlass :96-117)
//   HGFilter.forward high_res  lib/model/HGFilters.py:179-181
// Until round 6 the ~ 160 launches of an image were issued one by one from Python through ctypes (encoder.py, kept as the
// readable mirror and as the path of the non-default operand splits); a C-ABI consumer had to re-implement that sequencing.
// Here it is native: the same launches in the same order on the same tiles - the outputs equal encoder.py's bit for bit
// (tests/test_gpu_encoder_net.py) -, intermediates in a caller-supplied workspace, no allocation, no stream creation (the side
// streams of the hourglass fork are lent by the caller; without them the branches run one behind the other).
//
// Memory plan: a bump allocator over the workspace.  The super-resolution net reuses three buffers per stage for its residual
// blocks (one stream: reuse is ordered); filter_lr alternates between two arenas per stack (stack s + 2 starts after every
// kernel of stack s has been joined).  surs_encoder_workspace_bytes() runs the same sequencing with a counting allocator.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "surs_common.h"

using namespace surs;

namespace {

struct Map {   // NHWC fp32 view with channel pitch + the GroupNorm(32) statistics its producer(s) left (st.sums null: none)
    float *p = nullptr;
    int h = 0, w = 0, c = 0, ld = 0;
    SursGnStats st = {nullptr, 0, 0, 0, {0, 0, 0}};
    void set_stats(double *sums, int slots) { st = SursGnStats{sums, slots, 0, 0, {slots, slots, slots}}; }
    Map slice(int c0, int n) const { Map m = *this; m.p = p + c0; m.c = n; m.set_stats(nullptr, 0); return m; }
};

struct Arena {
    char *base = nullptr;
    size_t off = 0, cap = 0, peak = 0;
    bool dry = false;
    void *take(size_t bytes) {
        off = align_up(off, 256);
        void *r = (dry ? reinterpret_cast<char *>(4096) : base) + off;   // (a dry run hands out addresses nobody dereferences)
        off += bytes;
        if (off > peak) peak = off;
        return r;
    }
    bool ok() const { return dry || off <= cap; }
};

struct Run {
    const SursEncoderNet *net;
    Arena *a;            // current arena
    hipStream_t st;      // current stream
    int parts;           // 2 (fp32-grade) or 1 (one f16 product per MAC in the 3x3 convolutions)
    bool dry;
    int rc = 0;

    Map map(int h, int w, int c) {
        Map m;
        m.h = h; m.w = w; m.c = c; m.ld = c;
        m.p = (float *)a->take((size_t)h * w * c * sizeof(float));
        return m;
    }
    double *stats_buf(int capacity) { return (double *)a->take((size_t)32 * capacity * 2 * sizeof(double)); }
    bool fail(int code) {
        if (code && !rc) rc = code;
        return rc != 0;
    }
};

inline bool aligned16(const void *p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

// native.conv2d of encoder.py: the split-f16 kernels where they apply, the fp32 MFMA / direct kernels otherwise
void conv(Run &r, const Map &x, const SursConv &cw, Map &out, int stride, int act, float slope, const Map *residual,
          const float *in_scale = nullptr, const float *in_shift = nullptr) {
    out.set_stats(nullptr, 0);
    if (r.dry || r.rc) return;
    const bool thin = cw.ksize == 3 && stride == 1 && !in_scale && cw.cout <= 4 && cw.cin == 32;
    const bool x3 = cw.w_split && !thin && (stride == 1 || (stride == 2 && cw.ksize == 3)) && x.c % 16 == 0 && x.ld % 4 == 0 && aligned16(x.p);
    const float *res = residual ? residual->p : nullptr;
    const int res_ld = residual ? residual->ld : 0;
    int rc;
    if (x3 && r.parts == 1 && cw.ksize == 3)
        rc = surs_conv2d_nhwc_x1(x.p, x.h, x.w, x.c, x.ld, cw.w_split, cw.bias, out.p, cw.cout, out.ld, cw.ksize, stride, in_scale, in_shift,
                                 act, slope, res, res_ld, r.st);
    else if (x3)
        rc = surs_conv2d_nhwc_x2(x.p, x.h, x.w, x.c, x.ld, cw.w_split, cw.bias, out.p, cw.cout, out.ld, cw.ksize, stride, in_scale, in_shift,
                                 act, slope, res, res_ld, r.st);
    else
        rc = surs_conv2d_nhwc(x.p, x.h, x.w, x.c, x.ld, cw.w_packed, cw.bias, out.p, cw.cout, out.ld, cw.ksize, stride, in_scale, in_shift,
                              act, slope, res, res_ld, r.st);
    r.fail(rc);
}

// native.conv2d_gn: stride 1, GroupNorm(32) statistics handed from kernel to kernel (x's: one slot count for all groups)
void conv_gn(Run &r, const Map &x, const SursConv &cw, Map &out, const SursGroupNorm *gn, bool want_stats, const Map *residual = nullptr) {
    const int cap = cw.ksize == 1 ? (out.h * out.w + 127) / 128 : ((out.w + 31) / 32) * ((out.h + 3) / 4);
    double *sb = want_stats ? r.stats_buf(cap) : nullptr;
    out.set_stats(sb, 0);
    if (r.dry || r.rc) return;
    if (gn && (!x.st.sums || x.st.g1 > 0)) {
        r.fail(fail(SURS_E_INVALID, "encoder: a GroupNorm input carries no (single-kernel) statistics"));
        return;
    }
    int slots = 0;
    const int rc = surs_conv2d_nhwc_gn(r.parts, x.p, x.h, x.w, x.c, x.ld, cw.w_split, cw.bias, out.p, cw.cout, out.ld, cw.ksize, 1,
                                       gn ? x.st.sums : nullptr, gn ? x.st.slots[0] : 0, gn ? gn->gamma : nullptr, gn ? gn->beta : nullptr,
                                       1e-5f, 0, 0.0f, residual ? residual->p : nullptr, residual ? residual->ld : 0, sb, cap, &slots, r.st);
    if (r.fail(rc)) return;
    out.set_stats(sb, sb ? slots : 0);
}

inline int ew_capacity(long long items) {
    const long long n = (items + 1023) / 1024;
    return (int)(n < 512 ? n : 512);
}

void add3(Run &r, const Map &a, const Map &b, Map &out, bool want_stats) {
    const int cap = ew_capacity((long long)a.h * a.w * (a.c / 4));
    double *sb = want_stats ? r.stats_buf(cap) : nullptr;
    out.set_stats(sb, 0);
    if (r.dry || r.rc) return;
    int slots = 0;
    const int rc = want_stats ? surs_add3_gn(a.p, a.ld, b.p, b.ld, nullptr, 0, a.h * a.w, a.c, out.p, out.ld, sb, cap, &slots, r.st)
                              : surs_add3(a.p, a.ld, b.p, b.ld, nullptr, 0, a.h * a.w, a.c, out.p, out.ld, r.st);
    if (r.fail(rc)) return;
    out.set_stats(sb, slots);
}

Map avgpool2(Run &r, const Map &x, bool want_stats) {
    Map out = r.map(x.h / 2, x.w / 2, x.c);
    const int cap = ew_capacity((long long)out.h * out.w * (x.c / 4));
    double *sb = want_stats ? r.stats_buf(cap) : nullptr;
    out.set_stats(sb, 0);
    if (r.dry || r.rc) return out;
    int slots = 0;
    const int rc = want_stats ? surs_avgpool2_gn(x.p, x.h, x.w, x.c, x.ld, out.p, out.ld, sb, cap, &slots, r.st)
                              : surs_avgpool2(x.p, x.h, x.w, x.c, x.ld, out.p, out.ld, r.st);
    if (r.fail(rc)) return out;
    out.set_stats(sb, slots);
    return out;
}

void bicubic_up2(Run &r, const Map &x, bool align_corners, const Map *addend, Map &out, bool want_stats) {
    const int cap = ew_capacity((long long)out.h * out.w * (x.c / 4));
    double *sb = want_stats ? r.stats_buf(cap) : nullptr;
    out.set_stats(sb, 0);
    if (r.dry || r.rc) return;
    int slots = 0;
    const float *ad = addend ? addend->p : nullptr;
    const int ad_ld = addend ? addend->ld : 0;
    const int rc = want_stats ? surs_bicubic_up2_gn(x.p, x.h, x.w, x.c, x.ld, align_corners, ad, ad_ld, out.p, out.ld, sb, cap, &slots, r.st)
                              : surs_bicubic_up2(x.p, x.h, x.w, x.c, x.ld, align_corners, ad, ad_ld, out.p, out.ld, r.st);
    if (r.fail(rc)) return;
    out.set_stats(sb, slots);
}

void pixel_shuffle2(Run &r, const Map &x, float slope, Map &out) {
    out.set_stats(nullptr, 0);
    if (r.dry || r.rc) return;
    r.fail(surs_pixel_shuffle2(x.p, x.h, x.w, x.c, x.ld, slope, out.p, out.ld, r.st));
}

constexpr int ACT = 1;
constexpr float LRELU = 0.2f, RELU = 0.0f;

// ---------------------------------------------------------------- SuRSSR_v3.forward (lib/model/SuRSSR_v3.py:143-181)
void super_res(Run &r, const Map &x, bool want_image, float *img_sr, float *feature_lr, float *feature_hr) {
    const SursEncoderNet &n = *r.net;
    const int H2 = 2 * x.h, W2 = 2 * x.w;
    Map fin = r.map(H2, W2, 64);               // cat(h, up3)
    Map new3 = r.map(x.h, x.w, 128);           // cat(d1_f, up2)
    Map new2;                                  // cat(d2_f, up1) -> feature_lr: the caller's buffer
    new2.p = feature_lr; new2.h = x.h / 2; new2.w = x.w / 2; new2.c = new2.ld = 256;
    Map new1 = r.map(x.h / 4, x.w / 4, 512);   // cat(d3_f, bo)
    Map up = r.map(H2, W2, 3);
    bicubic_up2(r, x, false, nullptr, up, false);
    Map h = fin.slice(0, 32);
    conv(r, up, n.head, h, 1, ACT, LRELU, nullptr);

    int body0 = 0;
    auto stage = [&](int i, const Map &src, Map dst) {
        const int ho = (src.h + 2 - 3) / 2 + 1, wo = (src.w + 2 - 3) / 2 + 1, c = n.down[i].cout;
        Map buf[3] = {r.map(ho, wo, c), r.map(ho, wo, c), r.map(ho, wo, c)};
        int d = 0;
        conv(r, src, n.down[i], buf[d], 2, ACT, LRELU, nullptr);
        if (n.residual) {
            for (int b = 0; b < n.n_block[i]; ++b) {
                const int t = (d + 1) % 3, d2 = (d + 2) % 3;
                conv(r, buf[d], n.body[2 * (body0 + b)], buf[t], 1, ACT, RELU, nullptr);
                conv(r, buf[t], n.body[2 * (body0 + b) + 1], buf[d2], 1, 0, 0.0f, &buf[d]);
                d = d2;
            }
        }
        body0 += n.n_block[i];
        const int t = (d + 1) % 3;
        conv(r, buf[d], n.tail0[i], buf[t], 1, ACT, LRELU, nullptr);
        conv(r, buf[t], n.tail2[i], dst, 1, ACT, LRELU, nullptr);
        return dst;
    };
    Map d1_f = stage(0, h, new3.slice(0, 64));
    Map d2_f = stage(1, d1_f, new2.slice(0, 128));
    Map d3_f = stage(2, d2_f, new1.slice(0, 256));
    Map bo = new1.slice(256, 256);
    conv(r, d3_f, n.bottleneck, bo, 1, ACT, LRELU, nullptr);
    // conv -> LeakyReLU -> PixelShuffle -> LeakyReLU (the second LeakyReLU is fused into the shuffle)
    {
        Map t = r.map(new1.h, new1.w, n.bott2.cout), o = new2.slice(128, 128);
        conv(r, new1, n.bott2, t, 1, ACT, LRELU, nullptr);
        pixel_shuffle2(r, t, 0.2f, o);
    }
    {
        Map t = r.map(new2.h, new2.w, n.ups2.cout), o = new3.slice(64, 64);
        conv(r, new2, n.ups2, t, 1, ACT, LRELU, nullptr);
        pixel_shuffle2(r, t, 0.2f, o);
    }
    {
        Map t = r.map(new3.h, new3.w, n.ups3.cout), o = fin.slice(32, 32);
        conv(r, new3, n.ups3, t, 1, ACT, LRELU, nullptr);
        pixel_shuffle2(r, t, 0.2f, o);
    }
    Map new_fin;
    new_fin.p = feature_hr; new_fin.h = H2; new_fin.w = W2; new_fin.c = new_fin.ld = 64;
    conv(r, fin, n.ups4, new_fin, 1, ACT, LRELU, nullptr);
    if (want_image) {
        Map t = r.map(H2, W2, n.last0.cout), o;
        o.p = img_sr; o.h = H2; o.w = W2; o.c = o.ld = 3;
        conv(r, new_fin, n.last0, t, 1, ACT, LRELU, nullptr);
        conv(r, t, n.last2, o, 1, 0, 0.0f, nullptr);
    }
}

// ---------------------------------------------------------------- ConvBlock (lib/model/HGFilters.py:29-74), in_planes == out_planes
// out = cat(o1, o2, o3) + x; GroupNorm + ReLU applied in each convolution's staging from statistics handed from kernel to kernel.
// THREE launches (default): every convolution writes its own value (the next one's input, with its statistics) AND its slice of the
// sum, with the sum's statistics for the ConvBlock that follows (surs_conv2d_nhwc_gn_sum) - no closing pass over the map.
// SURS_ENC_SEPARATE_SUM (net->flags): the four-launch form of rounds 4 - 5 (convolutions into the slices, then surs_add3_gn), whose
// bits encoder.py's sequencing reproduces.  A first block (x without statistics) takes GroupNorm coefficients from
// surs_groupnorm_coeffs' two launches in front of its first convolution (in front of all three in the separate-sum form).
Map conv_block(Run &r, const SursConvBlock &b, const Map &x, bool want_stats) {
    const int c = x.c;
    Map out = r.map(x.h, x.w, c);
    Map o1 = out.slice(0, c / 2), o2 = out.slice(c / 2, c / 4), o3 = out.slice(3 * c / 4, c / 4);
    auto eligible = [&](const Map &t, const SursConv &cw) {
        return (cw.ksize == 1 || cw.ksize == 3) && cw.w_split && t.c % 32 == 0 && t.ld % 4 == 0 && aligned16(t.p);
    };
    const bool fused = c % 128 == 0 && eligible(x, b.conv[0]) && eligible(o1, b.conv[1]) && eligible(o2, b.conv[2]);
    auto coeffs = [&](const Map &t, const SursGroupNorm &g, float *&sc, float *&sh) {
        sc = (float *)r.a->take(sizeof(float) * t.c);
        sh = (float *)r.a->take(sizeof(float) * t.c);
        void *scratch = r.a->take(surs_groupnorm_scratch_bytes());
        if (!r.dry && !r.rc) r.fail(surs_groupnorm_coeffs_ws(t.p, t.h * t.w, t.c, t.ld, 32, 1e-5f, g.gamma, g.beta, sc, sh, scratch, r.st));
    };
    // (the second output is made by the kernels' whole-tile epilogue: maps of whole 8-row x 32-column x 64-channel tiles - every map
    //  of a 512 x 512 image's hourglass; smaller ones take the separate sum)
    const bool sum_in_conv = fused && !(r.net->flags & SURS_ENC_SEPARATE_SUM) && b.conv[0].ksize == 3 && b.conv[1].ksize == 3 &&
                             b.conv[2].ksize == 3 && x.w % 32 == 0 && x.h % 8 == 0 && c % 256 == 0;
    if (sum_in_conv) {
        const int cap = ((x.w + 31) / 32) * ((x.h + 3) / 4), cg = c / 32;
        Map raw1 = r.map(x.h, x.w, c / 2), raw2 = r.map(x.h, x.w, c / 4);
        double *S = want_stats ? r.stats_buf(cap) : nullptr;
        SursGnStats st1 = {r.stats_buf(cap), cap, 0, 0, {0, 0, 0}}, st2 = {r.stats_buf(cap), cap, 0, 0, {0, 0, 0}};
        float *sc = nullptr, *sh = nullptr;
        if (!x.st.sums) coeffs(x, b.bn[0], sc, sh);
        int sl[3] = {0, 0, 0};
        out.st = SursGnStats{S, cap, (c / 2) / cg, (3 * c / 4) / cg, {0, 0, 0}};
        if (r.dry || r.rc) return out;
        r.fail(surs_conv2d_nhwc_gn_sum(r.parts, x.p, x.h, x.w, c, x.ld, b.conv[0].w_split, b.conv[0].bias, x.st.sums ? &x.st : nullptr, sc, sh,
                                       b.bn[0].gamma, b.bn[0].beta, 1e-5f, raw1.p, c / 2, raw1.ld, &st1, x.p, x.ld, o1.p, out.ld, S, cap, 0, cg,
                                       &sl[0], r.st));
        if (!r.rc)
            r.fail(surs_conv2d_nhwc_gn_sum(r.parts, raw1.p, x.h, x.w, c / 2, raw1.ld, b.conv[1].w_split, b.conv[1].bias, &st1, nullptr, nullptr,
                                           b.bn[1].gamma, b.bn[1].beta, 1e-5f, raw2.p, c / 4, raw2.ld, &st2, x.p + c / 2, x.ld, o2.p, out.ld, S,
                                           cap, (c / 2) / cg, cg, &sl[1], r.st));
        if (!r.rc)
            r.fail(surs_conv2d_nhwc_gn_sum(r.parts, raw2.p, x.h, x.w, c / 4, raw2.ld, b.conv[2].w_split, b.conv[2].bias, &st2, nullptr, nullptr,
                                           b.bn[2].gamma, b.bn[2].beta, 1e-5f, nullptr, c / 4, 0, nullptr, x.p + 3 * c / 4, x.ld, o3.p, out.ld, S,
                                           cap, (3 * c / 4) / cg, cg, &sl[2], r.st));
        out.st.slots[0] = sl[0]; out.st.slots[1] = sl[1]; out.st.slots[2] = sl[2];
        if (!S) out.set_stats(nullptr, 0);
        return out;
    }
    if (fused && x.st.sums && x.st.g1 == 0) {
        conv_gn(r, x, b.conv[0], o1, &b.bn[0], true);
        conv_gn(r, o1, b.conv[1], o2, &b.bn[1], true);
        conv_gn(r, o2, b.conv[2], o3, &b.bn[2], false);
        add3(r, out, x, out, want_stats);
        return out;
    }
    const Map ins[3] = {x, o1, o2};
    Map outs[3] = {o1, o2, o3};
    for (int k = 0; k < 3; ++k) {
        float *sc, *sh;
        coeffs(ins[k], b.bn[k], sc, sh);
        conv(r, ins[k], b.conv[k], outs[k], 1, 0, 0.0f, nullptr, sc, sh);
    }
    add3(r, out, x, out, want_stats && fused);
    return out;
}

struct Events {   // per host thread: the events of the hourglass forks (created once; an event is reusable once its waits are enqueued)
    hipEvent_t e[16] = {};
    int n = 0;
    hipEvent_t get(int i) {
        if (i >= 16) return nullptr;
        if (!e[i] && hipEventCreateWithFlags(&e[i], hipEventDisableTiming) != hipSuccess) e[i] = nullptr;
        return e[i];
    }
};
thread_local Events t_events;

// HourGlass._forward (lib/model/HGFilters.py:96-117).  The two branches of a level are independent until their sum: with side
// streams lent by the caller the low-resolution one (pool -> ConvBlock -> [next level] -> ConvBlock) runs beside the
// full-resolution ConvBlock, ordered by events at the fork and the join.
Map hourglass(Run &r, const SursConvBlock *blocks, int depth, const Map &x, const SursEncoderStreams *ss) {
    // block order (encoder.EncoderWeights / include/surs.h): b1_d, b2_d, [level d - 1 ...], b2_plus_1, b3_1, ..., b3_d
    int next_block = 0;
    struct Level { const SursConvBlock *b1, *b2, *b2_plus, *b3; };
    Level lv[8];
    {
        // gen(level): b1, b2, then gen(level - 1) or b2_plus, then b3
        struct Gen {
            const SursConvBlock *blocks; int *next; Level *lv;
            void run(int level) {
                lv[level].b1 = blocks + (*next)++;
                lv[level].b2 = blocks + (*next)++;
                lv[level].b2_plus = nullptr;
                if (level > 1) run(level - 1); else lv[level].b2_plus = blocks + (*next)++;
                lv[level].b3 = blocks + (*next)++;
            }
        } g{blocks, &next_block, lv};
        g.run(depth);
    }
    int ev = 0;
    struct Fwd {
        Run &r; Level *lv; const SursEncoderStreams *ss; int *ev;
        Map low_branch(int level, const Map &inp) {
            Map pooled = avgpool2(r, inp, true);
            Map low1 = conv_block(r, *lv[level].b2, pooled, true);
            Map low2;
            if (level > 1) low2 = run(level - 1, low1);
            else low2 = conv_block(r, *lv[level].b2_plus, low1, true);
            return conv_block(r, *lv[level].b3, low2, false);
        }
        Map run(int level, const Map &inp) {
            hipStream_t side = (ss && level <= 4) ? (hipStream_t)ss->side[level - 1] : nullptr;
            hipEvent_t e_fork = nullptr, e_join = nullptr;
            if (side && !r.dry) {
                e_fork = t_events.get((*ev)++);
                e_join = t_events.get((*ev)++);
                if (!e_fork || !e_join) side = nullptr;
            }
            Map up1, low3;
            if (side && !r.dry && !r.rc) {
                hipStream_t cur = r.st;
                if (hipEventRecord(e_fork, cur) != hipSuccess || hipStreamWaitEvent(side, e_fork, 0) != hipSuccess)
                    r.fail(fail(SURS_E_HIP, "encoder: fork of the hourglass streams failed"));
                r.st = side;
                low3 = low_branch(level, inp);
                r.st = cur;
                up1 = conv_block(r, *lv[level].b1, inp, false);
                if (hipEventRecord(e_join, side) != hipSuccess || hipStreamWaitEvent(cur, e_join, 0) != hipSuccess)
                    r.fail(fail(SURS_E_HIP, "encoder: join of the hourglass streams failed"));
            } else {
                up1 = conv_block(r, *lv[level].b1, inp, false);
                low3 = low_branch(level, inp);
            }
            Map out = r.map(2 * low3.h, 2 * low3.w, low3.c);
            bicubic_up2(r, low3, true, &up1, out, true);   // up1 + up2
            return out;
        }
    } f{r, lv, ss, &ev};
    return f.run(depth, x);
}

// HGFilter.forward, low_res (lib/model/HGFilters.py:183-206).  outs[s]: where stack s's output goes (NULL: not wanted; the last
// stack's is always wanted).  The tail of a stack is two launches (three where the stack's output is wanted): conv_last leaves
// bn_end's statistics, the pointwise convolutions behind it fold them; previous + bl(t') + al(l(t')) is ONE pointwise convolution
// (`next`, packed by the caller: W = W_bl + W_al W_l) with the sum in its epilogue.
void filter_lr(Run &r, const Map &feature_lr, float *const *outs, const SursEncoderStreams *ss, Arena *arenas /* [2] */) {
    const SursEncoderNet &n = *r.net;
    int per_stack = 0;
    for (int l = n.hg_depth; l >= 1; --l) per_stack += 3;
    per_stack += 1;   // b2_plus_1
    r.a = &arenas[1];
    Map previous = conv_block(r, n.conv2, feature_lr, true);
    for (int s = 0; s < n.num_stack; ++s) {
        r.a = &arenas[s & 1];
        r.a->off = 0;   // (the arena of stack s - 2: every kernel of that stack was joined before stack s - 1 started)
        Map hg = hourglass(r, n.hg + (size_t)s * per_stack, n.hg_depth, previous, ss);
        Map ll = conv_block(r, n.top_m[s], hg, false);
        const bool last = s == n.num_stack - 1;
        Map t = r.map(ll.h, ll.w, n.conv_last[s].cout);
        conv_gn(r, ll, n.conv_last[s], t, nullptr, true);
        if (outs[s]) {
            Map o;
            o.p = outs[s]; o.h = t.h; o.w = t.w; o.c = o.ld = n.l[s].cout;
            conv_gn(r, t, n.l[s], o, &n.bn_end[s], false);
        }
        if (!last) {
            Map nx = r.map(t.h, t.w, n.next[s].cout);
            conv_gn(r, t, n.next[s], nx, &n.bn_end[s], true, &previous);
            previous = nx;
        }
    }
}

int check_net(const SursEncoderNet *n) {
    SURS_REQUIRE(n, "null network");
    SURS_REQUIRE(n->num_stack >= 1 && n->num_stack <= 16 && n->hg_depth >= 1 && n->hg_depth <= 4, "1..16 stacks, hourglass depth 1..4");
    SURS_REQUIRE(n->parts == 1 || n->parts == 2, "parts: 2 (fp32-grade) or 1");
    SURS_REQUIRE((n->flags & ~SURS_ENC_SEPARATE_SUM) == 0, "unknown flags");
    for (int i = 0; i < 3; ++i) SURS_REQUIRE(n->n_block[i] >= 0 && n->n_block[i] <= 64, "bad n_block");
    return 0;
}

Map input_map(const float *x, int h, int w, int c, int ld) {
    Map m;
    m.p = const_cast<float *>(x); m.h = h; m.w = w; m.c = c; m.ld = ld;
    return m;
}

}  // namespace

extern "C" size_t surs_encoder_workspace_bytes(const SursEncoderNet *net, int h, int w) {
    if (!net || h <= 0 || w <= 0 || check_net(net)) return 0;
    // the stages run one after the other on one workspace: the largest of them; filter_lr = two arenas
    Arena a;
    a.dry = true;
    Run r{net, &a, nullptr, net->parts, true};
    super_res(r, input_map(nullptr, h, w, 3, 3), true, nullptr, reinterpret_cast<float *>(4096), reinterpret_cast<float *>(4096));
    const size_t sr = align_up(a.peak, 256);
    Arena ar[2];
    ar[0].dry = ar[1].dry = true;
    Run r2{net, &ar[0], nullptr, net->parts, true};
    float *outs[16];
    for (int s = 0; s < 16; ++s) outs[s] = reinterpret_cast<float *>(4096);
    filter_lr(r2, input_map(reinterpret_cast<const float *>(4096), h / 2, w / 2, 256, 256), outs, nullptr, ar);
    const size_t half = align_up(ar[0].peak > ar[1].peak ? ar[0].peak : ar[1].peak, 256);
    return (sr > 2 * half ? sr : 2 * half) + 256;
}

extern "C" int surs_encoder_super_res(const SursEncoderNet *net, const float *x, int h, int w, int x_ld, int want_image, float *img_sr,
                                      float *feature_lr, float *feature_hr, void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = check_net(net)) return rc;
    SURS_REQUIRE(x && feature_lr && feature_hr && workspace && (!want_image || img_sr), "null argument");
    SURS_REQUIRE(h > 0 && w > 0 && h % 4 == 0 && w % 4 == 0 && x_ld >= 3,
                 "input image height/width must be multiples of 4 (three stride-2 stages), got %dx%d", h, w);
    Arena a;
    a.base = (char *)align_up((size_t)workspace, 256);
    a.cap = workspace_bytes - (size_t)(a.base - (char *)workspace);
    {   // enough room?  (the same sequencing, counted)
        Arena d;
        d.dry = true;
        Run rd{net, &d, nullptr, net->parts, true};
        super_res(rd, input_map(nullptr, h, w, 3, 3), want_image != 0, nullptr, reinterpret_cast<float *>(4096), reinterpret_cast<float *>(4096));
        SURS_REQUIRE(d.peak <= a.cap, "workspace too small: %zu bytes needed", d.peak + 256);
    }
    Run r{net, &a, as_stream(stream), net->parts, false};
    super_res(r, input_map(x, h, w, 3, x_ld), want_image != 0, img_sr, feature_lr, feature_hr);
    return r.rc;
}

extern "C" int surs_encoder_filter_lr(const SursEncoderNet *net, const float *feature_lr, int h, int w, int ld, float *const *outs,
                                      void *workspace, size_t workspace_bytes, const SursEncoderStreams *streams, void *stream) {
    if (int rc = check_net(net)) return rc;
    SURS_REQUIRE(feature_lr && outs && workspace && outs[net->num_stack - 1], "null argument (the last stack's output is always wanted)");
    SURS_REQUIRE(h > 0 && w > 0 && h % (1 << net->hg_depth) == 0 && w % (1 << net->hg_depth) == 0, "feature_lr size must be a multiple of 2^hg_depth");
    SURS_REQUIRE(ld >= 256 && ld % 4 == 0 && aligned16(feature_lr), "feature_lr: 256 channels, 16-byte aligned pixels");
    Arena dr[2];
    dr[0].dry = dr[1].dry = true;
    {
        Run rd{net, &dr[0], nullptr, net->parts, true};
        filter_lr(rd, input_map(reinterpret_cast<const float *>(4096), h, w, 256, ld), outs, nullptr, dr);
    }
    const size_t half = align_up(dr[0].peak > dr[1].peak ? dr[0].peak : dr[1].peak, 256);
    char *base = (char *)align_up((size_t)workspace, 256);
    SURS_REQUIRE(2 * half + (size_t)(base - (char *)workspace) <= workspace_bytes, "workspace too small: %zu bytes needed", 2 * half + 256);
    Arena ar[2];
    ar[0].base = base; ar[0].cap = half;
    ar[1].base = base + half; ar[1].cap = half;
    Run r{net, &ar[0], as_stream(stream), net->parts, false};
    filter_lr(r, input_map(feature_lr, h, w, 256, ld), outs, streams, ar);
    return r.rc;
}

extern "C" int surs_encoder_filter_hr(const SursEncoderNet *net, const float *feature_hr, int h, int w, int ld, float *out, void *stream) {
    if (int rc = check_net(net)) return rc;
    SURS_REQUIRE(feature_hr && out && h > 0 && w > 0, "null argument");
    Arena a;
    Run r{net, &a, as_stream(stream), net->parts, false};
    Map o;
    o.p = out; o.h = h; o.w = w; o.c = o.ld = net->conv5.cout;
    conv(r, input_map(feature_hr, h, w, net->conv5.cin, ld), net->conv5, o, 1, 0, 0.0f, nullptr);
    return r.rc;
}

extern "C" int surs_encoder_forward(const SursEncoderNet *net, const float *image, int h, int w, int x_ld, float *feature_lr,
                                    float *feature_hr, float *im_feat_lr, float *im_feat_hr, void *workspace, size_t workspace_bytes,
                                    const SursEncoderStreams *streams, void *stream) {
    if (int rc = check_net(net)) return rc;
    SURS_REQUIRE(im_feat_lr && im_feat_hr, "null argument");
    int rc = surs_encoder_super_res(net, image, h, w, x_ld, 0, nullptr, feature_lr, feature_hr, workspace, workspace_bytes, stream);
    if (rc) return rc;
    if ((rc = surs_encoder_filter_hr(net, feature_hr, 2 * h, 2 * w, 64, im_feat_hr, stream))) return rc;
    float *outs[16] = {};
    outs[net->num_stack - 1] = im_feat_lr;
    return surs_encoder_filter_lr(net, feature_lr, h / 2, w / 2, 256, outs, workspace, workspace_bytes, streams, stream);
}
