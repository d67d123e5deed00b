/*
 * ORACLE - test infrastructure only.  Never linked into or called by the
 * product path (surs_amd / csrc); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker.
 *
 * Plain-C restatement (fp32, NCHW like the reference) of the arithmetic on the
 * SuRS occupancy-query path.  Each function cites the reference lines it
 * follows.  Pinned against outputs of the reference itself, captured in the
 * build container by tools/gen_golden.py into tests/golden/ (the reference has
 * no tests of its own: SURVEY.md section 4).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ encoder primitives */

/* nn.Conv2d(k in {1,3,7}, stride, padding=k/2), NCHW, weight [Cout][Cin][k][k].
 * /root/reference/lib/net_util.py:94-97 (conv3x3), lib/model/SuRSSR_v3.py:46-138,
 * lib/model/HGFilters.py:153-174 (1x1 convs).  bias may be NULL. */
void orc_conv2d(const float *x, int cin, int h, int w, const float *wt, const float *bias, int cout, int k,
                int stride, float *y) {
    const int pad = k / 2;
    const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
#pragma omp parallel for schedule(static)
    for (int co = 0; co < cout; ++co) {
        float *yo = y + (size_t)co * ho * wo;
        const float b = bias ? bias[co] : 0.0f;
        for (int i = 0; i < ho * wo; ++i) yo[i] = b;
        for (int ci = 0; ci < cin; ++ci) {
            const float *xi = x + (size_t)ci * h * w;
            const float *wk = wt + ((size_t)co * cin + ci) * k * k;
            for (int ky = 0; ky < k; ++ky)
                for (int kx = 0; kx < k; ++kx) {
                    const float wv = wk[ky * k + kx];
                    for (int oy = 0; oy < ho; ++oy) {
                        const int iy = oy * stride + ky - pad;
                        if (iy < 0 || iy >= h) continue;
                        /* ox range with 0 <= ox*stride + kx - pad < w */
                        int ox0 = 0;
                        while (ox0 < wo && ox0 * stride + kx - pad < 0) ++ox0;
                        int ox1 = wo;
                        while (ox1 > ox0 && (ox1 - 1) * stride + kx - pad >= w) --ox1;
                        const float *xr = xi + (size_t)iy * w + (kx - pad);
                        float *yr = yo + (size_t)oy * wo;
                        if (stride == 1)
                            for (int ox = ox0; ox < ox1; ++ox) yr[ox] += wv * xr[ox];
                        else
                            for (int ox = ox0; ox < ox1; ++ox) yr[ox] += wv * xr[ox * stride];
                    }
                }
        }
    }
}

/* nn.GroupNorm(G, C), eps, affine, biased variance.  lib/model/HGFilters.py:41-45,140,164 */
void orc_group_norm(const float *x, int c, int hw, int groups, const float *gamma, const float *beta, float eps,
                    float *y) {
    const int cg = c / groups;
#pragma omp parallel for schedule(static)
    for (int g = 0; g < groups; ++g) {
        const float *xg = x + (size_t)g * cg * hw;
        const size_t n = (size_t)cg * hw;
        double s = 0.0, ss = 0.0;
        for (size_t i = 0; i < n; ++i) s += xg[i];
        const double mean = s / (double)n;
        for (size_t i = 0; i < n; ++i) { double d = xg[i] - mean; ss += d * d; }
        const double rstd = 1.0 / sqrt(ss / (double)n + (double)eps);
        for (int ci = 0; ci < cg; ++ci) {
            const int ch = g * cg + ci;
            const float sc = (float)(rstd * gamma[ch]);
            const float sh = (float)(beta[ch] - mean * rstd * gamma[ch]);
            const float *xi = x + (size_t)ch * hw;
            float *yi = y + (size_t)ch * hw;
            for (int i = 0; i < hw; ++i) yi[i] = xi[i] * sc + sh;
        }
    }
}

/* F.avg_pool2d(x, 2, stride=2).  lib/model/HGFilters.py:101 */
void orc_avg_pool2(const float *x, int c, int h, int w, float *y) {
    const int ho = h / 2, wo = w / 2;
#pragma omp parallel for schedule(static)
    for (int ch = 0; ch < c; ++ch)
        for (int oy = 0; oy < ho; ++oy)
            for (int ox = 0; ox < wo; ++ox) {
                const float *p = x + ((size_t)ch * h + 2 * oy) * w + 2 * ox;
                y[((size_t)ch * ho + oy) * wo + ox] = (p[0] + p[1] + p[w] + p[w + 1]) * 0.25f;
            }
}

static void cubic_coeffs(float t, float c[4]) {
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

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Bicubic x2 upsampling (A=-0.75, border-clamped taps), both alignment flavours:
 *  align_corners=1: F.interpolate(scale_factor=2, mode='bicubic', align_corners=True)  lib/model/HGFilters.py:115
 *  align_corners=0: nn.Upsample(scale_factor=2, mode='bicubic', align_corners=False)   lib/model/SuRSSR_v3.py:140,144 */
void orc_bicubic_up2(const float *x, int c, int h, int w, int align_corners, float *y) {
    const int ho = 2 * h, wo = 2 * w;
    const float sy = align_corners ? (ho > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.0f) : 0.5f;
    const float sx = align_corners ? (wo > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.0f) : 0.5f;
#pragma omp parallel for schedule(static)
    for (int ch = 0; ch < c; ++ch) {
        const float *xi = x + (size_t)ch * h * w;
        float *yo = y + (size_t)ch * ho * wo;
        for (int oy = 0; oy < ho; ++oy) {
            const float ry = align_corners ? sy * (float)oy : sy * ((float)oy + 0.5f) - 0.5f;
            const int iy = (int)floorf(ry);
            float cy[4];
            cubic_coeffs(ry - (float)iy, cy);
            for (int ox = 0; ox < wo; ++ox) {
                const float rx = align_corners ? sx * (float)ox : sx * ((float)ox + 0.5f) - 0.5f;
                const int ix = (int)floorf(rx);
                float cx[4];
                cubic_coeffs(rx - (float)ix, cx);
                float acc = 0.0f;
                for (int i = 0; i < 4; ++i) {
                    const float *row = xi + (size_t)clampi(iy - 1 + i, 0, h - 1) * w;
                    float r = 0.0f;
                    for (int j = 0; j < 4; ++j) r += cx[j] * row[clampi(ix - 1 + j, 0, w - 1)];
                    acc += cy[i] * r;
                }
                yo[(size_t)oy * wo + ox] = acc;
            }
        }
    }
}

/* ------------------------------------------------------------------ point query */

typedef struct {
    /* SurfaceClassifier weights, row-major [out][in] as Conv1d(k=1) stores them (last dim 1 dropped)
     * lib/model/SurfaceClassifier.py:30-43 ; dims[0] = 321 (lr) or 322 (hr) */
    const float *w[5];
    const float *b[5];
    int dims[6];
} orc_mlp_t;

#define PB 64 /* points per block */

/* SurfaceClassifier.forward, no_residual=False, res_layers {2,3,4}, leaky_relu(0.01), no last_op here.
 * lib/model/SurfaceClassifier.py:53-81.  feat: [c0][PB] (channel-major like the reference's [C,N]);
 * out: pre-sigmoid logit per point. */
static void mlp_block(const orc_mlp_t *m, const float *feat, int np, float *logit, float *bufa, float *bufb) {
    const int c0 = m->dims[0];
    const float *cur = feat;
    int ccur = c0;
    float *outb = bufa;
    for (int l = 0; l < 5; ++l) {
        const int cout = m->dims[l + 1];
        const int skip = (l >= 2);
        const int cin = ccur + (skip ? c0 : 0);
        for (int o = 0; o < cout; ++o) {
            float acc[PB];
            const float bo = m->b[l][o];
            for (int p = 0; p < np; ++p) acc[p] = bo;
            const float *wr = m->w[l] + (size_t)o * cin;
            for (int c = 0; c < ccur; ++c) {
                const float wv = wr[c];
                const float *xr = cur + (size_t)c * PB;
                for (int p = 0; p < np; ++p) acc[p] += wv * xr[p];
            }
            if (skip)
                for (int c = 0; c < c0; ++c) {
                    const float wv = wr[ccur + c];
                    const float *xr = feat + (size_t)c * PB;
                    for (int p = 0; p < np; ++p) acc[p] += wv * xr[p];
                }
            if (l < 4)
                for (int p = 0; p < np; ++p) outb[(size_t)o * PB + p] = acc[p] > 0.0f ? acc[p] : 0.01f * acc[p];
            else
                for (int p = 0; p < np; ++p) logit[p] = acc[p];
        }
        cur = outb;
        ccur = cout;
        outb = (outb == bufa) ? bufb : bufa;
    }
}

/* grid_sample(bilinear, zeros padding, align_corners=True) of one point.  lib/geometry.py:4-12 */
static void bilinear_point(const float *feat, int c, int h, int w, float u, float v, float *out, int ostride) {
    const float ix = ((u + 1.0f) / 2.0f) * (float)(w - 1);
    const float iy = ((v + 1.0f) / 2.0f) * (float)(h - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wnw = ((float)x1 - ix) * ((float)y1 - iy), wne = (ix - (float)x0) * ((float)y1 - iy);
    const float wsw = ((float)x1 - ix) * (iy - (float)y0), wse = (ix - (float)x0) * (iy - (float)y0);
    const int vx0 = x0 >= 0 && x0 < w, vx1 = x1 >= 0 && x1 < w, vy0 = y0 >= 0 && y0 < h, vy1 = y1 >= 0 && y1 < h;
    for (int ch = 0; ch < c; ++ch) {
        const float *f = feat + (size_t)ch * h * w;
        float a = 0.0f;
        if (vy0 && vx0) a += f[(size_t)y0 * w + x0] * wnw;
        if (vy0 && vx1) a += f[(size_t)y0 * w + x1] * wne;
        if (vy1 && vx0) a += f[(size_t)y1 * w + x0] * wsw;
        if (vy1 && vx1) a += f[(size_t)y1 * w + x1] * wse;
        out[(size_t)ch * ostride] = a;
    }
}

/* query_mr + query_sr + get_preds for num_views = 1.
 *   orthogonal                  lib/geometry.py:15-31   (calib: row-major 4x4, rows 0..2 used)
 *   in_img, z_feat, index, cat  lib/model/SuRSNet.py:131-187, lib/model/DepthNormalizer.py:18
 *   get_preds                   lib/model/BaseSuRSNet.py:80-85
 * points [3][n]; feat_lr [c_lr][hl][wl]; feat_hr [c_hr][hh][wh]; zmul = loadSize//2, zdiv = z_size.
 * Outputs (each [n], any may be NULL): pred_hr, pred_lr (masked sigmoid), logit_hr, logit_lr. */
void orc_query(const float *points, int n, const float *calib, float zmul, float zdiv, const float *feat_lr,
               int c_lr, int hl, int wl, const float *feat_hr, int c_hr, int hh, int wh, const orc_mlp_t *mlp_lr,
               const orc_mlp_t *mlp_hr, float *pred_hr, float *pred_lr, float *logit_hr, float *logit_lr) {
    const int cf = c_lr + c_hr; /* 320 */
    const int nblk = (n + PB - 1) / PB;
#pragma omp parallel
    {
        float *feat = (float *)malloc(sizeof(float) * (size_t)(cf + 2) * PB);
        float *bufa = (float *)malloc(sizeof(float) * 1024 * PB);
        float *bufb = (float *)malloc(sizeof(float) * 1024 * PB);
        float lg_lr[PB], lg_hr[PB], inimg[PB];
#pragma omp for schedule(dynamic, 4)
        for (int blk = 0; blk < nblk; ++blk) {
            const int p0 = blk * PB, np = (n - p0 < PB) ? n - p0 : PB;
            for (int p = 0; p < np; ++p) {
                const float px = points[p0 + p], py = points[(size_t)n + p0 + p], pz = points[2 * (size_t)n + p0 + p];
                const float X = calib[3] + (calib[0] * px + calib[1] * py + calib[2] * pz);
                const float Y = calib[7] + (calib[4] * px + calib[5] * py + calib[6] * pz);
                const float Z = calib[11] + (calib[8] * px + calib[9] * py + calib[10] * pz);
                inimg[p] = (X >= -1.0f && X <= 1.0f && Y >= -1.0f && Y <= 1.0f) ? 1.0f : 0.0f;
                bilinear_point(feat_lr, c_lr, hl, wl, X, Y, feat + p, PB);
                bilinear_point(feat_hr, c_hr, hh, wh, X, Y, feat + (size_t)c_lr * PB + p, PB);
                feat[(size_t)cf * PB + p] = Z * zmul / zdiv;
            }
            mlp_block(mlp_lr, feat, np, lg_lr, bufa, bufb);
            for (int p = 0; p < np; ++p)
                feat[(size_t)(cf + 1) * PB + p] = inimg[p] * (1.0f / (1.0f + expf(-lg_lr[p])));
            mlp_block(mlp_hr, feat, np, lg_hr, bufa, bufb);
            for (int p = 0; p < np; ++p) {
                if (pred_lr) pred_lr[p0 + p] = feat[(size_t)(cf + 1) * PB + p];
                if (pred_hr) pred_hr[p0 + p] = inimg[p] * (1.0f / (1.0f + expf(-lg_hr[p])));
                if (logit_lr) logit_lr[p0 + p] = lg_lr[p];
                if (logit_hr) logit_hr[p0 + p] = lg_hr[p];
            }
        }
        free(feat); free(bufa); free(bufb);
    }
}

/* Grid points of create_grid, flat order x-major / z fastest, float64 -> float32 as eval_func does.
 * lib/sdf.py:4-29, lib/mesh_util.py:20-24.  idx range [i0, i1) of the flat index; out [3][i1-i0]. */
void orc_grid_points(int rx, int ry, int rz, const double *bmin, const double *bmax, long long i0, long long i1,
                     float *out) {
    const long long n = i1 - i0;
    const double sx = (bmax[0] - bmin[0]) / rx, sy = (bmax[1] - bmin[1]) / ry, sz = (bmax[2] - bmin[2]) / rz;
#pragma omp parallel for schedule(static)
    for (long long t = 0; t < n; ++t) {
        const long long f = i0 + t;
        const long long k = f % rz, j = (f / rz) % ry, i = f / ((long long)rz * ry);
        out[t] = (float)(sx * (double)i + bmin[0]);
        out[n + t] = (float)(sy * (double)j + bmin[1]);
        out[2 * n + t] = (float)(sz * (double)k + bmin[2]);
    }
}

int orc_num_threads(void) {
    int n = 1;
#pragma omp parallel
    {
#pragma omp master
        {
#ifdef _OPENMP
            extern int omp_get_num_threads(void);
            n = omp_get_num_threads();
#endif
        }
    }
    return n;
}
