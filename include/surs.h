/*
 * surs.h - C ABI of the MI355X-native SuRS occupancy-query hot path.
 *
 * The reference (marcopesavento/Super-resolution-3D-Human-Shape-from-a-Single-
 * Low-Resolution-Image) is pure Python on PyTorch: it has no FFI of its own.
 * The entry points below are what a binding for this path replaces, one per
 * ATen / scikit-image call on the path:
 *
 *   surs_conv2d_nhwc          nn.Conv2d 3x3 / 1x1        lib/net_util.py:94-97, lib/model/SuRSSR_v3.py:46-138,
 *                                                        lib/model/HGFilters.py:60-64,153-174
 *   surs_groupnorm_coeffs     nn.GroupNorm(32, C) stats  lib/model/HGFilters.py:41-45,140,164
 *   surs_avgpool2             F.avg_pool2d(x,2,2)        lib/model/HGFilters.py:101
 *   surs_bicubic_up2          bicubic x2, both alignments lib/model/HGFilters.py:115, lib/model/SuRSSR_v3.py:140
 *   surs_pixel_shuffle2       PixelShuffle(2)+LeakyReLU   lib/model/SuRSSR_v3.py:111-115
 *   surs_add3                 residual adds / cat         lib/model/HGFilters.py:66-74,117,203-206
 *   surs_query_points         query_mr+query_sr+get_preds lib/model/SuRSNet.py:131-187, lib/model/BaseSuRSNet.py:80-85,
 *                                                        lib/geometry.py:4-31, lib/model/DepthNormalizer.py:18,
 *                                                        lib/model/SurfaceClassifier.py:53-81
 *   surs_query_grid           create_grid + eval_grid + eval_func  lib/sdf.py:4-52, lib/mesh_util.py:16-34
 *   surs_mc_*                 measure.marching_cubes_lewiner(sdf, 0.5)  lib/mesh_util.py:40,45
 *                             (scikit-image 0.17.2, skimage/measure/_marching_cubes_lewiner.py)
 *
 * Conventions: plain pointers and sizes, no torch types.  Unless a parameter
 * is marked HOST, every pointer is a DEVICE pointer valid on the current HIP
 * device; work is enqueued on `stream` (a hipStream_t passed as void*, NULL =
 * the null stream) and the call returns without synchronising unless stated.
 * Return value: 0 on success, a negative SURS_E_* code otherwise;
 * surs_last_error() returns a thread-local message.  No exceptions cross the
 * ABI.  Image tensors are NHWC fp32 with an explicit channel pitch (`ld`, in
 * floats, >= C) so that a channel slice of a wider tensor (the reference's
 * torch.cat) is addressed in place.
 */
#ifndef SURS_H
#define SURS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SURS_ABI_VERSION 1

enum {
    SURS_OK = 0,
    SURS_E_INVALID = -1,   /* bad argument */
    SURS_E_HIP = -2,       /* HIP runtime error (message has the hipError string) */
    SURS_E_UNSUPPORTED = -3,
    SURS_E_LEVEL_RANGE = -4, /* marching cubes: level outside [min,max]  (skimage ValueError) */
    SURS_E_NO_SURFACE = -5,  /* marching cubes: no surface              (skimage RuntimeError) */
    SURS_E_CAPACITY = -6,    /* marching cubes: output buffers too small; counts are reported */
    SURS_E_NONFINITE = -7    /* marching cubes: the volume contains NaN (counts->vmin / vmax are NaN).  skimage would go on
                                silently; here it is how an overflow of the fp32-grade sweep's f16 range surfaces */
};

enum { SURS_F32 = 0, SURS_BF16 = 1, SURS_F16 = 2,  /* arithmetic of the dense MLP contractions */
       SURS_F32_GEMM = 3 };  /* surs_query_grid only: fp32 on the per-point layer kernels (bf16 x 3 split operands: fp32's exponent
                                range) even where the fp32-grade column kernel (f16 x 2 split: |activation| < 65504) applies */

int surs_abi_version(void);
const char *surs_last_error(void);
/* number of compute units / gcn arch name of the current device (arch_out: >= 32 bytes, HOST) */
int surs_device_info(int *cu_count, char *arch_out);

/* Options: every experiment / A-B switch of the library by name, in one table (csrc/surs_api.cpp) - nothing else in the library reads
 * the environment.  An option's environment variable (SURS_<NAME>; surs_set_option accepts either spelling) is read ONCE, when the
 * table is first used, as its initial value; surs_set_option changes it at any time, process-wide.  surs_option_name(i) /
 * surs_option_help(i), i = 0, 1, ... until NULL, list them.  Defaults are the product's configuration: none of them needs setting. */
int surs_set_option(const char *name, int value);
int surs_get_option(const char *name, int *value);
const char *surs_option_name(int index);
const char *surs_option_help(int index);

/* ------------------------------------------------------------------ encoder primitives */

/* y[:, :, 0:cout] (pitch y_ld) = act( conv_k(pre(x))[...] + bias ) (+ residual)
 *   pre(x) = relu(x * in_scale[c] + in_shift[c]) if in_scale != NULL (fused GroupNorm-apply + ReLU, zero padding
 *            applied AFTER it, as nn.Conv2d pads the normalised tensor), else x.
 *   ksize 1 or 3, padding ksize/2, stride 1 or 2.  wpacked: surs_conv_pack_weights layout.  bias nullable.
 *   act: 0 none, 1 leaky-relu with `slope` (slope 0 = ReLU), applied before the residual add.
 *   residual (nullable): same spatial size as y, pitch res_ld. */
int surs_conv2d_nhwc(const float *x, int h, int w, int cin, int x_ld, const float *wpacked, const float *bias, float *y,
                     int cout, int y_ld, int ksize, int stride, const float *in_scale, const float *in_shift, int act,
                     float slope, const float *residual, int res_ld, void *stream);
/* The same 3x3 / stride-1 convolution on the bf16 matrix pipe with fp32 accuracy: every operand is split into three bf16
 * parts (exactly) and the six significant partial products are accumulated in fp32 - 2.7x the fp32 MFMA rate.
 * wsplit: surs_conv_pack_weights_x3 layout (device). */
int surs_conv2d_nhwc_x3(const float *x, int h, int w, int cin, int x_ld, const void *wsplit, const float *bias, float *y,
                        int cout, int y_ld, int ksize, int stride, const float *in_scale, const float *in_shift, int act,
                        float slope, const float *residual, int res_ld, void *stream);
/* HOST helper for it: [3 parts][k*k][cin_pad/16][cout_pad][16] uint16 (bf16).  Returns bytes (query with out == NULL). */
size_t surs_conv_pack_weights_x3(const float *w, int cout, int cin, int ksize, void *out);
/* The same convolution with every operand as TWO f16 parts (hi + lo) and three products per MAC: half the matrix work of the
 * three-part form, 22 significant bits, operands below 65504 in magnitude; the encoder's default.  wsplit:
 * surs_conv_pack_weights_x2 layout ([2 parts][k*k][cin_pad/16][cout_pad][16] uint16, f16). */
int surs_conv2d_nhwc_x2(const float *x, int h, int w, int cin, int x_ld, const void *wsplit, const float *bias, float *y,
                        int cout, int y_ld, int ksize, int stride, const float *in_scale, const float *in_shift, int act,
                        float slope, const float *residual, int res_ld, void *stream);
size_t surs_conv_pack_weights_x2(const float *w, int cout, int cin, int ksize, void *out);
/* The same entry with ONE f16 part per operand (part 0 of the surs_conv_pack_weights_x2 image) and one product per MAC in the 3x3
 * kernels: 11 significant bits - NOT fp32-grade; a third of the matrix work.  The encoder of --precision bf16, whose sweep
 * rounds the features' contributions to 8 bits anyway; held to the acceptance bounds of tests/test_gpu_precision.py.  1x1 convolutions (HBM-bound) run the two-part kernel. */
int surs_conv2d_nhwc_x1(const float *x, int h, int w, int cin, int x_ld, const void *wsplit, const float *bias, float *y,
                        int cout, int y_ld, int ksize, int stride, const float *in_scale, const float *in_shift, int act,
                        float slope, const float *residual, int res_ld, void *stream);
/* The split-operand 3x3 kernels come in two tiles (8 rows x 64 channels, 4 x 32 for maps too small to fill the chip with the
 * first) that sum their partial products in different orders.  surs_conv_tile_scale(num, den) makes the calling thread's following
 * convolutions choose the tile as if their maps were num / den times as wide: a column strip of an image then reproduces the bits
 * of the full image's columns (one rank's share of a sharded encoder).  (1, 1) restores the default. */
int surs_conv_tile_scale(int num, int den);
/* HOST helper: repack a PyTorch [cout][cin][k][k] weight into the kernel layout [k*k][cin_pad][cout_pad] (floats).
 * Returns the number of floats written (query with out == NULL). */
size_t surs_conv_pack_weights(const float *w, int cout, int cin, int ksize, float *out);

/* GroupNorm statistics folded with the affine parameters into per-channel coefficients:
 * scale[c] = gamma[c]*rstd[g(c)], shift[c] = beta[c] - mean[g]*rstd[g]*gamma[c]   (biased variance, eps). */
int surs_groupnorm_coeffs(const float *x, int hw, int c, int x_ld, int groups, float eps, const float *gamma,
                          const float *beta, float *scale, float *shift, void *stream);
/* The same with the scratch of the partial sums supplied by the caller (surs_groupnorm_scratch_bytes() bytes, device): nothing is
 * allocated inside, so the two launches can be captured into a HIP graph (hipMalloc is not permitted on a capturing stream). */
size_t surs_groupnorm_scratch_bytes(void);
int surs_groupnorm_coeffs_ws(const float *x, int hw, int c, int x_ld, int groups, float eps, const float *gamma,
                             const float *beta, float *scale, float *shift, void *scratch, void *stream);
/* y = relu?(x*scale + shift) elementwise (GroupNorm apply when it is not followed by a conv) */
int surs_scale_shift_act(const float *x, int hw, int c, int x_ld, const float *scale, const float *shift, int relu,
                         float *y, int y_ld, void *stream);
int surs_avgpool2(const float *x, int h, int w, int c, int x_ld, float *y, int y_ld, void *stream);
/* y = bicubic_x2(x) (+ addend, nullable, pitch add_ld): A=-0.75, border-clamped taps */
int surs_bicubic_up2(const float *x, int h, int w, int c, int x_ld, int align_corners, const float *addend, int add_ld,
                     float *y, int y_ld, void *stream);
/* y[2h+i][2w+j][c] = lrelu(x[h][w][4c+2i+j], slope)  (slope 1 = no activation) */
int surs_pixel_shuffle2(const float *x, int h, int w, int c4, int x_ld, float slope, float *y, int y_ld, void *stream);
/* y = a + b (+ c, nullable) */
int surs_add3(const float *a, int a_ld, const float *b, int b_ld, const float *c, int c_ld, int hw, int ch, float *y,
              int y_ld, void *stream);
/* GroupNorm(32, C) handed from kernel to kernel (ConvBlock, lib/model/HGFilters.py:57-74: three GroupNorm + ReLU + 3x3 convolutions
 * and a sum - four launches instead of ten).  A kernel that writes a map leaves the statistics of exactly the values it stored as
 * partial sums gn_out[32 groups][*gn_out_slots][2] (sum, sum of squares; doubles; *gn_out_slots <= gn_out_capacity is written to
 * the HOST int; one slot per workgroup, so the layout is a function of the shapes alone); the 3x3 convolution that consumes the map
 * folds them - every workgroup, in the same fixed order - into surs_groupnorm_coeffs' coefficients (same formulas) and applies
 * GroupNorm(gamma, beta, eps) + ReLU while staging.  Deterministic: no atomics.
 *   surs_conv2d_nhwc_gn: surs_conv2d_nhwc_x2 (parts = 2) / _x1 (parts = 1), 3x3 or 1x1 (1x1: the two-part kernel whatever `parts`),
 *   activation and residual as there; gn_in (nullable) = the statistics of x with their slot count, gamma / beta [cin], 32 | cin;
 *   gn_out (nullable) as above = the statistics of y after activation and residual, cout / 32 a power of two <= 32 (a buffer of
 *   ceil(w / 32) * ceil(h / 4) slots always suffices for 3x3, ceil(h * w / 128) for 1x1).
 *   surs_avgpool2_gn / surs_bicubic_up2_gn / surs_add3_gn: the elementwise kernels, the same values bit for bit, plus gn_out
 *   (C a power of two in [32, 1024], 16-byte aligned rows; at most 512 slots). */
int surs_conv2d_nhwc_gn(int parts, const float *x, int h, int w, int cin, int x_ld, const void *wsplit, const float *bias, float *y,
                        int cout, int y_ld, int ksize, int stride, const double *gn_in, int gn_in_slots, const float *gamma,
                        const float *beta, float eps, int act, float slope, const float *residual, int res_ld, double *gn_out,
                        int gn_out_capacity, int *gn_out_slots, void *stream);
int surs_avgpool2_gn(const float *x, int h, int w, int c, int x_ld, float *y, int y_ld, double *gn_out, int gn_out_capacity,
                     int *gn_out_slots, void *stream);
int surs_bicubic_up2_gn(const float *x, int h, int w, int c, int x_ld, int align_corners, const float *addend, int add_ld, float *y,
                        int y_ld, double *gn_out, int gn_out_capacity, int *gn_out_slots, void *stream);
int surs_add3_gn(const float *a, int a_ld, const float *b, int b_ld, const float *c, int c_ld, int hw, int ch, float *y, int y_ld,
                 double *gn_out, int gn_out_capacity, int *gn_out_slots, void *stream);
/* Input stage of the test path (lib/data/EvalDataset_LR_v2.py:227-243): rgb uint8 [h][w][3], mask uint8 [h][w] (device) ->
 * y[h][w][0:3] (pitch y_ld) = (mask / 255) * ((rgb / 255 - 0.5) / 0.5), float32, the reference's operations in its order
 * (ToTensor, Normalize(0.5, 0.5), mask multiply): bit-identical to its img_LR, already in the encoder's NHWC layout. */
int surs_image_prepare(const unsigned char *rgb, const unsigned char *mask, int h, int w, float *y, int y_ld, void *stream);
/* NCHW <-> NHWC(pitch ld) copies for the boundary tensors */
int surs_nchw_to_nhwc(const float *x, int c, int h, int w, float *y, int y_ld, void *stream);
int surs_nhwc_to_nchw(const float *x, int c, int h, int w, int x_ld, float *y, void *stream);

/* GroupNorm(32) statistics of a map as the kernels that wrote it left them: partial sums [32 groups][pitch][2] doubles (sum, sum of
 * squares).  A map written by ONE kernel has slots[0] slots in every group (g1 = g2 = 0).  A map written slice by slice by several
 * kernels, each with its own pixel tiling - a ConvBlock's cat(o1, o2, o3) + x -: groups [0, g1) hold slots[0], [g1, g2) slots[1],
 * [g2, 32) slots[2] slots. */
typedef struct SursGnStats { double *sums; int pitch, g1, g2; int slots[3]; } SursGnStats;
/* One 3x3 / stride-1 convolution of a ConvBlock (lib/model/HGFilters.py:57-73) with the block's closing sum in its epilogue:
 *   v  = conv(pre(x)) + bias,  pre = GroupNorm(gamma, beta, eps) + ReLU from gn_in's statistics, or from in_scale / in_shift, or none
 *   y  = v                      (nullable: the last convolution's own value is not read by anybody) + its statistics in gn_out
 *        (nullable; on entry gn_out->sums and ->pitch = capacity in slots, on return the slot counts)
 *   y2 = v + residual           the slice of out = cat(o1, o2, o3) + x this convolution makes, and - gn_out2 not NULL - the statistics
 *        of those values in the numbering of the WHOLE sum: group gn2_g0 + channel / gn2_cg, rows gn2_pitch slots apart,
 *        *gn2_slots = the slots this launch wrote per group.
 * The same tiles and bits as surs_conv2d_nhwc_gn followed by surs_add3_gn on the slice. */
int surs_conv2d_nhwc_gn_sum(int parts, const float *x, int h, int w, int cin, int x_ld, const void *wsplit, const float *bias,
                            const SursGnStats *gn_in, const float *in_scale, const float *in_shift, const float *gamma, const float *beta,
                            float eps, float *y, int cout, int y_ld, SursGnStats *gn_out, const float *residual, int res_ld, float *y2,
                            int y2_ld, double *gn_out2, int gn2_pitch, int gn2_g0, int gn2_cg, int *gn2_slots, void *stream);

/* ------------------------------------------------------------------ the encoder as ONE call per network
 * SuRSNet.super_res / filter_hr / filter_lr (lib/model/SuRSNet.py:101-129) = SuRSSR_v3.forward (lib/model/SuRSSR_v3.py:143-181),
 * HGFilter.forward high_res (lib/model/HGFilters.py:179-181) and low_res (:183-206; ConvBlock :29-74, HourGlass :76-120), sequenced
 * inside the library (csrc/surs_encoder_net.cpp): the launches of the primitives above in the reference's order, intermediates in
 * the caller's workspace, nothing allocated, no stream created.  The results equal the per-primitive sequencing of the host mirror
 * (encoder.py) bit for bit.  All weight pointers are DEVICE pointers in the layouts of the pack functions above; the struct itself
 * is HOST memory and is only read during the call. */
enum { SURS_ENC_SEPARATE_SUM = 1 };  /* SursEncoderNet.flags: a ConvBlock's closing sum as a pass of its own (surs_add3_gn: the four-launch
                                         form of rounds 4 - 5, whose bits the host mirror's per-launch sequencing reproduces) instead of in
                                         the three convolutions' epilogues (surs_conv2d_nhwc_gn_sum: the default) */
typedef struct SursConv {
    const void *w_split;    /* surs_conv_pack_weights_x2 image (3x3 and 1x1), or NULL: only the fp32 kernel applies */
    const float *w_packed;  /* surs_conv_pack_weights image (the fp32 MFMA / direct kernels: 3 -> 32 head, 32 -> 3 tail) */
    const float *bias;      /* [cout] or NULL */
    int cin, cout, ksize, reserved;
} SursConv;
typedef struct SursGroupNorm { const float *gamma, *beta; } SursGroupNorm;           /* GroupNorm(32, C), eps 1e-5 */
typedef struct SursConvBlock { SursConv conv[3]; SursGroupNorm bn[3]; } SursConvBlock; /* ConvBlock with in_planes == out_planes */
typedef struct SursEncoderNet {
    int residual;           /* opt.residual: the ResBlocks of the super-resolution stages run */
    int n_block[3];         /* opt.n_block */
    int num_stack, hg_depth;
    int parts;              /* 2 = fp32-grade (two f16 parts, three products per MAC); 1 = one f16 product in the 3x3 convolutions */
    int flags;              /* SURS_ENC_* */
    /* super_resolution.* (SuRSSR_v3): head.0, down{1,2,3}.0, tail{1,2,3}.0 / .2, bottleneck.0, bott2.0, ups2.0, ups3.0, ups4.0,
     * last.0, last.2; body: body{i}.{b}.body.0, .body.2 for i = 1..3, b = 0..n_block[i-1]-1, in that order */
    SursConv head, down[3], tail0[3], tail2[3], bottleneck, bott2, ups2, ups3, ups4, last0, last2;
    const SursConv *body;
    SursConv conv5;         /* image_filter_hr.conv5 */
    SursConvBlock conv2;    /* image_filter_lr.conv2 */
    /* image_filter_lr.m{s}: per stack 3 * hg_depth + 1 blocks in module order b1_d, b2_d, [b1_{d-1}, b2_{d-1}, ...], b2_plus_1, b3_1, .., b3_d */
    const SursConvBlock *hg;
    const SursConvBlock *top_m;   /* [num_stack] */
    const SursConv *conv_last, *l, *next;   /* [num_stack]; next[s] = bl{s} + al{s} o l{s} merged (W_bl + W_al W_l), unused for the last stack */
    const SursGroupNorm *bn_end;  /* [num_stack] */
} SursEncoderNet;
/* Streams the caller lends for the low-resolution branch of hourglass level 1..4 (NULL entries / NULL struct: the branches run one
 * behind the other on `stream`).  The library never creates a stream (a new stream shifts the hardware-queue assignment of every later one). */
typedef struct SursEncoderStreams { void *side[4]; } SursEncoderStreams;
/* bytes of workspace the calls below need for an h x w input image (0: bad arguments) */
size_t surs_encoder_workspace_bytes(const SursEncoderNet *net, int h, int w);
/* x [h][w][3] (pitch x_ld) -> feature_lr [h/2][w/2][256], feature_hr [2h][2w][64] and, if want_image, img_sr [2h][2w][3] (dense) */
int surs_encoder_super_res(const SursEncoderNet *net, const float *x, int h, int w, int x_ld, int want_image, float *img_sr,
                           float *feature_lr, float *feature_hr, void *workspace, size_t workspace_bytes, void *stream);
/* feature_lr [h][w][256] (pitch ld) -> outs[s] [h][w][last_ch] for every stack s with outs[s] != NULL (HOST array of num_stack device
 * pointers; the last one is required - eval keeps only it, training keeps all) */
int surs_encoder_filter_lr(const SursEncoderNet *net, const float *feature_lr, int h, int w, int ld, float *const *outs,
                           void *workspace, size_t workspace_bytes, const SursEncoderStreams *streams, void *stream);
/* feature_hr [h][w][64] (pitch ld) -> out [h][w][64] */
int surs_encoder_filter_hr(const SursEncoderNet *net, const float *feature_hr, int h, int w, int ld, float *out, void *stream);
/* the three in the order gen_mesh runs them (lib/train_util.py:57-59): image [h][w][3] -> feature_lr, feature_hr (kept: the caller's
 * buffers), im_feat_lr [h/2][w/2][last_ch] (last stack), im_feat_hr [2h][2w][64] */
int surs_encoder_forward(const SursEncoderNet *net, const float *image, int h, int w, int x_ld, float *feature_lr, float *feature_hr,
                         float *im_feat_lr, float *im_feat_hr, void *workspace, size_t workspace_bytes,
                         const SursEncoderStreams *streams, void *stream);

/* ------------------------------------------------------------------ point evaluator */

/* HOST: pack the two SurfaceClassifier MLPs (lr: 321-1024-512-256-128-1, hr: 322-..., skip-concat at layers
 * 2,3,4; w[l] = Conv1d weight [out][in], b[l] = bias) into one blob that surs_query_* consume.  `dtype` selects
 * the element type of the dense cores used by the grid kernel (SURS_BF16 or SURS_F16).  Call with blob == NULL
 * to get the size in bytes.  The caller uploads the blob to device memory (256-byte aligned). */
size_t surs_mlp_pack(const float *const w_lr[5], const float *const b_lr[5], const float *const w_hr[5],
                     const float *const b_hr[5], int dtype, void *blob);

/* How the fp32 point path (surs_query_points, _views, surs_query_grid_indexed, the general-calibration sweep) carries its fp32
 * operands through the bf16 / f16 matrix pipe, process-wide: 2 = two f16 parts, three products per MAC (default: 1.6x the
 * rate, 22 significant bits, |activation| and |feature| < 65504), 3 = three bf16 parts, six products (24 bits, fp32's exponent
 * range), 0 = back to the default (or the SURS_SPLIT environment variable).  Both meet the 1e-4 logit tolerance. */
int surs_set_operand_split(int parts);
/* The same for the calling host thread only (0 = back to the process-wide setting); takes precedence over it.  The host mirror
 * uses it to repeat a query or a sweep on three bf16 parts after an f16 overflow (non-finite results).  parts = 1 (this call only):
 * surs_query_points / surs_query_points_hr run ONE f16 product per MAC (one f16 part per operand, 11 significant bits, a third of the
 * matrix work) - NOT fp32-grade: what SuRSNet.query_mr / query_sr of `--precision bf16 | fp16` evaluate arbitrary points with, as the
 * reference's MLP would in half precision (lib/model/SurfaceClassifier.py:53-81); every other entry point ignores it. */
int surs_set_operand_split_local(int parts);

/* Column kernel of surs_query_grid, process-wide (A/B comparisons and regression tests; a per-call choice goes through
 * surs_query_grid_opt): 0 = default (or the SURS_GRID_KERNEL / SURS_GRID_F32_KERNEL environment variables); reduced precision
 * 3 (dense layer 1), 10 (layer 1 restated along the column, eight waves), 12 (the default: 10's arithmetic and bits with layer 1
 * streamed into layer 2, two workgroups per compute unit); fp32-grade 5 (dense), 11 (restated) - DESIGN.md section 4. */
int surs_set_grid_kernel(int version);

/* How many of the 1024 layer-0 channels the default column kernels (layer 1 restated along the column) would
 * list per z tile (`tile` = 128 for the reduced precisions, 64 for SURS_F32) on this sweep: evaluated on the ry columns of axis-0
 * plane `i_plane` of the grid `mat` describes - every slab of one grid gives the same answer.  listed[0]: mean over (column,
 * tile) for the lr classifier, listed[1]: an upper bound for the hr classifier; both -1 where the column kernels do not apply.
 * `listed` is host memory; the call synchronises the stream.  The host mirror runs the dense column kernels (version 3 / 5) when
 * listed[0] exceeds 400.  Workspace as for surs_query_grid. */
int surs_query_grid_probe(int i_plane, int ry, int rz, int tile, const double *mat, const float *calib, float zmul, float zdiv,
                          const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh, const void *mlp_blob,
                          void *workspace, size_t workspace_bytes, float *listed, void *stream);

/* bytes of device workspace the two query entry points need for `max_points` points per call / grid batch */
size_t surs_query_workspace_bytes(int max_points);

/* query_mr + query_sr + get_preds for one view, fp32 arithmetic (f32 MFMA).
 *   points  [3][n] fp32 (x row, y row, z row), calib HOST [12] = rows 0..2 of the 4x4 calibration,
 *   zmul = loadSize/2 (integer division done by the caller), zdiv = z_size,
 *   feat_lr NHWC [hl][wl][c_lr=256] pitch c_lr, feat_hr NHWC [hh][wh][c_hr=64],
 *   outputs [n] each; logit_* nullable (pre-sigmoid, pre-mask). */
int surs_query_points(const float *points, int n, const float *calib, float zmul, float zdiv, const float *feat_lr,
                      int hl, int wl, const float *feat_hr, int hh, int wh, const void *mlp_blob, void *workspace,
                      size_t workspace_bytes, float *pred_hr, float *pred_lr, float *logit_hr, float *logit_lr,
                      void *stream);

/* query_sr alone (lib/model/SuRSNet.py:161-187), for callers that pass it OTHER points than the preceding query_mr (the reference
 * only requires the same n): the hr classifier on `points`, its last input channel taken from p_lr [n] (device; the masked lr
 * occupancies query_mr left behind) instead of from an lr evaluation of these points.  Arguments as surs_query_points. */
int surs_query_points_hr(const float *points, int n, const float *calib, float zmul, float zdiv, const float *feat_lr,
                         int hl, int wl, const float *feat_hr, int hh, int wh, const void *mlp_blob, void *workspace,
                         size_t workspace_bytes, const float *p_lr, float *pred_hr, float *logit_hr, void *stream);

/* surs_query_points for point arrays that come as RUNS of equal (x, y): what the reference's dense sweep loop hands
 * query_mr / query_sr - 50 000 consecutive points of the flattened grid per call, z fastest (lib/sdf.py:32-45 batch_eval,
 * lib/mesh_util.py:20-28 eval_func) - i.e. ~ 98 columns of up to 512 points with one image position each.  Such a run is a column of
 * the sweep: the restated column kernels of surs_query_grid evaluate it (per-run constants from one gather + GEMM, layer 1 as the
 * affine part + the residuals of the listed channels), every point with its own z read from `points`.  No grid is assumed: runs are
 * found in the data (bit-equal x and y, z monotonic inside a run, cut at 4096 points).
 *   points [3][n] with row pitch ld >= n (a piece of a longer array), n <= 262 144; dtype = the blob's use: SURS_F32 (kernel v11,
 *   fp32-grade: logits within 1e-4 of surs_query_points'), SURS_BF16 / SURS_F16 (kernel v10: surs_query_grid's arithmetic);
 *   *columns = the number of runs evaluated, or 0 - NOTHING WAS WRITTEN, call surs_query_points - when the array holds fewer than
 *   2048 points or more than one run per 16 (SURS_F32) / 32 (SURS_BF16, SURS_F16) points - the layer kernels are the faster evaluator there -, z is not monotonic inside the runs, or the calibration lets the image position
 *   depend on z (calib[2], calib[6]) or the depth on x, y (calib[8], calib[9]).  Synchronises the stream once (the run count). */
/* Its run finder alone (tests, diagnostics): colstart[c] / kcount[c] = first point and length of run c (ints, room for n each), tiles =
 * (run, z tile) pairs of `tile` = 64 | 128 points (room for 2 n ints), meta[4] = {runs, work items - 0 and no lengths / work items when the
 * array holds more than one run per tile / 4 points -, z ascending violated, z descending violated}.  Device pointers; no synchronisation. */
int surs_point_runs(const float *points, long long ld, int n, int tile, int *colstart, int *kcount, int *tiles, int *meta, void *stream);
/* flag[0] (device) = 1 if a[0..n) or b[0..n) (b nullable) holds a NaN or an infinity, else 0: the check behind every query of the host
 * mirror (an activation beyond the f16 range of the two-part operand split surfaces as NaN; the query is then repeated on three bf16
 * parts).  One small launch; no synchronisation. */
int surs_nonfinite(const float *a, const float *b, long long n, int *flag, void *stream);
size_t surs_query_points_columns_workspace_bytes(void);
int surs_query_points_columns(const float *points, long long ld, int n, const float *calib, float zmul, float zdiv,
                              const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh, const void *mlp_blob,
                              int dtype, void *workspace, size_t workspace_bytes, float *pred_hr, float *pred_lr, int *columns,
                              void *stream);

/* The same for num_views > 1 and / or the perspective projection (SurfaceClassifier.forward's view mean after layer 2,
 * lib/model/SurfaceClassifier.py:70-76; reshape_sample_tensor, lib/train_util.py:40-51; perspective,
 * lib/geometry.py:34-48).  One subject (batch 1), V views:
 *   points  [V][3][n] (the caller repeats the samples per view as reshape_sample_tensor does), calibs HOST [V][12],
 *   projection 0 = orthogonal, 1 = perspective (x, y divided by the projected z),
 *   feat_lr NHWC [V][hl][wl][256], feat_hr NHWC [V][hh][wh][64],
 *   pred_hr / pred_lr [V][n]: the one prediction of the view-mean network under each view's in-image mask
 *   (`in_img[:, None].float() * mlp(...)`, SuRSNet.py:156,183), logit_* [n] nullable. */
int surs_query_points_views(const float *points, int n, int num_views, int projection, const float *calibs, float zmul,
                            float zdiv, const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh,
                            const void *mlp_blob, void *workspace, size_t workspace_bytes, float *pred_hr, float *pred_lr,
                            float *logit_hr, float *logit_lr, void *stream);

/* The dense sweep of such a model as one call: eval_grid's batch loop (lib/sdf.py:32-52) over eval_func (lib/mesh_util.py:20-28 -
 * every batch of grid points repeated per view, query_mr + query_sr, view 0's predictions kept) for the slab [i0, i1) of the grid
 * `mat` (rows 0..2 of create_grid's matrix, HOST float64): the voxels are generated in the gather for every view's calibration,
 * 262 144 at a time; vol_* [(i1-i0)][ry][rz].  Same kernels and bits as surs_query_points_views on create_grid's points. */
size_t surs_query_grid_views_workspace_bytes(int num_views);
int surs_query_grid_views(int i0, int i1, int ry, int rz, const double *mat, int num_views, int projection, const float *calibs,
                          float zmul, float zdiv, const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh,
                          const void *mlp_blob, void *workspace, size_t workspace_bytes, float *vol_hr, float *vol_lr, void *stream);
size_t surs_query_views_workspace_bytes(int max_points, int num_views);

/* Dense grid sweep: voxel (i,j,k), i in [i0,i1), j in [0,ry), k in [0,rz) has world position
 * p = float32( mat[:,0]*i + mat[:,1]*j + mat[:,2]*k + mat[:,3] )  (mat HOST [12] doubles = create_grid's
 * coords_matrix rows 0..2, evaluated in double like np.matmul on the float64 grid, then cast as eval_func does).
 * vol_hr / vol_lr: [(i1-i0)][ry][rz] fp32, z fastest (the flattening of lib/sdf.py:14-15,28).
 * dtype SURS_F32: fp32-grade results (logits within 1e-4 of the reference): on an axis-aligned orthographic sweep (the projected
 * X, Y do not depend on k: true for gen_mesh's calib) the fused column kernel with split-f16 operands, otherwise - and for
 * SURS_F32_GEMM - the arithmetic of surs_query_points.  SURS_BF16 / SURS_F16: the reduced-precision column kernel (axis-aligned
 * sweeps only; otherwise returns SURS_E_UNSUPPORTED). */
int surs_query_grid(int i0, int i1, int ry, int rz, const double *mat, const float *calib, float zmul, float zdiv,
                    const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh, const void *mlp_blob,
                    int dtype, void *workspace, size_t workspace_bytes, float *vol_hr, float *vol_lr, void *stream);
size_t surs_query_grid_workspace_bytes(int ry, int rz, int dtype);

/* The same sweep with per-call choices instead of process-wide ones (nothing global is read for a field that is set, nothing
 * global is written): `kernel` = column-kernel version (0 = the process setting / default; reduced precision 3, 10, 12;
 * fp32-grade 5, 11 - DESIGN.md 4), `operand_parts` = operand split of the fp32-grade GEMMs behind the sweep (0 = process
 * setting, 2 = two f16 parts, 3 = three bf16 parts).  opt == NULL behaves as surs_query_grid.  Safe to call from several host
 * threads on different streams. */
typedef struct SursGridOptions {
    int kernel;
    int operand_parts;
    int reserved[6];   /* must be zero */
} SursGridOptions;
int surs_query_grid_opt(int i0, int i1, int ry, int rz, const double *mat, const float *calib, float zmul, float zdiv,
                        const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh, const void *mlp_blob,
                        int dtype, void *workspace, size_t workspace_bytes, float *vol_hr, float *vol_lr,
                        const SursGridOptions *opt, void *stream);

/* Octree sweep (eval_grid_octree, lib/sdf.py:55-120), one level at a time; volumes are float64 [R][R][R] like the
 * reference's numpy arrays, `dirty` is uint8 [R][R][R].
 *   surs_octree_select     lattice points of stride `reso` that are still dirty -> idx[] (flat voxel indices, any
 *                          order), count (host; synchronises)                                 sdf.py:68-71
 *   surs_query_grid_indexed evaluate those voxels (fp32 arithmetic, as surs_query_points)       sdf.py:73
 *   surs_octree_scatter    sdf[idx] = pred, dirty[idx] = 0                                     sdf.py:73-74
 *   surs_octree_cells      the cell walk: blocks whose 8 corners span < threshold are set to (max+min)/2 and marked
 *                          clean, HR and LR sharing the one dirty mask                          sdf.py:81-117
 *   surs_f64_to_f32        the cast marching_cubes_lewiner applies to its input */
int surs_octree_select(const unsigned char *dirty, int R, int reso, long long *idx, int cap, int *count_dev, int *count_host,
                       void *stream);
/* select + evaluate + scatter of one level in ONE call, on the sweep's fp32-grade COLUMN kernel (axis-aligned orthographic sweeps:
 * the lattice points of stride `reso` form columns along axis 2 that share their image position, so the per-column constants
 * and the restated layer 1 of surs_query_grid apply).  A column's work items are its dirty lattice points in ascending order, 64
 * per tile; exactly the points lib/sdf.py:68-74 evaluates are evaluated and written (sdf = value, dirty = 0).  A point's value
 * depends on the point, on `reso` and - in the last bits, through the tile it shares - on which other points of its column are
 * dirty: the same dirty set gives the same bits.  kmid: axis-2 voxel index where the kernel takes its per-column LeakyReLU
 * branches (R / 2; any value gives the same function up to rounding).  counts (HOST, nullable, 3 values): dirty lattice points
 * evaluated, lattice columns that held them, 64-point tiles run.  Synchronises the stream once.  SURS_E_UNSUPPORTED for a
 * general calibration: use the three calls above and below. */
int surs_octree_level_columns(double *sdf_hr, double *sdf_lr, unsigned char *dirty, int R, int reso, int kmid, const double *mat,
                              const float *calib, float zmul, float zdiv, const float *feat_lr, int hl, int wl,
                              const float *feat_hr, int hh, int wh, const void *mlp_blob, void *workspace, size_t workspace_bytes,
                              long long *counts, void *stream);
/* The same with the level's evaluator chosen: dtype SURS_F32 = the fp32-grade column kernel (what surs_octree_level_columns runs),
 * SURS_BF16 / SURS_F16 = the 16-bit column kernel on the blob's cores - the octree sweep of `--precision bf16 | fp16`. */
int surs_octree_level_columns_dt(double *sdf_hr, double *sdf_lr, unsigned char *dirty, int R, int reso, int kmid, const double *mat,
                                 const float *calib, float zmul, float zdiv, const float *feat_lr, int hl, int wl,
                                 const float *feat_hr, int hh, int wh, const void *mlp_blob, int dtype, void *workspace,
                                 size_t workspace_bytes, long long *counts, void *stream);
size_t surs_octree_columns_workspace_bytes(int R);
int surs_query_grid_indexed(const long long *idx, int n, int ry, int rz, const double *mat, const float *calib, float zmul,
                            float zdiv, const float *feat_lr, int hl, int wl, const float *feat_hr, int hh, int wh,
                            const void *mlp_blob, void *workspace, size_t workspace_bytes, float *pred_hr, float *pred_lr,
                            void *stream);
int surs_octree_scatter(const long long *idx, int n, const float *pred_hr, const float *pred_lr, double *sdf_hr, double *sdf_lr,
                        unsigned char *dirty, void *stream);
size_t surs_octree_workspace_bytes(int R, int reso);
int surs_octree_cells(double *sdf_hr, double *sdf_lr, unsigned char *dirty, int R, int reso, double threshold, void *workspace,
                      size_t workspace_bytes, void *stream);
int surs_f64_to_f32(const double *a, float *b, long long n, void *stream);

/* HOST: write a Wavefront OBJ exactly as save_obj_mesh does (lib/mesh_util.py:53-61): 'v %.4f %.4f %.4f' per vertex,
 * then 'f a c b' per face with 1-based indices and the winding swapped.  verts float64 [n_verts][3], faces int32
 * [n_faces][3] (host pointers).  threads <= 0: use all hardware threads for the formatting. */
int surs_save_obj_mesh(const char *path, const double *verts, long long n_verts, const int32_t *faces, long long n_faces,
                       int threads);

/* Measurement aid (not on the reference's path): when enabled, every launch of the dominant kernel of
 * surs_query_grid's reduced-precision mode is bracketed by HIP events on its launch stream.  surs_profile_read
 * returns the number of timed launches, the sum of their durations (ms) and the voxels they evaluated, and resets. */
int surs_profile_enable(int on);
int surs_profile_read(double *launches, double *total_ms, double *points);
/* Column kernel v7 runs a data-dependent number of layer-1 k-steps.  tile_mlps: (z tile, MLP) pairs the timed launches
 * processed; ksteps: residual k-steps (16 listed channels each) they ran, not counting the affine k-step of every pair.
 * Call before surs_profile_read, which resets the counters. */
int surs_profile_read_ksteps(double *tile_mlps, double *ksteps);

/* ------------------------------------------------------------------ Lewiner marching cubes */

typedef struct {
    int32_t n_verts;   /* vertices produced (also when SURS_E_CAPACITY) */
    int32_t n_faces;   /* triangles produced */
    float vmin, vmax;  /* data range of the volume */
} surs_mc_counts;

size_t surs_mc_workspace_bytes(int n0, int n1, int n2);
/* Extract the level set of vol [n0][n1][n2] (fp32, axis 2 fastest) exactly as
 * skimage.measure.marching_cubes_lewiner(vol, level) with default arguments does: same vertices (fp32,
 * (axis0, axis1, axis2) order), same vertex numbering, same faces (int32, rows reversed for
 * gradient_direction='descent'), normals and values.  Outputs are device buffers of capacity cap_verts /
 * cap_faces; counts is a HOST struct filled before return (this call synchronises the stream).
 * normals / values nullable.  verts == NULL or faces == NULL: count-only call (fills counts, writes nothing). */
int surs_mc_lewiner(const float *vol, int n0, int n1, int n2, double level, void *workspace, size_t workspace_bytes,
                    float *verts, float *normals, float *values, int cap_verts, int32_t *faces, int cap_faces,
                    surs_mc_counts *counts, void *stream);

/* The same extraction, incrementally, for a volume that is produced slab by slab along axis 0 (the dense sweep writes
 * whole axis-0 planes in order): processes the cell layers [layer_begin, layer_end) - they read the voxel planes
 * layer_begin .. layer_end, which must be final - and appends their vertices / faces at run->n_verts / run->n_faces
 * (HOST struct, in/out; initialise to {0, 0, +FLT_MAX, -FLT_MAX}).  Called with contiguous increasing ranges that end
 * at n0 - 1, the outputs are identical to one surs_mc_lewiner call: Lewiner's sweep has axis 0 outermost, so vertex and
 * face numbering of a layer depend only on the layers before it.  Normals / values of a vertex keep accumulating from
 * later layers: read them after the last range and surs_mc_normalize.  The level-range / no-surface checks are the
 * caller's, from run->vmin / vmax / n_verts after the last range.  Synchronises the stream once per call.
 * SURS_E_CAPACITY leaves run advanced to the sizes needed so far. */
int surs_mc_lewiner_range(const float *vol, int n0, int n1, int n2, int layer_begin, int layer_end, double level,
                          void *workspace, size_t workspace_bytes, float *verts, float *normals, float *values, int cap_verts,
                          int32_t *faces, int cap_faces, surs_mc_counts *run, void *stream);
int surs_mc_normalize(float *normals, int n_verts, void *stream);

/* Slab mode (the grid split into contiguous axis-0 slabs over ranks, marching cubes per slab; the reference
 * has no counterpart - it runs lib/mesh_util.py:40,45 on the whole volume).  `vol` is ONE slab [n0][n1][n2] whose plane 0 is
 * plane `z_offset` of the whole grid and whose last plane is the next slab's first plane (the halo), except for the top
 * slab.  surs_mc_lewiner_range_slab is surs_mc_lewiner_range (vertices and faces only) with two differences: vertex
 * coordinates are those of the whole grid (bit-identical to the one-piece extraction), and for z_offset > 0 the first cell
 * layer does not create the vertices of the x- / y-edges in plane 0 - the slab below owns them - but references them as
 * -(2 + slot), slot = axis * n1 * n2 + y * n2 + x.  Vertex / face numbers are local to the slab (run starts at 0).
 * Afterwards: surs_mc_slab_top_ids copies the ids (local numbering) of the x- / y-edge vertices in the slab's last plane to
 * ids[2][n1][n2] (entries of edges the surface does not cross are undefined) - they go to the slab above; and
 * surs_mc_slab_fixup rewrites the slab's faces to the whole mesh's numbering: v >= 0 -> v + own_offset, v < 0 ->
 * below_ids[-v - 2] + below_offset, the offsets being the exclusive sums of the slabs' vertex counts.  Concatenating the
 * slabs' vertices and faces in slab order then gives exactly the one-piece result. */
int surs_mc_lewiner_range_slab(const float *vol, int n0, int n1, int n2, int layer_begin, int layer_end, double level,
                               void *workspace, size_t workspace_bytes, float *verts, int cap_verts, int32_t *faces, int cap_faces,
                               surs_mc_counts *run, int z_offset, void *stream);
int surs_mc_slab_top_ids(const void *workspace, size_t workspace_bytes, int n0, int n1, int n2, int32_t *ids, void *stream);
int surs_mc_slab_fixup(int32_t *faces, long long n_faces, int own_offset, const int32_t *below_ids, int below_offset, void *stream);

/* out[i] = mat[:3,:3] @ verts[i] + mat[:3,3] in float64 (mat HOST [12] doubles, rows 0..2 of the 4x4 index->world
 * matrix): the vertex transform of lib/mesh_util.py:42-43,47-48.  verts fp32 [n][3], out fp64 [n][3]. */
int surs_transform_points(const float *verts, int n, const double *mat, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SURS_H */
