/*
 * sda_hip.h -- C ABI of libsda_hip.so: the MI355X (gfx950) kernels behind the
 * posterior-sampling hot path of francois-rozet/sda.
 *
 * The reference has no FFI/plugin boundary of its own (it is pure Python on
 * torch/ATen); its boundary is the nn.Module call protocol of sda/score.py and
 * sda/nn.py.  Each entry point below therefore names the reference arithmetic
 * (file:line under /root/reference) that it replaces.  The Python host side
 * (sda_amd/score.py, sda_amd/nn.py) keeps the reference's class names, call
 * signatures and state_dict keys and binds these symbols with ctypes
 * (sda_amd/_lib.py); INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated otherwise
 *   - strides are in ELEMENTS
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it,
 *     nothing synchronises, nothing allocates => every call is hipGraph-capturable
 *   - return value: 0 on success, <0 = SDA_E_* (argument/shape not supported),
 *     >0 = a hipError_t from the launch
 *   - internal activations are PLANAR: [n][c][h][w] contiguous (1-D nets: h = 1)
 */
#ifndef SDA_HIP_H
#define SDA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDA_ABI_VERSION 13

enum {
    SDA_OK = 0,
    SDA_E_BADARG = -1,      /* null pointer / non-positive size               */
    SDA_E_UNSUPPORTED = -2, /* shape outside what the gfx950 kernels tile     */
    SDA_E_LDS = -3          /* tile does not fit the 160 KiB LDS budget        */
};

/* activation ids: sda/utils.py:19-25 (ACTIVATIONS) */
enum { SDA_ACT_NONE = 0, SDA_ACT_SILU = 1, SDA_ACT_RELU = 2, SDA_ACT_ELU = 3, SDA_ACT_GELU = 4, SDA_ACT_SELU = 5 };

int sda_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
 *
 * Replaces nn.Conv1d/nn.Conv2d as used by sda/nn.py:113-129,131-142,148-176 (heads, tails,
 * residue convs; padding = k//2; padding_mode zeros|circular; stride 1|2), and -- with
 * transposed/flipped packed weights -- their backward-data (what torch.autograd.grad computes
 * through them at sda/score.py:394).  Fused into the input loader (so no tensor is
 * materialised for them):
 *   - the sliding-window view of MCScoreNet.unfold           (sda/score.py:146-153)  [two-level batch stride]
 *   - the broadcast context/forcing channel concat           (sda/score.py:87, experiments/kolmogorov/utils.py:45-46)
 *   - the time modulation add + zuko LayerNorm over channels (sda/nn.py:28,137,163)   [mod, ln_mean, ln_rstd]
 *   - the activation between the two residue convs           (sda/nn.py:139)          [act_in]
 *   - nn.Upsample(nearest) before the tail conv              (sda/nn.py:164)          [up]
 *   - zero insertion (transposed stride-2 conv, backward of sda/nn.py:152-159)        [zins]
 * Fused into the epilogue: bias, multiply by act'(z) (backward through sda/nn.py:139),
 * residual/skip add (sda/nn.py:28,202).
 *
 * out[n][co][oy][ox] = bias[co] + sum_{ci,dy,dx} W[dy*kw+dx][ci][co] * V(n, ci, oy*stride_h+dy-kh/2, ox*stride_w+dx-kw/2)
 * V = virtual input: source x (and ctx channels) after  +mod -> LN -> act_in,  seen through
 *     nearest-upsample `up` or zero-insertion `zins`, padded circularly or with zeros.
 * ------------------------------------------------------------------------------------------ */
typedef struct sda_conv_desc {
    /* source tensor */
    const float* x;
    int64_t x_sn_outer, x_sn_inner; /* image n -> with m = n + x_n_off: (m / n_inner) * sn_outer + (m % n_inner) * sn_inner */
    int32_t n_inner;                /* >=1; windows per trajectory for the unfold view, else 1 (use sn_inner=0) */
    int32_t x_n_off;                /* image index offset applied before the n -> address map (chunked execution) */
    int64_t x_sc, x_sy, x_sx;       /* channel / row / column strides */
    int32_t cx;                     /* channels taken from x */
    const float* ctx;               /* optional extra channels appended after cx: planar [cctx][hs][ws] */
    int64_t ctx_sn;                 /* batch stride of ctx (0 = broadcast over n) */
    int32_t cctx;
    int32_t n;                      /* number of images */
    int32_t hs, ws;                 /* source spatial size */
    int32_t up_h, up_w;             /* >=1  nearest upsample of the source, per axis (1-D nets: up_h = 1) */
    int32_t zins_h, zins_w;         /* >=1  zero insertion (virtual[y][x] = src[y/zh][x/zw] iff both divisible) */
    /* loader transforms (all optional) */
    const float* mod;               /* [*][cx] additive per-channel modulation */
    int64_t mod_sn;                 /* per-image stride of mod (0 = shared) */
    const float* ln_mean;           /* [n][hs*ws] channel-LayerNorm statistics of (x + mod) */
    const float* ln_rstd;
    int32_t act_in;                 /* SDA_ACT_* applied after LN */
    /* filter */
    int32_t kh, kw, stride_h, stride_w, circular;
    const float* w;                 /* packed [kh*kw][cin_pad][cout_pad], see sda_pack_conv_weight */
    int32_t cin_pad, cout_pad;
    const float* bias;              /* [cout] or NULL */
    /* output: planar contiguous [n][cout][ho][wo] */
    float* out;
    int32_t cout, ho, wo;
    /* epilogue (tensors shaped like out) */
    const float* dact_z;            /* out *= act'(dact_z)  (NULL = off) */
    int32_t act_d;
    const float* res;               /* out += res           (NULL = off) */
    /* tiling: cout tile = 32*mt (mt in 1..4); weights must be packed with cout_pad % (32*mt) == 0 */
    int32_t mt;
    /* optional Winograd F(2x2,3x3) weights [16][cin_pad][cout_pad] (sda_pack_conv_weight_wino); when non-NULL and the
     * layer is eligible (3x3, stride 1, no zero insertion, no ctx, cout % 96 == 0, even output size, mt == 3) the
     * transform-domain kernel is used: 2.25x fewer multiplies, fp32 round-off-level error */
    const float* w_wino;
    /* optional overrides (all zero = the defaults above, so a zero-initialised descriptor keeps its meaning).
     *   explicit_pad != 0: taps read in[o*stride + t - pad_h|pad_w]; default kh/2, kw/2 (odd kernels).  Even kernels
     *   need it.
     *   out_sn / out_sc / out_sy / out_sx: element strides of `out` (and of dact_z / res, which always share its
     *   layout) per image / channel / row / pixel; all zero = planar contiguous.  With pad they let the VJP of a
     *   stride-2 convolution run as one small stride-1 convolution per output parity class, each writing its own
     *   interleaved quarter of the gradient, instead of convolving a zero-inserted tensor (4x the multiplies). */
    int32_t explicit_pad, pad_h, pad_w;
    int64_t out_sn, out_sc, out_sy, out_sx;
    /* optional Winograd weights for the second-generation kernel (sda_pack_conv_weight_wino4: [cin_pad/8][16][cout/16][64][2],
     * U fragments in MFMA lane order).  Taken when the layer is Winograd-eligible as above, its output height is a multiple
     * of 8 and its width of 16 (workgroup tile = 96 couts x 16 x 8 pixels of one image), and the loader fusions are one of
     * none / SiLU / LayerNorm / modulation + LayerNorm; otherwise w_wino / the direct kernel serve the launch.
     * (ABI v13) cout % 32 == 0 suffices for THIS kernel: its cout tile is 96 where cout % 96 == 0, else 64 where cout % 64 == 0 (the
     * reference's default widths (64, 128, 256), experiments/kolmogorov/utils.py:52), else 32 (sda/nn.py:99's (32, 64, 128)). */
    const float* w_wino4;
    /* optional output pooling (0 / 1 = off): out is [n][cout][ho / pool_h][wo / pool_w] and receives the SUM of each pool_h x pool_w
     * cell of the convolution's ho x wo output -- the input VJP of `Upsample(nearest) -> conv` (the tails, sda/nn.py:161-169) in one
     * launch, without the full-resolution gradient in between.  Served for (2, 2) by the w_wino4 kernel on plain launches (no loader
     * fusion, no epilogue operand): 7 of the 16 Winograd positions have weight zero in the cell sum and are never multiplied.
     * Anything else: SDA_E_UNSUPPORTED (run the plain launch and pool in the reader, sda_ln_bwd's pool arguments). */
    int32_t pool_h, pool_w;
    /* optional (may be NULL): the w_wino4 weights re-packed for the zero-position kernels (sda_pack_conv_weight_wino4_zp) -- per
     * (K stage, cout tile) the 9 live Winograd positions only, 28 KiB instead of the 36 KiB of the six position pairs that hold
     * them (96-cout tile; 20 / 12 KiB for the 64- / 32-cout tiles).  With it the 2 x 2 up-sampled / pooled launches above run their zero-position form; without it the up-sampled launch
     * runs the full kernel (same result bit for bit) and the pooled launch is SDA_E_UNSUPPORTED. */
    const float* w_wino4_zp;
    /* OPT-IN (ABI v11; all zero = off): the fp32 multiply emulated on the f16 matrix cores, fp32 accumulation (csrc/conv_h2.hip).
     *   w_h2: sda_pack_conv_weight_h2's packing of the layer (every weight as two halves hi + lo of s_w w), w_h2_scale = s_w
     *         (sda_conv_h2_scale(max |w|)).
     *   x_amax: device scalar, max |value the loader feeds the multiply| (after modulation + LayerNorm / activation) or any upper
     *         bound of it -- sda_absmax of the tensor, or the out_amax of the launch that produced it; NULL: x_amax_static (a bound
     *         known on the host: sqrt(channels) behind a LayerNorm).  The kernel derives the power-of-two input scale from it.
     *   out_amax: optional device scalar, atomically maxed (as uint bits; the caller zeroes it) with max |out| of this launch.
     * Also served, with their own packings in w_h2 (see sda_pack_conv_weight_h2_up / _rows below): up_h = up_w = 2 (the tails), pool_h =
     * pool_w = 2 (their VJP summed over the cells), stride_h = stride_w = 2 (the level heads; w_h2 = the four parity classes'
     * sda_pack_conv_weight_h2_rows packings back to back, class (py, px) = taps dy in ((1), (0, 2))[py] x dx alike, rows = cout, k = cin),
     * zins_h = zins_w = 2 (the heads' input VJP: the same with rows = forward cin, k = forward cout).
     * Served by sda_conv_h2 for 3 x 3 / stride 1 / cin % 32 == 0 / cout % 96 == 0 or cout % 64 == 0 (ABI v13; % 96 both before) / 16 x 16-tileable planar launches with the loader
     * fusions none / activation / (modulation +) LayerNorm and the epilogues bias, x act'(z), + res; sda_conv_igemm ignores the
     * fields.  Error against float64 equals the fp32 Winograd kernel's (3e-7, tools/f16_split_numerics.py). */
    const void* w_h2;
    float w_h2_scale;
    float x_amax_static;
    const float* x_amax;
    float* out_amax;
} sda_conv_desc;

int sda_conv_igemm(const sda_conv_desc* d, void* stream);
/* Backward-data of a stride-2 3 x 3 convolution (the gradient through the level heads, sda/nn.py:152-159) as ONE launch: the four
 * output parity classes of the split formulation above together.  d = the class-(0,0) launch (kh = kw = 2, explicit_pad with pad 0,
 * out / res = the class-(0,0) strided views of the planar [n][cout][2 ho][2 wo] gradient / skip tensors) with d->w = the four
 * classes' sda_pack_conv_weight packings (transpose = 1) back to back in the order (0,0), (0,1), (1,0), (1,1): [9][cin_pad][cout_pad].
 * SDA_E_UNSUPPORTED outside the kernel's range (cout % 32 -- 96- / 64- / 32-cout tiles as sda_conv_desc.w_wino4 --, ho % 8, wo % 16,
 * loader fusions): run the four class launches. */
int sda_conv_parity4(const sda_conv_desc* d, void* stream);
/* which kernel family would serve the launch (pure planning, nothing is launched): 2 = one-wave-per-SIMD Winograd
 * (w_wino4), 1 = Winograd (w_wino), 3 = the single-round-trip small 1-D kernel, 4 = the 3 x 3 kernel for <= 16 output
 * channels, 5 = the w_wino4 kernel in its zero-position form (2 x 2 up-sampled source or pooled output: 54 of the 96 multiplies
 * per stage), 0 = direct implicit GEMM; <0 error */
int sda_conv_igemm_path(const sda_conv_desc* d);
/* bytes of dynamic LDS the launch would use (or <0 error), for planning / tests */
int64_t sda_conv_igemm_lds_bytes(const sda_conv_desc* d);

/* ------------------------------------------------------------------------------------------
 * One modulated residual block of a 1-D U-Net in ONE launch (sda/nn.py:18-28, 113-176 with spatial = 1):
 *     y = a + conv2(act(conv1(LN(a + mod)))),   both convolutions c -> c, kernel 3, stride 1, same padding mode;
 * and its input VJP  gx = g + LN^T(conv1^T(act'(z) . conv2^T(g))).  For the latency-bound nets of the Lorenz experiments
 * (c <= 64): three launches forward / three backward become one each.  a, y, z, g, gx: planar [n][c][len] contiguous;
 * mean / rstd: [n][len].  w1 / w2: sda_pack_conv_weight packings [3][k_pad][m_pad] -- FORWARD form for sda_block1d_fwd,
 * BACKWARD-DATA form (transpose = 1) for sda_block1d_bwd.  SDA_E_UNSUPPORTED when the shape is outside the kernel's range
 * (callers fall back to sda_ln_stats + sda_conv_igemm + sda_ln_bwd). */
typedef struct sda_block1d_desc {
    int32_t n, c, len;
    int32_t circular, act, unbiased;   /* padding mode of both convolutions; SDA_ACT_*; LayerNorm variance convention */
    float eps;
    int32_t k_pad, m_pad;
    const float* a;
    const float* mod;                  /* [*][c] additive modulation (NULL = none) */
    int64_t mod_sn;                    /* per-image stride of mod (0 = shared) */
    const float* w1; const float* b1;  /* b1 / b2 may be NULL; unused by the VJP */
    const float* w2; const float* b2;
    float* z;                          /* fwd: out (pre-activation of conv1, NULL = not kept); bwd: in */
    float* mean; float* rstd;          /* fwd: out (NULL = not kept); bwd: in */
    float* y;                          /* fwd: out */
    const float* g;                    /* bwd: in */
    float* gx;                         /* bwd: out */
} sda_block1d_desc;
int sda_block1d_fwd(const sda_block1d_desc* d, void* stream);
int sda_block1d_bwd(const sda_block1d_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * A whole SINGLE-LEVEL 1-D U-Net in ONE launch (the Lorenz score networks, experiments/lorenz/utils.py:26-42;
 * sda/nn.py:184-206 with one level: head convolution cin -> c, `nblocks` modulated residual blocks (descent then ascent,
 * sda/nn.py:18-28), tail convolution c -> cout; every convolution kernel 3, stride 1, one padding mode), and its input VJP
 * in one more.  A workgroup owns a run of positions of one sequence and recomputes a halo of one position per convolution
 * and side (csrc/net1d.hip), so no layer ever waits for another workgroup.
 *   x / out: addressed as base + image * sn + channel * sc + position * sx (elements): (B, C, L) and (B, L, C) tensors alike.
 *   w: every convolution as an sda_pack_conv_weight packing [3][64][64] (k_pad = m_pad = 64: zero padded), 2 + 2 nblocks of them
 *   back to back in EXECUTION order; bias: [2 + 2 nblocks][64] zero padded, same order (NULL = none).
 *     sda_net1d_fwd: FORWARD packings: head, (conv1, conv2) of block 0 .. nblocks - 1, tail.
 *     sda_net1d_bwd: BACKWARD-DATA packings (transpose = 1): tail^T (cout -> c), (conv2^T, conv1^T) of block nblocks - 1 .. 0,
 *                    head^T (c -> cin); x = the incoming cotangent (cin = its channels), out = the input gradient (cout = its
 *                    channels); mod[k] is still the modulation of forward block k.
 *   a_save / z_save [nblocks][n][c][len], mean_save / rstd_save [nblocks][n][len] (block strides save_stride / stat_stride):
 *   written by the forward when non-NULL (all four or none), read by the VJP.
 * SDA_E_UNSUPPORTED outside the kernel's range (callers fall back to the per-block kernels). */
#define SDA_NET1D_MAXB 8
typedef struct sda_net1d_desc {
    int32_t n, len;
    int32_t cin, c, cout;
    int32_t nblocks;
    int32_t circular, act, unbiased;
    float eps;
    const float* x; int64_t x_sn, x_sc, x_sx;
    float* out; int64_t out_sn, out_sc, out_sx;
    const float* w;
    const float* bias;
    const float* mod[SDA_NET1D_MAXB];  /* [*][c] additive modulation of block k (NULL = none) */
    int64_t mod_sn;                    /* per-image stride of every mod (0 = shared) */
    float* a_save; float* z_save; int64_t save_stride;
    float* mean_save; float* rstd_save; int64_t stat_stride;
} sda_net1d_desc;
int sda_net1d_fwd(const sda_net1d_desc* d, void* stream);
int sda_net1d_bwd(const sda_net1d_desc* d, void* stream);

/* One Gaussian-guided score evaluation (GaussianScore.forward, sda/score.py:375-396) of such a network in TWO launches -- the
 * dependency minimum: every position of eps feeds the likelihood, whose cotangent feeds the VJP at every position -- and the
 * predictor / corrector bookkeeping of VPSDE.sample (sda/score.py:250-261) in their epilogues (ABI v7):
 *   sda_net1d_fwd_fused: out = eps = (cx0 + cx1 sigma) x + cn net(x, t) and
 *                        ghat = A^T((y - A x_hat) / (std^2 + gamma (sigma/mu)^2)), x_hat = (x - sigma eps) / mu   (score.py:387-392)
 *                        for A = x[..., p_start:p_stop:p_step, c_start:c_stop:c_step] (experiments/lorenz/eval.py:75), y
 *                        [n or 1][positions][channels] observed values;
 *   sda_net1d_bwd_fused: x = ghat; out = eps - (sigma/mu)(ghat - sigma J_eps^T ghat)   (score.py:394-396), then by `mode`
 *                        0: write out;  1: x <- r x + c1 out in place (score.py:252-253; out is not written);
 *                        2: write out and partial[image][tile] = sum of out^2 over the tile (score.py:259: delta = tau / mean(eps^2)).
 * (cx0, cx1, cn) = (0, 0, 1) is a bare network; other values serve estimators of the form eps = a(t) x + c net(x, t) (bench.py's
 * synthetic score).  x, eps, ghat, out share the (sn, sc, sx) strides of the descriptor's `out`; cin == cout.
 * coef: device {mu(t), sigma(t)}; step_coef: device {r, c1}.  Arithmetic = the unfused kernels' (same operations, same order). */
typedef struct sda_net1d_fuse {
    float cx0, cx1, cn;
    const float* coef;
    const float* y; int64_t y_sn;      /* fwd: observation; per-image stride (0 = one observation shared by the batch) */
    int32_t p_start, p_step, p_stop;   /* fwd: observed positions */
    int32_t c_start, c_step, c_stop;   /* fwd: observed channels */
    float std, gamma;
    float* ghat;                       /* fwd: out */
    const float* eps;                  /* bwd: in (the forward's out) */
    int32_t mode;                      /* bwd */
    float* xs;                         /* bwd mode 1: x, in place */
    const float* step_coef;            /* bwd mode 1 */
    float* partial; int32_t partial_stride;   /* bwd mode 2: [n][partial_stride >= sda_net1d_tiles(d)] */
} sda_net1d_fuse;
int sda_net1d_fwd_fused(const sda_net1d_desc* d, const sda_net1d_fuse* f, void* stream);
int sda_net1d_bwd_fused(const sda_net1d_desc* d, const sda_net1d_fuse* f, void* stream);
int sda_net1d_tiles(const sda_net1d_desc* d);          /* tiles per sequence of the launch serving `d` (host, no launch) */

/* Everything a predictor-corrector step needs before its score evaluations, in ONE launch (sda/score.py:250-253 scalars,
 * TimeEmbedding score.py:15-35, every block's `project` nn.py:132-135), for up to two time values:
 *   table != NULL: row = table[istep[0]] = {t, t - dt, r, c1, sigma(t - dt)} (the host-evaluated schedule of VPSDE.sample);
 *                  times = {t, t - dt};  istep[0] is incremented AFTER everything is written (single workgroup: the only reader);
 *   table == NULL: times = t_dev[0 .. nt).
 * out_coef (floats): {mu(t0), sigma(t0), mu(t1), sigma(t1), r, c1, sigma_next, t0, t1};  out_step (int64, optional): the step index
 * the row was read at;  mod: [nt][cp] modulation vectors -- or, with cp = 0 (wp NULL: a ScoreNet has no projections), the time
 * embedding itself, [nt][e].  Arithmetic identical to sda_vp_schedule / sda_time_embed /
 * sda_linear_small. */
int sda_step1d_prologue(const float* table, int row_len, int64_t* istep, const float* t_dev, int nt,
                        int alpha_kind, float eta, float k, int sigma_kind,
                        const float* freqs, int nf, const float* w0, const float* b0, int hidden, const float* w2, const float* b2, int e,
                        const float* wp, const float* bp, int cp,
                        float* out_coef, int64_t* out_step, float* mod, void* stream);
/* sda_pc_correct with z = the row-keyed N(0, 1) draw of sda_randn_rows generated in the kernel (same Philox counters: identical
 * values), draw index = draw_dev[0] * draw_mul + draw_add  (sda/score.py:257-261). */
int sda_pc_correct_keyed(float* x, const float* eps, int b, int64_t per_sample, const float* partial, int nchunk, float tau,
                         float sigma, const float* coef_dev, uint64_t seed, int64_t row0, const int64_t* draw_dev, int64_t draw_mul,
                         int64_t draw_add, void* stream);

/* Repack torch-layout conv weights [cout][cin][kh][kw] for sda_conv_igemm.
 *   transpose = 0: forward          dst[tap][ci][co]        = w[co][ci][dy][dx]
 *   transpose = 1: backward-data    dst[tap'][co][ci]       = w[co][ci][dy][dx], tap' = (kh-1-dy)*kw + (kw-1-dx)
 *                  (the packed "cin" axis is then the forward cout and vice versa)
 * cin_keep: for transpose=1, only the first cin_keep forward input channels are produced (drops ctx grads).
 * Rows/cols beyond the real sizes are zero filled. */
int sda_pack_conv_weight(const float* w, int cout, int cin, int kh, int kw, int transpose, int cin_keep,
                         float* dst, int k_pad, int m_pad, void* stream);

/* Winograd-domain weights U[4*xi+nu][k][m] = (G g G^T)[xi][nu] for 3x3 filters; transpose / cin_keep as above. */
int sda_pack_conv_weight_wino(const float* w, int cout, int cin, int transpose, int cin_keep, float* dst, int k_pad,
                              int m_pad, void* stream);

/* U for the second-generation Winograd kernel: dst[k_pad/8][16][m_pad/16][64][2], k_pad % 8 == 0, m_pad % 32 == 0 (ABI v13; % 96 before:
 * the layout is per 16-cout fragment and does not depend on the cout tile the kernel then picks from m_pad). */
int sda_pack_conv_weight_wino4(const float* w, int cout, int cin, int transpose, int cin_keep, float* dst, int k_pad,
                               int m_pad, void* stream);
/* The zero-position packing of a sda_pack_conv_weight_wino4 buffer (sda_conv_desc.w_wino4_zp): dst holds
 * sda_wino4_zp_floats(k_pad, m_pad) = (k_pad / 8) * (m_pad / T) * Z floats with (T, Z) = (96, 7168) for m_pad % 96 == 0, else (64, 5120)
 * for m_pad % 64 == 0, else (32, 3072) (ABI v13), [K stage][cout tile][Z]; for the 96-cout tile: the three position pairs
 * whose two positions are both live (6 x 256 float4 each), the live halves of the pairs (2, 3) and (6, 7) interleaved into one such block,
 * the live half of pair (14, 15) as 6 x 256 float2, zero padding to 28 KiB -- the order the kernel's LDS stage buffer has, so that a
 * helper wave copies seven linear 1-KiB pieces.  Values are copied, not recomputed: the kernels stay bit-identical to the full ones. */
int sda_pack_conv_weight_wino4_zp(const float* w_wino4, int k_pad, int m_pad, float* dst, void* stream);
int64_t sda_wino4_zp_floats(int k_pad, int m_pad);

/* ------------------------------------------------------------------------------------------
 * zuko.nn.LayerNorm(dim=-(spatial+1)) statistics: per pixel, over channels, of (x + mod).
 * Call sites sda/nn.py:137,163; algorithm zuko==0.1.4 (not in tree): mean, var (unbiased),
 * rstd = 1/sqrt(var+eps).  x planar [n][c][hw].  The normalisation itself is applied inside
 * the consuming conv's loader (sda_conv_igemm) or sda_ln_apply.
 * ------------------------------------------------------------------------------------------ */
int sda_ln_stats(const float* x, int n, int c, int hw, const float* mod, int64_t mod_sn, float eps, int unbiased,
                 float* mean, float* rstd, void* stream);
/* y = (x + mod - mean) * rstd   (materialised LN; used by tests and non-fused callers) */
int sda_ln_apply(const float* x, int n, int c, int hw, const float* mod, int64_t mod_sn, const float* mean,
                 const float* rstd, float* y, void* stream);
/* Backward of h = LN_c(x + mod) w.r.t. x (what autograd computes through sda/nn.py:137,163):
 *   gx = (res ? res : 0) + rstd * (gh - mean_c(gh) - h * sum_c(gh*h)/(c-1|c))
 * pool_h x pool_w: gh is given at that multiple of the resolution ([n][c][pool_h*h][pool_w*w]) and summed over each cell
 * first (backward of nn.Upsample nearest, sda/nn.py:164): 1x1 none, 2x2 for 2-D nets, 1x2 for 1-D nets (h == 1 there, but
 * h == 1 does not imply a 1-D net: the deepest level of a 2-D net may be one row high). */
int sda_ln_bwd(const float* gh, const float* x, int n, int c, int h, int w, const float* mod, int64_t mod_sn,
               const float* mean, const float* rstd, int unbiased, int pool_h, int pool_w, const float* res, float* gx,
               void* stream);
/* (ABI v12) the same, and amax[0] = max |gx| (device scalar, written by this call): the input scale of the OPT-IN f16 x 2 convolution
 * that reads gx next (sda_conv_desc.x_amax) without an sda_absmax pass -- reduced in the kernel's own epilogue on the U-Net levels'
 * 16-byte layouts, by an sda_absmax pass over gx otherwise. */
int sda_ln_bwd_amax(const float* gh, const float* x, int n, int c, int h, int w, const float* mod, int64_t mod_sn,
                    const float* mean, const float* rstd, int unbiased, int pool_h, int pool_w, const float* res, float* gx,
                    float* amax, void* stream);

/* ------------------------------------------------------------------------------------------
 * TimeEmbedding + every block's `project` Linear in one go (sda/score.py:15-35, sda/nn.py:132-135):
 *   feat = [cos(freqs*t), sin(freqs*t)]; emb = W2 silu(W0 feat + b0) + b2;  mod = Wp emb + bp
 * t: [nt] ; emb out: [nt][e]; mod out: [nt][cp]  (cp = sum of block widths, Wp = rows concatenated)
 * ------------------------------------------------------------------------------------------ */
int sda_time_embed(const float* t, int nt, const float* freqs, int nf, const float* w0, const float* b0, int hidden,
                   const float* w2, const float* b2, int e, float* emb, void* stream);
int sda_linear_small(const float* x, int rows, int in_f, const float* w, const float* b, int out_f, float* y,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * Fully-connected layers of ScoreNet / ResMLP (the Lorenz local kernel; sda/nn.py:31-71, sda/score.py:38-63), row-major
 * (rows, features) fp32, on the fp32 matrix cores:
 *   y = act_out( act_in(x) . Wop + b ) * act'(dact_z) + res
 *   trans_w = 0: Wop = w^T with w = torch's [out_f][in_f]  (forward of nn.Linear)
 *   trans_w = 1: Wop = w   with w = [in_f][out_f]          (backward-data of nn.Linear: gx = gy . W, pass in_f = W's out)
 * and zuko.nn.LayerNorm over the last axis (call site sda/nn.py:61) with its input gradient; one 64-lane wavefront per
 * row, shuffle reductions.
 * ------------------------------------------------------------------------------------------ */
int sda_linear(const float* x, int rows, int in_f, const float* w, const float* b, int out_f, int trans_w, int act_in,
               int act_out, const float* dact_z, int act_d, const float* res, float* y, void* stream);
int sda_row_ln(const float* x, int rows, int f, float eps, int unbiased, float* y, float* mean, float* rstd, void* stream);
int sda_row_ln_bwd(const float* gh, const float* x, int rows, int f, const float* mean, const float* rstd, int unbiased,
                   const float* res, float* gx, void* stream);

/* A whole ResMLP (sda/nn.py:31-71: per width step an optional Linear, then x + Lin2(act(Lin1(LN(x))))) in ONE launch, and its
 * input VJP in one more (csrc/mlp1d.hip; ABI v7) -- the Lorenz local score kernel of experiments/lorenz/utils.py:45-59.
 * The network is a list of GEMMs in forward order: kind 0 = nn.Linear; kind 1 = first half of a residual block (LayerNorm -> Linear ->
 * activation), always followed by kind 2 = its second half (Linear + residual).  Widths <= 128.
 *   w:    per GEMM one SLAB at float offset w_off[g] (a multiple of 4): the matrix Wp = W zero padded to [16 mf][16 kq] (mf = 1 if out <= 16
 *         else 8 output fragments; kq = 1 / 4 / 8 K quads for in <= 16 / 64 / 128) in MFMA A-operand order [m mf][sq kq][lane 64][4] --
 *         element e of lane (k = lane >> 4, li = lane & 15) = Wp[16 m + li][16 sq + 4 k + e] --, zero padded to a multiple of 4096 floats;
 *         sda_mlp_slab_floats(in, out) = its length.  sda_mlp_fwd: W = torch's [out][in] weight; sda_mlp_bwd: W = its transpose (the slab
 *         of (out -> in)), same offsets table;
 *   bias: [16 mf] zero padded at float offset b_off[g] (a multiple of 4; forward only);
 *   x / out: row-major (rows, features) with row strides x_ld / out_ld (sda_mlp_bwd: x = cotangent rows of width out_f[last], out =
 *         input-gradient rows of width in_f[0]).
 *   a_save / z_save [nres][rows][save_ld >= 128], mean_save / rstd_save [nres][rows] (block strides save_stride / stat_stride): written by
 *         the forward when non-NULL (all four or none), read by the VJP.
 * SDA_E_UNSUPPORTED outside the kernel's range (callers run the per-layer kernels sda_linear / sda_row_ln). */
#define SDA_MLP_MAXG 32
typedef struct sda_mlp_desc {
    int32_t rows, ngemm;
    int32_t act, unbiased;
    float eps;
    int32_t kind[SDA_MLP_MAXG];
    int32_t in_f[SDA_MLP_MAXG], out_f[SDA_MLP_MAXG];
    int32_t w_off[SDA_MLP_MAXG], b_off[SDA_MLP_MAXG];
    const float* w; const float* bias;
    const float* x; int64_t x_ld;
    float* out; int64_t out_ld;
    float* a_save; float* z_save; int64_t save_stride; int32_t save_ld;
    float* mean_save; float* rstd_save; int64_t stat_stride;
} sda_mlp_desc;
int sda_mlp_fwd(const sda_mlp_desc* d, void* stream);
int sda_mlp_bwd(const sda_mlp_desc* d, void* stream);
int sda_mlp_slab_floats(int in_f, int out_f);         /* host, no launch */

/* The same two launches as the halves of a Gaussian-guided evaluation of a LOCAL score network -- MCScoreNet over a ScoreNet kernel
 * (sda/score.py:134-164, 53-63; experiments/lorenz/utils.py:45-59) inside GaussianScore (score.py:375-396) --, the rows being the
 * nw = len - 2k windows of each of rows / nw trajectories x (B, len, c), (2k + 1) c <= 16:
 *   sda_mlp_fwd_win: the loader gathers a window's (2k + 1) c consecutive values and appends the time embedding emb[emb_n]
 *                    (`unfold`, `cat`: in_f[0] = (2k + 1) c + emb_n); the epilogue applies `fold` (score.py:155-164), forms
 *                    eps = (cx0 + cx1 sigma) x + cn s and the likelihood cotangent ghat exactly as sda_net1d_fwd_fused does, and writes both
 *                    as (B, len, c) tensors (d.x / d.out are unused);
 *   sda_mlp_bwd_win: the loader applies fold's adjoint to cn ghat; the window part of the input gradient leaves as gwin [rows][16];
 *   sda_mc_finish:   sums the overlapping windows (unfold's adjoint), adds the affine part, forms the guided score
 *                    eps - (sigma/mu)(ghat - sigma J_eps^T ghat) and, by mode (as sda_net1d_bwd_fused): 0 writes it; 1 x <- r x + c1 . in
 *                    place; 2 writes it and partial[b] = its sum of squares over the trajectory (one chunk per sample). */
typedef struct sda_mlp_win {
    int32_t nw, len, c, emb_n;
    const float* x; const float* emb;
    float cx0, cx1, cn;
    const float* coef;                 /* device {mu(t), sigma(t)} */
    const float* y; int64_t y_sn;
    int32_t p_start, p_step, p_stop, c_start, c_step, c_stop;
    float std, gamma;
    float* eps; float* ghat;           /* fwd: out; bwd: ghat in */
    float* gwin;                       /* bwd: out */
} sda_mlp_win;
int sda_mlp_fwd_win(const sda_mlp_desc* d, const sda_mlp_win* w, void* stream);
int sda_mlp_bwd_win(const sda_mlp_desc* d, const sda_mlp_win* w, void* stream);
int sda_mc_finish(const float* eps, const float* ghat, const float* gwin, int b, int nw, int k, int c, float cx0, float cx1,
                  const float* coef, int mode, float* out, float* xs, const float* step_coef, float* partial, void* stream);

/* ------------------------------------------------------------------------------------------
 * MCScoreNet.fold (sda/score.py:155-164): selective gather, NOT an overlap-add.
 *   s: [b][nw][(2k+1)*c][hw] -> out: [b][nw+2k][c][hw]
 * and the adjoints needed for the guidance gradient (sda/score.py:394):
 *   fold_adjoint:   g_out [b][l][c][hw] -> g_s [b][nw][(2k+1)c][hw]  (zero where fold does not read)
 *   unfold_adjoint: g_win [b][nw][(2k+1)c][hw] -> g_x [b][l][c][hw] (+= overlapping windows; the one place overlaps sum)
 * ------------------------------------------------------------------------------------------ */
int sda_fold(const float* s, int b, int nw, int k, int c, int hw, float* out, void* stream);
int sda_fold_adjoint(const float* g_out, int b, int nw, int k, int c, int hw, float* g_s, void* stream);
int sda_unfold_adjoint(const float* g_win, int b, int nw, int k, int c, int hw, int64_t win_sc_total, float* g_x,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * Predictor-corrector updates of VPSDE.sample (sda/score.py:250-261).
 *   predict:  x = r*x + c1*eps                                                (score.py:252-253)
 *   sumsq:    partial[b][j] = sum over chunk j of eps[b]^2  (deterministic two-stage mean, score.py:259)
 *   correct:  delta = tau / mean(eps^2); x = x - (delta*eps + sqrt(2 delta)*z)*sigma   (score.py:259-261)
 * coef: device pointer to {r, c1} / {sigma} when coef_dev != NULL (graph replay), else the by-value scalars.
 * ------------------------------------------------------------------------------------------ */
/* out2 = {mu(t), sigma(t)} of the VP / sub-VP / sub-sub-VP schedules (sda/score.py:195-210, 279-302) for a device scalar t.
 * alpha_kind 0 'lin', 1 'cos' (k = acos(sqrt(eta))), 2 'exp' (k = log(eta)); sigma_kind 0 VPSDE, 1 SubVPSDE, 2 SubSubVPSDE. */
int sda_vp_schedule(const float* t, int alpha_kind, float eta, float k, int sigma_kind, float* out2, void* stream);
int sda_pc_predict(float* x, const float* eps, int64_t numel, float r, float c1, const float* coef_dev, void* stream);
int sda_sumsq_partial(const float* eps, int b, int64_t per_sample, float* partial, int nchunk, void* stream);
int sda_pc_correct(float* x, const float* eps, const float* z, int b, int64_t per_sample, const float* partial,
                   int nchunk, float tau, float sigma, const float* coef_dev, void* stream);

/* Corrector noise z ~ N(0, I) (the torch.randn_like(x) of sda/score.py:257) keyed per ROW, for batch-sharded runs:
 *   out[r][j], r in [0, rows), depends on (seed, row0 + r, draw, j) only -- Philox4x32-10 with counter
 *   {j/4, row, draw} and key = seed, Box-Muller on 24-bit uniforms -- so every world size draws the same noise for the
 *   same trajectory and each rank generates its own rows only.  draw_dev (optional): the draw index is
 *   draw_dev[0] * draw_mul + draw_add, read on the device (a step counter) => the launch is hipGraph-replayable.
 * sda_philox_words: the raw generator words for counter {i, c1, c2, c3}, i in [0, n) (4 uint32 each; tests). */
int sda_randn_rows(float* out, int rows, int64_t per_row, uint64_t seed, int64_t row0, int64_t draw,
                   const int64_t* draw_dev, int64_t draw_mul, int64_t draw_add, void* stream);
int sda_philox_words(uint32_t* out, int64_t n, uint64_t seed, uint32_t c1, uint32_t c2, uint32_t c3, void* stream);

/* Gaussian-guidance elementwise pieces (sda/score.py:387,396):
 *   xhat = (x - sigma*eps)/mu ;   out = eps - (sigma/mu)*(ghat - sigma*vjp),  vjp = J_eps^T ghat
 * coef_dev (optional): device {mu, sigma} overriding the by-value scalars (no host sync, graph replay). */
int sda_denoise(const float* x, const float* eps, int64_t numel, float mu, float sigma, const float* coef_dev,
                float* xhat, void* stream);
int sda_guided_combine(const float* eps, const float* ghat, const float* vjp, int64_t numel, float mu, float sigma,
                       const float* coef_dev, float* out, void* stream);
/* out = (y - ax) / (std^2 + gamma (sigma/mu)^2): the cotangent of log N(y | A x_hat, var) (sda/score.py:389-392) for scalar
 * std / gamma; y broadcasts over the leading axis (y_numel divides numel); (mu, sigma) from coef_dev when non-NULL */
int sda_gauss_cotangent(const float* y, int64_t y_numel, const float* ax, int64_t numel, float std, float gamma, float mu,
                        float sigma, const float* coef_dev, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Linear observation operators A of the reference's experiments and their adjoints A^T, so that GaussianScore
 * (sda/score.py:387-394) needs no autograd through A: d log p / d x_hat = A^T((y - A x_hat)/var).
 *   subsample: a 5-D strided slice x[..., start::step, ...] (experiments/lorenz/eval.py:75; kolmogorov/figures.ipynb#cell30-39)
 *   coarsen:   KolmogorovFlow.coarsen, block mean over f x f cells      (sda/mcs.py:340-347)
 *   vorticity: KolmogorovFlow.vorticity, periodic central differences   (sda/mcs.py:361-375); x = [pairs][2][h][w]
 * size5/start5/step5: host int[5] (leading dims padded with size 1, start 0, step 1).
 * ------------------------------------------------------------------------------------------ */
int sda_obs_subsample(const float* x, const int* size5, const int* start5, const int* step5,
                      const int* stop5 /* exclusive ends per dim (a crop, figures.ipynb#cell16,23), or NULL */, float* out, void* stream);
int sda_obs_subsample_adjoint(const float* r, const int* size5, const int* start5, const int* step5, const int* stop5, float* gx,
                              void* stream);
/* g = A^T((y - A((x - sigma eps)/mu)) / (std^2 + gamma (sigma/mu)^2)) for the subsampling A above, scalar std / gamma, in one
 * launch (sda/score.py:387-394); y broadcasts over the leading axis; (mu, sigma) from coef_dev when non-NULL */
int sda_obs_subsample_guidance(const float* x, const float* eps, const float* y, int64_t y_numel, const int* size5,
                               const int* start5, const int* step5, const int* stop5 /* exclusive ends per dim, or NULL */,
                               float std, float gamma, float mu, float sigma, const float* coef_dev, float* g, void* stream);
int sda_obs_coarsen(const float* x, int64_t planes, int h, int w, int f, float* out, void* stream);
int sda_obs_coarsen_adjoint(const float* r, int64_t planes, int h, int w, int f, float* gx, void* stream);
int sda_obs_vorticity(const float* x, int64_t pairs, int h, int w, float* out, void* stream);
int sda_obs_vorticity_adjoint(const float* r, int64_t pairs, int h, int w, float* gx, void* stream);
/* The non-linear / masked / coupled observations of the reference's experiments with the VJP of their
 * linearisation, so that the guidance gradient d log p / d x_hat = J_A(x_hat)^T((y - A x_hat)/var) (sda/score.py:389-394)
 * needs no autograd through A:
 *   pointwise kind 1: w / (1 + |w|)  (the saturating sensor of kolmogorov/figures.ipynb#cell23), 2: tanh, 3: w^2, 4: |w|;
 *             _vjp: gx = r * f'(x)
 *   mask:     out = x * m, m broadcast over the leading dims (m_numel divides n)   (figures.ipynb#cell4); self-adjoint
 *   timediff: x [outer][len][inner] -> x[:, i] - x[:, j]  (the loop closure x[:, 0] - x[:, -1] of figures.ipynb#cell43);
 *             _adjoint: gx = +r at index i, -r at index j, 0 elsewhere */
int sda_obs_pointwise(const float* x, int64_t n, int kind, float* out, void* stream);
int sda_obs_pointwise_vjp(const float* x, const float* r, int64_t n, int kind, float* gx, void* stream);
int sda_obs_mask(const float* x, int64_t n, const float* m, int64_t m_numel, float* out, void* stream);
int sda_obs_timediff(const float* x, int64_t outer, int len, int64_t inner, int i, int j, float* out, void* stream);
int sda_obs_timediff_adjoint(const float* r, int64_t outer, int len, int64_t inner, int i, int j, float* gx, void* stream);

/* ------------------------------------------------------------------------------------------
 * Evaluation metrics of the sampling experiments.
 *   sda_pairwise_dist   : out[i][j] = |x_i - y_j|^2 (take_sqrt = 0) or |x_i - y_j| (1); x [m][d], y [n][d], out [m][n].
 *                         Replaces torch.cdist in emd (sda/utils.py:215-219) and the Gram-matrix expansion of mmd
 *                         (sda/utils.py:236-250).
 *   sda_mmd_kernel_sums : partial[b] = sum over a grid-stride slice of d2[0..count) of
 *                         sum_{sigma in 1e-3..1e3} exp(-d2/sigma)   (sda/utils.py:252-261); partial: nblocks doubles.
 *   sda_assignment_cost : HOST pointers.  min over permutations of sum_i cost[i][p(i)] -- the optimal transport LP of
 *                         ot.emd2 (sda/utils.py:215) for uniform weights and equally many samples (POT, a third-party CPU
 *                         solver, is not vendored by the reference); col_of_row (optional) receives the permutation.
 * ------------------------------------------------------------------------------------------ */
int sda_pairwise_dist(const float* x, int m, const float* y, int n, int64_t d, int take_sqrt, float* out, void* stream);
int sda_mmd_kernel_sums(const float* d2, int64_t count, double* partial, int nblocks, void* stream);
int sda_assignment_cost(const float* cost, int n, double* total, int* col_of_row);
/*   sda_transport_cost : HOST pointers.  The same LP for m != n samples (uniform marginals 1/m, 1/n; sda/utils.py:203-219 with
 *       unequal sample counts): *total = min_P <P, cost>, an integral min-cost flow after scaling by m n.  cost: m x n row-major,
 *       finite and >= 0. */
int sda_transport_cost(const float* cost, int m, int n, double* total);

/* ------------------------------------------------------------------------------------------
 * 3-D convolutions: `UNet(spatial=3)` (sda/nn.py:114-118 picks nn.Conv3d; heads / residual blocks / tails as nn.py:148-206).
 * One gather kernel (implicit GEMM on v_mfma_f32_16x16x4_f32, B operand read from the planar tensor) for every launch of the
 * forward pass and of the input VJP.  Per axis a (d, h, w order):  v = o * stride[a] + tap - pad[a];  circular: v mod V,
 * zeros: taps outside [0, V) contribute nothing;  up[a] > 1: source = v / up[a]  (nn.Upsample(nearest) in the loader,
 * nn.py:164);  dil[a] > 1: source = v / dil[a] iff dil[a] divides v  (zero insertion: the transposed stride-2 convolution,
 * backward of nn.py:152-159).  V = in * up (or, zero-inserted: in * dil circular / (in - 1) * dil + 1 zeros).
 * Loader: act_in(x) on the gathered values (padding stays zero).
 * Epilogue: + bias[cout]; then x act'(z) if z else act(.); then + res.   x: [n][cin][in_size d,h,w], out / z / res:
 * [n][cout][out_size d,h,w], all contiguous.  w: sda_pack_conv3d_weight's layout (transpose = 1: the VJP operator --
 * taps flipped, cin <-> cout; `cin` / `cout` of the descriptor are then the operator's own).
 *   sda_pool3d_sum: adjoint of nearest up-sampling -- out[nc][d][h][w] = sum over the fd x fh x fw cell of g.
 * ------------------------------------------------------------------------------------------ */
typedef struct sda_conv3d_desc {
    const float* x;
    const float* w;
    const float* bias;      /* NULL: none */
    const float* z;         /* NULL: apply act; else multiply by act'(z) */
    const float* res;       /* NULL: none */
    float* out;
    int32_t n, cin, cout;
    int32_t in_size[3], out_size[3];
    int32_t k[3], pad[3], stride[3], up[3], dil[3];
    int32_t circular, act;
    int32_t act_in;         /* activation applied to x by the loader (the conv behind the block's activation, nn.py:139) */
} sda_conv3d_desc;
int sda_conv3d(const sda_conv3d_desc* d, void* stream);
int64_t sda_conv3d_packed_floats(int cout, int cin, int kd, int kh, int kw, int transpose);
int sda_pack_conv3d_weight(const float* w, int cout, int cin, int kd, int kh, int kw, int transpose, float* dst, void* stream);
int sda_pool3d_sum(const float* g, int64_t nc, int d, int h, int w, int fd, int fh, int fw, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement support (ABI v10; no counterpart in the reference, which has no measurement code -- SURVEY.md section 6):
 * the shader clock the fp32 matrix-core stream sustains on this device.  `blocks` workgroups of 256 threads each issue
 * iters x 8 v_mfma_f32_16x16x4_f32 per wave from registers; out[2 b] = shader cycles (s_memtime), out[2 b + 1] = 100 MHz ticks
 * (s_memrealtime) workgroup b's first wave saw across its stream: clock = cycles / ticks x 100 MHz.  `sink`: one float of scratch.
 * bench.py runs it after the warm-up steps and reports roofline.frac next to the clock it was measured at.
 * ------------------------------------------------------------------------------------------ */
int sda_clock_probe(unsigned long long* out, int blocks, float* sink, int iters, void* stream);

/* ------------------------------------------------------------------------------------------
 * OPT-IN f16 x 2 emulation of the fp32 multiply (ABI v11; csrc/conv_h2.hip; see sda_conv_desc.w_h2).  Same layers as the
 * w_wino4 kernel -- the block convolutions nn.Conv2d(3 x 3) of sda/nn.py:131-142 and their backward-data -- as a DIRECT
 * convolution on v_mfma_f32_16x16x32_f16: x w ~ hi_x hi_w + hi_x lo_w + lo_x hi_w, fp32 accumulation.
 *   sda_conv_h2: the launch (SDA_E_UNSUPPORTED outside the range above: run sda_conv_igemm); sda_conv_h2_supported: 1 / 0, no launch.
 *   sda_pack_conv_weight_h2: torch-layout [cout][cin][3][3] -> fragments in MFMA lane order (transpose = 1: the input-VJP operator,
 *     taps flipped, cin <-> cout), w_amax = max |w| (host); sda_conv_h2_packed_bytes: size of dst (0: shape not served).
 *   sda_conv_h2_scale: the power of two s with s amax in [2^10, 2^11].   sda_absmax: amax[0] = max |x| (device scalar).
 * ------------------------------------------------------------------------------------------ */
int sda_conv_h2(const sda_conv_desc* d, void* stream);
int sda_conv_h2_supported(const sda_conv_desc* d);
int sda_pack_conv_weight_h2(const float* w, int cout, int cin, int transpose, float w_amax, void* dst, void* stream);
int64_t sda_conv_h2_packed_bytes(int cout, int cin, int transpose);
/* (ABI v12) the f16 x 2 form of a 3 x 3 convolution over a 2 x 2 nearest-up-sampled source (sda_conv_desc.up_h = up_w = 2: the tails,
 * sda/nn.py:161-169).  Output pixel (2 i + py, 2 j + px) sees only the 2 x 2 source pixels (i - 1 + py + a, j - 1 + px + b): each output
 * parity class is a 2 x 2-tap convolution of the low-resolution image with the taps that fall on one source pixel summed -- 4 / 9 of the
 * multiplies.  wsum: [4 classes (2 py + px)][cout][cin][4 taps (2 a + b)], summed in fp32 by the caller (class 0 rows: dy {0} | {1, 2};
 * class 1 rows: {0, 1} | {2}; columns alike), w_amax = max |wsum|.  A descriptor with up_h = up_w = 2 takes w_h2 = this packing;
 * served for cin % 32 == 0, cout a multiple of 96 or of 64 (ABI v13), a 16 x 16-tileable SOURCE grid, loader none / (modulation +) LayerNorm, epilogues bias / + res. */
int sda_pack_conv_weight_h2_up(const float* wsum, int cout, int cin, float w_amax, void* dst, void* stream);
int64_t sda_conv_h2_up_packed_bytes(int cout, int cin);
/* (ABI v12) generic form of the packing: w [rows][k][ntap] (ntap 1, 2, 4 or 9; rows a multiple of 96 or of 64 -- the cout tile the kernel runs --, k % 32 == 0: ABI v13).  The VJP of an up-sampled tail
 * summed over the 2 x 2 up-sampling cells (sda_conv_desc.pool_h = pool_w = 2) runs as a 2 x 2-tap convolution over the four parity planes
 * of the fine-resolution gradient: rows = the forward cin, k = 4 classes x the forward cout (class-major), ntap = 4,
 * w[ci][class * cout + co][tap] = wsum[class][co][ci][tap] of sda_pack_conv_weight_h2_up.  Served for a 32 x 32-tileable source, plain loader,
 * no bias / epilogue operand. */
int sda_pack_conv_weight_h2_rows(const float* w, int rows, int k, int ntap, float w_amax, void* dst, void* stream);
int64_t sda_conv_h2_rows_packed_bytes(int rows, int k, int ntap);
float sda_conv_h2_scale(float amax);
int sda_absmax(const float* x, int64_t numel, float* amax, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDA_HIP_H */
