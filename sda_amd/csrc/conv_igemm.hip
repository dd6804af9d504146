// Implicit-GEMM convolution for gfx950 on the fp32 matrix cores.
//
//   D[co][pix] = sum_{tap,ci} W[tap][ci][co] * V[ci][pix + tap]        (M = cout, N = pixels, K = taps*cin)
//
// Design:
//   * activations are PLANAR ([n][c][h][w]); a workgroup owns 128 output pixels (tn images x tr rows x tw cols,
//     all powers of two) x one cout tile of 32*MT channels;
//   * per K-stage (CK input channels) the workgroup stages the HALO TILE of those channels once into LDS
//     ([ck][tn][in_rows][in_cols]) and the weight slab [tap][ck][cout-tile]; all kh*kw taps are then served
//     from LDS -- the input is read from HBM once per cout tile, not once per tap;
//   * the loader applies, on the fly: the sliding-window view (two-level batch stride), the context-channel
//     concat, time-modulation add, channel LayerNorm, activation, nearest upsample, zero insertion and
//     circular / zero padding -- none of these is ever materialised;
//   * 4 wavefronts (64 lanes each); wave w owns pixels [32w, 32w+32) x all MT cout sub-tiles and issues
//     v_mfma_f32_32x32x2_f32: A = W[co = lane&31][k = lane>>5]  (LDS, conflict-free: 32 consecutive floats),
//                             B = V[k = lane>>5][pix = lane&31] (LDS, consecutive columns of the halo tile);
//   * epilogue: D col = lane&31 = pixel, D row = (r&3) + 8*(r>>2) + 4*(lane>>5) = cout  => for each accumulator
//     register 32 lanes store 128 contiguous bytes of one output-channel row.
//
// Everything that is index arithmetic lives in __host__ __device__ functions so that the very same code is
// replayed on the CPU by the emulator at the bottom (built only into libsda_emu.so for tests/).
#include "sda_common.hpp"
#include <stdlib.h>

#ifndef SDA_CONV_CK
#define SDA_CONV_CK 8
#endif
// This file is compiled once per SDA_CONV_PART (sda_amd/build.py) so that the ~60 kernel instantiations build in parallel:
//   part 0: the C ABI, planner, generic kernel, parity / 32-channel-stage variants;   part 1: 3x3 kernels, cout tiles 32/64;
//   part 2: 3x3 kernels, cout tiles 96/128;   part 3: 1x3 kernels (1-D nets)
#ifndef SDA_CONV_PART
#define SDA_CONV_PART 0
#endif
#define SDA_CONV_BP 128
#define SDA_CONV_THREADS 256
#define SDA_CONV_MAXPOS 4

struct ConvGeom {
    int cin;             // cx + cctx
    int hv, wv;          // virtual input size
    int pad_h, pad_w;
    int tn, tr, tw;      // tile: images x rows x cols (powers of two, product 128)
    int tr_shift, tw_shift;
    int tiles_x, tiles_y, tiles_n;
    int n_pt, n_ct, grid;
    int in_rows, in_cols, plane;   // halo tile per (channel, image)
    int S;                         // positions per channel = tn * plane
    int ntaps;
    int bm;                        // cout tile = 32*mt
    int nstage;
    int wide_out;                  // 4-pixel groups of the output are contiguous + 16-B aligned: dwordx4 epilogue
    int fast32;                    // producer offsets relative to the tile base fit 32 bits (always, in practice)
    int debug;                     // perf-ablation bits from $SDA_CONV_DEBUG (0 in production)
    int64_t o_sn, o_sc, o_sy, o_sx; // element strides of out / dact_z / res (image, channel, row, pixel)
    int64_t lds_bytes;
};

static inline int ilog2_pow2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }

static int pick_pow2_tile(int extent, int budget) {
    // largest power of two <= budget whose padded extent wastes <= 25 %; else the smallest that covers, else 1
    int best = 1;
    for (int t = budget; t >= 1; t >>= 1) {
        long padded = (long)((extent + t - 1) / t) * t;
        if (padded * 4 <= (long)extent * 5) { best = t; break; }
    }
    return best;
}

static int conv_plan(const sda_conv_desc* d, ConvGeom* g, int bp = SDA_CONV_BP, int max_pos = SDA_CONV_MAXPOS * SDA_CONV_THREADS,
                     bool shrink_tn = true) {
    if (!d || !d->x || !d->w || !d->out) return SDA_E_BADARG;
    if (d->n <= 0 || d->cx <= 0 || d->cout <= 0 || d->hs <= 0 || d->ws <= 0 || d->ho <= 0 || d->wo <= 0) return SDA_E_BADARG;
    if (d->kh <= 0 || d->kw <= 0) return SDA_E_UNSUPPORTED;
    if (!d->explicit_pad && (!(d->kh & 1) || !(d->kw & 1))) return SDA_E_UNSUPPORTED;     // even kernels: explicit pad
    if (d->explicit_pad && (d->pad_h < 0 || d->pad_w < 0 || d->pad_h >= d->kh || d->pad_w >= d->kw)) return SDA_E_BADARG;
    if (d->stride_h < 1 || d->stride_w < 1 || d->up_h < 1 || d->up_w < 1 || d->zins_h < 1 || d->zins_w < 1) return SDA_E_UNSUPPORTED;
    if ((d->up_h > 1 || d->up_w > 1) && (d->zins_h > 1 || d->zins_w > 1)) return SDA_E_UNSUPPORTED;
    if (d->mt < 1 || d->mt > 4) return SDA_E_UNSUPPORTED;
    if (d->cctx > 0 && !d->ctx) return SDA_E_BADARG;
    if ((d->ln_mean == nullptr) != (d->ln_rstd == nullptr)) return SDA_E_BADARG;
    if (d->n_inner < 1) return SDA_E_BADARG;
    g->cin = d->cx + (d->cctx > 0 ? d->cctx : 0);
    g->hv = d->hs * d->up_h * d->zins_h;
    g->wv = d->ws * d->up_w * d->zins_w;
    g->pad_h = d->explicit_pad ? d->pad_h : d->kh / 2;
    g->pad_w = d->explicit_pad ? d->pad_w : d->kw / 2;
    const bool dense_out = !d->out_sn && !d->out_sc && !d->out_sy && !d->out_sx;
    g->o_sn = dense_out ? (int64_t)d->cout * d->ho * d->wo : d->out_sn;
    g->o_sc = dense_out ? (int64_t)d->ho * d->wo : d->out_sc;
    g->o_sy = dense_out ? (int64_t)d->wo : d->out_sy;
    g->o_sx = dense_out ? 1 : d->out_sx;
    g->bm = 32 * d->mt;
    if (d->cout_pad % g->bm || d->cout_pad < d->cout) return SDA_E_BADARG;
    if (d->cin_pad % SDA_CONV_CK || d->cin_pad < g->cin) return SDA_E_BADARG;
    // (a stride-2 tile reads (2 tw + 1) x (2 tr + 1) inputs: a 2 x 128 tile costs 5.0 input pixels per output and does not fit
    //  the 1280-position stage, a 4 x 64 one 4.5 -- the 96 -> 192 level head at 256^2 went from the 128-pixel tile family, 0.53
    //  of the matrix peak, to the 256-pixel one its 128^2 sibling already used, 0.7)
    const int tw_cap = (d->stride_w > 1 && d->ho > 1) ? 64 : 128;
    g->tw = pick_pow2_tile(d->wo, bp < tw_cap ? bp : tw_cap);
    g->tr = pick_pow2_tile(d->ho, bp / g->tw);
    g->tn = bp / (g->tw * g->tr);
    // tiny images: a tile of many images can need more halo positions than the loader covers -- take fewer images per tile
    // (the tile then has idle pixel slots; only the generic kernel accepts that)
    while (shrink_tn && g->tn > 1 &&
           (long)g->tn * ((g->tr - 1) * d->stride_h + d->kh) * ((g->tw - 1) * d->stride_w + d->kw) > max_pos)
        g->tn >>= 1;
    g->tw_shift = ilog2_pow2(g->tw);
    g->tr_shift = ilog2_pow2(g->tr);
    g->tiles_x = (d->wo + g->tw - 1) / g->tw;
    g->tiles_y = (d->ho + g->tr - 1) / g->tr;
    g->tiles_n = (d->n + g->tn - 1) / g->tn;
    long npt = (long)g->tiles_x * g->tiles_y * g->tiles_n;
    g->n_ct = d->cout_pad / g->bm;
    // drop cout tiles that lie wholly beyond cout (packed padding)
    while (g->n_ct > 1 && (g->n_ct - 1) * g->bm >= d->cout) --g->n_ct;
    if (npt * g->n_ct > 0x7fffffffL) return SDA_E_UNSUPPORTED;
    g->n_pt = (int)npt;
    g->grid = g->n_pt * g->n_ct;
    g->in_rows = (g->tr - 1) * d->stride_h + d->kh;
    g->in_cols = (g->tw - 1) * d->stride_w + d->kw;
    g->plane = g->in_rows * g->in_cols;
    g->S = g->tn * g->plane;
    if (g->S > max_pos) return SDA_E_UNSUPPORTED;
    g->ntaps = d->kh * d->kw;
    g->nstage = d->cin_pad / SDA_CONV_CK;
    g->wide_out = dense_out && (d->wo % 4 == 0) && (g->tw >= 4) && ((reinterpret_cast<uintptr_t>(d->out) & 15) == 0) &&
                  (!d->res || (reinterpret_cast<uintptr_t>(d->res) & 15) == 0) &&
                  (!d->dact_z || (reinterpret_cast<uintptr_t>(d->dact_z) & 15) == 0);
    {   // worst-case |offset| of a halo element relative to its tile's first image, channel 0
        auto ab = [](int64_t v) { return v < 0 ? -v : v; };
        int64_t span = (int64_t)g->tn * (ab(d->x_sn_outer) + ab(d->x_sn_inner)) + (int64_t)d->hs * ab(d->x_sy) +
                       (int64_t)d->ws * ab(d->x_sx);
        bool nonneg = d->x_sn_outer >= 0 && d->x_sn_inner >= 0 && d->x_sy >= 0 && d->x_sx >= 0 &&
                      (g->tn == 1 || d->n_inner == 1 || d->x_sn_outer >= d->x_sn_inner * (int64_t)(d->n_inner - 1));
        g->fast32 = (span < (1LL << 30)) && nonneg;
    }
    { static const int dbg = sda_debug_env(); g->debug = dbg; }            // (0 in the product build)
    g->lds_bytes = ((int64_t)g->ntaps * SDA_CONV_CK * g->bm + (int64_t)SDA_CONV_CK * g->S) * 4;
    if (g->lds_bytes > 160 * 1024) return SDA_E_LDS;
    return SDA_OK;
}

// ---------------------------------------------------------------- index helpers (host + device)

// XCD-aware block remap: hardware places block b on XCD b % 8; give each XCD a contiguous range of logical
// tiles so that the cout tiles of one pixel tile (and neighbouring halos) share that XCD's L2.  Bijective for any G.
__host__ __device__ inline int conv_logical_block(int b, int G) {
    int xcd = b & 7, slot = b >> 3;
    int q = G >> 3, r = G & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

__host__ __device__ inline void conv_decode_block(const ConvGeom& g, int logical, int& ct, int& n0, int& oy0, int& ox0) {
    ct = logical % g.n_ct;
    int pt = logical / g.n_ct;
    int txi = pt % g.tiles_x;
    int r = pt / g.tiles_x;
    int tyi = r % g.tiles_y;
    int tni = r / g.tiles_y;
    n0 = tni * g.tn;
    oy0 = tyi * g.tr;
    ox0 = txi * g.tw;
}

__host__ __device__ inline int conv_wrap(int v, int m) {
    v %= m;
    return v < 0 ? v + m : v;
}

struct ConvPos {
    int64_t xoff;   // element offset of channel 0 of this position inside x, or -1 if the tap reads padding / nothing
    int64_t coff;   // offset inside ctx (channel 0 of the context block)
    int64_t stat;   // index into ln_mean / ln_rstd
    int nimg;
};

// halo position `pos` (0..S) of the tile -> where it comes from
__host__ __device__ inline ConvPos conv_decode_pos(const sda_conv_desc& d, const ConvGeom& g, int pos, int n0, int oy0, int ox0) {
    ConvPos r;
    r.xoff = -1; r.coff = 0; r.stat = 0; r.nimg = 0;
    int tni = pos / g.plane;
    int rem = pos - tni * g.plane;
    int ry = rem / g.in_cols;
    int rx = rem - ry * g.in_cols;
    int n = n0 + tni;
    if (n >= d.n) return r;
    int vy = oy0 * d.stride_h + ry - g.pad_h;
    int vx = ox0 * d.stride_w + rx - g.pad_w;
    if (d.circular) {
        vy = conv_wrap(vy, g.hv);
        vx = conv_wrap(vx, g.wv);
    } else if (vy < 0 || vy >= g.hv || vx < 0 || vx >= g.wv) {
        return r;
    }
    if ((vy % d.zins_h) || (vx % d.zins_w)) return r;      // zero insertion: only multiples carry data
    const int sy = vy / (d.zins_h * d.up_h);                 // (one of zins / up is 1 per axis)
    const int sx = vx / (d.zins_w * d.up_w);
    const int ng = n + d.x_n_off;
    int64_t nbase = (int64_t)(ng / d.n_inner) * d.x_sn_outer + (int64_t)(ng % d.n_inner) * d.x_sn_inner;
    r.xoff = nbase + (int64_t)sy * d.x_sy + (int64_t)sx * d.x_sx;
    r.coff = (int64_t)n * d.ctx_sn + (int64_t)sy * d.ws + sx;
    r.stat = (int64_t)n * d.hs * d.ws + (int64_t)sy * d.ws + sx;
    r.nimg = n;
    return r;
}

// value of virtual-input channel c at a decoded position, after modulation / LayerNorm / activation
__host__ __device__ inline float conv_load_value(const sda_conv_desc& d, const ConvGeom& g, const ConvPos& ps, int c,
                                                 float mean, float rstd) {
    if (ps.xoff < 0 || c >= g.cin) return 0.f;
    float v;
    if (c < d.cx) {
        v = d.x[ps.xoff + (int64_t)c * d.x_sc];
        if (d.mod) v += d.mod[(int64_t)ps.nimg * d.mod_sn + c];
        if (d.ln_mean) v = (v - mean) * rstd;
    } else {
        v = d.ctx[ps.coff + (int64_t)(c - d.cx) * d.hs * d.ws];
    }
    if (d.act_in) v = sda_act(d.act_in, v);
    return v;
}

// LDS offset (in floats, inside one channel plane set) of the (dy=0,dx=0) tap of output pixel p of the tile
__host__ __device__ inline int conv_pix_lds_base(const sda_conv_desc& d, const ConvGeom& g, int p) {
    int tx = p & (g.tw - 1);
    int ty = (p >> g.tw_shift) & (g.tr - 1);
    int tni = p >> (g.tw_shift + g.tr_shift);
    if (tni >= g.tn) tni = 0;                    // idle slot of a shrunken tile: reads staged data, result discarded
    return (tni * g.in_rows + ty * d.stride_h) * g.in_cols + tx * d.stride_w;
}

// output element offset of pixel p, channel 0; -1 if the pixel is outside the tensor
__host__ __device__ inline int64_t conv_pix_out_base(const sda_conv_desc& d, const ConvGeom& g, int p, int n0, int oy0, int ox0) {
    int tx = p & (g.tw - 1);
    int ty = (p >> g.tw_shift) & (g.tr - 1);
    int tni = p >> (g.tw_shift + g.tr_shift);
    int n = n0 + tni, oy = oy0 + ty, ox = ox0 + tx;
    if (tni >= g.tn || n >= d.n || oy >= d.ho || ox >= d.wo) return -1;
    return (int64_t)n * g.o_sn + (int64_t)oy * g.o_sy + (int64_t)ox * g.o_sx;
}

__host__ __device__ inline void conv_epilogue_store(const sda_conv_desc& d, const ConvGeom& g, int64_t obase, int co, float acc) {
    if (obase < 0 || co >= d.cout) return;
    int64_t off = obase + (int64_t)co * g.o_sc;
    float v = acc;
    if (d.bias) v += d.bias[co];
    if (d.dact_z) v *= sda_dact(d.act_d, d.dact_z[off]);
    if (d.res) v += d.res[off];
    d.out[off] = v;
}

// D-fragment row of accumulator register r for v_mfma_f32_32x32x2_f32 (col = lane & 31)
__host__ __device__ inline int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------- the kernel
#ifndef SDA_HOST_EMU

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MT, int NPOS>
__global__ __launch_bounds__(SDA_CONV_THREADS) void conv_igemm_kernel(const sda_conv_desc d, const ConvGeom g) {
    constexpr int CK = SDA_CONV_CK;
    constexpr int BM = MT * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_w = smem;                          // [ntaps][CK][BM]
    float* s_in = smem + g.ntaps * CK * BM;     // [CK][S]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int khalf = lane >> 5;

    int ct, n0, oy0, ox0;
    conv_decode_block(g, conv_logical_block(blockIdx.x, gridDim.x), ct, n0, oy0, ox0);
    const int co0 = ct * BM;

    // per-thread halo positions (fixed for the whole kernel; only the channel slab changes per stage)
    ConvPos ps[NPOS];
    float pmean[NPOS], prstd[NPOS];
#pragma unroll
    for (int i = 0; i < NPOS; ++i) {
        int pos = tid + i * SDA_CONV_THREADS;
        ps[i].xoff = -1; ps[i].coff = 0; ps[i].stat = 0; ps[i].nimg = 0;
        pmean[i] = 0.f; prstd[i] = 1.f;
        if (pos < g.S) {
            ps[i] = conv_decode_pos(d, g, pos, n0, oy0, ox0);
            if (d.ln_mean && ps[i].xoff >= 0) {
                pmean[i] = d.ln_mean[ps[i].stat];
                prstd[i] = d.ln_rstd[ps[i].stat];
            }
        }
    }

    const int pix = wave * 32 + l31;
    const int pixbase = conv_pix_lds_base(d, g, pix);

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    for (int st = 0; st < g.nstage; ++st) {
        const int c0 = st * CK;
        // ---- stage the weight slab: [tap][ck][BM] <- w[tap][c0+ck][co0 .. co0+BM)
        if (!(SDA_DBG(g, 2) && st > 0)) {
            constexpr int ROW4 = BM / 4;
            const int total4 = g.ntaps * CK * ROW4;
            for (int f = tid; f < total4; f += SDA_CONV_THREADS) {
                int row = f / ROW4;
                int c4 = f - row * ROW4;
                int tap = row / CK;
                int ck = row - tap * CK;
                const float* src = d.w + ((int64_t)tap * d.cin_pad + c0 + ck) * d.cout_pad + co0 + c4 * 4;
                *reinterpret_cast<f32x4*>(s_w + row * BM + c4 * 4) = *reinterpret_cast<const f32x4*>(src);
            }
        }
        // ---- stage the input halo tile with every loader-side fusion applied
        if (!(SDA_DBG(g, 2) && st > 0)) {
            float v[NPOS][CK];
#pragma unroll
            for (int i = 0; i < NPOS; ++i)
#pragma unroll
                for (int ck = 0; ck < CK; ++ck) v[i][ck] = conv_load_value(d, g, ps[i], c0 + ck, pmean[i], prstd[i]);
#pragma unroll
            for (int i = 0; i < NPOS; ++i) {
                int pos = tid + i * SDA_CONV_THREADS;
                if (pos < g.S) {
#pragma unroll
                    for (int ck = 0; ck < CK; ++ck) s_in[ck * g.S + pos] = v[i][ck];
                }
            }
        }
        __syncthreads();
        // ---- MFMA over all taps of this channel slab
        if (!SDA_DBG(g, 4)) {
            int dy = 0, dx = 0;
            for (int tap = 0; tap < g.ntaps; ++tap) {
                const int toff = dy * g.in_cols + dx + pixbase;
                const float* wrow = s_w + tap * CK * BM + l31;
#pragma unroll
                for (int k2 = 0; k2 < CK / 2; ++k2) {
                    const int ck = 2 * k2 + khalf;
                    const float b = s_in[ck * g.S + toff];
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const float a = wrow[ck * BM + m * 32];
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
                    }
                }
                if (++dx == d.kw) { dx = 0; ++dy; }
            }
        }
        __syncthreads();
    }

    // ---- epilogue
    const int64_t obase = conv_pix_out_base(d, g, pix, n0, oy0, ox0);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) conv_epilogue_store(d, g, obase, co0 + m * 32 + mfma32_row(r, lane), acc[m][r]);
}

// ---------------------------------------------------------------- v2: wave-specialised, double-buffered kernel
//
// 8 wavefronts per workgroup: waves 0-3 are CONSUMERS (nothing but ds_read + v_mfma, every LDS offset an immediate),
// waves 4-7 are PRODUCERS (global loads, loader-side fusions, LDS writes; the weight slab goes HBM/L2 -> LDS directly
// with global_load_lds_dwordx4).  Two LDS buffers: producers fill stage s+1 while consumers multiply stage s; one
// workgroup barrier per stage.  The matrix pipe and the VALU are separate pipes of a SIMD, so the producers' index
// math and LayerNorm/activation arithmetic run beside the consumers' MFMAs instead of in front of them (v1 measured:
// matrix pipe busy 65 %, SQ_WAIT_ANY 45 % of wave cycles, 5 VALU + 4 SALU instructions per MFMA).
//
// Compile-time: MT (cout tile = 32 MT), NT (pixel tile = 128 NT; each consumer wave owns 32 NT pixels x all MT cout
// sub-tiles), SPAD (LDS stride between channel planes of the halo tile, >= S), KH x KW.
template <int MT, int NT, int SPAD, int KH, int KW, int CK = SDA_CONV_CK>
__global__ __launch_bounds__(512, ((CK == SDA_CONV_CK && NT == 1 && MT <= 3 && SPAD <= 392) ? 4 : 2)) void conv_igemm_ws_kernel(const sda_conv_desc d, const ConvGeom g) {
    constexpr int BM = MT * 32;
    constexpr int NTAPS = KH * KW;
    constexpr int NPOS = (SPAD + 255) / 256;
    constexpr int WSZ = NTAPS * CK * BM;     // floats of one weight slab
    constexpr int BUF = WSZ + CK * SPAD;     // floats of one stage buffer
    // wide (dwordx4) epilogue through a wave-private LDS slab: needs NT = 2 (64-pixel runs) and 8 KiB more LDS
    constexpr bool WIDE_EPI = (NT == 2) && (2 * BUF * 4 + 4 * 16 * 32 * NT * 4 <= 160 * 1024);
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int khalf = lane >> 5;
    const bool producer = __builtin_amdgcn_readfirstlane(wave) >= 4;

    // Persistent workgroups: gridDim.x (a multiple of 8) workgroups walk all g.grid tiles.  Hardware places workgroup
    // b on XCD b % 8; XCD x owns a contiguous range of tiles and its workgroups take them round-robin, so at any
    // moment one XCD's L2 serves neighbouring pixel tiles (shared halos) and the cout tiles of one pixel tile.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int tq = g.grid >> 3, tr_ = g.grid & 7;
    const int t_begin = xcd * tq + (xcd < tr_ ? xcd : tr_);
    const int t_end = t_begin + tq + (xcd < tr_ ? 1 : 0);

    if (producer) {
        // ------------------------------------------------ producer waves: HBM/L2 -> (fusions) -> LDS
        // Lean by construction: everything that does not depend on the channel is computed once per workgroup
        // (per-lane 32-bit offsets relative to a wave-uniform tile base, LayerNorm statistics, validity); per stage a
        // lane issues its loads against SGPR bases, applies <= 5 VALU per element and writes LDS at immediates.
        const int ptid = tid & 255;
        int gs = 0;                                    // running stage counter across tiles: buffer = gs & 1
        for (int tile = t_begin + slot; tile < t_end; tile += per_xcd) {
        int ct, n0, oy0, ox0;
        conv_decode_block(g, tile, ct, n0, oy0, ox0);
        const int co0 = ct * BM;
        // tile base (uniform): address of image n0, channel 0, pixel (0,0)
        const int ng0 = n0 + d.x_n_off;
        const int64_t nb0 = (int64_t)(ng0 / d.n_inner) * d.x_sn_outer + (int64_t)(ng0 % d.n_inner) * d.x_sn_inner;
        const float* xtile = d.x + nb0;
        unsigned poff[NPOS];
        int coff[NPOS], nimg[NPOS];
        bool pvalid[NPOS];
        float pmean[NPOS], prstd[NPOS];
        int64_t xabs[NPOS];
#pragma unroll
        for (int i = 0; i < NPOS; ++i) {
            const int pos = ptid + i * 256;
            poff[i] = 0; coff[i] = 0; nimg[i] = 0; pvalid[i] = false; pmean[i] = 0.f; prstd[i] = 1.f; xabs[i] = -1;
            if (pos < g.S) {
                const ConvPos ps = conv_decode_pos(d, g, pos, n0, oy0, ox0);
                xabs[i] = ps.xoff; coff[i] = (int)ps.coff; nimg[i] = ps.nimg;
                if (ps.xoff >= 0) {
                    pvalid[i] = true;
                    poff[i] = (unsigned)(ps.xoff - nb0);
                    if (d.ln_mean) {
                        pmean[i] = d.ln_mean[ps.stat];
                        prstd[i] = d.ln_rstd[ps.stat];
                    }
                }
            }
        }
        constexpr int ROW4 = BM / 4;
        constexpr int TOTAL4 = NTAPS * CK * ROW4;
        constexpr int NIT = (TOTAL4 + 255) / 256;
        unsigned woff[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int f = ptid + it * 256;
            const int row = f / ROW4, c4 = f - row * ROW4;
            const int tap = row / CK, ck = row - tap * CK;
            woff[it] = (unsigned)((tap * d.cin_pad + ck) * d.cout_pad + co0 + c4 * 4);
        }
        const bool fast = g.fast32 && (d.mod == nullptr || d.mod_sn == 0);

        auto produce = [&](int st, float* buf) {
            const int c0 = st * CK;
            // weight slab: [tap][ck][BM] <- w[tap][c0+ck][co0 .. co0+BM), straight into LDS (lane-linear image)
            const float* wst = d.w + (int64_t)c0 * d.cout_pad;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int f = ptid + it * 256;
                if (it + 1 < NIT || f < TOTAL4) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wst + woff[it]),
                                                     (__attribute__((address_space(3))) void*)(buf + (f - lane) * 4), 16, 0, 0);
                }
            }
            // halo tile with every loader-side fusion applied
            float v[NPOS][CK];
            if (fast && c0 + CK <= d.cx) {
                const float* xs = xtile + (int64_t)c0 * d.x_sc;
#pragma unroll
                for (int ck = 0; ck < CK; ++ck) {
                    const float* xc = xs + (int64_t)ck * d.x_sc;            // wave-uniform channel base
#pragma unroll
                    for (int i = 0; i < NPOS; ++i) v[i][ck] = xc[poff[i]];
                }
                if (d.ln_mean) {
#pragma unroll
                    for (int ck = 0; ck < CK; ++ck) {
                        const float m = d.mod ? d.mod[c0 + ck] : 0.f;       // shared modulation: scalar
#pragma unroll
                        for (int i = 0; i < NPOS; ++i) v[i][ck] = ((v[i][ck] + m) - pmean[i]) * prstd[i];
                    }
                } else if (d.mod) {
#pragma unroll
                    for (int ck = 0; ck < CK; ++ck) {
                        const float m = d.mod[c0 + ck];
#pragma unroll
                        for (int i = 0; i < NPOS; ++i) v[i][ck] += m;
                    }
                }
                // activation id is wave-uniform: branch once per stage, not once per element (SiLU inlined, rest generic)
                if (d.act_in == SDA_ACT_SILU) {
#pragma unroll
                    for (int ck = 0; ck < CK; ++ck)
#pragma unroll
                        for (int i = 0; i < NPOS; ++i) v[i][ck] = sda_act(SDA_ACT_SILU, v[i][ck]);
                } else if (d.act_in) {
#pragma unroll
                    for (int ck = 0; ck < CK; ++ck)
#pragma unroll
                        for (int i = 0; i < NPOS; ++i) v[i][ck] = sda_act(d.act_in, v[i][ck]);
                }
#pragma unroll
                for (int ck = 0; ck < CK; ++ck)
#pragma unroll
                    for (int i = 0; i < NPOS; ++i) v[i][ck] = pvalid[i] ? v[i][ck] : 0.f;
            } else {
#pragma unroll
                for (int i = 0; i < NPOS; ++i) {
                    ConvPos ps;
                    ps.xoff = xabs[i]; ps.coff = coff[i]; ps.stat = 0; ps.nimg = nimg[i];
#pragma unroll
                    for (int ck = 0; ck < CK; ++ck) v[i][ck] = conv_load_value(d, g, ps, c0 + ck, pmean[i], prstd[i]);
                }
            }
            float* s_in = buf + WSZ;
#pragma unroll
            for (int i = 0; i < NPOS; ++i) {
                const int pos = ptid + i * 256;
                if (pos < g.S) {
#pragma unroll
                    for (int ck = 0; ck < CK; ++ck) s_in[ck * SPAD + pos] = v[i][ck];
                }
            }
        };

        for (int st = 0; st < g.nstage; ++st, ++gs) {
            if (!(SDA_DBG(g, 2) && gs > 0)) produce(st, smem + (gs & 1) * BUF);
            __syncthreads();
        }
        }   // tiles
        __syncthreads();                               // pairs with the consumers' final barrier
    } else {
        // ------------------------------------------------ consumer waves: ds_read + MFMA only
        __builtin_amdgcn_s_setprio(2);                 // the matrix-pipe feeders outrank the loaders on their SIMD
        int pixbase[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q) pixbase[q] = conv_pix_lds_base(d, g, (wave * NT + q) * 32 + l31);
        int gs = 0;
        __syncthreads();                               // stage 0 of the first tile has landed
        for (int tile = t_begin + slot; tile < t_end; tile += per_xcd) {
        f32x16 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

        for (int st = 0; st < g.nstage; ++st, ++gs) {
            const float* buf = smem + (gs & 1) * BUF;
            if SDA_DBG(g, 4) { __syncthreads(); continue; }
            const float* bw = buf + khalf * BM + l31;
            const float* bin = buf + WSZ + khalf * SPAD;
            // operands of one (tap, 8-channel chunk) live in registers; the next one's are fetched while this one's MFMAs run
            // (stages deeper than 8 channels walk their chunks tap by tap: the register cost does not grow with CK)
            constexpr int KC2 = CK >= 8 ? 4 : CK / 2, NCH = (CK / 2) / KC2, NSEQ = NTAPS * NCH;
            float av[2][KC2][MT], bv[2][KC2][NT];
            auto fetch = [&](int u, int slot_) {
                const int tap = u % NTAPS, kc = u / NTAPS;
                const int dy = tap / KW, dx = tap - dy * KW;
                const int toff = dy * g.in_cols + dx;
#pragma unroll
                for (int k2 = 0; k2 < KC2; ++k2) {
                    const int kk = kc * KC2 + k2;
#pragma unroll
                    for (int q = 0; q < NT; ++q) bv[slot_][k2][q] = bin[pixbase[q] + toff + 2 * kk * SPAD];
#pragma unroll
                    for (int m = 0; m < MT; ++m) av[slot_][k2][m] = bw[(tap * CK + 2 * kk) * BM + m * 32];
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int u = 0; u < NSEQ; ++u) {
                if (u + 1 < NSEQ) fetch(u + 1, (u + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);       // keep the next chunk's ds_reads ahead of this chunk's MFMAs
#pragma unroll
                for (int k2 = 0; k2 < KC2; ++k2)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q)
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u & 1][k2][m], bv[u & 1][k2][q], acc[m][q], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                           // producers may now refill this buffer; next stage is ready
        }

        // epilogue of this tile (the producers are already staging the next tile's second stage)
        int ct, n0, oy0, ox0;
        conv_decode_block(g, tile, ct, n0, oy0, ox0);
        const int co0 = ct * BM;
        if SDA_DBG(g, 8) {                              // ablation: no epilogue memory traffic at all
            float keep = 0.f;
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) keep += acc[m][q][r];
            if (keep == 123.456f) d.out[0] = keep;
            continue;
        }
        if (WIDE_EPI && g.wide_out) {
            // Store-issue is what an MFMA epilogue pays for (~400+ cycles per store instruction with every CU storing),
            // so each 16-cout x 64-pixel fragment is transposed through a wave-private LDS slab and leaves as
            // global_store_dwordx4 of 4 consecutive pixels per lane: 24 stores per wave and tile instead of 96.
            float* slab = smem + 2 * BUF + wave * (16 * 32 * NT);
            const int64_t hw_w = g.o_sc;
            const int px4 = (lane & 15) * 4;                       // 16 lanes cover the wave's 32*NT-pixel run (NT = 2)
            const int64_t gb = conv_pix_out_base(d, g, wave * 32 * NT + px4, n0, oy0, ox0);
            const bool gvalid = gb >= 0;
            const int64_t gbs = gvalid ? gb : 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int q = 0; q < NT; ++q)
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr)
                            slab[((rr & 3) + 8 * (rr >> 2) + 4 * khalf) * (32 * NT) + q * 32 + l31] = acc[m][q][8 * h + rr];
#pragma unroll
                    for (int it0 = 0; it0 < 4; it0 += 2) {
                        f32x4 val[2], zz[2], rs[2];
                        float bia[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int lr = (it0 + u) * 4 + (lane >> 4);
                            const int co = co0 + m * 32 + 16 * h + lr;
                            const bool ok = gvalid && co < d.cout;
                            const int64_t off = gbs + (int64_t)(ok ? co : 0) * hw_w;
                            val[u] = *reinterpret_cast<const f32x4*>(slab + lr * (32 * NT) + px4);
                            bia[u] = (d.bias && ok) ? d.bias[co] : 0.f;
                            if (d.dact_z) zz[u] = *reinterpret_cast<const f32x4*>(d.dact_z + off);
                            if (d.res) rs[u] = *reinterpret_cast<const f32x4*>(d.res + off);
                        }
                        if (d.dact_z) {                  // activation id is wave-uniform
                            if (d.act_d == SDA_ACT_SILU) {
#pragma unroll
                                for (int u = 0; u < 2; ++u)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) zz[u][e] = sda_dact(SDA_ACT_SILU, zz[u][e]);
                            } else {
#pragma unroll
                                for (int u = 0; u < 2; ++u)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) zz[u][e] = sda_dact(d.act_d, zz[u][e]);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int lr = (it0 + u) * 4 + (lane >> 4);
                            const int co = co0 + m * 32 + 16 * h + lr;
                            f32x4 v = val[u] + bia[u];
                            if (d.dact_z) v *= zz[u];
                            if (d.res) v += rs[u];
                            if (gvalid && co < d.cout) *reinterpret_cast<f32x4*>(d.out + gbs + (int64_t)co * hw_w) = v;
                        }
                    }
                }
            }
            continue;
        }
        // All loads of a 16-register fragment (bias / act'(z) operand / residual) are issued before its first store:
        // `out` may alias nothing here, but the compiler cannot know, and a load placed after a store waits for it.
        const int64_t hw_o = g.o_sc;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int64_t obase = conv_pix_out_base(d, g, (wave * NT + q) * 32 + l31, n0, oy0, ox0);
            const bool pvalid = obase >= 0;
            const int64_t ob = pvalid ? obase : 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int cb = co0 + m * 32 + 4 * khalf;          // co = cb + (r & 3) + 8 * (r >> 2)
                float bia[16], zv[16], rv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = cb + (r & 3) + 8 * (r >> 2);
                    const bool ok = pvalid && co < d.cout;
                    const int64_t off = ob + (int64_t)(ok ? co : 0) * hw_o;
                    bia[r] = (d.bias && ok) ? d.bias[co] : 0.f;
                    zv[r] = d.dact_z ? d.dact_z[off] : 0.f;
                    rv[r] = d.res ? d.res[off] : 0.f;
                }
                if (d.dact_z) {
                    if (d.act_d == SDA_ACT_SILU) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) zv[r] = sda_dact(SDA_ACT_SILU, zv[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) zv[r] = sda_dact(d.act_d, zv[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) zv[r] = 1.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = cb + (r & 3) + 8 * (r >> 2);
                    const float v = (acc[m][q][r] + bia[r]) * zv[r] + rv[r];
                    if (pvalid && co < d.cout) d.out[ob + (int64_t)co * hw_o] = v;
                }
            }
        }
        }   // tiles
    }
}

template <int MT, int NT, int SPAD, int KH, int KW, int CK = SDA_CONV_CK>
static int conv_launch_ws(const sda_conv_desc* d, const ConvGeom& g, hipStream_t stream) {
    constexpr int BUF = KH * KW * CK * MT * 32 + CK * SPAD;
    constexpr int slab = (NT == 2 && 2 * BUF * 4 + 4 * 16 * 32 * NT * 4 <= 160 * 1024) ? 4 * 16 * 32 * NT * 4 : 0;
    constexpr int lds = 2 * BUF * 4 + slab;
    static_assert(lds <= 160 * 1024, "stage buffers exceed the LDS");
    auto kern = conv_igemm_ws_kernel<MT, NT, SPAD, KH, KW, CK>;
    static bool attr_set[SDA_MAX_DEVICES];           // per device: a process may use several GPUs
    if (lds > 48 * 1024) {
        const int rc = sda_raise_dyn_lds(reinterpret_cast<const void*>(kern), lds, attr_set);
        if (rc != SDA_OK) return rc;
    }
    // persistent grid: as many workgroups as stay co-resident
    const int cus = sda_cu_count();
    if (!cus) return SDA_E_BADARG;
    const int per_cu = (CK == SDA_CONV_CK && NT == 1 && 2 * lds <= 160 * 1024 && MT <= 3 && SPAD <= 392) ? 2 : 1;
    int grid = cus * per_cu;
    grid -= grid % 8;
    const int need = (g.grid + 7) / 8 * 8;            // never launch more workgroups than there are tiles (x8 for the XCD map)
    if (grid > need) grid = need;
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), (size_t)lds, stream, *d, g);
    return sda_launch_status();
}

// NT = 2 (256-pixel tiles, one workgroup per CU): SPAD classes 400 / 520 / 1280
// NT = 1 (128-pixel tiles, two workgroups per CU when LDS allows): SPAD classes 272 / 392 / 1024
template <int MT, int KH, int KW>
static int conv_launch_ws_s(const sda_conv_desc* d, const ConvGeom& g, hipStream_t stream) {
    if (g.tn * g.tr * g.tw == 128) {
        if (g.S <= 272) return conv_launch_ws<MT, 1, 272, KH, KW>(d, g, stream);
        if (g.S <= 392) return conv_launch_ws<MT, 1, 392, KH, KW>(d, g, stream);
        if (MT <= 3 && g.S <= 1024) return conv_launch_ws<MT, 1, 1024, KH, KW>(d, g, stream);
        return SDA_E_LDS;
    }
    if (g.S <= 400) return conv_launch_ws<MT, 2, 400, KH, KW>(d, g, stream);
    if (g.S <= 520) return conv_launch_ws<MT, 2, 520, KH, KW>(d, g, stream);
    if (g.S <= 1280) return conv_launch_ws<MT, 2, 1280, KH, KW>(d, g, stream);
    return SDA_E_LDS;
}

// the big instantiation families live in their own translation units (parts 1-3)
int sda_conv_ws_k33_lo(const sda_conv_desc* d, const ConvGeom& g, hipStream_t stream);    // 3x3, mt 1..2
int sda_conv_ws_k33_hi(const sda_conv_desc* d, const ConvGeom& g, hipStream_t stream);    // 3x3, mt 3..4
int sda_conv_ws_k13(const sda_conv_desc* d, const ConvGeom& g, hipStream_t stream);       // 1x3, mt 1..4
#if SDA_CONV_PART == 1
int sda_conv_ws_k33_lo(const sda_conv_desc* d, const ConvGeom& g, hipStream_t stream) {
    return d->mt == 1 ? conv_launch_ws_s<1, 3, 3>(d, g, stream) : conv_launch_ws_s<2, 3, 3>(d, g, stream);
}
#elif SDA_CONV_PART == 2
int sda_conv_ws_k33_hi(const sda_conv_desc* d, const ConvGeom& g, hipStream_t stream) {
    return d->mt == 3 ? conv_launch_ws_s<3, 3, 3>(d, g, stream) : conv_launch_ws_s<4, 3, 3>(d, g, stream);
}
#elif SDA_CONV_PART == 3
int sda_conv_ws_k13(const sda_conv_desc* d, const ConvGeom& g, hipStream_t stream) {
    switch (d->mt) {
        case 1: return conv_launch_ws_s<1, 1, 3>(d, g, stream);
        case 2: return conv_launch_ws_s<2, 1, 3>(d, g, stream);
        case 3: return conv_launch_ws_s<3, 1, 3>(d, g, stream);
        default: return conv_launch_ws_s<4, 1, 3>(d, g, stream);
    }
}
#endif

#if SDA_CONV_PART == 0

template <int MT, int NPOS>
static int conv_launch_t(const sda_conv_desc* d, const ConvGeom& g, hipStream_t stream) {
    auto kern = conv_igemm_kernel<MT, NPOS>;
    if (g.lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)g.lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(g.grid), dim3(SDA_CONV_THREADS), (size_t)g.lds_bytes, stream, *d, g);
    return sda_launch_status();
}

template <int MT>
static int conv_launch_m(const sda_conv_desc* d, const ConvGeom& g, hipStream_t stream) {
    int npos = (g.S + SDA_CONV_THREADS - 1) / SDA_CONV_THREADS;
    if (npos <= 2) return conv_launch_t<MT, 2>(d, g, stream);
    return conv_launch_t<MT, SDA_CONV_MAXPOS>(d, g, stream);
}

// Winograd F(2x2,3x3) path (conv_wino.hip)
struct WinoGeom;
int sda_wino_try(const sda_conv_desc* d, hipStream_t stream);   // SDA_E_UNSUPPORTED -> use the direct kernel
int sda_wino4_try(const sda_conv_desc* d, hipStream_t stream);  // one-wave-per-SIMD Winograd (conv_wino4.hip)
int sda_small1d_try(const sda_conv_desc* d, hipStream_t stream); // small 1-D layers: one round trip per launch (conv_small1d.hip)
int sda_few_try(const sda_conv_desc* d, hipStream_t stream);     // 3 x 3, <= 16 output channels (conv_few.hip)

extern "C" int sda_conv_igemm(const sda_conv_desc* d, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (d && d->w_wino4) {
        const int rc4 = sda_wino4_try(d, s);
        if (rc4 != SDA_E_UNSUPPORTED) return rc4;
    }
    // (pooled output exists in the w_wino4 kernel only: the caller runs the plain launch + a pooling reader instead)
    if (d && (d->pool_h > 1 || d->pool_w > 1)) return SDA_E_UNSUPPORTED;
    if (d && d->w_wino) {
        const int rcw = sda_wino_try(d, s);
        if (rcw != SDA_E_UNSUPPORTED) return rcw;
    }
    if (d && d->kh == 1 && d->kw == 3) {
        const int rcs = sda_small1d_try(d, s);
        if (rcs != SDA_E_UNSUPPORTED) return rcs;
    }
    if (d && d->kh == 3 && d->kw == 3 && d->cout <= 16) {
        const int rcf = sda_few_try(d, s);
        if (rcf != SDA_E_UNSUPPORTED) return rcf;
    }
    static const bool force_v1 = getenv("SDA_CONV_V1") != nullptr;
    const bool parity_shape = d && d->mt == 3 && d->kh >= 1 && d->kh <= 2 && d->kw >= 1 && d->kw <= 2;
    if (!force_v1 && d && ((d->kh == 3 && d->kw == 3) || (d->kh == 1 && d->kw == 3) || parity_shape)) {
        ConvGeom g2;
        // 256-pixel tiles (one workgroup per CU) unless the problem is too small to give every CU a tile, or forced
        static const int nt_env = getenv("SDA_CONV_NT") ? atoi(getenv("SDA_CONV_NT")) : 0;
        int rc2 = nt_env == 1 ? SDA_E_LDS : conv_plan(d, &g2, 256, 1280, false);
        if (nt_env != 2 && (rc2 != SDA_OK || g2.grid < 256)) rc2 = conv_plan(d, &g2, 128, 1024, false);
        // 1-D nets with a tile or less per CU are latency-bound: every K-stage costs one exposed global-load round trip
        // (12 of the 21 us of a 64-channel, 64-pixel Lorenz layer), so they take 32-channel stages -- a quarter of the trips
        static const bool no_ck32 = getenv("SDA_CONV_CK32") != nullptr && atoi(getenv("SDA_CONV_CK32")) == 0;
        if (rc2 == SDA_OK && !no_ck32 && d->kh == 1 && d->kw == 3 && g2.grid <= 256 && g2.tn * g2.tr * g2.tw == 128 &&
            g2.S <= 392 && d->cin_pad % 32 == 0 && d->mt <= 2) {
            g2.nstage = d->cin_pad / 32;
            if (d->mt == 1) return g2.S <= 272 ? conv_launch_ws<1, 1, 272, 1, 3, 32>(d, g2, s) : conv_launch_ws<1, 1, 392, 1, 3, 32>(d, g2, s);
            return g2.S <= 272 ? conv_launch_ws<2, 1, 272, 1, 3, 32>(d, g2, s) : conv_launch_ws<2, 1, 392, 1, 3, 32>(d, g2, s);
        }
        // The parity classes of a stride-2 head's VJP (1 x 1 .. 2 x 2 taps): an 8-channel stage is only 1-4 taps of MFMAs
        // (1.5-6 k cycles) against one HBM round trip of the stage behind it (~4-5 k cycles, the pipeline is one stage deep):
        // they take 16-channel stages -- half the barriers, twice the multiply time per round trip.
        static const bool no_ck16 = getenv("SDA_CONV_CK16") != nullptr && atoi(getenv("SDA_CONV_CK16")) == 0;
        if (rc2 == SDA_OK && parity_shape && !no_ck16 && d->cin_pad % 16 == 0 && g2.tn * g2.tr * g2.tw == 256 && g2.S <= 520) {
            const bool s400 = g2.S <= 400;
            if (d->kh == 2 && d->kw == 2) {
                g2.nstage = d->cin_pad / 16;
                return s400 ? conv_launch_ws<3, 2, 400, 2, 2, 16>(d, g2, s) : conv_launch_ws<3, 2, 520, 2, 2, 16>(d, g2, s);
            }
            if (d->cin_pad % 32 == 0 && s400) {              // 1 and 2 taps: 32-channel stages (6-12 k cycles of MFMAs per round trip)
                g2.nstage = d->cin_pad / 32;
                if (d->kh == 1 && d->kw == 1) return conv_launch_ws<3, 2, 400, 1, 1, 32>(d, g2, s);
                if (d->kh == 1 && d->kw == 2) return conv_launch_ws<3, 2, 400, 1, 2, 32>(d, g2, s);
                return conv_launch_ws<3, 2, 400, 2, 1, 32>(d, g2, s);
            }
            g2.nstage = d->cin_pad / 16;
            if (d->kh == 1 && d->kw == 1) return s400 ? conv_launch_ws<3, 2, 400, 1, 1, 16>(d, g2, s) : conv_launch_ws<3, 2, 520, 1, 1, 16>(d, g2, s);
            if (d->kh == 1 && d->kw == 2) return s400 ? conv_launch_ws<3, 2, 400, 1, 2, 16>(d, g2, s) : conv_launch_ws<3, 2, 520, 1, 2, 16>(d, g2, s);
            return s400 ? conv_launch_ws<3, 2, 400, 2, 1, 16>(d, g2, s) : conv_launch_ws<3, 2, 520, 2, 1, 16>(d, g2, s);
        }
        if (rc2 == SDA_OK) {
            if (d->kw == 3) rc2 = d->kh == 3 ? (d->mt <= 2 ? sda_conv_ws_k33_lo(d, g2, s) : sda_conv_ws_k33_hi(d, g2, s)) : sda_conv_ws_k13(d, g2, s);
            else if (d->kh == 1) rc2 = d->kw == 1 ? conv_launch_ws_s<3, 1, 1>(d, g2, s) : conv_launch_ws_s<3, 1, 2>(d, g2, s);
            else rc2 = d->kw == 1 ? conv_launch_ws_s<3, 2, 1>(d, g2, s) : conv_launch_ws_s<3, 2, 2>(d, g2, s);
            if (rc2 != SDA_E_LDS) return rc2;
        } else if (rc2 != SDA_E_UNSUPPORTED && rc2 != SDA_E_LDS) {
            return rc2;
        }
    }
    ConvGeom g;
    int rc = conv_plan(d, &g);
    if (rc != SDA_OK) return rc;
    switch (d->mt) {
        case 1: return conv_launch_m<1>(d, g, s);
        case 2: return conv_launch_m<2>(d, g, s);
        case 3: return conv_launch_m<3>(d, g, s);
        default: return conv_launch_m<4>(d, g, s);
    }
}

// which kernel family sda_conv_igemm would serve this launch with: 2 one-wave-per-SIMD Winograd, 1 Winograd, 3 the small 1-D
// kernel (conv_small1d.hip), 4 the few-output-channel 3 x 3 kernel (conv_few.hip), 0 the direct implicit-GEMM kernels
struct Wino4Geom;
int sda_wino4_path(const sda_conv_desc* d);
int sda_wino_path(const sda_conv_desc* d);
int sda_small1d_path(const sda_conv_desc* d);
int sda_few_path(const sda_conv_desc* d);
extern "C" int sda_conv_igemm_path(const sda_conv_desc* d) {
    if (!d) return SDA_E_BADARG;
    if (d->w_wino4) {
        const int p4 = sda_wino4_path(d);
        if (p4) return p4 == 2 ? 5 : 2;
    }
    if (d->pool_h > 1 || d->pool_w > 1) return SDA_E_UNSUPPORTED;
    if (d->w_wino && sda_wino_path(d)) return 1;
    if (d->kh == 1 && d->kw == 3 && sda_small1d_path(d)) return 3;
    if (d->kh == 3 && d->kw == 3 && d->cout <= 16 && sda_few_path(d)) return 4;
    return 0;
}

extern "C" int64_t sda_conv_igemm_lds_bytes(const sda_conv_desc* d) {
    ConvGeom g;
    int rc = conv_plan(d, &g);
    return rc != SDA_OK ? (int64_t)rc : g.lds_bytes;
}

// ---------------------------------------------------------------- weight repack (device side, one-off per layer)

__global__ void pack_conv_weight_kernel(const float* __restrict__ w, int cout, int cin, int kh, int kw, int transpose,
                                        int cin_keep, float* __restrict__ dst, int k_pad, int m_pad) {
    const int64_t total = (int64_t)kh * kw * k_pad * m_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int mm = (int)(i % m_pad);
        int64_t r = i / m_pad;
        int kk = (int)(r % k_pad);
        int tap = (int)(r / k_pad);
        int dy = tap / kw, dx = tap % kw;
        float v = 0.f;
        if (!transpose) {
            int ci = kk, co = mm;
            if (ci < cin && co < cout) v = w[(((int64_t)co * cin + ci) * kh + dy) * kw + dx];
        } else {
            int co = kk, ci = mm;
            int sdy = kh - 1 - dy, sdx = kw - 1 - dx;
            if (ci < cin_keep && ci < cin && co < cout) v = w[(((int64_t)co * cin + ci) * kh + sdy) * kw + sdx];
        }
        dst[i] = v;
    }
}

extern "C" int sda_pack_conv_weight(const float* w, int cout, int cin, int kh, int kw, int transpose, int cin_keep,
                                    float* dst, int k_pad, int m_pad, void* stream) {
    if (!w || !dst || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || k_pad <= 0 || m_pad <= 0) return SDA_E_BADARG;
    int64_t total = (int64_t)kh * kw * k_pad * m_pad;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, cout, cin, kh, kw,
                       transpose, cin_keep, dst, k_pad, m_pad);
    return sda_launch_status();
}

#endif  // SDA_CONV_PART == 0
#endif  // !SDA_HOST_EMU

// ---------------------------------------------------------------- CPU emulator (tests only; libsda_emu.so)
#ifdef SDA_HOST_EMU
#include <vector>
// Replays conv_igemm_kernel<MT,*> on the host with HOST pointers: same planner, same index helpers, same staging
// order, same MFMA lane<->element maps (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D row = mfma32_row(r,l), col = l&31).
extern "C" int sda_conv_igemm_emulate(const sda_conv_desc* dp) {
    ConvGeom g;
    int rc = conv_plan(dp, &g);
    if (rc != SDA_OK) return rc;
    const sda_conv_desc& d = *dp;
    constexpr int CK = SDA_CONV_CK;
    const int BM = g.bm, MT = d.mt;
    std::vector<float> s_w((size_t)g.ntaps * CK * BM), s_in((size_t)CK * g.S);
    std::vector<float> acc((size_t)SDA_CONV_THREADS * MT * 16);
    for (int b = 0; b < g.grid; ++b) {
        int ct, n0, oy0, ox0;
        conv_decode_block(g, conv_logical_block(b, g.grid), ct, n0, oy0, ox0);
        const int co0 = ct * BM;
        std::fill(acc.begin(), acc.end(), 0.f);
        for (int st = 0; st < g.nstage; ++st) {
            const int c0 = st * CK;
            for (int tid = 0; tid < SDA_CONV_THREADS; ++tid) {
                const int ROW4 = BM / 4, total4 = g.ntaps * CK * ROW4;
                for (int f = tid; f < total4; f += SDA_CONV_THREADS) {
                    int row = f / ROW4, c4 = f - row * ROW4, tap = row / CK, ck = row - tap * CK;
                    const float* src = d.w + ((int64_t)tap * d.cin_pad + c0 + ck) * d.cout_pad + co0 + c4 * 4;
                    for (int q = 0; q < 4; ++q) s_w[(size_t)row * BM + c4 * 4 + q] = src[q];
                }
                for (int pos = tid; pos < g.S; pos += SDA_CONV_THREADS) {
                    ConvPos ps = conv_decode_pos(d, g, pos, n0, oy0, ox0);
                    float mean = 0.f, rstd = 1.f;
                    if (d.ln_mean && ps.xoff >= 0) { mean = d.ln_mean[ps.stat]; rstd = d.ln_rstd[ps.stat]; }
                    for (int ck = 0; ck < CK; ++ck) s_in[(size_t)ck * g.S + pos] = conv_load_value(d, g, ps, c0 + ck, mean, rstd);
                }
            }
            for (int wave = 0; wave < 4; ++wave) {
                int dy = 0, dx = 0;
                for (int tap = 0; tap < g.ntaps; ++tap) {
                    for (int k2 = 0; k2 < CK / 2; ++k2) {
                        for (int m = 0; m < MT; ++m) {
                            float A[32][2], B[2][32];
                            for (int lane = 0; lane < 64; ++lane) {
                                int l31 = lane & 31, kh_ = lane >> 5, ck = 2 * k2 + kh_;
                                int pix = wave * 32 + l31;
                                int toff = dy * g.in_cols + dx + conv_pix_lds_base(d, g, pix);
                                B[kh_][l31] = s_in[(size_t)ck * g.S + toff];
                                A[l31][kh_] = s_w[((size_t)tap * CK + ck) * BM + m * 32 + l31];
                            }
                            for (int lane = 0; lane < 64; ++lane)
                                for (int r = 0; r < 16; ++r) {
                                    int i = mfma32_row(r, lane), j = lane & 31;
                                    float& c = acc[((size_t)(wave * 64 + lane) * MT + m) * 16 + r];
                                    c = fmaf(A[i][0], B[0][j], c);
                                    c = fmaf(A[i][1], B[1][j], c);
                                }
                        }
                    }
                    if (++dx == d.kw) { dx = 0; ++dy; }
                }
            }
        }
        for (int tid = 0; tid < SDA_CONV_THREADS; ++tid) {
            int lane = tid & 63, wave = tid >> 6, pix = wave * 32 + (lane & 31);
            int64_t obase = conv_pix_out_base(d, g, pix, n0, oy0, ox0);
            for (int m = 0; m < MT; ++m)
                for (int r = 0; r < 16; ++r)
                    conv_epilogue_store(d, g, obase, co0 + m * 32 + mfma32_row(r, lane), acc[((size_t)tid * MT + m) * 16 + r]);
        }
    }
    return SDA_OK;
}

extern "C" void sda_pack_conv_weight_host(const float* w, int cout, int cin, int kh, int kw, int transpose, int cin_keep,
                                          float* dst, int k_pad, int m_pad) {
    const int64_t total = (int64_t)kh * kw * k_pad * m_pad;
    for (int64_t i = 0; i < total; ++i) {
        int mm = (int)(i % m_pad);
        int64_t r = i / m_pad;
        int kk = (int)(r % k_pad), tap = (int)(r / k_pad), dy = tap / kw, dx = tap % kw;
        float v = 0.f;
        if (!transpose) {
            if (kk < cin && mm < cout) v = w[(((int64_t)mm * cin + kk) * kh + dy) * kw + dx];
        } else {
            int sdy = kh - 1 - dy, sdx = kw - 1 - dx;
            if (mm < cin_keep && mm < cin && kk < cout) v = w[(((int64_t)kk * cin + mm) * kh + sdy) * kw + sdx];
        }
        dst[i] = v;
    }
}
#endif  // SDA_HOST_EMU
