// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores: 2.25x fewer multiplies than the direct implicit GEMM at
// fp32 round-off-level error (measured ~1e-6 relative), used for the stride-1 3x3 layers whose cout is a multiple of 96
// (every block / tail convolution of the reference Kolmogorov net and their backward-data forms).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//   U[p][ci][co] = (G g G^T)[xi][nu],  p = 4 xi + nu            (precomputed: sda_pack_conv_weight_wino)
//   V[p][ci][t]  = (B^T d B)[xi][nu]   for tile t              (computed by the producer waves while staging)
//   M[p][co][t]  = sum_ci U[p][ci][co] V[p][ci][t]              (16 independent GEMMs on v_mfma_f32_32x32x2_f32)
//
// Same machine as conv_igemm_ws_kernel (8 wavefronts: 4 MFMA consumers + 4 producers, two LDS stage buffers of 8 input
// channels, persistent XCD-aware tiles, LDS-DMA weight slabs, loader-side LayerNorm / modulation / activation /
// upsample / padding), with:
//   * workgroup tile = 32 Winograd tiles (128 output pixels) x 96 couts;
//   * producer thread (ck, t) loads its 4x4 patch, applies the loader fusions, transforms it in registers (32 adds) and
//     writes V[0..15][ck][t];
//   * consumer wave w owns the four positions p = 4w .. 4w+3 (xi = w): accumulators [4][3] 32x32 fragments = 192 VGPRs;
//   * epilogue: the nu-part of A^T . A is lane-local; the xi-part crosses waves through a 32 KiB LDS exchange buffer
//     (two workgroup barriers per 32-cout slab, mirrored by the producers so that barrier counts match), after which
//     every lane owns 2x2 outputs of one (cout, tile) and stores float2 rows (consecutive lanes = consecutive tiles).
#include "sda_common.hpp"
#include <stdlib.h>

#define WINO_CK 8
#define WINO_BM 96
#define WINO_T 32
#define WINO_MT 3

struct WinoGeom {
    int cin, hv, wv;
    int TX, TY;                     // output tiles per image (2x2 pixels each)
    int ttx, tty, ttn, ttx_shift, tty_shift;
    int tiles_x, tiles_y, tiles_n;  // workgroup tiles over (TX, TY, n)
    int n_ct, grid, nstage, debug;
    int hrows, hcols, sh;           // halo of one workgroup tile per channel: ttn x hrows x hcols = sh positions
};

static inline int wino_ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }
static int wino_pick(int extent, int budget) {
    for (int t = budget; t >= 1; t >>= 1) {
        long padded = (long)((extent + t - 1) / t) * t;
        if (padded * 4 <= (long)extent * 5) return t;
    }
    return 1;
}

// eligibility + geometry.  Returns SDA_E_UNSUPPORTED when the direct kernel must be used.
int sda_wino_plan(const sda_conv_desc* d, WinoGeom* g) {
    if (!d || !d->x || !d->w_wino || !d->out) return SDA_E_UNSUPPORTED;
    if (d->kh != 3 || d->kw != 3 || d->stride_h != 1 || d->stride_w != 1 || d->zins_h != 1 || d->zins_w != 1)
        return SDA_E_UNSUPPORTED;
    if (d->explicit_pad || d->out_sn || d->out_sc || d->out_sy || d->out_sx) return SDA_E_UNSUPPORTED;
    if (d->cctx > 0 || d->mt != WINO_MT || d->cout % WINO_BM || d->cout_pad != d->cout || d->cin_pad % WINO_CK)
        return SDA_E_UNSUPPORTED;
    if ((d->ho & 1) || (d->wo & 1) || d->ho != d->hs * d->up_h || d->wo != d->ws * d->up_w) return SDA_E_UNSUPPORTED;
    if (d->up_h > 2 || d->up_w > 2) return SDA_E_UNSUPPORTED;                        // the loader shifts by log2(up)
    if (d->mod && d->mod_sn != 0) return SDA_E_UNSUPPORTED;            // per-image modulation: direct path
    if ((reinterpret_cast<uintptr_t>(d->out) & 7) || (d->res && (reinterpret_cast<uintptr_t>(d->res) & 7)) ||
        (d->dact_z && (reinterpret_cast<uintptr_t>(d->dact_z) & 7)))
        return SDA_E_UNSUPPORTED;
    // 32-bit producer offsets relative to the tile's first image
    {
        auto ab = [](int64_t v) { return v < 0 ? -v : v; };
        int64_t span = 32 * (ab(d->x_sn_outer) + ab(d->x_sn_inner)) + (int64_t)d->hs * ab(d->x_sy) + (int64_t)d->ws * ab(d->x_sx);
        if (span >= (1LL << 30) || d->x_sn_outer < 0 || d->x_sn_inner < 0 || d->x_sy < 0 || d->x_sx < 0) return SDA_E_UNSUPPORTED;
        if (d->n_inner > 1 && d->x_sn_outer < d->x_sn_inner * (int64_t)(d->n_inner - 1)) return SDA_E_UNSUPPORTED;
    }
    g->cin = d->cx;
    g->hv = d->ho; g->wv = d->wo;
    g->TX = d->wo / 2; g->TY = d->ho / 2;
    g->ttx = wino_pick(g->TX, WINO_T);
    g->tty = wino_pick(g->TY, WINO_T / g->ttx);
    g->ttn = WINO_T / (g->ttx * g->tty);
    g->ttx_shift = wino_ilog2(g->ttx);
    g->tty_shift = wino_ilog2(g->tty);
    g->tiles_x = (g->TX + g->ttx - 1) / g->ttx;
    g->tiles_y = (g->TY + g->tty - 1) / g->tty;
    g->tiles_n = (d->n + g->ttn - 1) / g->ttn;
    g->n_ct = d->cout / WINO_BM;
    long total = (long)g->tiles_x * g->tiles_y * g->tiles_n * g->n_ct;
    if (total > 0x7fffffffL) return SDA_E_UNSUPPORTED;
    g->grid = (int)total;
    g->nstage = d->cin_pad / WINO_CK;
    g->hrows = 2 * g->tty + 2;
    g->hcols = 2 * g->ttx + 2;
    g->sh = g->ttn * g->hrows * g->hcols;
    if (g->sh > 288) return SDA_E_UNSUPPORTED;       // (degenerate 2x2-pixel images)
    { static const int dbg = sda_debug_env(); g->debug = dbg; }            // (0 in the product build)
    return SDA_OK;
}

__device__ inline void wino_decode_tile(const WinoGeom& g, int tile, int& ct, int& n0, int& ty0, int& tx0) {
    ct = tile % g.n_ct;
    int pt = tile / g.n_ct;
    int txi = pt % g.tiles_x;
    int r = pt / g.tiles_x;
    int tyi = r % g.tiles_y;
    int tni = r / g.tiles_y;
    n0 = tni * g.ttn; ty0 = tyi * g.tty; tx0 = txi * g.ttx;
}

// tile slot t (0..31) of a workgroup tile -> image / tile coordinates; false if outside the tensor
__device__ inline bool wino_slot(const sda_conv_desc& d, const WinoGeom& g, int t, int n0, int ty0, int tx0, int& n, int& ty,
                                 int& tx) {
    const int ix = t & (g.ttx - 1);
    const int iy = (t >> g.ttx_shift) & (g.tty - 1);
    const int in = t >> (g.ttx_shift + g.tty_shift);
    n = n0 + in; ty = ty0 + iy; tx = tx0 + ix;
    return n < d.n && ty < g.TY && tx < g.TX;
}

__device__ inline int conv_wrap_i(int v, int m) { v %= m; return v < 0 ? v + m : v; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NH: halo slots per producer lane, ceil(2 sh / 64) -- 6 covers the usual 8 x 4-tile workgroup tile (sh = 180), 9 the maximum (288)
template <int NH>
__global__ __launch_bounds__(512, 2) void conv_wino_kernel(const sda_conv_desc d, const WinoGeom g) {
    constexpr int CK = WINO_CK, BM = WINO_BM, MT = WINO_MT;
    // LDS carries only V (the transformed input patches).  U is read by exactly one wave each (wave w owns positions
    // 4w..4w+3), so staging it in LDS would buy no reuse: the consumers stream their U fragments straight from L2 into
    // registers as one dwordx4 per (position, m-tile) -- U is packed [stage][p][khalf][cout][k2] for that purpose.
    constexpr int VSZ = 16 * CK * WINO_T;        //  4096 floats: V slab of one stage, layout [p][khalf][t][k2]
    constexpr int BUF = VSZ;                     // 16 KiB
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const xch = smem + 2 * BUF;           // [xi 4][j 2][co 32][t 32] exchange buffer (32 KiB)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, khalf = lane >> 5;
    const bool producer = __builtin_amdgcn_readfirstlane(wave) >= 4;

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int tq = g.grid >> 3, tr_ = g.grid & 7;
    const int t_begin = xcd * tq + (xcd < tr_ ? xcd : tr_);
    const int t_end = t_begin + tq + (xcd < tr_ ? 1 : 0);
    const int EPI_BARRIERS = 2 * MT;

    if (producer) {
        const int ptid = tid - 256;
        const int t = ptid & 31;
        const int ck = 2 * (__builtin_amdgcn_readfirstlane(wave) - 4) + khalf;     // = ptid >> 5
        int pb = 0;                                  // barriers executed so far
        int gs = 0, tile_idx = 0;
        int t2 = 0, s2 = 0;                          // tile and stage-in-tile of global stage gs - 2
        // halo slot e = lane + 64 i of this wave's two channels -> (channel select, image in tile, halo row, halo column):
        // tile-independent, so the divisions are paid once per kernel, not once per tile
        int hpk[NH];
        bool hinb[NH];
        unsigned hsel = 0;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int e = lane + 64 * i;
            hinb[i] = e < 2 * g.sh;
            const int chsel = e >= g.sh ? 1 : 0;
            const int hp = e - chsel * g.sh;
            const int plane = g.hrows * g.hcols;
            const int in = hp / plane;
            const int rem = hp - in * plane;
            const int hy = rem / g.hcols, hx = rem - hy * g.hcols;
            hpk[i] = (in << 20) | (hy << 10) | hx;
            hsel |= (unsigned)chsel << i;
        }
        const int up_sh_h = d.up_h == 2 ? 1 : 0, up_sh_w = d.up_w == 2 ? 1 : 0;      // up factors are 1 or 2 (plan)
        for (int tile = t_begin + slot; tile < t_end; tile += per_xcd, ++tile_idx) {
            int ct, n0, ty0, tx0;
            wino_decode_tile(g, tile, ct, n0, ty0, tx0);
            const int co0 = ct * BM;
            const int ng0 = n0 + d.x_n_off;
            const int64_t nb0 = (int64_t)(ng0 / d.n_inner) * d.x_sn_outer + (int64_t)(ng0 % d.n_inner) * d.x_sn_inner;
            const float* xtile = d.x + nb0;
            // Halo of the workgroup tile, two channels per producer wave (ck = 2 pw, 2 pw + 1), staged through a
            // wave-private LDS area: every input pixel is fetched once per channel with row-contiguous loads (a patch-wise
            // gather would fetch it four times with 8-byte lane strides), the loader fusions run once per pixel, and each
            // lane then reads its 4x4 patch back from LDS.
            const int pw = __builtin_amdgcn_readfirstlane(wave) - 4;    // scalar: channel bases / mod values become SGPR work
            float* priv = smem + 2 * BUF + 4 * 2 * 32 * WINO_T + pw * (2 * 288);
            unsigned hoff[NH], hoffx[NH];                  // hoffx: + one channel stride for the wave's second channel
            float hmean[NH], hrstd[NH];
            bool hlive[NH];                                // per-tile lane mask: the position carries data
            unsigned hmask = 0;
            const int q0 = ng0 / d.n_inner, r0 = ng0 - q0 * d.n_inner;      // wave-uniform: image n0 of the tile
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                hoff[i] = 0; hoffx[i] = 0; hmean[i] = 0.f; hrstd[i] = 1.f; hlive[i] = false;
                if (hinb[i]) {
                    const int in = hpk[i] >> 20, hy = (hpk[i] >> 10) & 1023, hx = hpk[i] & 1023;
                    const int n = n0 + in;
                    int vy = 2 * ty0 - 1 + hy, vx = 2 * tx0 - 1 + hx;
                    bool ok = n < d.n;
                    if (d.circular) {                      // |overshoot| is at most a tile: compare-and-add instead of %
                        if (vy < 0) vy += g.hv;
                        if (vx < 0) vx += g.wv;
                        while (vy >= g.hv) vy -= g.hv;
                        while (vx >= g.wv) vx -= g.wv;
                    } else {
                        ok = ok && vy >= 0 && vy < g.hv && vx >= 0 && vx < g.wv;
                    }
                    if (ok) {
                        const int sy = vy >> up_sh_h, sx = vx >> up_sh_w;
                        int q = q0, r = r0 + in;           // (n + x_n_off) / n_inner and % n_inner, carried from image n0
                        while (r >= d.n_inner) { r -= d.n_inner; ++q; }
                        const int64_t nb = (int64_t)q * d.x_sn_outer + (int64_t)r * d.x_sn_inner;
                        hoff[i] = (unsigned)(nb - nb0 + (int64_t)sy * d.x_sy + (int64_t)sx * d.x_sx);
                        hoffx[i] = hoff[i] + (((hsel >> i) & 1u) ? (unsigned)d.x_sc : 0u);
                        hlive[i] = true;
                        hmask |= 1u << i;
                        if (d.ln_mean) {
                            const int64_t st = (int64_t)n * d.hs * d.ws + (int64_t)sy * d.ws + sx;
                            hmean[i] = d.ln_mean[st];
                            hrstd[i] = d.ln_rstd[st];
                        }
                    }
                }
            }
            // every existing slot of this wave carries data (always, for circular padding away from the batch end): the
            // per-slot zero-select can be skipped for the whole tile
            bool some_dead = false;
#pragma unroll
            for (int i = 0; i < NH; ++i) some_dead |= hinb[i] && !hlive[i];
            const bool all_live = __builtin_amdgcn_readfirstlane(__any((int)some_dead)) == 0;
            // where this lane's patch starts inside the wave-private halo
            int pbase;
            {
                const int ix = t & (g.ttx - 1);
                const int iy = (t >> g.ttx_shift) & (g.tty - 1);
                const int in = t >> (g.ttx_shift + g.tty_shift);
                pbase = (lane >> 5) * g.sh + (in * g.hrows + 2 * iy) * g.hcols + 2 * ix;
            }
            for (int st = 0; st < g.nstage; ++st, ++gs) {
                // this ring slot was last read in global stage gs-2: wait for the barrier the consumers pass after it
                if (gs >= 2) {
                    const int need = 1 + (gs - 2) + t2 * EPI_BARRIERS;      // index of that barrier; t2 = (gs - 2) / nstage
                    while (pb <= need) { __syncthreads(); ++pb; }
                    if (++s2 == g.nstage) { s2 = 0; ++t2; }                  // (kept incrementally: no division per stage)
                }
                float* buf = smem + (gs & 1) * BUF;
                const int c0 = st * CK;
                if (!(SDA_DBG(g, 2) && gs > 0)) {
                    if SDA_DBG(g, 32) continue;
                    const int ca = c0 + 2 * pw, cb = ca + 1;                 // this wave's two channels
                    // channel offsets relative to the tile base (padded channels >= cin read channel 0 and are zeroed)
                    const unsigned offa = ca < g.cin ? (unsigned)((int64_t)ca * d.x_sc) : 0u;
                    const unsigned offb = cb < g.cin ? (unsigned)((int64_t)cb * d.x_sc) : 0u;
                    float hv_[NH];
                    const bool full = cb < g.cin;              // both channels real (always, unless cin is not a multiple of 8)
                    if (full) {
                        const float* xa = xtile + (int64_t)ca * d.x_sc;        // wave-uniform base: no per-stage address VALU
#pragma unroll
                        for (int i = 0; i < NH; ++i) hv_[i] = xa[hoffx[i]];
                    } else {
#pragma unroll
                        for (int i = 0; i < NH; ++i) hv_[i] = xtile[hoff[i] + (((hsel >> i) & 1u) ? offb : offa)];
                    }
                    {
                        // (each fusion is a wave-uniform branch: a launch without it does not pay its VALU slots)
                        if (d.mod) {
                            const float ma = ca < g.cin ? d.mod[ca] : 0.f;
                            const float mb = cb < g.cin ? d.mod[cb] : 0.f;
#pragma unroll
                            for (int i = 0; i < NH; ++i) hv_[i] += ((hsel >> i) & 1u) ? mb : ma;
                        }
                        if (d.ln_mean) {
#pragma unroll
                            for (int i = 0; i < NH; ++i) hv_[i] = (hv_[i] - hmean[i]) * hrstd[i];
                        }
                        if (d.act_in == SDA_ACT_SILU) {
#pragma unroll
                            for (int i = 0; i < NH; ++i) hv_[i] = sda_act(SDA_ACT_SILU, hv_[i]);
                        } else if (d.act_in) {
#pragma unroll
                            for (int i = 0; i < NH; ++i) hv_[i] = sda_act(d.act_in, hv_[i]);
                        }
                        // padding / out-of-range positions (and, in a partial last stage, padded channels) stage zeros
                        // (slots beyond 2 sh still fall inside this wave's 2 x 288-float private area and are never read back:
                        // the stores need no bounds predicate -- 9 exec-mask round trips and 18 SGPRs less per stage)
                        static_assert(NH * 64 <= 2 * 288, "halo slots must stay inside the wave-private area");
                        if (full && all_live) {
#pragma unroll
                            for (int i = 0; i < NH; ++i) priv[lane + 64 * i] = hv_[i];
                        } else if (full) {
#pragma unroll
                            for (int i = 0; i < NH; ++i) priv[lane + 64 * i] = hlive[i] ? hv_[i] : 0.f;
                        } else {
                            const unsigned livem = ca < g.cin ? hmask & ~hsel : 0u;
#pragma unroll
                            for (int i = 0; i < NH; ++i) priv[lane + 64 * i] = ((livem >> i) & 1u) ? hv_[i] : 0.f;
                        }
                    }
                    // patch rows as float2 pairs (pbase and hcols are even: two 8-byte LDS reads per row); B^T d B on packed
                    // pairs -- v_pk_add_f32 does two of the 32 adds per instruction
                    f32x2 lo[4], hi[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const f32x2* row = reinterpret_cast<const f32x2*>(priv + pbase + a * g.hcols);
                        lo[a] = row[0]; hi[a] = row[1];
                    }
                    // rows: u0 = v0 - v2, u1 = v1 + v2, u2 = v2 - v1, u3 = v1 - v3
                    const f32x2 ul[4] = {lo[0] - lo[2], lo[1] + lo[2], lo[2] - lo[1], lo[1] - lo[3]};
                    const f32x2 uh[4] = {hi[0] - hi[2], hi[1] + hi[2], hi[2] - hi[1], hi[1] - hi[3]};
                    // V[p][khalf = ck & 1][t][k2 = ck >> 1]
                    float* s_v = buf + ((ck & 1) * WINO_T + t) * 4 + (ck >> 1);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        // columns: (w0, w1) = (u0 - u2, u1 + u2);  (w2, w3) = (u2 - u1, u1 - u3)
                        const f32x2 w01 = ul[a] + f32x2{-uh[a][0], uh[a][0]};
                        const f32x2 w23 = f32x2{-ul[a][1], ul[a][1]} + f32x2{uh[a][0], -uh[a][1]};
                        s_v[(a * 4 + 0) * (2 * WINO_T * 4)] = w01[0];
                        s_v[(a * 4 + 1) * (2 * WINO_T * 4)] = w01[1];
                        s_v[(a * 4 + 2) * (2 * WINO_T * 4)] = w23[0];
                        s_v[(a * 4 + 3) * (2 * WINO_T * 4)] = w23[1];
                    }
                }
            }
        }
        // drain: match every remaining consumer barrier
        const int total = 1 + gs + tile_idx * EPI_BARRIERS;
        while (pb < total) { __syncthreads(); ++pb; }
    } else {
        __builtin_amdgcn_s_setprio(2);
        int gs = 0;
        __syncthreads();                                   // barrier 0: the first stage has landed
        for (int tile = t_begin + slot; tile < t_end; tile += per_xcd) {
            int ct, n0, ty0, tx0;
            wino_decode_tile(g, tile, ct, n0, ty0, tx0);
            const int co0 = ct * BM;
            f32x16 acc[4][MT];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[p][m][r] = 0.f;

            // U fragments: [stage][p][khalf][cout][k2] (one float4 per lane, m-tile and position), streamed L2 -> VGPRs one
            // position ahead of their MFMAs -- across stage boundaries too, since weights do not depend on the barrier.
            const float* ubase = d.w_wino + ((int64_t)(wave * 4 * 2 + khalf) * d.cout_pad + co0 + l31) * 4;
            const int64_t ustage = (int64_t)(16 * 2 * 4) * d.cout_pad;
            f32x4 av[2][MT], bv[2];
            auto fetch_a = [&](int st_, int pl, int s_) {
                const float* u = ubase + st_ * ustage + ((int64_t)pl * 2 * d.cout_pad) * 4;
#pragma unroll
                for (int m = 0; m < MT; ++m) av[s_][m] = *reinterpret_cast<const f32x4*>(u + m * 32 * 4);
            };
            if (!SDA_DBG(g, 4)) fetch_a(0, 0, 0);
            for (int st = 0; st < g.nstage; ++st, ++gs) {
                const float* buf = smem + (gs & 1) * BUF;
                if SDA_DBG(g, 4) { __syncthreads(); continue; }
                const float* vst = buf + ((wave * 4 * 2 + khalf) * WINO_T + l31) * 4;
                bv[0] = *reinterpret_cast<const f32x4*>(vst);
#pragma unroll
                for (int pl = 0; pl < 4; ++pl) {
                    if (pl + 1 < 4) {
                        bv[(pl + 1) & 1] = *reinterpret_cast<const f32x4*>(vst + (pl + 1) * (2 * WINO_T * 4));
                        fetch_a(st, pl + 1, (pl + 1) & 1);
                    } else if (st + 1 < g.nstage) {
                        fetch_a(st + 1, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k2 = 0; k2 < CK / 2; ++k2)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            acc[pl][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[pl & 1][m][k2], bv[pl & 1][k2], acc[pl][m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
            }

            // ---------------- epilogue: A^T M A, nu in registers, xi through the LDS exchange buffer
            int n, ty, tx;
            const bool tv = wino_slot(d, g, l31, n0, ty0, tx0, n, ty, tx);
            const int hw_o = d.ho * d.wo;
            const int64_t pix = tv ? ((int64_t)n * d.cout * d.ho + 2 * ty) * d.wo + 2 * tx : 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (!SDA_DBG(g, 8)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * khalf;
                        const float m0 = acc[0][m][r], m1 = acc[1][m][r], m2 = acc[2][m][r], m3 = acc[3][m][r];
                        xch[((wave * 2 + 0) * 32 + row) * WINO_T + l31] = (m0 + m1) + m2;
                        xch[((wave * 2 + 1) * 32 + row) * WINO_T + l31] = (m1 - m2) - m3;
                    }
                }
                __syncthreads();
                if (!SDA_DBG(g, 8)) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = wave * 8 + it * 2 + khalf;
                        const int co = co0 + m * 32 + row;
                        float z[4][2];
#pragma unroll
                        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
                            for (int j = 0; j < 2; ++j) z[xi][j] = xch[((xi * 2 + j) * 32 + row) * WINO_T + l31];
                        const float bias = d.bias ? d.bias[co] : 0.f;
                        f32x2 y0, y1;
                        y0[0] = (z[0][0] + z[1][0]) + z[2][0] + bias; y0[1] = (z[0][1] + z[1][1]) + z[2][1] + bias;
                        y1[0] = (z[1][0] - z[2][0]) - z[3][0] + bias; y1[1] = (z[1][1] - z[2][1]) - z[3][1] + bias;
                        const int64_t off = pix + (int64_t)co * hw_o;
                        if (tv) {
                            if (d.dact_z) {
                                const f32x2 q0 = *reinterpret_cast<const f32x2*>(d.dact_z + off);
                                const f32x2 q1 = *reinterpret_cast<const f32x2*>(d.dact_z + off + d.wo);
                                if (d.act_d == SDA_ACT_SILU) {
                                    y0[0] *= sda_dact(SDA_ACT_SILU, q0[0]); y0[1] *= sda_dact(SDA_ACT_SILU, q0[1]);
                                    y1[0] *= sda_dact(SDA_ACT_SILU, q1[0]); y1[1] *= sda_dact(SDA_ACT_SILU, q1[1]);
                                } else {
                                    y0[0] *= sda_dact(d.act_d, q0[0]); y0[1] *= sda_dact(d.act_d, q0[1]);
                                    y1[0] *= sda_dact(d.act_d, q1[0]); y1[1] *= sda_dact(d.act_d, q1[1]);
                                }
                            }
                            if (d.res) {
                                y0 += *reinterpret_cast<const f32x2*>(d.res + off);
                                y1 += *reinterpret_cast<const f32x2*>(d.res + off + d.wo);
                            }
                            *reinterpret_cast<f32x2*>(d.out + off) = y0;
                            *reinterpret_cast<f32x2*>(d.out + off + d.wo) = y1;
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
}

template <int NH>
static int wino_launch_nh(const sda_conv_desc* d, const WinoGeom& g, int grid, hipStream_t stream) {
    constexpr int lds = (2 * (16 * WINO_CK * WINO_T) + 4 * 2 * 32 * WINO_T + 4 * 2 * 288) * 4;   // 2 V stages + exchange + halos = 73 KiB
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_set[SDA_MAX_DEVICES];           // per device: a process may use several GPUs
    const int rc = sda_raise_dyn_lds(reinterpret_cast<const void*>(conv_wino_kernel<NH>), lds, attr_set);
    if (rc != SDA_OK) return rc;
    hipLaunchKernelGGL(conv_wino_kernel<NH>, dim3(grid), dim3(512), (size_t)lds, stream, *d, g);
    return sda_launch_status();
}

int sda_wino_launch(const sda_conv_desc* d, const WinoGeom& g, hipStream_t stream) {
    const int cus = sda_cu_count();
    if (!cus) return SDA_E_BADARG;
    int grid = cus - cus % 8;
    const int need = (g.grid + 7) / 8 * 8;
    if (grid > need) grid = need;
    if (grid < 8) grid = 8;
    // the usual 8 x 4-tile workgroup tile has sh = 180 -> 6 halo slots per lane; only odd shapes need all 9
    return 2 * g.sh <= 6 * 64 ? wino_launch_nh<6>(d, g, grid, stream) : wino_launch_nh<9>(d, g, grid, stream);
}

static bool wino_disabled() {
    static const bool off = getenv("SDA_CONV_WINO") && atoi(getenv("SDA_CONV_WINO")) == 0;
    return off;
}

int sda_wino_path(const sda_conv_desc* d) {
    WinoGeom g;
    return !wino_disabled() && sda_wino_plan(d, &g) == SDA_OK;
}

int sda_wino_try(const sda_conv_desc* d, hipStream_t stream) {
    if (wino_disabled()) return SDA_E_UNSUPPORTED;
    WinoGeom g;
    const int rc = sda_wino_plan(d, &g);
    if (rc != SDA_OK) return rc;
    return sda_wino_launch(d, g, stream);
}

// ---------------------------------------------------------------- Winograd weight transform (one-off per layer)
// dst[stage][p][khalf][mm][k2] with kk = 8 stage + 2 k2 + khalf, p = 4 xi + nu:  forward (transpose=0): kk = ci, mm = co, g = w[co][ci];
// backward-data (transpose=1): kk = co, mm = ci, g[dy][dx] = w[co][ci][2-dy][2-dx].
__global__ void pack_wino_kernel(const float* __restrict__ w, int cout, int cin, int transpose, int cin_keep,
                                 float* __restrict__ dst, int k_pad, int m_pad) {
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int64_t total = (int64_t)k_pad * m_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int mm = (int)(i % m_pad), kk = (int)(i / m_pad);
        float gk[3][3];
        bool live;
        int co, ci;
        if (!transpose) { ci = kk; co = mm; live = ci < cin && co < cout; }
        else { co = kk; ci = mm; live = ci < cin_keep && ci < cin && co < cout; }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int sy = transpose ? 2 - dy : dy, sx = transpose ? 2 - dx : dx;
                gk[dy][dx] = live ? w[(((int64_t)co * cin + ci) * 3 + sy) * 3 + sx] : 0.f;
            }
        float tmp[4][3];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) tmp[xi][dx] = G[xi][0] * gk[0][dx] + G[xi][1] * gk[1][dx] + G[xi][2] * gk[2][dx];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                const float u = tmp[xi][0] * G[nu][0] + tmp[xi][1] * G[nu][1] + tmp[xi][2] * G[nu][2];
                // [stage = kk/8][p][khalf = kk&1][m][k2 = (kk&7)>>1]
                const int st = kk >> 3, kh_ = kk & 1, k2 = (kk & 7) >> 1;
                dst[((((int64_t)st * 16 + (xi * 4 + nu)) * 2 + kh_) * m_pad + mm) * 4 + k2] = u;
            }
    }
}

extern "C" int sda_pack_conv_weight_wino(const float* w, int cout, int cin, int transpose, int cin_keep, float* dst,
                                         int k_pad, int m_pad, void* stream) {
    if (!w || !dst || cout <= 0 || cin <= 0 || k_pad <= 0 || m_pad <= 0) return SDA_E_BADARG;
    int64_t total = (int64_t)k_pad * m_pad;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_wino_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, cout, cin, transpose, cin_keep,
                       dst, k_pad, m_pad);
    return sda_launch_status();
}
