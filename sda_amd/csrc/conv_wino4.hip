// Winograd F(2x2, 3x3) convolution, second generation: ONE wavefront per SIMD (4 waves x 512 registers per workgroup), every
// wave stages + transforms + multiplies in one software-pipelined instruction stream, and every wave owns ALL 16
// transform-domain positions of its output fragment, so the inverse transform is lane-local (no LDS exchange, no epilogue
// barriers).  Same arithmetic as conv_wino.hip (which stays as the fallback for shapes this kernel does not tile):
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A         d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//   U[p][ci][co] = (G g G^T)[xi][nu],  p = 4 xi + nu        (sda_pack_conv_weight_wino4, once per layer)
//   V[p][ci][t]  = (B^T d B)[xi][nu]   of tile t            (computed here, into LDS)
//   M[p][co][t]  = sum_ci U[p][ci][co] V[p][ci][t]          (16 GEMMs on v_mfma_f32_16x16x4_f32: exact fp32)
//
// Workgroup tile = 96 couts x (8 x 4 Winograd tiles = 16 x 8 output pixels of one image); K-stage = 16 input channels.
//   * wave (wm, wn) owns couts 48 wm .. +48 (three 16-row MFMA fragments) x Winograd tiles 16 wn .. +16 for all 16
//     positions: 16 x 3 accumulator fragments = 192 registers (the AGPR half of the file; the VGPR half is free for the
//     operand streams and the producer work).  Per (position, 4-channel k-step): 3 A + 1 B operand registers, 3 MFMAs.
//   * A operand (U) never touches LDS: it is packed [stage][p][cout fragment][lane][k-step] and streamed L2 -> registers,
//     one fully coalesced dwordx4 per (position, cout fragment) and stage, one position ahead of its MFMAs.
//   * B operand (V): LDS [p][kq][tile][k4], two 32 KiB stage buffers.  Wave w produces the four channels 4 w .. 4 w + 3 of
//     a stage (kq = w): it stages their 18 x 10 halo into a wave-private LDS area with row-contiguous loads (each input
//     pixel fetched once; the loader fusions -- modulation, LayerNorm, activation, nearest upsample, circular / zero
//     padding -- applied once per pixel), then lane (t, h) transforms the patches of tile t for channels 2 h, 2 h + 1 and
//     writes V[p][w][t][2 h .. 2 h + 1] (ds_write_b64); consumers read one conflict-free ds_read_b128 per position.
//   * the stage pipeline runs across tiles: while stage q is multiplied, stage q + 1 (possibly the next tile's first) is
//     committed and transformed and the global loads of stage q + 2 are issued, all spread over the 16 position steps of
//     stage q in the shadow of its 192 MFMAs (32 cycles each); one workgroup barrier per stage (the four waves are
//     symmetric, so they arrive together).  The producer is stateless across tiles: halo addresses, liveness and the
//     LayerNorm statistics of a stage are derived when its loads are issued and travel with them in registers.
//   * epilogue: A^T M A in registers, bias / act'(z) / residual fused, float2 row stores.
// Roofline: fp32 matrix pipe, 157.3 TFLOP/s; issued flops = algorithmic (direct-convolution) flops / 2.25.
#include "sda_common.hpp"
#include <stdlib.h>
#include <type_traits>

#define W4_CK 16
#define W4_BM 96
#define W4_T 32                        // 8 x 4 Winograd tiles
#define W4_HC 18                       // halo columns (16 pixels + 2)
#define W4_HRW 10                      // halo rows (8 pixels + 2)
#define W4_HS 24                       // halo row stride in LDS: tile-row stride 48 floats = 48 banks, so the four tile rows a
                                       // 32-lane half reads with ds_read_b64 cover the 64 banks exactly once
#define W4_HPLANE (W4_HRW * W4_HS)     // 240 floats per channel
#define W4_NSLOT 3                     // ceil(18 * 10 / 64) halo positions per lane and channel
#define W4_PSTR (4 * W4_T * 4)         // floats per position in a V buffer: [kq 4][t 32][k4 4]
#define W4_VBUF (16 * W4_PSTR)         // 8192 floats = 32 KiB: V of one stage
#define W4_LDS_BYTES ((2 * W4_VBUF + 4 * 4 * W4_HPLANE) * 4)

struct Wino4Geom {
    int cin, hv, wv;                   // real input channels, virtual (= output) image size
    int bx_n, by_n;                    // 16 x 8-pixel blocks per image
    int n_ct, grid, nstage, debug, mtiles;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int I, int N, class F>
__device__ __forceinline__ void w4_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        w4_static_for<I + 1, N>(f);
    }
}

static int wino4_config(const sda_conv_desc* d);

// eligibility + geometry.  SDA_E_UNSUPPORTED -> the caller falls back to conv_wino / the direct kernel.
int sda_wino4_plan(const sda_conv_desc* d, Wino4Geom* g) {
    if (!d || !d->x || !d->w_wino4 || !d->out) return SDA_E_UNSUPPORTED;
    if (d->kh != 3 || d->kw != 3 || d->stride_h != 1 || d->stride_w != 1 || d->zins_h != 1 || d->zins_w != 1)
        return SDA_E_UNSUPPORTED;
    if (d->explicit_pad || d->out_sn || d->out_sc || d->out_sy || d->out_sx) return SDA_E_UNSUPPORTED;
    if (d->cctx > 0 || d->cout % W4_BM || d->cout_pad != d->cout) return SDA_E_UNSUPPORTED;
    if ((d->ho & 7) || (d->wo & 15) || d->ho != d->hs * d->up_h || d->wo != d->ws * d->up_w) return SDA_E_UNSUPPORTED;
    if (d->up_h > 2 || d->up_w > 2 || d->up_h < 1 || d->up_w < 1) return SDA_E_UNSUPPORTED;
    if (d->mod && d->mod_sn != 0) return SDA_E_UNSUPPORTED;
    if (d->act_in != SDA_ACT_NONE && d->act_in != SDA_ACT_SILU) return SDA_E_UNSUPPORTED;       // (other activations: conv_wino)
    if ((d->ln_mean == nullptr) != (d->ln_rstd == nullptr)) return SDA_E_UNSUPPORTED;
    if (wino4_config(d) < 0) return SDA_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(d->out) & 7) || (d->res && (reinterpret_cast<uintptr_t>(d->res) & 7)) ||
        (d->dact_z && (reinterpret_cast<uintptr_t>(d->dact_z) & 7)) || (reinterpret_cast<uintptr_t>(d->w_wino4) & 15))
        return SDA_E_UNSUPPORTED;
    if (d->x_sc < 0 || d->x_sy < 0 || d->x_sx < 0 || d->n_inner != 1) return SDA_E_UNSUPPORTED;      // (no window view)
    // 32-bit BYTE offsets inside one image (channel base included)
    if ((int64_t)d->cx * d->x_sc + (int64_t)d->hs * d->x_sy + (int64_t)d->ws * d->x_sx >= (1LL << 30)) return SDA_E_UNSUPPORTED;
    if ((int64_t)d->cout * d->ho * d->wo >= (1LL << 30) || (int64_t)d->n * d->hs * d->ws >= (1LL << 31)) return SDA_E_UNSUPPORTED;
    g->cin = d->cx;
    g->hv = d->ho; g->wv = d->wo;
    g->bx_n = d->wo / 16; g->by_n = d->ho / 8;
    g->n_ct = d->cout / W4_BM;
    g->mtiles = d->cout / 16;
    const int64_t total = (int64_t)g->bx_n * g->by_n * d->n * g->n_ct;
    if (total > 0x3fffffffLL || total < 1) return SDA_E_UNSUPPORTED;
    g->grid = (int)total;
    g->nstage = (d->cx + W4_CK - 1) / W4_CK;
    if ((int64_t)g->nstage * total > 0x7fffffffLL) return SDA_E_UNSUPPORTED;
    { static const int dbg = getenv("SDA_CONV_DEBUG") ? atoi(getenv("SDA_CONV_DEBUG")) : 0; g->debug = dbg; }
    return SDA_OK;
}

// (tile id of the flat XCD-ordered list) -> cout tile, image, block row / column
struct W4Tile { int ct, n, by, bx; };
__device__ __forceinline__ W4Tile w4_decode(const Wino4Geom& g, int tile) {
    W4Tile t;
    t.ct = tile % g.n_ct;
    int rest = tile / g.n_ct;
    t.bx = rest % g.bx_n; rest /= g.bx_n;
    t.by = rest % g.by_n;
    t.n = rest / g.by_n;
    return t;
}

// MOD / LN / SILU: the loader fusions of the launch (modulation add, LayerNorm, SiLU), compile-time so that the stage body is
// ONE basic block: a runtime branch inside it would fence the producer work off from the MFMAs it is meant to hide behind
// VAR: tuning / ablation variant (0 = the shipped schedule; others exist only for tools/wino4_check.py --variants)
template <bool MOD, bool LN, bool SILU, int VAR = 0>
__global__ __launch_bounds__(256, 1) void conv_wino4_kernel(const sda_conv_desc d, const Wino4Geom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, kq = lane >> 4;
    float* const priv = smem + 2 * W4_VBUF + wave * (4 * W4_HPLANE);       // this wave's 4-channel halo

    // persistent, XCD-aware tile walk: XCD (blockIdx & 7) owns a contiguous range of the tile list, and each of its
    // workgroups a contiguous sub-range -- consecutive tiles are the cout tiles of one block, then the next block along
    // the row, so a tile's input halo was (mostly) just read into this XCD's L2, and stepping to the next tile is an
    // increment with carries (no division in the stage body)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int tq = g.grid >> 3, tr_ = g.grid & 7;
    const int t_begin = xcd * tq + (xcd < tr_ ? xcd : tr_);
    const int t_cnt = tq + (xcd < tr_ ? 1 : 0);
    const int sq = t_cnt / per_xcd, sr = t_cnt % per_xcd;
    const int first = t_begin + slot * sq + (slot < sr ? slot : sr);
    const int my_tiles = sq + (slot < sr ? 1 : 0);
    if (my_tiles <= 0) return;
    const int Q = my_tiles * g.nstage;                     // stages this workgroup runs, across all of its tiles

    // halo slot i of this lane: position lane + 64 i of the 10 x 18 halo -> (row, column); tile independent
    int hy[W4_NSLOT], hx[W4_NSLOT], lidx[W4_NSLOT];
#pragma unroll
    for (int i = 0; i < W4_NSLOT; ++i) {
        const int pos = lane + 64 * i;
        const bool valid = pos < W4_HRW * W4_HC;
        hy[i] = valid ? pos / W4_HC : -1000;               // (never live)
        hx[i] = valid ? pos - W4_HC * (pos / W4_HC) : 0;
        lidx[i] = valid ? (pos / W4_HC) * W4_HS + hx[i] : W4_HS - 1;       // (column 23 of row 0 is padding)
    }
    const int up_sh_h = d.up_h == 2 ? 1 : 0, up_sh_w = d.up_w == 2 ? 1 : 0;
    const bool circ = d.circular != 0;
    // producer lane (t, h): tile t = lane & 31 = (ty, tx) = (t >> 3, t & 7), channels 2 h and 2 h + 1 of the wave's four
    const int ph = lane >> 5;
    const int pbase = (2 * ((lane & 31) >> 3)) * W4_HS + 2 * (lane & 7) + 2 * ph * W4_HPLANE;
    const int vwr = (wave * W4_T + (lane & 31)) * 4 + 2 * ph;
    // consumer: B fragment of position p = V[p][kq][16 wn + li][0..3]
    const int vrd = (kq * W4_T + 16 * wn + li) * 4;

    // ---- producer state that travels from the issue of a stage's loads to their commit
    float hv_[4][W4_NSLOT];
    float hmean[W4_NSLOT], hrstd[W4_NSLOT];
    unsigned hlive = 0;                                    // bit i: slot i carries data
    unsigned hoff[W4_NSLOT];                               // BYTE offsets inside a channel plane (scalar base + 32-bit lane offset)
    const float* ximg = d.x;
    float u[2][4][4];                                      // [channel of the pair][row a][column]: row-transformed patches

    // stage cursor (all scalar): stage in tile + the decoded tile; kept incrementally for q (consume), q + 1 (commit /
    // transform) and q + 2 (issue)
    struct Cur { int st, ct, bx, by, n; };
    auto advance = [&](Cur& c) {
        const bool t_next = c.st + 1 == g.nstage;
        c.st = t_next ? 0 : c.st + 1;
        const bool b_next = t_next && c.ct + 1 == g.n_ct;
        c.ct = t_next ? (b_next ? 0 : c.ct + 1) : c.ct;
        const bool y_next = b_next && c.bx + 1 == g.bx_n;
        c.bx = b_next ? (y_next ? 0 : c.bx + 1) : c.bx;
        const bool n_next = y_next && c.by + 1 == g.by_n;
        c.by = y_next ? (n_next ? 0 : c.by + 1) : c.by;
        c.n += n_next ? 1 : 0;
    };

    // geometry of a stage's halo (once per stage, at issue time): addresses, liveness, LayerNorm statistics
    auto halo_geometry = [&](const Cur& t, auto I0, auto I1) {
        constexpr int i0 = decltype(I0)::value, i1 = decltype(I1)::value;
        if constexpr (i0 == 0) {
            ximg = d.x + (int64_t)(t.n + d.x_n_off) * d.x_sn_outer;
            hlive = 0;
        }
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            // (selects only: no branch may split the stage body)
            const int vy0 = 8 * t.by - 1 + hy[i], vx0 = 16 * t.bx - 1 + hx[i];
            const bool inside = vy0 >= 0 && vy0 < g.hv && vx0 >= 0 && vx0 < g.wv;
            const int vyw = vy0 < 0 ? vy0 + g.hv : (vy0 >= g.hv ? vy0 - g.hv : vy0);
            const int vxw = vx0 < 0 ? vx0 + g.wv : (vx0 >= g.wv ? vx0 - g.wv : vx0);
            const bool ok = (hy[i] >= 0) && (circ || inside);
            const int vy = ok ? vyw : 0, vx = ok ? vxw : 0;
            const int sy = vy >> up_sh_h, sx = vx >> up_sh_w;
            hoff[i] = (unsigned)(sy * (int)d.x_sy + sx * (int)d.x_sx) * 4u;
            hlive |= ok ? (1u << i) : 0u;
            if constexpr (LN) {
                const int st = (t.n * d.hs + sy) * d.ws + sx;
                hmean[i] = d.ln_mean[st];
                hrstd[i] = d.ln_rstd[st];
            }
        }
    };
    auto halo_issue = [&](const Cur& c, auto CH) {
        constexpr int ch = decltype(CH)::value;
        const int cc = W4_CK * c.st + 4 * wave + ch;
        const int cce = cc < g.cin ? cc : 0;               // (padded channels read channel 0 and are zeroed at commit)
        const char* xc = reinterpret_cast<const char*>(ximg + (int64_t)cce * d.x_sc);
#pragma unroll
        for (int i = 0; i < W4_NSLOT; ++i) hv_[ch][i] = *reinterpret_cast<const float*>(xc + hoff[i]);
    };
    auto halo_commit = [&](const Cur& c, auto CH) {
        constexpr int ch = decltype(CH)::value;
        const int cc = W4_CK * c.st + 4 * wave + ch;
        const bool real = cc < g.cin;                      // wave uniform (false only in a partial last stage)
        float* dst = priv + ch * W4_HPLANE;
        float mv = 0.f;
        if constexpr (MOD) mv = d.mod[real ? cc : 0];
        const unsigned keep = real ? hlive : 0u;
#pragma unroll
        for (int i = 0; i < W4_NSLOT; ++i) {
            float v = hv_[ch][i];
            if constexpr (MOD) v += mv;
            if constexpr (LN) v = (v - hmean[i]) * hrstd[i];
            if constexpr (SILU) v = sda_act(SDA_ACT_SILU, v);
            // padding / out-of-range positions and padded channels stage zeros
            dst[lidx[i]] = ((keep >> i) & 1u) ? v : 0.f;
        }
    };
    // patch of this lane's channel `e` of its pair -> rows transformed: u0 = d0 - d2, u1 = d1 + d2, u2 = d2 - d1, u3 = d1 - d3
    auto patch_rows = [&](auto E) {
        constexpr int e = decltype(E)::value;
        const float* src = priv + pbase + e * W4_HPLANE;
        f32x2 lo[4], hi[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            lo[a] = *reinterpret_cast<const f32x2*>(src + a * W4_HS);
            hi[a] = *reinterpret_cast<const f32x2*>(src + a * W4_HS + 2);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float d0 = c < 2 ? lo[0][c & 1] : hi[0][c & 1], d1 = c < 2 ? lo[1][c & 1] : hi[1][c & 1];
            const float d2 = c < 2 ? lo[2][c & 1] : hi[2][c & 1], d3 = c < 2 ? lo[3][c & 1] : hi[3][c & 1];
            u[e][0][c] = d0 - d2;
            u[e][1][c] = d1 + d2;
            u[e][2][c] = d2 - d1;
            u[e][3][c] = d1 - d3;
        }
    };
    // row a of both patches -> columns transformed -> V[4 a + b][wave][t][2 h .. 2 h + 1], b = 0..3
    auto patch_cols_store = [&](float* vb, auto A) {
        constexpr int a = decltype(A)::value;
        f32x2 w0, w1, w2, w3;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            w0[e] = u[e][a][0] - u[e][a][2];
            w1[e] = u[e][a][1] + u[e][a][2];
            w2[e] = u[e][a][2] - u[e][a][1];
            w3[e] = u[e][a][1] - u[e][a][3];
        }
        float* dst = vb + vwr + (4 * a) * W4_PSTR;
        *reinterpret_cast<f32x2*>(dst) = w0;
        *reinterpret_cast<f32x2*>(dst + 1 * W4_PSTR) = w1;
        *reinterpret_cast<f32x2*>(dst + 2 * W4_PSTR) = w2;
        *reinterpret_cast<f32x2*>(dst + 3 * W4_PSTR) = w3;
    };

    // ---- consumer operand streams.  A: U packed [stage][p][cout fragment][lane][k4]
    const int64_t u_pstride = (int64_t)g.mtiles * 256, u_sstride = 16 * u_pstride;
    constexpr int UD = VAR == 3 ? 7 : 3, UB = UD + 1;                     // U fragments are fetched UD position steps ahead (ring of UB)
    static_assert(16 % UB == 0, "the ring index must be static across stages");
    f32x4 av[UB][3], bv[2];
    const unsigned lane16 = lane * 16u;
    const char* up_cur = nullptr;                          // (scalar) U of the position whose fragments are fetched next
    auto u_stage = [&](const Cur& c) {
        return reinterpret_cast<const char*>(d.w_wino4 + (int64_t)(c.ct * 6 + 3 * wm) * 256 + c.st * u_sstride);
    };
    auto fetch_a = [&](int s_) {
#pragma unroll
        for (int m = 0; m < 3; ++m) av[s_][m] = *reinterpret_cast<const f32x4*>((up_cur + m * 1024) + lane16);   // scalar base + lane offset
        up_cur += u_pstride * 4;
    };
    auto fetch_b = [&](const float* vb, int p, int s_) {
        bv[s_] = *reinterpret_cast<const f32x4*>(vb + vrd + p * W4_PSTR);
    };

    f32x4 acc[16][3];

    // ---- prologue: produce stage 0 into V buffer 0, issue the loads of stage 1
    Cur c0;
    {
        const W4Tile t0 = w4_decode(g, first);
        c0.st = 0; c0.ct = t0.ct; c0.bx = t0.bx; c0.by = t0.by; c0.n = t0.n;
    }
    Cur c1 = c0, c2 = c0;                                  // c1 / c2: stages q + 1 / q + 2, clamped to the last one
    if (Q > 1) { advance(c1); advance(c2); }
    if (Q > 2) advance(c2);
    halo_geometry(c0, std::integral_constant<int, 0>{}, std::integral_constant<int, W4_NSLOT>{});
    w4_static_for<0, 4>([&](auto CH) { halo_issue(c0, CH); });
    w4_static_for<0, 4>([&](auto CH) { halo_commit(c0, CH); });
    w4_static_for<0, 2>([&](auto E) { patch_rows(E); });
    w4_static_for<0, 4>([&](auto A) { patch_cols_store(smem, A); });
    halo_geometry(c1, std::integral_constant<int, 0>{}, std::integral_constant<int, W4_NSLOT>{});
    w4_static_for<0, 4>([&](auto CH) { halo_issue(c1, CH); });
    up_cur = u_stage(c0);
#pragma unroll
    for (int i = 0; i < UD; ++i) fetch_a(i);
    __syncthreads();

    // one stage: 16 position steps of 12 MFMAs, each carrying its slice of the production of stage q + 1 and of the loads of
    // stage q + 2.  The body is ONE basic block, the same for every stage: at the very end of the workgroup's run the
    // producer cursors are clamped to the last stage, whose (valid) data is produced once more into a buffer nobody reads --
    // two redundant slices per launch instead of a peeled loop whose copies of the accumulators would have to be merged.
    // sched_barrier fences keep every slice (and the operand prefetch, which the scheduler would otherwise hoist to the top
    // and hold 190 registers with) inside its step; inside a step the group barriers ask for MFMA / others / MFMA / ...
    // so the slice issues in the 32-cycle shadows of the step's MFMAs.
    int q = 0;
    for (int tl = 0; tl < my_tiles; ++tl) {
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int m = 0; m < 3; ++m) acc[p][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int st = 0; st < g.nstage; ++st, ++q) {
        const float* vb = smem + (q & 1) * W4_VBUF;
        float* vnext = smem + ((q + 1) & 1) * W4_VBUF;
        fetch_b(vb, 0, 0);
        w4_static_for<0, 16>([&](auto P) {
            constexpr int p = decltype(P)::value;
            __builtin_amdgcn_sched_barrier(0);
            // B of the next position; A of position p + UD (of the next stage's first positions at the end: at a tile
            // change the U pointer is re-based, inside a tile the next stage simply follows in memory)
            constexpr bool NO_PROD = VAR == 4 || VAR == 6, NO_A = VAR == 5 || VAR == 6;
            if constexpr (p + 1 < 16) fetch_b(vb, p + 1, (p + 1) & 1);
            if constexpr (p + UD == 16) up_cur = u_stage(c1);
            if constexpr (!NO_A) fetch_a((p + UD) % UB);
            // the producer slice of this position step
            // (the geometry of stage q + 2 overwrites what the commits of stage q + 1 read: it starts after them)
            if constexpr (NO_PROD) { }
            else if constexpr (p < 4) halo_commit(c1, std::integral_constant<int, p>{});
            else if constexpr (p == 4) halo_geometry(c2, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            else if constexpr (p == 5 || p == 6) patch_rows(std::integral_constant<int, p - 5>{});
            else if constexpr (p >= 7 && p < 11) patch_cols_store(vnext, std::integral_constant<int, p - 7>{});
            else if constexpr (p == 11) halo_geometry(c2, std::integral_constant<int, 2>{}, std::integral_constant<int, W4_NSLOT>{});
            else if constexpr (p >= 12) halo_issue(c2, std::integral_constant<int, p - 12>{});
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                for (int m = 0; m < 3; ++m)
                    acc[p][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[p % UB][m][k4], bv[p & 1][k4], acc[p][m], 0, 0, 0);
            // fillers per MFMA shadow: the slice of the step spread over its 12 MFMAs
            constexpr int NF = VAR == 1 ? ((p == 4 || p == 11) ? 6 : 5)
                                        : ((p == 4 || p == 11) ? 6 : (p < 4 && (SILU || LN)) ? 5 : 3);
            if constexpr (VAR != 2) {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x496, NF, 0);     // a few VALU / SALU / VMEM / DS / TRANS
                }
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        // stage hand-off: every wave's V writes of stage q + 1 have landed and its V reads of stage q have returned
        // (lgkmcnt only -- a __syncthreads() would also drain the global loads in flight for the coming stages: the U
        // fragments of the next three positions and the halo of stage q + 2, i.e. expose an HBM round trip per stage)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (st + 1 < g.nstage) advance(c0);                // (c0 stays on the tile for its epilogue)
        if (q + 2 < Q) advance(c1);
        if (q + 3 < Q) advance(c2);
      }
        {
            // ---- epilogue of tile c0.tile: Y = A^T M A per (cout, tile), lane local.  acc[4 xi + nu][m][r]:
            //      cout = 96 ct + 48 wm + 16 m + 4 kq + r,  tile = 16 wn + li
            const Cur& tt = c0;
            const int hw_o = d.ho * d.wo;
            const int t = 16 * wn + li;
            const int oy = 8 * tt.by + 2 * (t >> 3), ox = 16 * tt.bx + 2 * (t & 7);
            const int64_t obase = (int64_t)tt.n * d.cout * hw_o + (int64_t)oy * d.wo + ox;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = W4_BM * tt.ct + 48 * wm + 16 * m + 4 * kq + r;
                    // rows (xi): s0 = M0 + M1 + M2, s1 = M1 - M2 - M3, for each nu
                    float s0[4], s1[4];
#pragma unroll
                    for (int nu = 0; nu < 4; ++nu) {
                        const float m0 = acc[nu][m][r], m1 = acc[4 + nu][m][r], m2 = acc[8 + nu][m][r], m3 = acc[12 + nu][m][r];
                        s0[nu] = (m0 + m1) + m2;
                        s1[nu] = (m1 - m2) - m3;
                    }
                    const float bias = d.bias ? d.bias[co] : 0.f;
                    f32x2 y0, y1;
                    y0[0] = (s0[0] + s0[1]) + s0[2] + bias; y0[1] = (s0[1] - s0[2]) - s0[3] + bias;
                    y1[0] = (s1[0] + s1[1]) + s1[2] + bias; y1[1] = (s1[1] - s1[2]) - s1[3] + bias;
                    const int64_t o = obase + (int64_t)co * hw_o;
                    if (d.dact_z) {
                        const f32x2 q0 = *reinterpret_cast<const f32x2*>(d.dact_z + o);
                        const f32x2 q1 = *reinterpret_cast<const f32x2*>(d.dact_z + o + d.wo);
                        if (d.act_d == SDA_ACT_SILU) {
                            y0[0] *= sda_dact(SDA_ACT_SILU, q0[0]); y0[1] *= sda_dact(SDA_ACT_SILU, q0[1]);
                            y1[0] *= sda_dact(SDA_ACT_SILU, q1[0]); y1[1] *= sda_dact(SDA_ACT_SILU, q1[1]);
                        } else {
                            y0[0] *= sda_dact(d.act_d, q0[0]); y0[1] *= sda_dact(d.act_d, q0[1]);
                            y1[0] *= sda_dact(d.act_d, q1[0]); y1[1] *= sda_dact(d.act_d, q1[1]);
                        }
                    }
                    if (d.res) {
                        y0 += *reinterpret_cast<const f32x2*>(d.res + o);
                        y1 += *reinterpret_cast<const f32x2*>(d.res + o + d.wo);
                    }
                    if (!(g.debug & 8)) {
                        *reinterpret_cast<f32x2*>(d.out + o) = y0;
                        *reinterpret_cast<f32x2*>(d.out + o + d.wo) = y1;
                    }
                }
            }
        }
        advance(c0);
    }
}

template <bool MOD, bool LN, bool SILU, int VAR>
static int wino4_launch_t(const sda_conv_desc* d, const Wino4Geom& g, int grid, hipStream_t stream) {
    static_assert(W4_LDS_BYTES <= 160 * 1024, "LDS");
    static bool attr_set[SDA_MAX_DEVICES];
    const int rc = sda_raise_dyn_lds(reinterpret_cast<const void*>(conv_wino4_kernel<MOD, LN, SILU, VAR>), W4_LDS_BYTES, attr_set);
    if (rc != SDA_OK) return rc;
    hipLaunchKernelGGL((conv_wino4_kernel<MOD, LN, SILU, VAR>), dim3(grid), dim3(256), (size_t)W4_LDS_BYTES, stream, *d, g);
    return sda_launch_status();
}

// the four loader configurations of the reference U-Net have a kernel: plain (backward-data convolutions), modulation +
// LayerNorm (first block convolution), SiLU (second block convolution), LayerNorm (upsampling tails); other combinations
// are served by conv_wino / the direct kernel
static int wino4_config(const sda_conv_desc* d) {
    const bool mod = d->mod != nullptr, ln = d->ln_mean != nullptr, silu = d->act_in == SDA_ACT_SILU;
    const int key = (mod ? 4 : 0) | (ln ? 2 : 0) | (silu ? 1 : 0);
    return (key == 0 || key == 1 || key == 2 || key == 6) ? key : -1;
}

int sda_wino4_launch(const sda_conv_desc* d, const Wino4Geom& g, hipStream_t stream) {
    const int cus = sda_cu_count();
    if (!cus) return SDA_E_BADARG;
    int grid = cus - cus % 8;
    const int need = (g.grid + 7) / 8 * 8;
    if (grid > need) grid = need;
    if (grid < 8) grid = 8;
    switch (wino4_config(d)) {
        case 0: {
#ifdef SDA_W4_VARIANTS
            const char* ev = getenv("SDA_W4_VAR");
            switch (ev ? atoi(ev) : 0) {
                case 1: return wino4_launch_t<false, false, false, 1>(d, g, grid, stream);
                case 2: return wino4_launch_t<false, false, false, 2>(d, g, grid, stream);
                case 3: return wino4_launch_t<false, false, false, 3>(d, g, grid, stream);
                case 4: return wino4_launch_t<false, false, false, 4>(d, g, grid, stream);
                case 5: return wino4_launch_t<false, false, false, 5>(d, g, grid, stream);
                case 6: return wino4_launch_t<false, false, false, 6>(d, g, grid, stream);
                default: break;
            }
#endif
            return wino4_launch_t<false, false, false, 0>(d, g, grid, stream);
        }
        case 1: return wino4_launch_t<false, false, true, 0>(d, g, grid, stream);
        case 2: return wino4_launch_t<false, true, false, 0>(d, g, grid, stream);
        case 6: return wino4_launch_t<true, true, false, 0>(d, g, grid, stream);
        default: return SDA_E_UNSUPPORTED;
    }
}

static bool wino4_disabled() {
    static const bool off = getenv("SDA_CONV_WINO4") && atoi(getenv("SDA_CONV_WINO4")) == 0;
    return off;
}

int sda_wino4_path(const sda_conv_desc* d) {
    Wino4Geom g;
    return !wino4_disabled() && sda_wino4_plan(d, &g) == SDA_OK;
}

int sda_wino4_try(const sda_conv_desc* d, hipStream_t stream) {
    if (wino4_disabled()) return SDA_E_UNSUPPORTED;
    Wino4Geom g;
    const int rc = sda_wino4_plan(d, &g);
    if (rc != SDA_OK) return rc;
    return sda_wino4_launch(d, g, stream);
}

// ---------------------------------------------------------------- weight transform for this kernel (one-off per layer)
// dst[stage][p][m tile][lane = 16 kq + i][k4]  <-  (G g G^T)[xi][nu] of the filter between contraction channel
// kk = 16 stage + 4 kq + k4 and output channel mm = 16 mtile + i;  forward (transpose = 0): kk = ci, mm = co;
// backward-data (transpose = 1): kk = co, mm = ci, filter flipped.  k_pad % 16 == 0, m_pad % 96 == 0; padding is zero.
__global__ void pack_wino4_kernel(const float* __restrict__ w, int cout, int cin, int transpose, int cin_keep,
                                  float* __restrict__ dst, int k_pad, int m_pad) {
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int64_t total = (int64_t)k_pad * m_pad;
    const int mtiles = m_pad / 16;
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
        const int st = kk >> 4, kq_ = (kk >> 2) & 3, k4 = kk & 3, mt = mm >> 4, ii = mm & 15;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                const float uu = tmp[xi][0] * G[nu][0] + tmp[xi][1] * G[nu][1] + tmp[xi][2] * G[nu][2];
                dst[((((int64_t)st * 16 + (xi * 4 + nu)) * mtiles + mt) * 64 + (kq_ * 16 + ii)) * 4 + k4] = uu;
            }
    }
}

extern "C" int sda_pack_conv_weight_wino4(const float* w, int cout, int cin, int transpose, int cin_keep, float* dst,
                                          int k_pad, int m_pad, void* stream) {
    if (!w || !dst || cout <= 0 || cin <= 0 || k_pad <= 0 || m_pad <= 0 || (k_pad & 15) || (m_pad % W4_BM)) return SDA_E_BADARG;
    int64_t total = (int64_t)k_pad * m_pad;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_wino4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, cout, cin, transpose, cin_keep,
                       dst, k_pad, m_pad);
    return sda_launch_status();
}
