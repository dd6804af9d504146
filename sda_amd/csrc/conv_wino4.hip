// Winograd F(2x2, 3x3) convolution, second generation.  Same arithmetic as conv_wino.hip (which stays as the fallback
// for shapes this kernel does not tile):
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A         d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//   U[p][ci][co] = (G g G^T)[xi][nu],  p = 4 xi + nu        (sda_pack_conv_weight_wino4, once per layer)
//   V[p][ci][t]  = (B^T d B)[xi][nu]   of tile t            (computed here, into LDS)
//   M[p][co][t]  = sum_ci U[p][ci][co] V[p][ci][t]          (16 GEMMs on v_mfma_f32_16x16x4_f32: exact fp32)
//
// What the design is built around (tools/mfma_shadow_gen.py, profiles/r02_mfma_shadow.txt): the fp32 MFMA runs on the
// vector datapath, so an instruction of the SAME wave issued between two MFMAs is not hidden -- a VALU costs 4 cycles plus
// ~10 for the switch, a global load ~18, an LDS read ~0.3-5, scalar instructions nothing; the LDS / scalar / vector-memory
// instructions of a SIBLING wave on the same SIMD cost the MFMA wave nothing, but its VALU instructions do not issue at all
// while the stream runs.  Hence:
//   * waves 0-3 (one per SIMD, "consumers") are pure multiply streams: MFMA + ds_read + scalar bookkeeping, nothing else in
//     the loop.  BOTH operands come from LDS.  Each owns ALL 16 transform-domain positions of a 48-cout x 16-tile fragment
//     (3 x 1 MFMA fragments x 16 positions = 192 accumulators of its 256 registers), so the inverse transform A^T M A is
//     lane-local: no LDS exchange between consumers.
//   * waves 4-7 (their siblings, "helpers") do everything else for the NEXT K-stages while the current one is multiplied:
//     the stage's U slab L2 -> registers -> LDS, the input halo (every pixel once; loader fusions -- modulation, LayerNorm,
//     SiLU, nearest upsample, circular / zero padding -- applied once per pixel), the B^T d B transform and the V writes;
//     their global loads are issued stages ahead from inline asm with hand-counted s_waitcnt vmcnt(N).  Their vector-ALU work
//     runs in a deliberate pause of the consumers (barrier M_q after step 3 of a stage); a second barrier E_q after step 6
//     hands the next stage's buffers over early.
//   * the epilogue operand (residual / skip tensor, or z of x act'(z)) reaches the consumers through the helpers' registers
//     and the U buffer the tile's last hand-off released (EPI kernels): a consumer-side global load queues behind everything
//     the helpers have in flight on the CU's vector-memory path.
//   * workgroup tile = 96 couts (64 for widths that are multiples of 64 but not of 96: template parameter MF, round 6) x (8 x 4
//     Winograd tiles = 16 x 8 output pixels of one image); K-stage = 8 input channels;
//     two (U 48 KiB + V 16 KiB) stage buffers + wave-private halo planes = 135.5 KiB of LDS; persistent over an XCD-contiguous
//     tile range.  The stage pipeline runs across tiles (a stage's halo addresses / liveness / LayerNorm statistics are
//     derived when its loads are issued and travel with them in registers), so only the epilogue itself is not covered by MFMAs.
//   * LDS layouts are conflict free: U [pair][cout fragment][lane][h][k4] (one ds_read_b128 per fragment and position pair),
//     V [pair][kq][k4][tile][h] (8-byte lane-linear writes, ds_read_b64).
// DESIGN.md 5.1c has the measurements behind each choice.
// Roofline: fp32 matrix pipe, 157.3 TFLOP/s; issued flops = algorithmic (direct-convolution) flops / 2.25.
#include "sda_common.hpp"
#include <stdlib.h>
#include <type_traits>

// U slab: L2 -> registers -> ds_write_b128 (0, shipped) or L2 -> LDS by inline-asm LDS-DMA (1; A/B builds:
// SDA_EXTRA_HIPCC_FLAGS=-DW4_UDMA=1).  Round 5 built and measured the DMA form (profiles/r05_w4_udma_ab.txt): parity-green, and
// SLOWER -- plain 96 -> 96 +2 %, x act'(z) +8 %, 384 -> 384 +11 %, the zero-position forms +7 / +8 % -- although leaving the U stores
// out altogether is worth -1 .. -6 % (profiles/r05_w4_udma_upper_bound.txt): a DMA piece costs the in-order helper ~100+ cycles of
// issue where a global load + a ds_write_b128 cost ~30, and that time sits in front of the work the consumers' pause waits for.
#ifndef W4_UDMA
#define W4_UDMA 0
#endif
#define W4_CK 8
// cout tile = 32 MF output channels (MF = cout fragments of 16 per consumer wave): MF = 3 -> 96 (the reference's training widths
// (96, 192, 384), kolmogorov/train.py:19), MF = 2 -> 64 (its DEFAULT widths (64, 128, 256), kolmogorov/utils.py:52: round 6), MF = 1 ->
// 32 (the remaining multiples of 32: UNet's own default (32, 64, 128), sda/nn.py:99 -- four MFMAs per consumer step, helper-bound).  A
// 128-cout tile (MF = 4) would need 256 accumulator registers per consumer; 128 / 256 couts run as 2 / 4 tiles of 64.
#define W4_BM_OF(MF) (32 * (MF))
#define W4_T 32                        // 8 x 4 Winograd tiles
#define W4_HC 18                       // halo columns (16 pixels + 2)
#define W4_HRW 10                      // halo rows (8 pixels + 2)
#define W4_HS 24                       // halo row stride in LDS: tile-row stride 48 floats = 48 banks, so the four tile rows a
                                       // 32-lane half reads with ds_read_b64 cover the 64 banks exactly once
#define W4_HPLANE (W4_HRW * W4_HS)     // 240 floats per channel
#define W4_NSLOT 3                     // ceil(18 * 10 / 64) halo positions per lane and channel
#define W4_UPP_OF(MF) (2 * (MF) * 64 * 4)   // floats per position PAIR in a U buffer: [cout fragment 2 MF][lane 64][h 2][k4 2]  (1536 | 1024)
#define W4_UBUF_OF(MF) (8 * W4_UPP_OF(MF))  // 12288 floats = 48 KiB | 8192 floats = 32 KiB
// floats of a stage's U slab in the zero-position packing (sda_pack_conv_weight_wino4_zp): the full pairs 0 / 2 / 6 at 0 / UPP / 2 UPP,
// the live halves of pairs 1 and 3 interleaved at 3 UPP, pair 7's at 4 UPP (float2 per lane: UPP / 2), zero padding behind it up to a
// whole number of 1-KiB pieces per helper wave: MF = 3: 6912 -> 7168 floats (28 KiB, seven pieces per helper), MF = 2: 4608 -> 5120
// (20 KiB, five pieces per helper), MF = 1: 2304 -> 3072 (12 KiB, three pieces per helper)
#define W4_NULZ_OF(MF) ((MF) == 3 ? 7 : (MF) == 2 ? 5 : 3)
#define W4_UZP_OF(MF) (W4_NULZ_OF(MF) * 4 * 256)
#define W4_VKQ 128                     // floats per kq plane of V: [k4 2][tile 32][h 2]
#define W4_VPP (4 * W4_VKQ)            // floats per position pair in a V buffer
#define W4_VBUF (8 * W4_VPP)           // 4096 floats = 16 KiB
#define W4_LDS_BYTES_OF(MF) ((2 * W4_UBUF_OF(MF) + 2 * W4_VBUF + 4 * 2 * W4_HPLANE) * 4)     // 135.5 KiB | 103.5 KiB
#define W4_LDS_ALLOC_OF(MF) (W4_LDS_BYTES_OF(MF) + (W4_UDMA ? 1024 : 0))      // + the landing KiB of the prologue's dummy DMA

struct Wino4Geom {
    int cin, hv, wv;                   // real input channels, virtual (= output) image size
    int bx_n, by_n;                    // 16 x 8-pixel blocks per image
    int n_ct, grid, nstage, debug, mtiles;
    int walk;                          // 1: the workgroups of an XCD walk its tile range interleaved (tile = slot + j * per_xcd), 0: each a contiguous sub-range
    int sc, sx, sy, sn;                // a workgroup's step from one tile to its next, as (cout tiles, columns, rows, images): mixed-radix digits
    long long* trace;                  // (tooling builds only) per-wave phase cycle sums
    int mf;                            // cout fragments per consumer wave: cout tile = 32 mf (3: cout % 96 == 0, else 2: cout % 64 == 0); host side only
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned w4_u32x2 __attribute__((ext_vector_type(2)));

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
    if (d->cctx < 0 || (d->cctx > 0 && !d->ctx) || (d->cout % 32) || d->cout_pad != d->cout || d->cin_pad % W4_CK || d->cin_pad < d->cx) return SDA_E_UNSUPPORTED;
    if ((d->ho & 7) || (d->wo & 15) || d->ho != d->hs * d->up_h || d->wo != d->ws * d->up_w) return SDA_E_UNSUPPORTED;
    if (d->up_h > 2 || d->up_w > 2 || d->up_h < 1 || d->up_w < 1) return SDA_E_UNSUPPORTED;
    // pooled output (2 x 2 cell sums at half resolution): plain launches without an epilogue operand only
    if (d->pool_h > 1 || d->pool_w > 1) {
        if (!d->w_wino4_zp || (reinterpret_cast<uintptr_t>(d->w_wino4_zp) & 15)) return SDA_E_UNSUPPORTED;
        if (d->pool_h != 2 || d->pool_w != 2 || d->mod || d->ln_mean || d->act_in != SDA_ACT_NONE || d->res || d->dact_z || d->cctx ||
            (reinterpret_cast<uintptr_t>(d->out) & 3))
            return SDA_E_UNSUPPORTED;
    }
    if (d->mod && d->mod_sn != 0) return SDA_E_UNSUPPORTED;
    if (d->act_in != SDA_ACT_NONE && d->act_in != SDA_ACT_SILU) return SDA_E_UNSUPPORTED;       // (other activations: conv_wino)
    if ((d->ln_mean == nullptr) != (d->ln_rstd == nullptr)) return SDA_E_UNSUPPORTED;
    if (wino4_config(d) < 0) return SDA_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(d->out) & 7) || (d->res && (reinterpret_cast<uintptr_t>(d->res) & 7)) ||
        (d->dact_z && (reinterpret_cast<uintptr_t>(d->dact_z) & 7)) || (reinterpret_cast<uintptr_t>(d->w_wino4) & 15) ||
        (d->bias && (reinterpret_cast<uintptr_t>(d->bias) & 15)))
        return SDA_E_UNSUPPORTED;
    if (d->x_sc < 0 || d->x_sy < 0 || d->x_sx < 0 || d->n_inner < 1) return SDA_E_UNSUPPORTED;
    // context channels (planar [cctx][hs][ws], appended after the cx source channels -- the forcing channel of the Kolmogorov
    // head convolution) share the halo offsets of the source: needs a source with planar rows and no loader fusion
    if (d->cctx > 0 && (d->x_sy != d->ws || d->x_sx != 1 || d->mod || d->ln_mean || d->act_in != SDA_ACT_NONE || d->up_h != 1 ||
                        d->up_w != 1 || d->ctx_sn < 0 || (int64_t)d->cctx * d->hs * d->ws >= (1LL << 28)))
        return SDA_E_UNSUPPORTED;
    // 32-bit BYTE offsets inside one image (channel base included)
    if ((int64_t)d->cx * d->x_sc + (int64_t)d->hs * d->x_sy + (int64_t)d->ws * d->x_sx >= (1LL << 30)) return SDA_E_UNSUPPORTED;
    if ((int64_t)48 * d->ho * d->wo * 4 >= (1LL << 31)) return SDA_E_UNSUPPORTED;            // (epilogue buffer descriptors: <= 48 planes)
    if ((int64_t)d->cout * d->ho * d->wo >= (1LL << 30) || (int64_t)d->n * d->hs * d->ws >= (1LL << 31)) return SDA_E_UNSUPPORTED;
    g->cin = d->cx + d->cctx;
    g->hv = d->ho; g->wv = d->wo;
    g->bx_n = d->wo / 16; g->by_n = d->ho / 8;
    g->mf = d->cout % 96 == 0 ? 3 : (d->cout % 64 == 0 ? 2 : 1);
#if W4_UDMA
    if (g->mf != 3) return SDA_E_UNSUPPORTED;          // (the measured-and-rejected DMA form exists for the 96-cout tile only)
#endif
    g->n_ct = d->cout / W4_BM_OF(g->mf);
    g->mtiles = d->cout / 16;
    const int64_t total = (int64_t)g->bx_n * g->by_n * d->n * g->n_ct;
    if (total > 0x3fffffffLL || total < 1) return SDA_E_UNSUPPORTED;
    g->grid = (int)total;
    g->nstage = d->cin_pad / W4_CK;
    if ((int64_t)g->nstage * total > 0x7fffffffLL) return SDA_E_UNSUPPORTED;
#ifdef SDA_W4_VARIANTS
    g->debug = sda_debug_env();      // (tooling build: re-read per launch)
#else
    { static const int dbg = sda_debug_env(); g->debug = dbg; }            // (0 in the product build)
#endif
    g->trace = nullptr;
    g->walk = 1; g->sc = 1; g->sx = 0; g->sy = 0; g->sn = 0;   // (set per launch: sda_wino4_launch)
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

// stage cursor (all scalar): stage in tile + the decoded tile.  Stepping to the next stage / tile is an increment with
// carries; a workgroup's next tile is (sc, sx, sy, sn) further on in (cout tile, column, row, image) order -- see the tile walk
// in the kernel.
struct W4Cur { int st, ct, bx, by, n; };
// 1 if a == n else 0, for a <= n: pure integer arithmetic.  (A bool conjunction of uniform compares is lowered through lane
// masks and a v_cndmask / v_readfirstlane pair -- vector instructions in a wave that must issue none.)
__device__ __forceinline__ int w4_eq(int a, int n) { return (int)(((unsigned)(n - a) - 1u) >> 31); }
// 1 if a >= n else 0, for 0 <= a < 2 n
__device__ __forceinline__ int w4_ge(int a, int n) { return (int)((unsigned)(n - 1 - a) >> 31); }
__device__ __forceinline__ void w4_advance(const Wino4Geom& g, W4Cur& c) {
    const int t_next = w4_eq(c.st + 1, g.nstage);
    c.st = (c.st + 1) * (1 - t_next);
    // mixed-radix add of the tile step: every digit of the step is below its radix, so one carry per digit suffices
    const int ct = c.ct + t_next * g.sc;
    const int cc = w4_ge(ct, g.n_ct);
    c.ct = ct - cc * g.n_ct;
    const int bx = c.bx + t_next * g.sx + cc;
    const int cx = w4_ge(bx, g.bx_n);
    c.bx = bx - cx * g.bx_n;
    const int by = c.by + t_next * g.sy + cx;
    const int cy = w4_ge(by, g.by_n);
    c.by = by - cy * g.by_n;
    c.n += t_next * g.sn + cy;
}

// phase stamps of the tracing variant (VAR == 11): T(k) adds the cycles since the previous stamp to phase k
#define W4_DBG(b) SDA_DBG(g, b)        // tooling builds only (sda_common.hpp): runtime ablation switches, timing experiments
#define W4_TRACE_DECL long long w4tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long w4tl_ = 0, w4tb_ = 0; (void)w4tr_; (void)w4tl_; (void)w4tb_; if constexpr (VAR == 11) w4tb_ = __builtin_readcyclecounter()
#define W4_T0() do { if constexpr (VAR == 11) w4tl_ = __builtin_readcyclecounter(); } while (0)
#define W4_STAMP(k) do { if constexpr (VAR == 11) if (!SDA_DBG(g, 4096)) { const long long n_ = __builtin_readcyclecounter(); w4tr_[k] += n_ - w4tl_; w4tl_ = n_; } } while (0)
// deferred stamps (VAR == 11, debug bit 8192): s_memtime into SGPRs, read only after the pause they bracket -- no wait in between
#define W4_MARK(v) do { if constexpr (VAR == 11) if SDA_DBG(g, 8192) asm volatile("s_memtime %0" : "=s"(v) :: "memory"); } while (0)
#define W4_MARK_ADD(k, a, b) do { if constexpr (VAR == 11) if SDA_DBG(g, 8192) w4tr_[k] += (long long)((b) - (a)); } while (0)
#define W4_MARK_WAIT() do { if constexpr (VAR == 11) if SDA_DBG(g, 8192) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#define W4_TRACE_OUT() do { if constexpr (VAR == 11) { w4tr_[7] = __builtin_readcyclecounter() - w4tb_; if (g.trace && lane == 0) for (int k_ = 0; k_ < 8; ++k_) g.trace[(blockIdx.x * 8 + wave) * 8 + k_] = w4tr_[k_]; } } while (0)

// Helper-wave global loads, hidden from the compiler's s_waitcnt bookkeeping.  The helpers keep loads in flight for several
// loop iterations (the halo comes from HBM: ~2.5 us under load); hipcc's waitcnt insertion loses count across the loop's
// control flow and falls back to vmcnt(0) before every use, which serialises every load with its consumer.  Issued from
// inline asm the loads are invisible to it, and the waits are written by hand with exact counts (vmcnt retires in order).
// `s_nop 4`: an SGPR base written by a scalar instruction needs 5 wait states before a vector memory instruction reads it,
// and the compiler does not know this statement is one.
__device__ __forceinline__ void w4_ld1(float& dst, const char* base, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void w4_ld2(f32x2& dst, const char* base, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory");
}
template <int OFF>
__device__ __forceinline__ void w4_ld4(f32x4& dst, const char* base, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(off), "s"(base), "n"(OFF) : "memory");
}


// SiLU'(z) = s (1 + z (1 - s)), s = 1 / (1 + e), e = exp(-z); with 1 - s = e s: s (1 + z e s).  On register PAIRS: the four
// multiplies / adds are packed instructions (v_pk_mul / v_pk_add / v_pk_fma: two values each), only exp and rcp are per
// value -- the consumers' epilogue arithmetic is time the matrix pipe stands still (the x act'(z) layers lose ~12 % to it).
// z is clamped at -80 first: below it e = exp(-z) overflows (z e -> -inf at z ~ -85, e = inf -> fma(-inf, 0, 1) = NaN at z ~ -89) where the
// true derivative is ~ z exp(z) -> -0 (sda_dact and torch's silu_backward return ~ -1e-35 / -0 there); at the clamp the value is -1.4e-33.
__device__ __forceinline__ f32x2 w4_dsilu2(f32x2 z) {
    z = f32x2{__builtin_fmaxf(z[0], -80.f), __builtin_fmaxf(z[1], -80.f)};
    const f32x2 t = z * f32x2{-1.4426950408889634f, -1.4426950408889634f};
    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    const f32x2 dn = e + f32x2{1.f, 1.f};
    const f32x2 sg = {__builtin_amdgcn_rcpf(dn[0]), __builtin_amdgcn_rcpf(dn[1])};
    const f32x2 u = z * e;
    const f32x2 w = __builtin_elementwise_fma(u, sg, f32x2{1.f, 1.f});
    return sg * w;
}

// MOD / LN / SILU: the loader fusions of the launch (modulation add, LayerNorm, SiLU) as compile-time switches
// EPI: the epilogue operand (residual or activation-derivative input) reaches the consumers through the helpers' registers and
//      LDS instead of their own global loads (see "epilogue operand" in the helpers)
// VAR: ablation variant (0 = shipped; others exist only under -DSDA_W4_VARIANTS for tools/wino4_check.py --variants)
// EPM: how the epilogue operand travels.  0 = the launch has none (plain / modulation + LayerNorm launches: the epilogue is the
//      inverse transform and the stores, nothing else is compiled in); 1 = through the helpers (EPI, above); 2 = consumer-side
//      buffer loads (any operand combination; tiles shorter than twelve stages).
template <int ZP>
__host__ __device__ constexpr bool w4_dead(int p) { return ZP != 0 && ((p >> 2) == 2 || (p & 3) == 2); }

// ZP:  "zero positions".  With B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1] a 4 x 4 input patch whose rows 1 and 2 are EQUAL has a zero
//      row 2 in B^T d B (columns alike), and with A^T = [1 1 1 0; 0 1 -1 -1] the SUM of a tile's 2 x 2 outputs is v^T M v with
//      v = (1, 2, 0, -1): in both cases the seven positions (xi, nu) with xi == 2 or nu == 2 carry nothing -- 54 MFMAs per stage
//      instead of 96, bit-identical results (the skipped products are exact zeros / have weight zero).
//      1 = the source is nearest-upsampled by 2 x 2 (the tails: LayerNorm -> Upsample -> conv, sda/nn.py:161-169; output tiles
//          start at even pixels, so patch rows / columns 1, 2 are the same source pixel);
//      2 = the output is summed over its 2 x 2 cells (sda_conv_desc.pool_h / pool_w: the input VJP of such a tail, the VJP of
//          the upsample fused into the epilogue): one value per (cout, tile), written at half resolution.
// MF:  cout fragments (16 couts each) per consumer wave: the workgroup tile is 32 MF couts x 32 Winograd tiles (W4_BM_OF above).  MF = 2
//      keeps everything of the MF = 3 design -- same helpers, same V side, same two barriers per stage -- with 8 instead of 12 MFMAs and 4
//      instead of 5 LDS reads per consumer step, a 32-KiB U slab (8 loads per helper) and the epilogue operand through consumer-side
//      loads (EPM 2; see wino4_epm for the measured-and-removed helper-fed form).
template <bool MOD, bool LN, bool SILU, int EPM, int VAR = 0, int ZP = 0, int MF = 3>
__global__ __launch_bounds__(512, 2) void conv_wino4_kernel(const sda_conv_desc d, const Wino4Geom g) {
    static_assert(MF >= 1 && MF <= 3, "cout tile = 32, 64 or 96");
    static_assert(MF == 3 || EPM != 1, "the helper-fed epilogue operand exists for the 96-cout tile only");
    static_assert(MF > 1 || ZP != 1, "the 32-cout tile has no up-sampled zero-position form");
    constexpr bool ZPOS = ZP != 0;                         // 9 live Winograd positions of 16 (see ZP above)
    constexpr bool EPI = EPM == 1;
    constexpr int W4_BM = W4_BM_OF(MF), W4_UPP = W4_UPP_OF(MF), W4_UBUF = W4_UBUF_OF(MF), W4_UZP = W4_UZP_OF(MF);
    constexpr int W4_LDS_BYTES = W4_LDS_BYTES_OF(MF);
    constexpr int NULZ = W4_NULZ_OF(MF);                   // 1-KiB pieces of the position-packed slab per helper
    constexpr int NUF = 4 * MF;                            // 1-KiB pieces of the full slab per helper (two position pairs x 2 MF fragments)
    (void)W4_BM; (void)W4_UZP; (void)W4_LDS_BYTES; (void)NULZ; (void)NUF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const ubuf = smem;                              // [2][W4_UBUF]
    float* const vbuf = smem + 2 * W4_UBUF;                // [2][W4_VBUF]

    // persistent, XCD-aware tile walk.  The tile list is ordered (image, block row, block column, cout tile), a block being
    // 16 x 8 output pixels.  XCD (blockIdx & 7) owns a contiguous eighth of it, and its per_xcd workgroups walk that range
    // INTERLEAVED (tile = begin + slot + j * per_xcd): at any moment the XCD works on ~per_xcd consecutive tiles, i.e. on ALL
    // cout tiles of a few consecutive blocks -- whole block rows of one image.  Everything the tiles share is then fetched into
    // this XCD's L2 once and hit there by the other readers within microseconds: the input of a block by its n_ct cout tiles,
    // the three 128-byte lines a halo row touches by the horizontal neighbours, the rows shared with the blocks above / below,
    // and a K-stage's U slab by the workgroups on the same cout tile (they run in near lockstep).  Round 2 gave each workgroup
    // a contiguous sub-range: a block's neighbour (and its next cout tile) then ran on the same CU one tile later, after the
    // XCD's 32 CUs had pulled twice the L2's size through it -- 5 to 16 times the algorithmic reads reached the fabric
    // (profiles/r03_w4_traffic.txt).  g.walk == 0 keeps that walk for A/B measurements.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int tq = g.grid >> 3, tr_ = g.grid & 7;
    const int t_begin = xcd * tq + (xcd < tr_ ? xcd : tr_);
    const int t_cnt = tq + (xcd < tr_ ? 1 : 0);
    int first, my_tiles;
    if (g.walk) {
        first = t_begin + slot;
        my_tiles = slot < t_cnt ? (t_cnt - slot + per_xcd - 1) / per_xcd : 0;
    } else {
        const int sq = t_cnt / per_xcd, sr = t_cnt % per_xcd;
        first = t_begin + slot * sq + (slot < sr ? slot : sr);
        my_tiles = sq + (slot < sr ? 1 : 0);
    }
    if (my_tiles <= 0) return;
    const int Q = my_tiles * g.nstage;                     // stages this workgroup runs, across all of its tiles
    W4Cur c0;
    {
        const W4Tile t0 = w4_decode(g, first);
        c0.st = 0; c0.ct = t0.ct; c0.bx = t0.bx; c0.by = t0.by; c0.n = t0.n;
    }

    if (wave >= 4) {
        // ================================================================== helpers (waves 4-7, the consumers' siblings)
        // A sibling of an MFMA-saturated wave issues roughly one instruction per 10-20 cycles (and an LDS-DMA instruction costs
        // it ~150), so a helper has room for ~200 instructions per 3072-cycle stage: the work is split evenly over the four of
        // them and kept lean.  Helper j, while stage q is multiplied:
        //   * V of stage q + 1 for the channels with kq = j (channel of a stage = 2 kq + k4): commits their 18 x 10 halo --
        //     loaded THREE iterations ago (HBM round trips take ~2.5 us under load) into a ring of four register sets, with the
        //     loader fusions applied once per pixel -- to its private LDS area, then lane (t, e) transforms the patch of tile
        //     t, channel k4 = e and writes V[p][j][t][e];
        //   * U of stage q + 1, positions 4 j .. 4 j + 3 (12 KiB): registers (loaded one iteration ago from L2, 12 coalesced
        //     dwordx4) -> LDS, lane-linear; then the loads of stage q + 2 into the same registers;
        //   * the halo loads of stage q + 4 last: vmcnt retires in order, so waiting for the U registers of the next iteration
        //     completes only halo loads that are at least two iterations old.
        // Halo addresses / liveness / LayerNorm statistics depend on the tile only: they are recomputed when the issue cursor
        // enters a new tile and travel with each register set.
        const int pw = wave - 4;
        if (W4_DBG(8192)) return;                          // (ablation: consumers alone; ended waves leave the barriers)
        float* const priv = smem + 2 * W4_UBUF + 2 * W4_VBUF + pw * (2 * W4_HPLANE);
        // halo slot i of this lane: position lane + 64 i of the 10 x 18 halo -> (row, column); tile independent
        int hy[W4_NSLOT], hx[W4_NSLOT], lidx[W4_NSLOT];
#pragma unroll
        for (int i = 0; i < W4_NSLOT; ++i) {
            const int pos = lane + 64 * i;
            const bool valid = pos < W4_HRW * W4_HC;
            hy[i] = valid ? pos / W4_HC : -1000;           // (never live)
            hx[i] = valid ? pos - W4_HC * (pos / W4_HC) : 0;
            lidx[i] = valid ? (pos / W4_HC) * W4_HS + hx[i] : W4_HS - 1;   // (column 23 of row 0 is padding)
        }
        const int up_sh_h = d.up_h == 2 ? 1 : 0, up_sh_w = d.up_w == 2 ? 1 : 0;
        const bool circ = d.circular != 0;
        // ZP == 1 (2 x 2 up-sampled source): the 10 x 18 halo is 6 x 10 SOURCE pixels -- one per lane (lanes 0 .. 59), loaded and
        // normalised once (slot 0) and committed to the up to four halo cells it covers: 4 loads per stage and lane instead of 12
        constexpr int NSL = ZP == 1 ? 1 : W4_NSLOT;
        const int sj = lane / 10, si = lane - 10 * (lane / 10);
        int lidx4[4];
        {
            const int r0 = 2 * sj - 1 < 0 ? 0 : 2 * sj - 1, r1 = 2 * sj > W4_HRW - 1 ? W4_HRW - 1 : 2 * sj;
            const int q0 = 2 * si - 1 < 0 ? 0 : 2 * si - 1, q1 = 2 * si > W4_HC - 1 ? W4_HC - 1 : 2 * si;
            const bool sv = lane < 60;
            lidx4[0] = sv ? r0 * W4_HS + q0 : W4_HS - 1; lidx4[1] = sv ? r0 * W4_HS + q1 : W4_HS - 1;
            lidx4[2] = sv ? r1 * W4_HS + q0 : W4_HS - 1; lidx4[3] = sv ? r1 * W4_HS + q1 : W4_HS - 1;
        }
        // lane (t, e): tile t = lane & 31 = (ty, tx) = (t >> 3, t & 7), channel e = lane >> 5 of the wave's pair
        const int pe = lane >> 5;
        const int pbase = pe * W4_HPLANE + (2 * ((lane & 31) >> 3)) * W4_HS + 2 * (lane & 7);
        const int vwr = pw * W4_VKQ + lane * 2;                 // [k4 = pe][tile][h]: lane-linear 8-byte stores, no bank conflicts
        const unsigned lane16 = lane * 16u;

        // per-tile halo geometry of the issue cursor
        unsigned goff[W4_NSLOT], gstat[W4_NSLOT] = {0u, 0u, 0u}, glive = 0;   // gstat: byte offset of the pixel's LN statistics
        const float* gimg = d.x;
        const float* gctx = d.ctx;                         // context planes of the tile's image (NULL = none)
        auto geometry = [&](const W4Cur& t) {
            const int m_img = t.n + d.x_n_off;             // image -> (trajectory, window) for the sliding-window view
            const int m_out = d.n_inner == 1 ? m_img : m_img / d.n_inner;
            gimg = d.x + (int64_t)m_out * d.x_sn_outer + (int64_t)(m_img - m_out * d.n_inner) * d.x_sn_inner;
            if (d.cctx > 0) gctx = d.ctx + (int64_t)t.n * d.ctx_sn;
            glive = 0;
            if constexpr (ZP == 1) {
                const int sy0 = 4 * t.by - 1 + sj, sx0 = 8 * t.bx - 1 + si;          // source pixel of this lane
                const bool inside = sy0 >= 0 && sy0 < d.hs && sx0 >= 0 && sx0 < d.ws;
                const int syw = sy0 < 0 ? sy0 + d.hs : (sy0 >= d.hs ? sy0 - d.hs : sy0);
                const int sxw = sx0 < 0 ? sx0 + d.ws : (sx0 >= d.ws ? sx0 - d.ws : sx0);
                const bool ok = lane < 60 && (circ || inside);
                const int sy = ok ? syw : 0, sx = ok ? sxw : 0;
                goff[0] = (unsigned)(sy * (int)d.x_sy + sx * (int)d.x_sx) * 4u;
                glive = ok ? 1u : 0u;
                if constexpr (LN) gstat[0] = (unsigned)((t.n * d.hs + sy) * d.ws + sx) * 4u;
                return;
            }
#pragma unroll
            for (int i = 0; i < W4_NSLOT; ++i) {
                const int vy0 = 8 * t.by - 1 + hy[i], vx0 = 16 * t.bx - 1 + hx[i];
                const bool inside = vy0 >= 0 && vy0 < g.hv && vx0 >= 0 && vx0 < g.wv;
                const int vyw = vy0 < 0 ? vy0 + g.hv : (vy0 >= g.hv ? vy0 - g.hv : vy0);
                const int vxw = vx0 < 0 ? vx0 + g.wv : (vx0 >= g.wv ? vx0 - g.wv : vx0);
                const bool ok = (hy[i] >= 0) && (circ || inside);
                const int vy = ok ? vyw : 0, vx = ok ? vxw : 0;
                const int sy = vy >> up_sh_h, sx = vx >> up_sh_w;
                goff[i] = (unsigned)(sy * (int)d.x_sy + sx * (int)d.x_sx) * 4u;
                glive |= ok ? (1u << i) : 0u;
                if constexpr (LN) gstat[i] = (unsigned)((t.n * d.hs + sy) * d.ws + sx) * 4u;
            }
        };
        // what travels from the issue of a stage's halo loads to their commit
        // (mv: the modulation of the wave's two channels; mean / rstd: the LayerNorm statistics of the set's pixels -- loaded
        // WITH the set, by the same hand-counted asm loads: fetched where they are used, through compiler-visible loads, each
        // commit would drain every load the helper has in flight -- s_waitcnt vmcnt(0) inside the pause)
        struct Halo { float v[2][W4_NSLOT]; float mean[W4_NSLOT], rstd[W4_NSLOT], mv[2]; unsigned live; };
        // (vector-memory and scalar instructions only: runs beside the consumers' MFMAs)
        auto issue = [&](const W4Cur& t, Halo& h) {
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int cc = W4_CK * t.st + 2 * pw + ch;
                const int cce = cc < g.cin ? cc : 0;       // (padded channels read channel 0 and are zeroed at commit)
                const char* xc = reinterpret_cast<const char*>(gimg + (int64_t)cce * d.x_sc);
                if (cce >= d.cx) xc = reinterpret_cast<const char*>(gctx + (int64_t)(cce - d.cx) * (d.hs * d.ws));   // (wave uniform)
#pragma unroll
                for (int i = 0; i < NSL; ++i) w4_ld1(h.v[ch][i], xc, goff[i]);
                if constexpr (MOD) w4_ld1(h.mv[ch], reinterpret_cast<const char*>(d.mod + cce), 0u);
            }
            if constexpr (LN) {
#pragma unroll
                for (int i = 0; i < NSL; ++i) {
                    w4_ld1(h.mean[i], reinterpret_cast<const char*>(d.ln_mean), gstat[i]);
                    w4_ld1(h.rstd[i], reinterpret_cast<const char*>(d.ln_rstd), gstat[i]);
                }
            }
        };
        // the geometry the set's loads were issued with (a VALU copy: part of the pause work)
        auto tag = [&](Halo& h) { h.live = glive; };
        // wait until at most N of this wave's loads are outstanding; the set's registers pass through the statement, so that
        // nothing that reads them can be scheduled above it
#define W4_WAIT_HALO(N, h)                                                                                                     \
    do {                                                                                                                       \
        asm volatile("s_waitcnt vmcnt(%6)" : "+v"((h).v[0][0]), "+v"((h).v[0][1]), "+v"((h).v[0][2]), "+v"((h).v[1][0]),         \
                     "+v"((h).v[1][1]), "+v"((h).v[1][2]) : "n"((N) > 63 ? 63 : (N)) : "memory");   /* (6-bit counter: 63 is stricter) */ \
        if constexpr (LN) asm volatile("" : "+v"((h).mean[0]), "+v"((h).mean[1]), "+v"((h).mean[2]), "+v"((h).rstd[0]),        \
                                       "+v"((h).rstd[1]), "+v"((h).rstd[2]) :: "memory");                                      \
        if constexpr (MOD) asm volatile("" : "+v"((h).mv[0]), "+v"((h).mv[1]) :: "memory");                                    \
    } while (0)
        // a launch without loader fusions whose halo positions and channels all carry data (circular padding, cin % 8 == 0 --
        // every backward-data convolution of the reference nets) commits with plain LDS stores: no VALU, so the commit moves
        // into the part of the iteration that runs beside the MFMAs
        const bool raw_commit = !MOD && !LN && !SILU && circ && (g.cin % W4_CK) == 0;
        auto commit_raw = [&](const Halo& h) {
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int i = 0; i < W4_NSLOT; ++i) priv[ch * W4_HPLANE + lidx[i]] = h.v[ch][i];
        };
        auto commit = [&](const W4Cur& t, const Halo& h) {
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int cc = W4_CK * t.st + 2 * pw + ch;
                const bool real = cc < g.cin;              // wave uniform (false only in a partial last stage)
                if constexpr (ZP == 1) {
                    float v = h.v[ch][0];
                    if constexpr (MOD) v += h.mv[ch];
                    if constexpr (LN) v = (v - h.mean[0]) * h.rstd[0];
                    if constexpr (SILU) v = sda_act(SDA_ACT_SILU, v);
                    v = (real && (h.live & 1u)) ? v : 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) priv[ch * W4_HPLANE + lidx4[k]] = v;
                    continue;
                }
                float val[W4_NSLOT];
#pragma unroll
                for (int i = 0; i < W4_NSLOT; ++i) {
                    float v = h.v[ch][i];
                    if constexpr (MOD) v += h.mv[ch];
                    if constexpr (LN) v = (v - h.mean[i]) * h.rstd[i];
                    if constexpr (SILU) v = sda_act(SDA_ACT_SILU, v);
                    val[i] = v;
                }
                if (circ && real) {
                    // every halo position carries data (a lane's slots beyond the 180 positions land on the padding cell)
#pragma unroll
                    for (int i = 0; i < W4_NSLOT; ++i) priv[ch * W4_HPLANE + lidx[i]] = val[i];
                } else {
                    // padding / out-of-range positions and padded channels stage zeros
                    const unsigned keep = real ? h.live : 0u;
#pragma unroll
                    for (int i = 0; i < W4_NSLOT; ++i) priv[ch * W4_HPLANE + lidx[i]] = ((keep >> i) & 1u) ? val[i] : 0.f;
                }
            }
        };
        // this lane's patch: LDS -> registers (no VALU) ...
        f32x2 plo[4], phi[4];
        auto patch_read = [&]() {
            const float* src = priv + pbase;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                plo[a] = *reinterpret_cast<const f32x2*>(src + a * W4_HS);
                phi[a] = *reinterpret_cast<const f32x2*>(src + a * W4_HS + 2);
            }
        };
        // ... -> B^T d B -> V[pair][pw][pe][t][h].  Sixteen packed adds (the compiler's own pairing needs a dozen moves on
        // top): rows on the column pairs the patch reads delivered, u0 = d0 - d2, u1 = d1 + d2, u2 = d2 - d1, u3 = d1 - d3;
        // columns with operand-half selects, (o0, o1) = (u0 - u2, u1 + u2) and (o2, o3) = (u2 - u1, u1 - u3) from the pairs
        // A = (u0, u1), B = (u2, u3).
        auto transform = [&](float* vb) {
            f32x2 ua[4], ub[4];                            // [row] -> columns (0, 1) and (2, 3)
            ua[0] = plo[0] - plo[2]; ub[0] = phi[0] - phi[2];
            ua[1] = plo[1] + plo[2]; ub[1] = phi[1] + phi[2];
            ua[2] = plo[2] - plo[1]; ub[2] = phi[2] - phi[1];
            ua[3] = plo[1] - plo[3]; ub[3] = phi[1] - phi[3];
            float* dst = vb + vwr;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (ZPOS && a == 2) continue;              // (positions 8 .. 11 are never read)
                f32x2 o01, o23;
                asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(o01) : "v"(ua[a]), "v"(ub[a]));
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(o23) : "v"(ub[a]), "v"(ua[a]));
                if constexpr (VAR == 14 || VAR == 15) {
                    // (tooling) what splitting a V value into three bf16 pieces costs the helpers: 8 VALU instructions per value
                    // (round hi, subtract, round mid, subtract, round lo, two packs), here as 8 dependent dummies per value
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        float t0 = o01[e], t1 = o23[e];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            asm volatile("v_and_b32 %0, 0xffff0000, %0\n\tv_sub_f32 %0, %1, %0" : "+v"(t0) : "v"(o01[e]));
                            asm volatile("v_and_b32 %0, 0xffff0000, %0\n\tv_sub_f32 %0, %1, %0" : "+v"(t1) : "v"(o23[e]));
                        }
                        o01[e] = t0; o23[e] = t1;
                    }
                }
                *reinterpret_cast<f32x2*>(dst + (2 * a + 0) * W4_VPP) = o01;       // positions 4 a + 0, 4 a + 1
                *reinterpret_cast<f32x2*>(dst + (2 * a + 1) * W4_VPP) = o23;       // positions 4 a + 2, 4 a + 3
            }
        };
        // U slab, position pairs 2 pw and 2 pw + 1 of a stage: 4 MF KiB = 4 MF wave-wide dwordx4 (one per cout fragment and pair)
        f32x4 ureg[12];                                    // (MF = 2 uses the first 8; W4_WAIT_U names all twelve)
        if constexpr (MF != 3) {
#pragma unroll
            for (int m = NUF; m < 12; ++m) ureg[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // ZP: 9 of the 16 positions are multiplied; the slab comes position-packed (W4_UZP: 28 KiB instead of the 36 KiB of the six pairs
        // that hold a live position) and in the stage buffer's own order -- helper pw copies the linear pieces 7 pw .. 7 pw + 6.  The
        // up-sampled / pooled layers are vector-memory bound on the helper side (two workgroups' U, halo and operand loads share the CU's
        // 64 B/clk address path): dropping three of the former nine loads per helper measured -16.5 / -19 % per layer
        // (profiles/r04_zp_packed_u.txt).
        auto u_load = [&](const W4Cur& t) {
            const int64_t pstride = (int64_t)g.mtiles * 1024;         // bytes between position pairs
            // (the immediate offset field reaches 4095: one scalar base per three consecutive 1-KiB pieces)
            if constexpr (ZPOS) {
                const char* b0 = reinterpret_cast<const char*>(d.w_wino4_zp + ((int64_t)t.st * g.n_ct + t.ct) * W4_UZP) + pw * (NULZ * 1024);
                const char* b1 = b0 + 3072;
                w4_ld4<0>(ureg[0], b0, lane16);
                w4_ld4<1024>(ureg[1], b0, lane16);
                w4_ld4<2048>(ureg[2], b0, lane16);
                if constexpr (NULZ >= 5) {
                    w4_ld4<0>(ureg[3], b1, lane16);
                    w4_ld4<1024>(ureg[4], b1, lane16);
                }
                if constexpr (NULZ == 7) {
                    const char* b2 = b0 + 6144;
                    w4_ld4<2048>(ureg[5], b1, lane16);
                    w4_ld4<0>(ureg[6], b2, lane16);
                }
                return;
            }
            const char* src = reinterpret_cast<const char*>(d.w_wino4 + ((int64_t)(t.st * 8 + 2 * pw) * g.mtiles + 2 * MF * t.ct) * 256);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const char* sp = src + pp * pstride;
                w4_ld4<0>(ureg[pp * 2 * MF + 0], sp, lane16);
                w4_ld4<1024>(ureg[pp * 2 * MF + 1], sp, lane16);
                if constexpr (MF >= 2) w4_ld4<2048>(ureg[pp * 2 * MF + 2], sp, lane16);
                if constexpr (MF == 3) {
                    const char* sq = sp + 3072;
                    w4_ld4<0>(ureg[pp * 6 + 3], sq, lane16);
                    w4_ld4<1024>(ureg[pp * 6 + 4], sq, lane16);
                    w4_ld4<2048>(ureg[pp * 6 + 5], sq, lane16);
                } else if constexpr (MF == 2) {
                    w4_ld4<3072>(ureg[pp * 4 + 3], sp, lane16);
                }
            }
        };
        // The same pieces by LDS-DMA (W4_UDMA): global_load_lds_dwordx4 writes [M0 + lane x 16], and both the packing and the stage
        // buffer are lane-linear 1-KiB pieces, so a piece goes straight to its place -- no registers, no ds_write_b128 (48 per stage
        // and workgroup through the VGPR -> LDS store path that the consumers' operand reads share), and hipcc does not see it (inline
        // asm: the waits are counted by hand like every other load of this wave).  Issued where u_store + u_load stood, for the stage
        // the consumers multiply NEXT, and waited for in front of that iteration's hand-off barrier.
        // (the instruction's immediate offset is added to the global AND the LDS address: three consecutive KiB per statement, one
        //  scalar base pair and one M0 value -- twelve separate bases spill scalars into the pause's vector work)
        auto u_dma3 = [&](const char* src, const float* dst) {
            const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) const float*)dst));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane16), "s"(src), "s"(la) : "memory");
        };
        auto u_dma1 = [&](const char* src, const float* dst) {
            const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) const float*)dst));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane16), "s"(src), "s"(la) : "memory");
        };
        auto u_dma = [&](const W4Cur& t, float* ub) {
            static_assert(!W4_UDMA || MF == 3, "the DMA form of the U slab exists for the 96-cout tile only");
            if constexpr (ZPOS) {
                const char* b0 = reinterpret_cast<const char*>(d.w_wino4_zp + ((int64_t)t.st * g.n_ct + t.ct) * W4_UZP) + pw * 7168;
                float* dz = ub + pw * 1792;
                u_dma3(b0, dz);
                u_dma3(b0 + 3072, dz + 768);
                u_dma1(b0 + 6144, dz + 1536);
                return;
            }
            const int64_t pstride = (int64_t)g.mtiles * 1024;
            const char* src = reinterpret_cast<const char*>(d.w_wino4 + ((int64_t)(t.st * 8 + 2 * pw) * g.mtiles + 6 * t.ct) * 256);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                u_dma3(src + pp * pstride, ub + (2 * pw + pp) * W4_UPP);
                u_dma3(src + pp * pstride + 3072, ub + (2 * pw + pp) * W4_UPP + 768);
            }
        };
        // (prologue only: NUL pieces onto the workgroup's scratch KiB behind the halo planes -- the static wait counts assume a second
        //  group of U loads there)
        auto u_dma_dummy = [&]() {
            const char* src = reinterpret_cast<const char*>(d.w_wino4);
#pragma unroll
            for (int m = 0; m < (ZPOS ? NULZ : NUF); ++m) u_dma1(src, smem + W4_LDS_BYTES / 4);
        };
#define W4_WAIT_U(N)                                                                                                           \
    asm volatile("s_waitcnt vmcnt(%14)" : "+v"(ureg[0]), "+v"(ureg[1]), "+v"(ureg[2]), "+v"(ureg[3]), "+v"(ureg[4]), "+v"(ureg[5]), \
                 "+v"(ureg[6]), "+v"(ureg[7]), "+v"(ureg[8]), "+v"(ureg[9]), "+v"(ureg[10]), "+v"(ureg[11]), "+v"(pfreg[0]),   \
                 "+v"(pfreg[1]) : "n"(N) : "memory");                                                                         \
    W4_PIN_EPI()
        auto u_store = [&](float* ub) {
            if constexpr (ZPOS) {
                float* dz = ub + pw * (NULZ * 256) + lane * 4;
#pragma unroll
                for (int m = 0; m < NULZ; ++m) *reinterpret_cast<f32x4*>(dz + m * 256) = ureg[m];
                return;
            }
            float* dst = ub + (2 * pw) * W4_UPP + lane * 4;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int m = 0; m < 2 * MF; ++m) *reinterpret_cast<f32x4*>(dst + pp * W4_UPP + m * 256) = ureg[pp * 2 * MF + m];
        };
        // hand-off: this wave's LDS writes have landed; its global loads stay in flight across the barrier
        auto handoff = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        // Epilogue operands (the residual / the activation-derivative input: one value per output element, read once, by the
        // consumers, with nothing to hide the HBM round trip behind) are pulled into the XCD's L2 ahead of time: while the
        // issue cursor is in the last three stages of a tile (the consumers are then 5 stages behind), each helper touches
        // one dword of 64 of the tile's 768 64-byte row segments per iteration (lane -> cout plane 8 pw + (lane >> 3) (+ 32
        // per stage), row lane & 7).  The destination registers are dummies (two, alternating: a load has retired by the U
        // wait two iterations later, where both are pinned); launches without such an operand issue the load all the same,
        // on the weights, so that the hand-counted vmcnt values do not depend on the launch.
        const float* const pf_t = d.res ? d.res : d.dact_z;
        const unsigned pf_off = (unsigned)(((pw * 8 + (lane >> 3)) * (d.ho * d.wo) + (lane & 7) * d.wo) * 4);
        constexpr int NPF = EPI ? 4 : 1;                   // loads per iteration: the operand window's (EPI) / one prefetch touch
        float pfreg[2 * NPF] = {0.f, 0.f};
        // EPI launches: the helpers LOAD the operand instead (wave pw for consumer wave pw, lane for lane: the 24 8-byte pairs
        // the consumer lane's epilogue needs -- plane 16 m + r, rows 0 / 1 of its 2 x 2 output block), four pairs per iteration
        // while the issue cursor is in the tile's last six stages, into 48 registers that wait for the tile's end.  There the
        // pairs go to LDS -- the U buffer the consumers released at the tile's last hand-off, [wave][pair 24][lane] x 8 bytes,
        // lane-linear both ways -- between two extra barriers: X1 (operand in LDS; the consumers arrive after the first
        // inverse transform) and X2 (operand read; the helper refills the buffer).  A consumer-side global load would queue
        // behind everything the helpers have in flight on the CU's vector-memory path: ~1 us per dependent round trip, three
        // of them per tile.  Outside the window the four loads are issued all the same (dummies on the weights: static counts).
        constexpr int NEOP = 8 * MF;
        f32x2 eop[EPI ? NEOP : 1];
#pragma unroll
        for (int j = 0; j < (EPI ? NEOP : 1); ++j) eop[j] = f32x2{0.f, 0.f};
        const int e_hw = d.ho * d.wo;
        const int e_t = 16 * (pw & 1) + (lane & 15);
        const unsigned e_lo0 = (unsigned)(((4 * (lane >> 4)) * e_hw + 2 * (e_t >> 3) * d.wo + 2 * (e_t & 7)) * 4);
        const unsigned e_lo1 = e_lo0 + (unsigned)d.wo * 4u;
        const int e_first = g.nstage - 6;                  // six window slots (stages) of a tile's operand loads
#define W4_PIN_EPI()                                                                                                           \
    do {                                                                                                                       \
        if constexpr (EPI) {                                                                                                   \
            asm volatile("" : "+v"(pfreg[2]), "+v"(pfreg[3]), "+v"(pfreg[4]), "+v"(pfreg[5]), "+v"(pfreg[6]), "+v"(pfreg[7]) :: "memory"); \
            asm volatile("" : "+v"(eop[0]), "+v"(eop[1]), "+v"(eop[2]), "+v"(eop[3]), "+v"(eop[4]), "+v"(eop[5]), "+v"(eop[6]),     \
                         "+v"(eop[7]), "+v"(eop[8]), "+v"(eop[9]), "+v"(eop[10]), "+v"(eop[11]) :: "memory");                        \
            asm volatile("" : "+v"(eop[12]), "+v"(eop[13]), "+v"(eop[14]), "+v"(eop[15]), "+v"(eop[16]), "+v"(eop[17]),            \
                         "+v"(eop[18]), "+v"(eop[19]), "+v"(eop[20]), "+v"(eop[21]), "+v"(eop[22]), "+v"(eop[23]) :: "memory");     \
        }                                                                                                                      \
    } while (0)
        // One straight-line statement per window slot K (its four loads are skipped INSIDE the asm text unless k == K): a C++
        // switch would put the 24 destination registers through phi copies -- vector moves of registers with loads in flight.
#define W4_EPI_SLOT(K)                                                                                                         \
    asm volatile("s_cmp_lg_u32 %4, " #K "\n\ts_cbranch_scc1 .Lw4epi%=\n\ts_nop 4\n\t"                                           \
                 "global_load_dwordx2 %0, %5, %7\n\tglobal_load_dwordx2 %1, %6, %7\n\t"                                        \
                 "global_load_dwordx2 %2, %5, %8\n\tglobal_load_dwordx2 %3, %6, %8\n.Lw4epi%=:"                                 \
                 : "+v"(eop[4 * K + 0]), "+v"(eop[4 * K + 1]), "+v"(eop[4 * K + 2]), "+v"(eop[4 * K + 3])                      \
                 : "s"(k), "v"(e_lo0), "v"(e_lo1), "s"(b0), "s"(b1) : "memory", "scc")
        auto epi_issue = [&](const W4Cur& t, float& dm0, float& dm1, float& dm2, float& dm3) {
            if constexpr (EPI) {
                const int k = t.st - e_first;              // window slot: planes 16 (k >> 1) + 2 (k & 1) and the next one
                const int kc = k < 0 ? 0 : k;
                const int64_t ps = (int64_t)e_hw * 4;
                const char* b0 = reinterpret_cast<const char*>(pf_t + ((int64_t)t.n * d.cout + W4_BM * t.ct + 48 * (pw >> 1)) * e_hw +
                                                               (8 * t.by) * d.wo + 16 * t.bx) + (16 * (kc >> 1) + 2 * (kc & 1)) * ps;
                const char* b1 = b0 + ps;
                W4_EPI_SLOT(0); W4_EPI_SLOT(1); W4_EPI_SLOT(2); W4_EPI_SLOT(3); W4_EPI_SLOT(4); W4_EPI_SLOT(5);
                // outside the window: four dummy loads on the weights (the hand-counted vmcnt values are static)
                asm volatile("s_cmp_lt_u32 %4, 6\n\ts_cbranch_scc1 .Lw4epi%=\n\ts_nop 4\n\t"
                             "global_load_dword %0, %5, %6\n\tglobal_load_dword %1, %5, %6\n\t"
                             "global_load_dword %2, %5, %6\n\tglobal_load_dword %3, %5, %6\n.Lw4epi%=:"
                             : "+v"(dm0), "+v"(dm1), "+v"(dm2), "+v"(dm3)
                             : "s"(k), "v"(lane16), "s"(reinterpret_cast<const char*>(d.w_wino4)) : "memory", "scc");
            }
        };
        // the tile's operand: registers -> LDS (the U buffer of stage q + 1, free since the hand-off E_(q-1)), X1, X2
        auto epi_store = [&](int q) {
            if constexpr (EPI) {
                float* rb = ubuf + ((q + 1) & 1) * W4_UBUF + (pw * NEOP * 64 + lane) * 2;
#pragma unroll
                for (int j = 0; j < NEOP; ++j) *reinterpret_cast<f32x2*>(rb + j * 128) = eop[j];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();              // X1
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();              // X2
                asm volatile("" ::: "memory");
            }
        };
        int cst = 0;                                       // stage-in-tile of the consumers' stage q
        const int pf_first = g.nstage > MF ? g.nstage - MF : 0;
        auto prefetch = [&](const W4Cur& t, float& dst) {
            const int k = t.st - pf_first;
            if (pf_t && k >= 0 && k < MF) {
                const float* base = pf_t + ((int64_t)t.n * d.cout + W4_BM * t.ct + 32 * k) * ((int64_t)d.ho * d.wo) +
                                    (8 * t.by) * d.wo + 16 * t.bx;
                w4_ld1(dst, reinterpret_cast<const char*>(base), pf_off);
            } else {
                w4_ld1(dst, reinterpret_cast<const char*>(d.w_wino4), lane16);
            }
        };
        // cursors: stage q + 2 (U load, halo commit) and stage q + 5 (halo issue), clamped to the last stage (which is then
        // produced again into buffers nobody reads)
        W4Cur c2 = c0, ci = c0;
        int q2 = 0, qi = 0;                                // stage indices of c2, ci
        auto step2 = [&]() { if (q2 + 1 < Q) { w4_advance(g, c2); ++q2; } };
        bool new_tile = false;
        auto step_issue_cursor = [&]() {                   // scalar only
            new_tile = false;
            if (qi + 1 < Q) {
                w4_advance(g, ci); ++qi;
                new_tile = ci.st == 0;                     // (one stage per tile: every time)
            }
        };
        auto step_issue = [&]() {
            step_issue_cursor();
            if (new_tile) geometry(ci);
        };
        W4_TRACE_DECL;
        constexpr int NHL = 2 * NSL + (LN ? 2 * NSL : 0) + (MOD ? 2 : 0);   // loads per halo set (+ its LN / modulation operands)
        constexpr int NUL = ZPOS ? NULZ : NUF;             // loads per helper's share of the U slab (NPF above: per prefetch / operand window slot)
        // ---- prologue: V and U of stage 0 into the buffers 0; the halo of stage 1 committed; U of stage 1 and the halo sets
        // of stages 2, 3, 4 in flight
        Halo h0, h1, h2, h3;
        geometry(ci);
        issue(ci, h0); tag(h0); step_issue();
        issue(ci, h1); tag(h1); step_issue();
        issue(ci, h2); tag(h2); step_issue();
        issue(ci, h3); tag(h3);
#if W4_UDMA
        u_dma(c2, ubuf);                                   // stage 0's U slab -> buffer 0
#else
        u_load(c2);
#endif
        W4_WAIT_HALO(3 * NHL + NUL, h0);
        commit(c2, h0);
        patch_read();
        transform(vbuf);
#if W4_UDMA
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        step2();
        W4Cur cu = c2;                                     // cursor of the stage whose U slab the next iteration copies (stage q + 1)
        u_dma_dummy();
#else
        W4_WAIT_U(0);
        u_store(ubuf);
        step2();
        u_load(c2);
#endif
        W4_WAIT_HALO(NUL, h1);
        commit(c2, h1);
        step2();
        step_issue();
        if constexpr (EPI) epi_issue(ci, pfreg[4], pfreg[5], pfreg[6], pfreg[7]);
        else prefetch(ci, pfreg[1]);
        issue(ci, h0); tag(h0);
        handoff();
        // One helper iteration, while the consumers multiply stage q.  The fp32 MFMA stream owns the SIMD's vector ALU: a
        // sibling's VALU instruction does not issue at all until the stream pauses (tools/mfma_shadow_gen.py, p_* / x_* rows),
        // while its LDS, scalar and vector-memory instructions do.  A stage therefore has TWO workgroup barriers:
        //   M_q  after the consumers' step 3: the deliberate pause.  The helpers arrive with their VALU work queued -- B^T d B
        //        of stage q + 1 (patch already in registers), the loader fusions of stage q + 2, the cursor / geometry update --
        //        which executes while the consumers wait here; its LDS writes drain behind the MFMAs of steps 4-6.
        //   E_q  after the consumers' step 6: the hand-off of the stage q + 1 buffers.  Nobody waits at it in the steady state
        //        (the helpers' remaining work is vector memory only), and the consumers fetch the first operands of stage q + 1
        //        under the MFMAs of step 7 -- no LDS round trip between stages.
        // Before M_q the helper runs what needs no VALU: U registers -> LDS, the U loads of stage q + 2, the patch reads of
        // stage q + 1; between M_q and E_q the halo loads of stage q + 5.
        auto iteration = [&](int q, Halo& hcommit, Halo& hissue, auto PAR_) {
            constexpr int PAR = decltype(PAR_)::value;
            float* ub = ubuf + ((q + 1) & 1) * W4_UBUF;
            float* vb = vbuf + ((q + 1) & 1) * W4_VBUF;
            if constexpr (EPI) {
                if (cst == 0 && q > 0) epi_store(q);       // the consumers are in the epilogue of the tile that just ended
                const int wrap = w4_eq(cst + 1, g.nstage);
                cst = (cst + 1) * (1 - wrap);
            }
            W4_T0();
#if W4_UDMA
            // (at most the NHL + NPF loads of the previous iteration's tail are outstanding here: that iteration waited for exactly that
            //  in front of its hand-off.  The dummy / operand registers pass through a statement, as at the former U wait.)
            asm volatile("" : "+v"(pfreg[0]), "+v"(pfreg[1]) :: "memory");
            W4_PIN_EPI();
            W4_STAMP(0);
            __builtin_amdgcn_sched_barrier(0);
            if (!W4_DBG(256)) u_dma(cu, ub);               // stage q + 1's U slab -> the buffer the hand-off E_(q-1) released
            __builtin_amdgcn_sched_barrier(0);
            cu = c2;
#else
            // the U registers were loaded one iteration ago, before that iteration's halo loads
            W4_WAIT_U(NHL + NPF);
            if (!W4_DBG(16)) u_store(ub);
            if constexpr (VAR == 11) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            W4_STAMP(0);                                       // U registers -> LDS (incl. the wait for their loads)
            __builtin_amdgcn_sched_barrier(0);
            if (!W4_DBG(256)) u_load(c2);                  // (the hand-written wait counts assume both load groups)
            __builtin_amdgcn_sched_barrier(0);
#endif
            // cursor arithmetic (scalar, ~60 instructions): here, beside the MFMAs, not in the pause
            const W4Cur ccommit = c2;
            step2();
            step_issue_cursor();
            __builtin_amdgcn_sched_barrier(0);
            if (!W4_DBG(64)) patch_read();   // halo of stage q + 1 (committed during the previous iteration)
            __builtin_amdgcn_sched_barrier(0);
            // the set committed now (stage q + 2) was issued three iterations ago; behind it: the (U + prefetch + halo) loads
            // of two iterations and this iteration's U loads
            W4_WAIT_HALO(2 * (NHL + NPF + NUL) + NUL, hcommit);
            if (raw_commit && !W4_DBG(128)) commit_raw(hcommit);
            W4_STAMP(1);                                       // U loads + patch reads (+ the VALU-free commit)
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long mk0 = 0, mk1 = 0, mk2 = 0, mk3 = 0;
            W4_MARK(mk0);
            __builtin_amdgcn_sched_barrier(0);
            if (!W4_DBG(32)) transform(vb);
            __builtin_amdgcn_sched_barrier(0);
            W4_MARK(mk1);
            __builtin_amdgcn_sched_barrier(0);
            if (!raw_commit) commit(ccommit, hcommit);
            if (new_tile) geometry(ci);
            tag(hissue);
            __builtin_amdgcn_sched_barrier(0);
            W4_MARK(mk2);
            W4_STAMP(2);                                       // the VALU part
            __builtin_amdgcn_sched_barrier(0);
            if (!W4_DBG(512)) __builtin_amdgcn_s_barrier();                  // M_q
            asm volatile("" ::: "memory");
            if constexpr (VAR == 11) if SDA_DBG(g, 8192) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(mk3), "+s"(mk0), "+s"(mk1), "+s"(mk2) :: "memory");
            W4_MARK_ADD(0, mk0, mk1); W4_MARK_ADD(1, mk1, mk2); W4_MARK_ADD(2, mk2, mk3);
            W4_STAMP(3);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (EPI) epi_issue(ci, pfreg[4 * PAR], pfreg[4 * PAR + 1], pfreg[4 * PAR + 2], pfreg[4 * PAR + 3]);
            else prefetch(ci, pfreg[PAR]);
            if (!W4_DBG(256)) issue(ci, hissue);
            __builtin_amdgcn_sched_barrier(0);
            W4_STAMP(4);                                       // halo loads
            // (tooling: what an LDS-DMA of the U slab would impose -- vmcnt retires in order, so waiting for a DMA issued at the top of
            //  this iteration also waits for every halo / operand load of the earlier iterations; only this iteration's may stay in flight)
            if (W4_UDMA || W4_DBG(1024)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NHL + NPF) > 63 ? 63 : (NHL + NPF)) : "memory");
            handoff();                                     // E_q
            W4_STAMP(5);
        };
        for (int q = 0; q < Q; q += 4) {
            iteration(q, h2, h1, std::integral_constant<int, 0>{});
            if (q + 1 < Q) iteration(q + 1, h3, h2, std::integral_constant<int, 1>{});
            if (q + 2 < Q) iteration(q + 2, h0, h3, std::integral_constant<int, 0>{});
            if (q + 3 < Q) iteration(q + 3, h1, h0, std::integral_constant<int, 1>{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (EPI) { W4_PIN_EPI(); epi_store(Q); }     // the last tile's operand
        W4_TRACE_OUT();
        return;
    }

    // ====================================================================== consumers (MFMA + LDS reads only in the loop)
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, kq = lane >> 4;
    // A fragments of (position pair s, cout fragment m): U[s][3 wm + m][lane][h][k4] -- one ds_read_b128 = the operands of
    // positions 2 s, 2 s + 1 and both K quads;  B fragments: V[s][kq][k4][16 wn + li][h], two 8-byte reads (one ds_read2_b64).  (ds_read_b128 moves 256 B per LDS
    // clock, ds_read2_b64 half of that: tools/w4_feed_gen.py -- the cheaper read is also worth ~5 % of shader clock here,
    // the kernel runs at the power limit.)
    int ard = (MF * wm * 64 + lane) * 4;
    int brd = kq * W4_VKQ + (16 * wn + li) * 2;
    f32x4 acc[16][MF];
    if constexpr (ZPOS) {                                  // (never written: constants for the epilogue, no registers)
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int m = 0; m < MF; ++m)
                if (w4_dead<ZP>(p)) acc[p][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    W4_TRACE_DECL;
    __syncthreads();                                       // stage 0 is in buffer 0
    int q = 0;
    W4_T0();
    // one K-stage: two positions per step -- 8 LDS reads, then 12 MFMAs; the operands of step s + 1 are read during step s,
    // those of the NEXT stage's step 0 during step 7 (after the hand-off barrier E_q, see the helpers).
    // FIRST: the tile's first stage starts its accumulators from the C operand instead of reading them -- zero, or the bias
    // for position p = 5 = (xi, nu) = (1, 1): A^T e_11 A is the all-ones 2 x 2 block, so a bias placed there comes out of the
    // inverse transform added to every output pixel (no zero-fill and no bias adds in the epilogue).
    f32x4 av[2][MF];
    f32x2 av7[MF];                                         // (ZP: step 7's A operands, position 15 alone)
    f32x2 bv[2][2];                                        // [buffer][k4] -> (h = 0, h = 1)
    auto fetch = [&](int qq, int s, int buf) {
        const float* ua = ubuf + (qq & 1) * W4_UBUF + ard;
        const float* va = vbuf + (qq & 1) * W4_VBUF + brd;
        // (volatile: keeps two ds_read_b64 with immediate offsets; merged into a ds_read2_b64 they would need a vector add
        // per step for the base -- the pair stride exceeds read2's offset range -- and read at half the LDS rate)
        typedef const volatile f32x2 __attribute__((address_space(3))) * lds_cv2;
        bv[buf][0] = *(lds_cv2)(va + s * W4_VPP);
        bv[buf][1] = *(lds_cv2)(va + s * W4_VPP + 64);
        if constexpr (ZPOS) {
            // the position-packed slab (W4_UZP): steps 0 / 2 / 6 read their pair, steps 1 and 3 the block that interleaves their live
            // positions 3 and 7 ((k4 0, k4 1) of position 3, then of position 7: step 3 finds its operands where a full pair has them),
            // step 7 the 8-byte values of position 15
            if (s == 7) {
                const float* u7 = ubuf + (qq & 1) * W4_UBUF + 4 * W4_UPP + (ard >> 1);
#pragma unroll
                for (int m = 0; m < MF; ++m) av7[m] = *reinterpret_cast<const f32x2*>(u7 + m * 128);
            } else {
                const int off = s == 0 ? 0 : (s == 2 ? W4_UPP : (s == 6 ? 2 * W4_UPP : 3 * W4_UPP));
#pragma unroll
                for (int m = 0; m < MF; ++m) av[buf][m] = *reinterpret_cast<const f32x4*>(ua + off + m * 256);
            }
            return;
        }
#pragma unroll
        for (int m = 0; m < MF; ++m) av[buf][m] = *reinterpret_cast<const f32x4*>(ua + s * W4_UPP + m * 256);
    };
    fetch(0, 0, 0);
    auto stage = [&](auto FIRST_, const f32x4 (&binit)[MF]) {
        constexpr bool FIRST = decltype(FIRST_)::value;
        w4_static_for<0, 8>([&](auto S) {
            constexpr int s = decltype(S)::value;
            // ZP: steps 4 and 5 (positions 8 .. 11: xi == 2) do not exist, steps 1, 3, 7 run their second position only; the
            // executed steps 0 1 2 3 6 7 still alternate between the two operand buffers
            if constexpr (ZPOS && (s == 4 || s == 5)) return;
            constexpr int sn = (ZPOS && s == 3) ? 6 : s + 1;
            constexpr int nmfma = (w4_dead<ZP>(2 * s) ? 0 : 2 * MF) + (w4_dead<ZP>(2 * s + 1) ? 0 : 2 * MF);
            if constexpr (sn < 8) fetch(q, sn, sn & 1);
            else fetch(q + 1, 0, 0);
            // MFMA order (k4, m, h): the two MFMAs of one accumulator are SIX apart.  (Round 2 ran (m, k4, h) -- two apart, so that
            // cout block m's fragments were needed late in the step; a dependent fp32 MFMA one instruction behind its producer
            // loses a few cycles to the accumulator hand-over: 1.0-1.7 % per layer, tools/w4_quick_bench.py.  The operands of a
            // step are all read during the previous one, so nothing is needed "late".)
#pragma unroll
            for (int k4 = 0; k4 < 2; ++k4)
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int p = 2 * s + h;
                        if (w4_dead<ZP>(p)) continue;
                        // (tooling, profiles/r04_bf16x6_prototype.txt: the pipeline with a 2x / 3x cheaper multiply -- results are wrong)
                        if constexpr (VAR == 12 || VAR == 14) { if (k4 == 1) continue; }
                        if constexpr (VAR == 13 || VAR == 15) { if (k4 == 1 || m == 2) continue; }
                        f32x4 c;
                        if (FIRST && k4 == 0) c = (p == 5) ? binit[m] : f32x4{0.f, 0.f, 0.f, 0.f};
                        else c = acc[p][m];
                        const float aop = (ZPOS && s == 7) ? av7[m][k4] : ((ZPOS && s == 1) ? av[s & 1][m][k4] : av[s & 1][m][2 * h + k4]);
                        acc[p][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop, bv[s & 1][k4][h], c, 0, 0, 0);
                    }
            // pin the software pipeline: the four LDS reads of the NEXT step are spread between this step's MFMAs (a read
            // issued right behind an MFMA costs the stream nothing, a group of four ~12 cycles) -- except in step 6, whose
            // reads must have returned at the hand-off barrier that follows it
            if constexpr (MF == 1) {
                // 32-cout tile: 3 LDS reads (1 x b128 of U, 2 x b64 of V) per 4 (full step) / 2 (half step) MFMAs
                if constexpr (s == 6) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                } else if constexpr (nmfma == 2) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
            } else if constexpr (MF == 2) {
                // 64-cout tile: 4 LDS reads (2 x b128 of U, 2 x b64 of V) per 8 (full step) / 4 (half step) MFMAs, spread alike
                if constexpr (s == 6) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                } else if constexpr (nmfma == 4) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                } else {
                    // (round 6, profiles/r06_mf2_sched_variants.txt: reads bunched early (1-2-1-1-1-1-5), all four behind the first MFMA,
                    //  or one per MFMA measured within +-1 % of this spread on every layer type)
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                }
            } else if constexpr (s == 6) {
                __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            } else if constexpr (nmfma == 6) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // inside a tile only the stage index of the consumers' cursor moves (the carries into cout tile / block happen in
            // w4_advance behind the epilogue)
            if constexpr (!FIRST && s == 0) c0.st += 1;
            if constexpr (s == 3) {
                // M_q: the helpers' vector-ALU work runs during this wait
                W4_STAMP(0);
                unsigned long long mc0 = 0, mc1 = 0;
                W4_MARK(mc0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (VAR == 11) if SDA_DBG(g, 8192) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(mc1), "+s"(mc0) :: "memory");
                W4_MARK_ADD(1, mc0, mc1);
                W4_STAMP(1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (s == 6) {
                // E_q: every operand of stage q is in registers (step 7's were read during step 6), so the helpers may
                // overwrite its buffers; stage q + 1 is complete in the other pair
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                W4_STAMP(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                W4_STAMP(3);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (s == 7) W4_STAMP(0);
        });
        ++q;
    };
    for (int tl = 0; tl < my_tiles; ++tl) {
        // bias of this lane's couts: 32 MF ct + 16 MF wm + 16 m + 4 kq + (0..3)
        // (per-lane addresses are rebuilt from a lane id the compiler cannot hoist out of the tile loop: kept live across the
        // multiply they are spilled, and a consumer's scratch reload queues behind everything the helpers have in flight)
        f32x4 binit[MF];
        int lane_b;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_b));
        if constexpr (EPM == 2) {
            // (the generic epilogue is the register-hungriest: the two LDS read offsets are rebuilt per tile as well, instead of
            // being spilled around it -- the next tile's first operands were fetched before the epilogue, from the old copies)
            ard = (MF * wm * 64 + lane_b) * 4;
            brd = (lane_b >> 4) * W4_VKQ + (16 * wn + (lane_b & 15)) * 2;
        }
#pragma unroll
        for (int m = 0; m < MF; ++m) {
            binit[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (d.bias) binit[m] = *reinterpret_cast<const f32x4*>(d.bias + W4_BM * c0.ct + 16 * MF * wm + 16 * m + 4 * (lane_b >> 4));
        }
        stage(std::true_type{}, binit);
        // 64- / 32-cout tiles (EPM 2): the epilogue operand -- one value per output element, act'(z)'s z if there is one, else the residual
        // -- is REQUESTED IN FRONT OF THE TILE'S LAST STAGE into registers this tile size has to spare (16 MF of them), so that its
        // round trips (two dependent ones per tile, ~1 us each behind the helpers' traffic: the whole cost of a residual on a 64-channel
        // layer, 16 %) run under that stage's multiplies.  The 96-cout tile has no registers for it (and feeds the operand through the
        // helpers instead).
        constexpr bool PRE = EPM == 2 && MF < 3;
        f32x2 pre0[PRE ? MF : 1][4], pre1[PRE ? MF : 1][4];
        for (int st = 1; st < g.nstage - (PRE ? 1 : 0); ++st) {
            stage(std::false_type{}, binit);               // (advances c0: it ends on the tile's last stage, the epilogue's tile)
        }
        if constexpr (PRE) {
            // (c0 names this tile already: only its stage index moves inside a tile)
            const int hw_p = d.ho * d.wo;
            int lane_p;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_p));
            const int tp_ = 16 * wn + (lane_p & 15);
            const int lo0p = ((4 * (lane_p >> 4)) * hw_p + (8 * c0.by + 2 * (tp_ >> 3)) * d.wo + 16 * c0.bx + 2 * (tp_ & 7)) * 4, lo1p = lo0p + d.wo * 4;
            const float* pp = d.dact_z ? d.dact_z : d.res;
            const auto r_pre = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pp + ((int64_t)c0.n * d.cout + W4_BM * c0.ct + 16 * MF * wm) * hw_p),
                                                                 (short)0, 16 * MF * hw_p * 4, 0x00020000);
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pre0[m][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_pre, lo0p, (16 * m + r) * hw_p * 4, 0));
                    pre1[m][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_pre, lo1p, (16 * m + r) * hw_p * 4, 0));
                }
            __builtin_amdgcn_sched_barrier(0);
            if (g.nstage > 1) stage(std::false_type{}, binit);  // the tile's last stage
        }
        {
            // ---- epilogue of tile c0: Y = A^T M A per (cout, tile), lane local.  acc[4 xi + nu][m][r]:
            //      cout = 32 MF ct + 16 MF wm + 16 m + 4 kq + r,  tile = 16 wn + li.  The arithmetic runs on the f32x4 fragments
            //      (four couts at once: register pairs -> packed adds, no shuffling); memory goes through buffer
            //      instructions: one descriptor per tile (this wave's 48 cout planes of image n), a per-lane byte offset
            //      and a scalar offset per cout -- no 64-bit vector address arithmetic.
            const W4Cur& tt = c0;
            const int hw_o = d.ho * d.wo;
            int lane_e;                                    // (rebuilt, as for the bias)
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
            const int li_e = lane_e & 15, kq_e = lane_e >> 4;
            const int t = 16 * wn + li_e;
            const int oy = 8 * tt.by + 2 * (t >> 3), ox = 16 * tt.bx + 2 * (t & 7);
            const int lo0 = ((4 * kq_e) * hw_o + oy * d.wo + ox) * 4, lo1 = lo0 + d.wo * 4;
            const int64_t sbase = ((int64_t)tt.n * d.cout + W4_BM * tt.ct + 16 * MF * wm) * hw_o;
            const int plane_bytes = 16 * MF * hw_o * 4;
            auto rsrc_of = [&](const float* p) {
                return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p + sbase), (short)0, plane_bytes, 0x00020000);
            };
            auto run = [&](auto DACT_, auto RES_) {
                // DACT_: 0 = none; 1 = SiLU' only (the EPI kernels: no run-time switch in the plane loop, so that the planes of a
                // fragment are independent exp / rcp chains the scheduler interleaves -- and the other activations' erf / expm1
                // expansions, 6 500 instructions, stay out of the hot kernels); 2 = any activation, switched at run time
                constexpr int DACT_KIND = decltype(DACT_)::value;
                constexpr bool DACT = DACT_KIND != 0, RES = decltype(RES_)::value;
                const auto r_out = rsrc_of(d.out);
                const auto r_z = rsrc_of(DACT ? d.dact_z : d.out);
                const auto r_res = rsrc_of(RES ? d.res : d.out);
                // EPI: the operand pairs of this lane in the U buffer the last stage released: [wave][pair 24][lane]
                const float* el = ubuf + ((q + 1) & 1) * W4_UBUF + (wave * 8 * MF * 64 + lane_e) * 2;
                f32x2 el0[MF > 1 ? 4 * (MF - 1) : 1], el1[MF > 1 ? 4 * (MF - 1) : 1];
#pragma unroll
                for (int m = 0; m < MF; ++m) {
                    if constexpr (EPM != 0) __builtin_amdgcn_sched_barrier(0);       // (one fragment at a time: register pressure)
                    // rows (xi): s0 = M0 + M1 + M2, s1 = M1 - M2 - M3 for each nu;  columns (nu): the same combination
                    f32x4 s0[4], s1[4];
#pragma unroll
                    for (int nu = 0; nu < 4; ++nu) {
                        s0[nu] = (acc[nu][m] + acc[4 + nu][m]) + acc[8 + nu][m];
                        s1[nu] = (acc[4 + nu][m] - acc[8 + nu][m]) - acc[12 + nu][m];
                    }
                    const f32x4 y00 = (s0[0] + s0[1]) + s0[2], y01 = (s0[1] - s0[2]) - s0[3];
                    const f32x4 y10 = (s1[0] + s1[1]) + s1[2], y11 = (s1[1] - s1[2]) - s1[3];
                    if constexpr (EPI) {
                        if (m == 0) {                                                // X1: the helpers have stored the operand
                            // (behind the first fragment's inverse transform: the helpers' stores take as long)
                            asm volatile("" :: "v"(y00), "v"(y01), "v"(y10), "v"(y11));
                            __builtin_amdgcn_sched_barrier(0);
                            __builtin_amdgcn_s_barrier();
                            asm volatile("" ::: "memory");
                        }
                    }
                    // the operands of the second and third fragment are read at once (the first fragment's accumulators are free
                    // by now): X2 releases the helpers after a third of the epilogue
                    if constexpr (EPI) {
                        if (m == 1) {
#pragma unroll
                            for (int r = 0; r < 4 * (MF - 1); ++r) {
                                el0[r] = *reinterpret_cast<const f32x2*>(el + ((4 + r) * 2 + 0) * 128);
                                el1[r] = *reinterpret_cast<const f32x2*>(el + ((4 + r) * 2 + 1) * 128);
                            }
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // X2: all of it is in registers
                            __builtin_amdgcn_s_barrier();
                            asm volatile("" ::: "memory");
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // (two planes at a time: four independent chains; all four would spill the accumulators still live)
                        if constexpr (DACT_KIND == 1) { if (r == 2) __builtin_amdgcn_sched_barrier(0); }
                        const int so = (16 * m + r) * hw_o * 4;                      // scalar byte offset of the cout plane
                        f32x2 y0 = {y00[r], y01[r]}, y1 = {y10[r], y11[r]};
                        f32x2 e0, e1;
                        if constexpr (EPI) {
                            if (m >= 1) { e0 = el0[4 * (m - 1) + r]; e1 = el1[4 * (m - 1) + r]; }
                            else {
                                e0 = *reinterpret_cast<const f32x2*>(el + ((4 * m + r) * 2 + 0) * 128);
                                e1 = *reinterpret_cast<const f32x2*>(el + ((4 * m + r) * 2 + 1) * 128);
                            }
                        }
                        if constexpr (DACT) {
                            f32x2 q0, q1;
                            if constexpr (EPI) { q0 = e0; q1 = e1; }
                            else if constexpr (PRE) { q0 = pre0[m][r]; q1 = pre1[m][r]; }
                            else {
                                q0 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_z, lo0, so, 0));
                                q1 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_z, lo1, so, 0));
                            }
                            if (DACT_KIND == 1 || d.act_d == SDA_ACT_SILU) {
                                y0 *= w4_dsilu2(q0);
                                y1 *= w4_dsilu2(q1);
                            } else {
                                y0[0] *= sda_dact(d.act_d, q0[0]); y0[1] *= sda_dact(d.act_d, q0[1]);
                                y1[0] *= sda_dact(d.act_d, q1[0]); y1[1] *= sda_dact(d.act_d, q1[1]);
                            }
                        }
                        if constexpr (RES) {
                            if constexpr (EPI) { y0 += e0; y1 += e1; }
                            else if constexpr (PRE && !DACT) { y0 += pre0[m][r]; y1 += pre1[m][r]; }     // (with act'(z): z was the prefetched one)
                            else {
                                y0 += __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_res, lo0, so, 0));
                                y1 += __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_res, lo1, so, 0));
                            }
                        }
                        if (!SDA_DBG(g, 8)) {
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(w4_u32x2, y0), r_out, lo0, so, 0);
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(w4_u32x2, y1), r_out, lo1, so, 0);
                        }
                    }
                }
            };
            using K0 = std::integral_constant<int, 0>;
            using K1 = std::integral_constant<int, 1>;
            using K2 = std::integral_constant<int, 2>;
            if constexpr (ZP == 2) {
                // pooled output: the sum of the tile's 2 x 2 block = v^T M v, v = (1, 2, 0, -1) -- one value per (cout, tile) at
                // half resolution: out[n][cout][ho / 2][wo / 2], tile (by, bx, t) -> pixel (4 by + (t >> 3), 8 bx + (t & 7)).
                // (A bias seeded at position (1, 1) arrives with weight 4: once per summed pixel.)
                const int wo_p = d.wo >> 1, hw_p = (d.ho >> 1) * wo_p;
                const int lo_p = ((4 * kq_e) * hw_p + (4 * tt.by + (t >> 3)) * wo_p + 8 * tt.bx + (t & 7)) * 4;
                const int64_t sbase_p = ((int64_t)tt.n * d.cout + W4_BM * tt.ct + 16 * MF * wm) * hw_p;
                const auto r_p = __builtin_amdgcn_make_buffer_rsrc(d.out + sbase_p, (short)0, 16 * MF * hw_p * 4, 0x00020000);
#pragma unroll
                for (int m = 0; m < MF; ++m) {
                    const f32x4 c0v = (acc[0][m] + 2.f * acc[4][m]) - acc[12][m];
                    const f32x4 c1v = (acc[1][m] + 2.f * acc[5][m]) - acc[13][m];
                    const f32x4 c3v = (acc[3][m] + 2.f * acc[7][m]) - acc[15][m];
                    const f32x4 y = (c0v + 2.f * c1v) - c3v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float yr = y[r];             // (a copy: bit_cast of a vector-element lvalue reads element 0)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yr), r_p, lo_p, (16 * m + r) * hw_p * 4, 0);
                    }
                }
            } else if constexpr (EPM == 0) {
                run(K0{}, std::false_type{});
            } else if constexpr (EPI) {                    // (exactly one of the two operands: the launch condition)
                if (d.dact_z) run(K1{}, std::false_type{});            // (SiLU': the launch condition as well)
                else run(K0{}, std::true_type{});
            } else if (d.dact_z) {
                if (d.res) run(K2{}, std::true_type{});
                else run(K2{}, std::false_type{});
            } else {
                if (d.res) run(K0{}, std::true_type{});
                else run(K0{}, std::false_type{});
            }
        }
        w4_advance(g, c0);
        W4_STAMP(2);                                           // epilogue
    }
    W4_TRACE_OUT();
}

#ifdef SDA_W4_VARIANTS
static long long* w4_trace_buf = nullptr;
static int w4_trace_grid = 0;
// tooling: phase cycle sums of the last SDA_W4_VAR=11 launch, averaged over workgroups: out[wave 8][phase 8]
extern "C" int sda_w4_trace_read(double* out) {
    if (!w4_trace_buf || !out) return SDA_E_BADARG;
    static long long host[256 * 64];
    if (hipMemcpy(host, w4_trace_buf, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return SDA_E_BADARG;
    for (int i = 0; i < 64; ++i) {
        double s = 0;
        for (int b = 0; b < w4_trace_grid; ++b) s += (double)host[b * 64 + i];
        out[i] = s / w4_trace_grid;
    }
    return SDA_OK;
}
#endif

template <bool MOD, bool LN, bool SILU, int EPI, int VAR, int ZP, int MF>
static int wino4_launch_mf(const sda_conv_desc* d, const Wino4Geom& g, int grid, hipStream_t stream) {
    static_assert(W4_LDS_ALLOC_OF(MF) <= 160 * 1024, "LDS");
    static bool attr_set[SDA_MAX_DEVICES];
    const int rc = sda_raise_dyn_lds(reinterpret_cast<const void*>(conv_wino4_kernel<MOD, LN, SILU, EPI, VAR, ZP, MF>), W4_LDS_ALLOC_OF(MF), attr_set);
    if (rc != SDA_OK) return rc;
    hipLaunchKernelGGL((conv_wino4_kernel<MOD, LN, SILU, EPI, VAR, ZP, MF>), dim3(grid), dim3(512), (size_t)W4_LDS_ALLOC_OF(MF), stream, *d, g);
    return sda_launch_status();
}

// (the 64-cout tile ships for the product variants only: the tooling variants VAR != 0 study the 96-cout kernel)
template <bool MOD, bool LN, bool SILU, int EPI, int VAR, int ZP = 0>
static int wino4_launch_t(const sda_conv_desc* d, const Wino4Geom& g, int grid, hipStream_t stream) {
    if constexpr ((VAR == 0 || VAR == 11) && !W4_UDMA && EPI != 1) {   // (11: the phase trace of tools/wino4_check.py, tooling builds)
        if (g.mf == 2) return wino4_launch_mf<MOD, LN, SILU, EPI, VAR, ZP, 2>(d, g, grid, stream);
        if constexpr (ZP != 1 && VAR == 0) {
            if (g.mf == 1) return wino4_launch_mf<MOD, LN, SILU, EPI, VAR, ZP, 1>(d, g, grid, stream);
        }
    }
    if (g.mf != 3) return SDA_E_UNSUPPORTED;
    return wino4_launch_mf<MOD, LN, SILU, EPI, VAR, ZP, 3>(d, g, grid, stream);
}

// the four loader configurations of the reference U-Net have a kernel: plain (backward-data convolutions), modulation +
// LayerNorm (first block convolution), SiLU (second block convolution), LayerNorm (upsampling tails); other combinations
// are served by conv_wino / the direct kernel
static int wino4_config(const sda_conv_desc* d) {
    const bool mod = d->mod != nullptr, ln = d->ln_mean != nullptr, silu = d->act_in == SDA_ACT_SILU;
    const int key = (mod ? 4 : 0) | (ln ? 2 : 0) | (silu ? 1 : 0);
    return (key == 0 || key == 1 || key == 2 || key == 6) ? key : -1;
}

static int wino4_epm(const sda_conv_desc* d, const Wino4Geom& g) {
    // the epilogue operand through the helpers (EPI, mode 1): one operand, tiles of at least twelve stages (the six-stage load
    // window of a tile must open after the previous tile's operand has left the registers: not before stage 5), SiLU' if it is an
    // act' launch.  The 64-cout tile loads its 16 pairs in a two-stage window: tiles of at least eight stages (64 input channels).
    static const bool epi_on = !(getenv("SDA_W4_EPI") && atoi(getenv("SDA_W4_EPI")) == 0);
    // 64- / 32-cout tiles: consumer-side loads (EPM 2) only.  The helper-fed route was built for the 64-cout tile (a two-slot window of
    // eight loads: an eight-stage tile leaves no room for more slots) and measured SLOWER than the consumers' own loads -- equal at 64
    // channels, +4 % at 128, +7 % at 256, +10 % on the up-sampled tail (profiles/r06_mf2_epm_ab.txt: 128 accumulators leave the consumers the
    // registers to keep their loads in flight, and the helpers lose the window loads of every stage) -- and removed.
    const bool epi = epi_on && g.mf == 3 && ((d->res != nullptr) != (d->dact_z != nullptr)) && g.nstage >= 12 &&
                     (!d->dact_z || d->act_d == SDA_ACT_SILU);
    return epi ? 1 : ((d->res || d->dact_z) ? 2 : 0);
}

// which zero-position kernel (ZP) serves the launch: 2 = pooled output, 1 = 2 x 2 up-sampled source (the reference tails'
// configuration: LayerNorm loader + skip operand through the helpers), 0 = the full kernels
static int wino4_zp(const sda_conv_desc* d, const Wino4Geom& g) {
    static const bool zp_on = !(getenv("SDA_W4_ZP") && atoi(getenv("SDA_W4_ZP")) == 0);
    if (d->pool_h > 1 || d->pool_w > 1) return 2;                                               // (the plan requires w_wino4_zp)
    if (!(zp_on && d->w_wino4_zp && !(reinterpret_cast<uintptr_t>(d->w_wino4_zp) & 15) && d->up_h == 2 && d->up_w == 2 && wino4_config(d) == 2))
        return 0;
    if (g.mf == 3) return wino4_epm(d, g) == 1 ? 1 : 0;
    // 64-cout tile: the same launch (one operand; SiLU' if it is the act' one) with consumer-side loads, any tile length
    return (g.mf == 2 && ((d->res != nullptr) != (d->dact_z != nullptr)) && (!d->dact_z || d->act_d == SDA_ACT_SILU)) ? 1 : 0;
}

int sda_wino4_launch(const sda_conv_desc* d, const Wino4Geom& g_in, hipStream_t stream) {
    const int cus = sda_cu_count();
    if (!cus) return SDA_E_BADARG;
    int grid = cus - cus % 8;
    const int need = (g_in.grid + 7) / 8 * 8;
    if (grid > need) grid = need;
    if (grid < 8) grid = 8;
    // the tile walk (see the kernel): interleaved within an XCD by default; SDA_W4_WALK=0 = contiguous sub-ranges (A/B runs)
    static const bool walk_on = !(getenv("SDA_W4_WALK") && atoi(getenv("SDA_W4_WALK")) == 0);
    Wino4Geom g = g_in;
    g.walk = walk_on ? 1 : 0;
    {
        int step = walk_on ? grid / 8 : 1;                 // tiles between a workgroup's consecutive tiles
        g.sc = step % g.n_ct; step /= g.n_ct;
        g.sx = step % g.bx_n; step /= g.bx_n;
        g.sy = step % g.by_n;
        g.sn = step / g.by_n;
    }
    const int epm = wino4_epm(d, g);
#define W4_LAUNCH3(MOD, LN, SILU)                                                                                              \
    (epm == 1 ? wino4_launch_t<MOD, LN, SILU, 1, 0>(d, g, grid, stream)                                                        \
              : epm == 2 ? wino4_launch_t<MOD, LN, SILU, 2, 0>(d, g, grid, stream)                                             \
                         : wino4_launch_t<MOD, LN, SILU, 0, 0>(d, g, grid, stream))
    // zero-position kernels (ZP, see the kernel): the pooled-output launch, and the 2 x 2 up-sampled LayerNorm + skip launch of the
    // reference tails; SDA_W4_ZP=0 runs the latter on the full kernels (A/B runs)
    const int zp = wino4_zp(d, g);
#ifdef SDA_W4_VARIANTS
    if (zp && getenv("SDA_W4_VAR") && atoi(getenv("SDA_W4_VAR")) == 11) {                      // phase tracing of the ZP kernels
        static long long* tbuf = nullptr;
        if (!tbuf && hipMalloc(&tbuf, 256 * 64 * sizeof(long long)) != hipSuccess) return SDA_E_BADARG;
        (void)hipMemsetAsync(tbuf, 0, 256 * 64 * sizeof(long long), stream);
        Wino4Geom gt = g; gt.trace = tbuf; w4_trace_buf = tbuf; w4_trace_grid = grid;
        return zp == 2 ? wino4_launch_t<false, false, false, 0, 11, 2>(d, gt, grid, stream)
                       : wino4_launch_t<false, true, false, 1, 11, 1>(d, gt, grid, stream);
    }
#endif
    if (zp == 2) return wino4_launch_t<false, false, false, 0, 0, 2>(d, g, grid, stream);      // (eligibility: the plan)
    // (64-cout tile: the tail's skip operand through consumer-side loads -- 2.07 -> 1.87 ms on the 128 -> 64 tail, neutral at 96 couts)
    if (zp == 1) return g.mf == 2 ? wino4_launch_mf<false, true, false, 2, 0, 1, 2>(d, g, grid, stream)
                                  : wino4_launch_t<false, true, false, 1, 0, 1>(d, g, grid, stream);
    switch (wino4_config(d)) {
        case 0: {
#ifdef SDA_W4_VARIANTS
            const char* ev = getenv("SDA_W4_VAR");
            switch (ev ? atoi(ev) : 0) {
                case 11: {                                                                      // phase tracing
                    static long long* tbuf = nullptr;
                    if (!tbuf && hipMalloc(&tbuf, 256 * 64 * sizeof(long long)) != hipSuccess) return SDA_E_BADARG;
                    (void)hipMemsetAsync(tbuf, 0, 256 * 64 * sizeof(long long), stream);
                    Wino4Geom gt = g; gt.trace = tbuf; w4_trace_buf = tbuf; w4_trace_grid = grid;
                    return epm == 1 ? wino4_launch_t<false, false, false, 1, 11>(d, gt, grid, stream)
                         : epm == 2 ? wino4_launch_t<false, false, false, 2, 11>(d, gt, grid, stream)
                                    : wino4_launch_t<false, false, false, 0, 11>(d, gt, grid, stream);
                }
                case 12: if (epm == 0) return wino4_launch_t<false, false, false, 0, 12>(d, g, grid, stream); break;
                case 13: if (epm == 0) return wino4_launch_t<false, false, false, 0, 13>(d, g, grid, stream); break;
                case 14: if (epm == 0) return wino4_launch_t<false, false, false, 0, 14>(d, g, grid, stream); break;
                case 15: if (epm == 0) return wino4_launch_t<false, false, false, 0, 15>(d, g, grid, stream); break;
                default: break;
            }
#endif
            return W4_LAUNCH3(false, false, false);
        }
        case 1: return W4_LAUNCH3(false, false, true);
        case 2: return W4_LAUNCH3(false, true, false);                                             // (up-sampling tails + skip)
        case 6: return epm == 0 ? wino4_launch_t<true, true, false, 0, 0>(d, g, grid, stream)
                                : wino4_launch_t<true, true, false, 2, 0>(d, g, grid, stream);
        default: return SDA_E_UNSUPPORTED;
    }
#undef W4_LAUNCH3
}

static bool wino4_disabled() {
    static const bool off = getenv("SDA_CONV_WINO4") && atoi(getenv("SDA_CONV_WINO4")) == 0;
    return off;
}

// 0 = not served; 1 = the full kernels; 2 = a zero-position kernel (54 of 96 MFMAs per stage)
int sda_wino4_path(const sda_conv_desc* d) {
    Wino4Geom g;
    if (wino4_disabled() || sda_wino4_plan(d, &g) != SDA_OK) return 0;
    return wino4_zp(d, g) ? 2 : 1;
}

int sda_wino4_try(const sda_conv_desc* d, hipStream_t stream) {
    if (wino4_disabled()) return SDA_E_UNSUPPORTED;
    Wino4Geom g;
    const int rc = sda_wino4_plan(d, &g);
    if (rc != SDA_OK) return rc;
    return sda_wino4_launch(d, g, stream);
}

// ---------------------------------------------------------------- weight transform for this kernel (one-off per layer)
// dst[stage][position pair p >> 1][m tile][lane = 16 kq + i][h = p & 1][k4]  <-  (G g G^T)[xi][nu] of the filter between contraction channel
// kk = 8 stage + 2 kq + k4 and output channel mm = 16 mtile + i;  forward (transpose = 0): kk = ci, mm = co;
// backward-data (transpose = 1): kk = co, mm = ci, filter flipped.  k_pad % 8 == 0, m_pad % 96 == 0 or m_pad % 64 == 0 (the cout tile
// the kernel then runs: W4_BM_OF; the layout itself is per 16-cout fragment and does not depend on it); padding is zero.
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
        const int st = kk >> 3, kq_ = (kk >> 1) & 3, k4 = kk & 1, mt = mm >> 4, ii = mm & 15;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                const float uu = tmp[xi][0] * G[nu][0] + tmp[xi][1] * G[nu][1] + tmp[xi][2] * G[nu][2];
                const int pp = 2 * xi + (nu >> 1), h = nu & 1;         // position p = 4 xi + nu = 2 pp + h
                dst[((((int64_t)st * 8 + pp) * mtiles + mt) * 64 + (kq_ * 16 + ii)) * 4 + 2 * h + k4] = uu;
            }
    }
}

extern "C" int sda_pack_conv_weight_wino4(const float* w, int cout, int cin, int transpose, int cin_keep, float* dst,
                                          int k_pad, int m_pad, void* stream) {
    if (!w || !dst || cout <= 0 || cin <= 0 || k_pad <= 0 || m_pad <= 0 || (k_pad & 7) || (m_pad % 32)) return SDA_E_BADARG;
    int64_t total = (int64_t)k_pad * m_pad;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_wino4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, cout, cin, transpose, cin_keep,
                       dst, k_pad, m_pad);
    return sda_launch_status();
}

// ---- the zero-position packing of a w_wino4 buffer (W4_UZP above; sda_hip.h: sda_pack_conv_weight_wino4_zp): values are copied
// mf: cout fragments per consumer (cout tile = 32 mf): upp = floats of a full position pair of the tile, uzp = of the packed slab
__global__ void pack_wino4_zp_kernel(const float* __restrict__ src, float* __restrict__ dst, int nstage, int n_ct, int mf) {
    const int upp = W4_UPP_OF(mf), uzp = W4_UZP_OF(mf);
    const int64_t total = (int64_t)nstage * n_ct * uzp;
    const int mtiles = 2 * mf * n_ct;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % uzp);
        const int64_t blk = i / uzp;
        const int ct = (int)(blk % n_ct), st = (int)(blk / n_ct);
        auto at = [&](int pair, int m, int lane, int e) {
            return src[((((int64_t)st * 8 + pair) * mtiles + 2 * mf * ct + m) * 64 + lane) * 4 + e];
        };
        float v = 0.f;
        if (j < 3 * upp) {                                 // full pairs 0, 2, 6
            const int fi = j / upp, r = j % upp;
            v = at(fi == 0 ? 0 : (fi == 1 ? 2 : 6), r / 256, (r % 256) / 4, r & 3);
        } else if (j < 4 * upp) {                          // positions 3 | 7: the h = 1 halves of pairs 1 and 3
            const int r = j - 3 * upp, e = r & 3;
            v = e < 2 ? at(1, r / 256, (r % 256) / 4, 2 + e) : at(3, r / 256, (r % 256) / 4, e);
        } else if (j < 4 * upp + upp / 2) {                // position 15: the h = 1 half of pair 7
            const int r = j - 4 * upp;
            v = at(7, r / 128, (r % 128) / 2, 2 + (r & 1));
        }
        dst[i] = v;
    }
}

// the cout tile a packing of m_pad output channels is made for: the kernel's own rule (sda_wino4_plan)
static int wino4_mf_of(int m_pad) { return m_pad % 96 == 0 ? 3 : (m_pad % 64 == 0 ? 2 : (m_pad % 32 == 0 ? 1 : 0)); }

extern "C" int64_t sda_wino4_zp_floats(int k_pad, int m_pad) {
    const int mf = wino4_mf_of(m_pad);
    if (k_pad <= 0 || m_pad <= 0 || (k_pad & 7) || !mf) return SDA_E_BADARG;
    return (int64_t)(k_pad / 8) * (m_pad / W4_BM_OF(mf)) * W4_UZP_OF(mf);
}

extern "C" int sda_pack_conv_weight_wino4_zp(const float* w_wino4, int k_pad, int m_pad, float* dst, void* stream) {
    const int mf = wino4_mf_of(m_pad);
    if (!w_wino4 || !dst || k_pad <= 0 || m_pad <= 0 || (k_pad & 7) || !mf) return SDA_E_BADARG;
    const int64_t total = sda_wino4_zp_floats(k_pad, m_pad);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_wino4_zp_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_wino4, dst, k_pad / 8, m_pad / W4_BM_OF(mf), mf);
    return sda_launch_status();
}
