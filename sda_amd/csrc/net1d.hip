// A whole single-level 1-D U-Net (the Lorenz score networks of experiments/lorenz/utils.py:26-42: head convolution, the
// descent + ascent modulated residual blocks of sda/nn.py:18-28, tail convolution -- sda/nn.py:184-206 with one level) in ONE
// launch, and its input VJP in one more.  The per-block kernels of block1d.hip left a network evaluation at 8 launches
// forward + 8 backward, each ~60 % launch latency + first global round trip; here the dependency between layers stays inside a
// workgroup:
//   * a workgroup owns TP consecutive positions of one sequence and computes every layer on NC = 16 NF columns = TP + a halo
//     of H = (number of convolutions) positions per side.  A k = 3 convolution makes one more column per side depend on data
//     beyond the tile, so after all H convolutions exactly the TP own columns are exact; the halo columns are recomputed by
//     the neighbouring workgroups (1.8x redundant multiplies at NF = 4 -- the nets are latency-bound, not MFMA-bound).
//     Positions outside the sequence are forced to zero at every convolution input (zero padding) or wrap (circular).
//   * wave w owns output channels 16 w .. 16 w + 15 of every convolution (v_mfma_f32_16x16x4_f32; A = weight fragments held
//     in registers, B from LDS).  Every convolution's weights are packed [tap][64][64] (zero padded) in EXECUTION order in one
//     buffer, so a fragment load is one lane offset + a scalar offset; the weights of convolution i + 1 are loaded (L2 ->
//     registers, 48 dwords per lane) while convolution i multiplies: two register sets, alternating.  Biases and modulation
//     vectors are staged in LDS once per launch.
//   * the residual stream lives in registers in MFMA D layout (channel 16 w + 4 kq + r, column 16 nf + li) between layers;
//     LayerNorm statistics are reduced lane-locally, across the four lane groups (shuffles), across the four waves (LDS).
//   * what the VJP needs (block inputs a, pre-activations z, mean / rstd per position) is written for the own columns only;
//     the backward kernel reads it for its halo columns too (written by the neighbours' forward).
//   * input / output are addressed through (image, channel, position) strides: the (B, L, C) <-> (B, C, L) transposes of
//     MCScoreWrapper (sda/score.py:104-110) cost nothing on either side.
//   * whole-sequence tiles (round 4): with zero padding a tile that holds an ENTIRE sequence needs no halo at all -- what lies beyond
//     its edge columns is the padding itself.  When every sequence fits 16 NF <= 80 columns and there are enough sequences to fill the
//     chip (the reference's evaluation job: 1024 trajectories of 65 positions, experiments/lorenz/eval.py:72-84) a workgroup takes one
//     whole sequence: 80 columns per sequence instead of two 64-column tiles (1.8x halo recompute), nothing narrowed (`whole`).
//   * the validity cone (round 4): convolution i (0 = first of the launch) is only exact -- and only needed -- on columns
//     [1 + i, NC - 1 - i).  The 16-column MFMA fragments are therefore mapped so that the LAST one holds the tile's outermost
//     columns [0, 8) + [NC - 8, NC) (fragment nf < NF - 1 holds columns 8 + 16 nf ..): from convolution 7 on nothing in it is needed
//     any more, and the rest of the launch multiplies, normalises and stores NF - 1 fragments (7 of 14 convolutions of the Lorenz
//     nets: -25 % of the MFMAs at NF = 2, -12.5 % at NF = 4).  A stage may only drop the fragment once the stage feeding it wrote
//     nothing the next one reads there: a block's LayerNorm / residual store goes narrow one block later than its convolutions.
#include "sda_common.hpp"
#include <type_traits>
#include <stdlib.h>

// LDS tiles are [column][channel] (68-float rows: 16-byte accesses of consecutive columns land 4 banks apart -> conflict free).
// The K index of an MFMA fragment is a free permutation as long as both operands agree: fragment cb holds channels
// { 16 kq + cb }, so a lane's 16 B values of one (column, tap) are CONSECUTIVE channels -- four ds_read_b128 feed sixteen MFMAs
// (the [channel][column] layout of block1d.hip needs one ds_read_b32 per MFMA: +15-28 % on the stream, profiles/r02_w4_feed.txt)
// -- and a lane's four D values of a column are consecutive channels too: one ds_write_b128 per column on the way back.
#define N1_LD 68
#define N1_MAXC 64
#define N1_MAXCOL 82                   // up to 80 conv columns + one edge column per side

typedef float n1_f32x4 __attribute__((ext_vector_type(4)));

#ifdef SDA_N1_TRACE                    // tooling (tools/net1d_trace.py): per-phase cycle sums of workgroup 0 / wave 0
__device__ long long n1_trace[16];
#define N1_T0() long long n1_tl = __builtin_readcyclecounter()
#define N1_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long n_ = __builtin_readcyclecounter(); n1_trace[k] += n_ - n1_tl; n1_tl = n_; } } while (0)
extern "C" int sda_n1_trace_read(long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(n1_trace), sizeof(long long) * 16) != hipSuccess) return SDA_E_BADARG;
    if (reset) { long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(n1_trace), z, sizeof(z)); }
    return SDA_OK;
}
#else
#define N1_T0() do {} while (0)
#define N1_STAMP(k) do {} while (0)
#endif

struct N1Ctx {
    int tid, lane, wave, kq, li, co0, n, p0, H, len;
    int col_outer;                     // this lane's column in the OUTER fragment: li < 8 ? li : NC - 16 + li
    int col_shift;                     // 0; 8 for whole-sequence tiles (fragment nf = columns 16 nf ..: the identity map)
    unsigned wlane;                    // this lane's element offset inside a [tap][64][64] weight slab: row 16 kq, column co0 + li
    bool circular;
};

// position of conv-output column j (0 .. NC-1): wrapped for circular padding; `inside` = carries data
__device__ __forceinline__ int n1_pos(const N1Ctx& c, int j, bool& inside) {
    int p = c.p0 - c.H + j;
    if (c.circular) {
        p %= c.len;
        if (p < 0) p += c.len;
    }
    inside = p >= 0 && p < c.len;
    return inside ? p : 0;
}

// conv-output column of this lane in fragment nf: inner fragments are consecutive runs from column 8, the last fragment is the tile's
// outermost 8 + 8 columns (see "validity cone" above)
template <int NF>
__device__ __forceinline__ int n1_col(const N1Ctx& c, int nf) { return nf < NF - 1 ? 8 - c.col_shift + 16 * nf + c.li : c.col_outer; }

// all A fragments of convolution `conv` for this wave: wreg[tap][cb] = W[conv][tap][k = 16 kq + cb][m = co0 + li]
// (one batch of loads: a per-lane offset against wave-uniform bases)
__device__ __forceinline__ void n1_load_w(const float* w, int conv, const N1Ctx& c, float (&wreg)[3][16]) {
    const float* wb = w + (size_t)conv * (3 * 64 * 64);
#ifdef SDA_N1_NOW                      // (tooling, tools/net1d_trace.py N1_FLAGS=-DSDA_N1_NOW: what the weight loads of convolutions >= 2 cost -- results wrong)
    if (conv > 1) return;
#endif
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int cb = 0; cb < 16; ++cb) wreg[tap][cb] = (wb + tap * 4096 + cb * 64)[c.wlane];
}

// acc[nf] = sum_{tap, cb} A(tap, cb) B[channel 16 kq + cb][column 16 nf + li + tap]   (tile column jj <-> conv column jj - 1).
// All 16 K fragments, unconditionally (see block1d.hip: a runtime trip count costs more than the surplus MFMAs).  Per tap the
// wave reads its 16 x NF operand values as 4 x NF ds_read_b128; consecutive MFMAs rotate over the NF accumulators.
// NFA <= NF: the fragments still inside the validity cone (the outer fragment is the last index); boff[nf] = this lane's float offset
// of (column n1_col(nf), channel 16 kq) in a tile
template <int NF, int NFA>
__device__ __forceinline__ void n1_mm(const float (&wreg)[3][16], const float* tile, const unsigned (&boff)[NF], n1_f32x4 (&acc)[NF]) {
    n1_f32x4 bv[NFA][4];
#pragma unroll
    for (int nf = 0; nf < NFA; ++nf) acc[nf] = n1_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
        // (one operand set: the ~100 cycles until a tap's reads return are exposed three times per 6000-cycle convolution)
#pragma unroll
        for (int nf = 0; nf < NFA; ++nf)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                bv[nf][q] = *reinterpret_cast<const n1_f32x4*>(tile + boff[nf] + tap * N1_LD + 4 * q);
#pragma unroll
        for (int cb = 0; cb < 16; ++cb)
#pragma unroll
            for (int nf = 0; nf < NFA; ++nf)
                acc[nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap][cb], bv[nf][cb >> 2][cb & 3], acc[nf], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                 // (or the scheduler hoists all three taps' reads: 192 live registers)
    }
}

// a strided (image, channel, position) tensor -> tile columns 1 .. NC (conv columns 0 .. NC-1), channels [0, 64): thread
// (column j = lane, channel group = wave): channels 16 sub .. 16 sub + 15.  Channels >= `channels` and columns outside the
// sequence are zero.  Offsets inside one image are 32-bit (checked by the launcher).
template <int NF>
__device__ __forceinline__ void n1_load_tile(const float* src, int64_t sn, int64_t sc, int64_t sx, int channels, const N1Ctx& c,
                                             float* tile, float scale = 1.f) {
    constexpr int NC = 16 * NF;
    const int sub = c.wave;
    for (int j = c.lane; j < NC; j += 64) {                // (80-column whole-sequence tiles: two passes)
        bool inside;
        const int ps = n1_pos(c, j, inside);
        const float* base = src + (int64_t)c.n * sn;
        const unsigned lo = (unsigned)(ps * (int)sx);
        n1_f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = 16 * sub + i, cic = ci < channels ? ci : channels - 1;
            v[i >> 2][i & 3] = base[lo + (unsigned)(cic * (int)sc)];
        }
        if (scale != 1.f) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i >> 2][i & 3] *= scale;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = 16 * sub + i;
            if (!(inside && ci < channels)) v[i >> 2][i & 3] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<n1_f32x4*>(tile + (1 + j) * N1_LD + 16 * sub + 4 * q) = v[q];
    }
}

// registers in D layout -> the tile channels of this wave (masked: columns outside the sequence and channels >= c are zero):
// a lane's four values of a column are consecutive channels -> one 16-byte store per column
template <int NF, int NFA>
__device__ __forceinline__ void n1_store_tile(const n1_f32x4 (&v)[NF], const bool (&inside)[NF], const bool (&rok)[4], const N1Ctx& c,
                                              float* tile) {
    const int cb = c.co0 + 4 * c.kq;
#pragma unroll
    for (int nf = 0; nf < NFA; ++nf) {
        n1_f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (inside[nf] && rok[r]) ? v[nf][r] : 0.f;
        *reinterpret_cast<n1_f32x4*>(tile + (1 + n1_col<NF>(c, nf)) * N1_LD + cb) = o;
    }
}

// Predicated 4-byte stores without branches (round 6).  `if (own && rok) p[off] = v` compiled to an exec-mask save / and / branch /
// restore around every store, the sixteen lane masks of a (fragment, channel row) grid living in SGPR pairs spilled to VGPR lanes
// (v_readlane per use): ~10 instructions and a branch per store, 1 000 cycles per block and saved tensor (tools/net1d_trace.py).  A
// raw buffer store whose offset lies beyond the descriptor's num_records is DROPPED by the hardware: the predicate goes into the
// offset (N1_OOB for lanes that must not write; channel rows >= c fall beyond a [c][len] plane by themselves).
#define N1_OOB 0x80000000u
typedef unsigned n1_u32;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t n1_rsrc(const float* p, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), (short)0, (int)(bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes), 0x00020000);
}
__device__ __forceinline__ void n1_bstore(float v, __amdgpu_buffer_rsrc_t rs, n1_u32 byte_off) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)byte_off, 0, 0);
}

// sum over the channels of every column: lane-local over r, across the 4 lane groups, across the 4 waves (LDS; one barrier)
template <int NF, int NFA>
__device__ __forceinline__ void n1_colsum(float (&s)[NF], float* red, const N1Ctx& c) {
    constexpr int NC = 16 * NF;
#pragma unroll
    for (int nf = 0; nf < NFA; ++nf) {
        s[nf] += __shfl_xor(s[nf], 16, 64);
        s[nf] += __shfl_xor(s[nf], 32, 64);
        if (c.kq == 0) red[c.wave * NC + 16 * nf + c.li] = s[nf];      // (slot 16 nf + li: any bijection serves the exchange)
    }
    __syncthreads();
#pragma unroll
    for (int nf = 0; nf < NFA; ++nf) {
        const int m = 16 * nf + c.li;
        s[nf] = (red[m] + red[NC + m]) + (red[2 * NC + m] + red[3 * NC + m]);
    }
}

// two independent column sums with ONE exchange (the LayerNorm backward's mean_c(gh) and mean_c(gh xh)): the same partial sums in the
// same order as two n1_colsum calls -- bit-identical -- and one barrier + LDS round trip fewer per block
template <int NF, int NFA>
__device__ __forceinline__ void n1_colsum2(float (&s)[NF], float (&t)[NF], float* red, const N1Ctx& c) {
    constexpr int NC = 16 * NF;
#pragma unroll
    for (int nf = 0; nf < NFA; ++nf) {
        s[nf] += __shfl_xor(s[nf], 16, 64);
        t[nf] += __shfl_xor(t[nf], 16, 64);
        s[nf] += __shfl_xor(s[nf], 32, 64);
        t[nf] += __shfl_xor(t[nf], 32, 64);
        if (c.kq == 0) { red[c.wave * NC + 16 * nf + c.li] = s[nf]; red[4 * NC + c.wave * NC + 16 * nf + c.li] = t[nf]; }
    }
    __syncthreads();
#pragma unroll
    for (int nf = 0; nf < NFA; ++nf) {
        const int m = 16 * nf + c.li;
        s[nf] = (red[m] + red[NC + m]) + (red[2 * NC + m] + red[3 * NC + m]);
        t[nf] = (red[4 * NC + m] + red[5 * NC + m]) + (red[6 * NC + m] + red[7 * NC + m]);
    }
}

__device__ __forceinline__ void n1_ctx(N1Ctx& c, const sda_net1d_desc& d, int ptiles, int tp) {
    c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6); c.kq = c.lane >> 4; c.li = c.lane & 15;
    c.co0 = 16 * c.wave; c.n = blockIdx.x / ptiles; c.p0 = (blockIdx.x - c.n * ptiles) * tp;
    c.H = 2 * d.nblocks + 2; c.len = d.len; c.circular = d.circular != 0;
    c.wlane = (unsigned)(16 * c.kq * 64 + c.co0 + c.li);
    c.col_outer = 0; c.col_shift = 0;  // (set by the kernels: needs NC)
}

// biases of every convolution and the modulation vectors of every block -> LDS (once per launch; read per block as one 16-byte
// LDS read per lane instead of dependent global round trips).  All loads are issued before the first LDS store: one round trip.
__device__ __forceinline__ void n1_stage_vectors(const sda_net1d_desc& d, const N1Ctx& c, float* sb, float* smod) {
    constexpr int NB = ((2 + 2 * SDA_NET1D_MAXB) * 64 + 255) / 256, NM = (SDA_NET1D_MAXB * 64 + 255) / 256;
    const int nconv = 2 + 2 * d.nblocks;
    float vb[NB], vm[NM];
    if (sb) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = c.tid + 256 * i;
            vb[i] = (d.bias && e < nconv * 64) ? d.bias[e] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        const int e = c.tid + 256 * i, k = e >> 6, ch = e & 63;
        const float* mp = k < d.nblocks ? d.mod[k] : nullptr;
        vm[i] = (mp && ch < d.c) ? mp[(int64_t)c.n * d.mod_sn + ch] : 0.f;
    }
    if (sb) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = c.tid + 256 * i;
            if (e < (2 + 2 * SDA_NET1D_MAXB) * 64) sb[e] = vb[i];
        }
    }
#pragma unroll
    for (int i = 0; i < NM; ++i) smod[c.tid + 256 * i] = vm[i];
}

// ------------------------------------------------------------------------------------------------------------ forward
// convolution order in d.w / d.bias: head, (conv1, conv2) of block 0 .. nblocks - 1, tail
// FUSED (sda_net1d_fwd_fused): the launch is one half of a Gaussian-guided score evaluation (sda/score.py:375-396).  Its epilogue
// forms eps = (cx0 + cx1 sigma) x + cn net(x, t) on the own columns, writes it, and writes the likelihood cotangent
//   ghat = A^T((y - A x_hat) / var),  x_hat = (x - sigma eps) / mu,  var = std^2 + gamma (sigma / mu)^2
// for the strided observation A = x[..., p_start:p_stop:p_step, c_start:c_stop:c_step] next to it: sda_denoise / sda_obs_subsample /
// sda_gauss_cotangent / sda_obs_subsample_adjoint (= sda_obs_subsample_guidance) without a launch of their own, in their arithmetic
// (same operations in the same order: bit-identical to the unfused path).
template <int NF, bool FUSED>
__global__ __launch_bounds__(256) void net1d_fwd_kernel(const sda_net1d_desc d, const sda_net1d_fuse f, int ptiles, int tp, int whole) {
    constexpr int NC = 16 * NF;
    constexpr int NR = NF > 1 ? NF - 1 : 1;                                   // fragments once the outer one has left the validity cone
    using FULL = std::integral_constant<int, NF>;
    using NARROW = std::integral_constant<int, NR>;
    __shared__ __attribute__((aligned(16))) float tin[(NC + 2) * N1_LD];      // input of the next convolution, [column jj <-> conv column jj - 1][channel]
    __shared__ __attribute__((aligned(16))) float tz[(NC + 2) * N1_LD];       // act(z) between the two convolutions of a block
    __shared__ __attribute__((aligned(16))) float sb[(2 + 2 * SDA_NET1D_MAXB) * 64];
    __shared__ __attribute__((aligned(16))) float smod[SDA_NET1D_MAXB * 64];
    __shared__ float red[2 * 4 * NC];
    N1Ctx c;
    n1_ctx(c, d, ptiles, tp);
    c.col_outer = c.li < 8 ? c.li : NC - 16 + c.li;
    if (whole) { c.H = 0; c.col_outer = NC - 16 + c.li; c.col_shift = 8; }     // columns 16 nf + li: no halo, nothing to narrow
    N1_T0();
    float wA[3][16], wB[3][16];
    n1_load_w(d.w, 0, c, wA);
    // the two edge columns of both tiles are never written again: they stand for data beyond the tile (zeros: whatever they
    // were, the columns they reach are halo columns that have lost their meaning by the time they matter)
    if (c.tid < 2 * N1_MAXC) {
        const int ch = c.tid >> 1, col = (c.tid & 1) ? NC + 1 : 0;
        tin[col * N1_LD + ch] = 0.f;
        tz[col * N1_LD + ch] = 0.f;
    }
    bool inside[NF], own[NF], rok[4];
    unsigned soff[NF], ooff[NF], boff[NF];
    const int cbase = c.co0 + 4 * c.kq;
#pragma unroll
    for (int r = 0; r < 4; ++r) rok[r] = cbase + r < d.c;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const int j = n1_col<NF>(c, nf);
        const int ps = n1_pos(c, j, inside[nf]);
        own[nf] = inside[nf] && j >= c.H && j < c.H + tp && c.p0 - c.H + j < d.len;      // (the un-wrapped position is this tile's)
        soff[nf] = (unsigned)(cbase * d.len + ps);                                       // planar [c][len] saves
        ooff[nf] = (unsigned)(cbase * (int)d.out_sc + ps * (int)d.out_sx);
        boff[nf] = (unsigned)(j * N1_LD + 16 * c.kq);
    }
    // byte offsets of this lane's (fragment nf, channel row cbase) element in a planar [c][len] save / in a [len] statistics row, N1_OOB
    // where the column is not this tile's: the store predicates live in the offsets (see n1_bstore)
    n1_u32 svoff[NF], stoff[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        svoff[nf] = own[nf] ? soff[nf] * 4u : N1_OOB;
        stoff[nf] = (own[nf] && c.wave == 0 && c.kq == 0) ? (soff[nf] - (unsigned)(cbase * d.len)) * 4u : N1_OOB;
    }
    const n1_u32 row_b = (n1_u32)d.len * 4u;
    n1_load_tile<NF>(d.x, d.x_sn, d.x_sc, d.x_sx, d.cin, c, tin);
    n1_stage_vectors(d, c, sb, smod);
    n1_load_w(d.w, 1, c, wB);
    N1_STAMP(0);                                           // address arithmetic + issue of the first loads
    __syncthreads();
    N1_STAMP(1);                                           // the input tile's round trip
    // ---- head convolution: a = conv(x) + b
    n1_f32x4 a[NF];
    n1_mm<NF, NF>(wA, tin, boff, a);
    {
        const n1_f32x4 bh = *reinterpret_cast<const n1_f32x4*>(sb + cbase);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) a[nf] += bh;
    }
    N1_STAMP(2);                                           // head convolution (waits for its weights)
    const bool silu = d.act == SDA_ACT_SILU;
    const float inv_c = 1.f / (float)d.c, inv_v = 1.f / (float)(d.unbiased ? d.c - 1 : d.c);
    const int64_t plane = (int64_t)d.c * d.len;
    // one modulated residual block; NFL fragments through the LayerNorm and its tile store, NFC through the two convolutions, their
    // epilogues and the residual update (NFC <= NFL; see "validity cone")
    auto block = [&](const int k, auto NFL_, auto NFC_) {
        constexpr int NFL = decltype(NFL_)::value, NFC = decltype(NFC_)::value;
        // ---- per-channel operands of the block (this lane's 4 channels)
        const n1_f32x4 mo = *reinterpret_cast<const n1_f32x4*>(smod + k * 64 + cbase);
        const n1_f32x4 b1 = *reinterpret_cast<const n1_f32x4*>(sb + (1 + 2 * k) * 64 + cbase);
        const n1_f32x4 b2 = *reinterpret_cast<const n1_f32x4*>(sb + (2 + 2 * k) * 64 + cbase);
        // ---- the block input is what the VJP differentiates through: save the own columns
        if (d.a_save) {
            const auto ras = n1_rsrc(d.a_save + (int64_t)k * d.save_stride + (int64_t)c.n * plane, plane * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nf = 0; nf < NFL; ++nf) n1_bstore(a[nf][r], ras, svoff[nf] + r * row_b);       // (rows >= c: beyond the plane)
        }
        // ---- LayerNorm over channels of u = a + mod (two passes over registers: mean, then centred sum of squares)
        n1_f32x4 u[NF];
        float s[NF];
#pragma unroll
        for (int nf = 0; nf < NFL; ++nf) {
            s[nf] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                u[nf][r] = rok[r] ? a[nf][r] + mo[r] : 0.f;
                s[nf] += u[nf][r];
            }
        }
        n1_colsum<NF, NFL>(s, red, c);
        N1_STAMP(3);                                       // block operands, a_save stores, first channel reduction
        float mean[NF], rstd[NF];
#pragma unroll
        for (int nf = 0; nf < NFL; ++nf) {
            mean[nf] = s[nf] * inv_c;
            s[nf] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dl = u[nf][r] - mean[nf];
                s[nf] += rok[r] ? dl * dl : 0.f;
            }
        }
        n1_colsum<NF, NFL>(s, red + 4 * NC, c);
#pragma unroll
        for (int nf = 0; nf < NFL; ++nf) {
            rstd[nf] = __builtin_amdgcn_rsqf(s[nf] * inv_v + d.eps);       // (v_rsq_f32: 1 ulp)
#pragma unroll
            for (int r = 0; r < 4; ++r) u[nf][r] = (u[nf][r] - mean[nf]) * rstd[nf];
        }
        if (d.mean_save) {
            const auto rms = n1_rsrc(d.mean_save + (int64_t)k * d.stat_stride + (int64_t)c.n * d.len, (int64_t)d.len * 4);
            const auto rrs = n1_rsrc(d.rstd_save + (int64_t)k * d.stat_stride + (int64_t)c.n * d.len, (int64_t)d.len * 4);
#pragma unroll
            for (int nf = 0; nf < NFL; ++nf) { n1_bstore(mean[nf], rms, stoff[nf]); n1_bstore(rstd[nf], rrs, stoff[nf]); }
        }
        n1_store_tile<NF, NFL>(u, inside, rok, c, tin);
        n1_load_w(d.w, 2 + 2 * k, c, wA);                  // conv2 of this block (set A is free: the previous conv2 / the head is done)
        __syncthreads();
        N1_STAMP(4);                                       // second reduction, normalised tile -> LDS, weight-load issue
        // ---- conv1: z = conv(LN) + b1 -> saved (own columns); act(z) -> LDS
        n1_f32x4 z[NF];
        n1_mm<NF, NFC>(wB, tin, boff, z);
        N1_STAMP(5);                                       // conv1 multiply
        // (no z_save: a descriptor of zero records drops every store)
        const auto rzs = n1_rsrc(d.z_save ? d.z_save + (int64_t)k * d.save_stride + (int64_t)c.n * plane : d.x, d.z_save ? plane * 4 : 0);
        auto conv1_epilogue = [&](auto SILU_) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int nf = 0; nf < NFC; ++nf) {
                    const float zv = z[nf][r] + b1[r];
                    n1_bstore(zv, rzs, svoff[nf] + r * row_b);
                    z[nf][r] = decltype(SILU_)::value ? sda_act(SDA_ACT_SILU, zv) : sda_act(d.act, zv);
                }
            }
        };
        if (silu) conv1_epilogue(std::true_type{});
        else conv1_epilogue(std::false_type{});
        n1_store_tile<NF, NFC>(z, inside, rok, c, tz);
        n1_load_w(d.w, 3 + 2 * k, c, wB);                  // conv1 of the next block, or the tail
        __syncthreads();
        N1_STAMP(6);                                       // conv1 epilogue: z stores, activation, tile -> LDS
        // ---- conv2 + b2 + residual
        n1_f32x4 y[NF];
        n1_mm<NF, NFC>(wA, tz, boff, y);
        N1_STAMP(7);                                       // conv2 multiply
#pragma unroll
        for (int nf = 0; nf < NFC; ++nf) a[nf] += y[nf] + b2;
        N1_STAMP(8);                                       // residual update
    };
    // convolution index of block k: 1 + 2 k (conv1), 2 + 2 k (conv2).  Outputs of convolution i are needed on columns [1 + i, NC - 1 - i):
    // the outer fragment (columns < 8 and >= NC - 8) is out of the multiply from i = 7 (block 3) and out of the LayerNorm + store --
    // conv1's INPUT, columns [1 + 2 k, ..) -- from block 4.  (The own columns are inner columns whenever a block >= 3 exists: H >= 8.)
    {
        int k = 0;
        const int nb = d.nblocks;
        for (; k < nb && (k < 3 || whole); ++k) block(k, FULL{}, FULL{});
        if (k < nb) { block(k, FULL{}, NARROW{}); ++k; }
        for (; k < nb; ++k) block(k, NARROW{}, NARROW{});
    }
    // ---- tail convolution (index 1 + 2 nblocks) -> out (own columns, through the output strides)
    n1_f32x4 o[NF];
    const int nbn = whole ? 0 : d.nblocks;                 // (whole-sequence tiles: every column is needed to the end)
    if (nbn >= 4) {
        n1_store_tile<NF, NR>(a, inside, rok, c, tin);
        __syncthreads();
        n1_mm<NF, NR>(wB, tin, boff, o);
    } else if (nbn == 3) {
        n1_store_tile<NF, NF>(a, inside, rok, c, tin);
        __syncthreads();
        n1_mm<NF, NR>(wB, tin, boff, o);
    } else {
        n1_store_tile<NF, NF>(a, inside, rok, c, tin);
        __syncthreads();
        n1_mm<NF, NF>(wB, tin, boff, o);
    }
    // (own columns of a fragment that was not multiplied do not exist: with nblocks >= 3 the halo is >= 8 columns)
    const int nfo = nbn >= 3 ? NR : NF;
    if constexpr (!FUSED) {
        const n1_f32x4 bt = *reinterpret_cast<const n1_f32x4*>(sb + (1 + 2 * d.nblocks) * 64 + cbase);
        float* ob = d.out + (int64_t)c.n * d.out_sn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* obr = ob + (int64_t)r * d.out_sc;       // (uniform)
            const bool cok = cbase + r < d.cout;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
                if (nf < nfo && own[nf] && cok) obr[ooff[nf]] = o[nf][r] + bt[r];
        }
    } else {
        const n1_f32x4 bt = *reinterpret_cast<const n1_f32x4*>(sb + (1 + 2 * d.nblocks) * 64 + cbase);
        const float mu = f.coef[0], sg = f.coef[1];
        const bool bare = f.cx0 == 0.f && f.cx1 == 0.f && f.cn == 1.f;
        const float cx = f.cx0 + f.cx1 * sg;
        const float rr = __fdiv_rn(sg, mu);
        const float var = __fadd_rn(__fmul_rn(f.std, f.std), __fmul_rn(f.gamma, __fmul_rn(rr, rr)));
        const int n_oc = (f.c_stop - f.c_start + f.c_step - 1) / f.c_step;
        const float* xb = d.x + (int64_t)c.n * d.x_sn;
        const float* yb = f.y + (int64_t)c.n * f.y_sn;
        float* eb = d.out + (int64_t)c.n * d.out_sn;
        float* gb = f.ghat + (int64_t)c.n * d.out_sn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = cbase + r;
            const bool cok = ch < d.cout;
            const int crel = ch - f.c_start;
            const bool c_obs = crel >= 0 && ch < f.c_stop && crel % f.c_step == 0;
            const int och = c_obs ? crel / f.c_step : 0;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                if (!(nf < nfo && own[nf] && cok)) continue;
                const int ps = (int)((soff[nf] - (unsigned)(cbase * d.len)));          // this column's position
                const float xv = xb[(unsigned)(ch * (int)d.x_sc + ps * (int)d.x_sx)];
                const float ov = o[nf][r] + bt[r];
                const float e = bare ? ov : (xv * cx) + (f.cn * ov);
                const unsigned oo = ooff[nf] + (unsigned)(r * (int)d.out_sc);
                eb[oo] = e;
                const int prel = ps - f.p_start;
                float gv = 0.f;
                if (c_obs && prel >= 0 && ps < f.p_stop && prel % f.p_step == 0) {
                    const float xh = (xv - sg * e) / mu;
                    gv = __fdiv_rn(yb[(prel / f.p_step) * n_oc + och] - xh, var);
                }
                gb[oo] = gv;
            }
        }
    }
    N1_STAMP(9);                                           // tail convolution + output stores
}

// ------------------------------------------------------------------------------------------------------------ input VJP
// d.w holds the BACKWARD-DATA packings (sda_pack_conv_weight with transpose = 1) in execution order: tail^T (cout -> c), then for
// k = nblocks - 1 .. 0: conv2^T, conv1^T of block k, then head^T (c -> cin).  x = incoming cotangent (cin = its channels), out =
// the input gradient (cout = its channels); d.bias is unused.
// FUSED (sda_net1d_bwd_fused): x = ghat (the cotangent the fused forward wrote); the tile is scaled by cn on the way in, and the
// epilogue finishes the guided score on the own columns,
//   vjp = (cx0 + cx1 sigma) ghat + J_net^T (cn ghat);   out = eps - (sigma / mu) (ghat - sigma vjp)            (sda_guided_combine)
// and, by f.mode, 0: writes out;  1: applies the predictor update x <- r x + c1 out in place (sda_pc_predict; safe: this launch
// reads x on its own columns only);  2: writes out and this tile's sum of out^2 into partial[image][tile] (a fixed slot: the
// Langevin step size of sda/score.py:259 stays deterministic) for sda_pc_correct / sda_pc_correct_keyed.
template <int NF, bool FUSED>
__global__ __launch_bounds__(256) void net1d_bwd_kernel(const sda_net1d_desc d, const sda_net1d_fuse f, int ptiles, int tp, int whole) {
    constexpr int NC = 16 * NF;
    constexpr int NR = NF > 1 ? NF - 1 : 1;
    using FULL = std::integral_constant<int, NF>;
    using NARROW = std::integral_constant<int, NR>;
    __shared__ __attribute__((aligned(16))) float tg[(NC + 2) * N1_LD];
    __shared__ __attribute__((aligned(16))) float tq[(NC + 2) * N1_LD];
    __shared__ __attribute__((aligned(16))) float smod[SDA_NET1D_MAXB * 64];
    __shared__ float red[2 * 4 * NC];
    N1Ctx c;
    n1_ctx(c, d, ptiles, tp);
    c.col_outer = c.li < 8 ? c.li : NC - 16 + c.li;
    if (whole) { c.H = 0; c.col_outer = NC - 16 + c.li; c.col_shift = 8; }
    float wA[3][16], wB[3][16];
    n1_load_w(d.w, 0, c, wA);
    if (c.tid < 2 * N1_MAXC) {
        const int ch = c.tid >> 1, col = (c.tid & 1) ? NC + 1 : 0;
        tg[col * N1_LD + ch] = 0.f;
        tq[col * N1_LD + ch] = 0.f;
    }
    bool inside[NF], own[NF], rok[4];
    unsigned poff[NF], ooff[NF], boff[NF];
    const int cbase = c.co0 + 4 * c.kq;
    // (loads of saved tensors clamp their channel row: lanes beyond c read a row that exists and are masked afterwards)
    unsigned roff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rok[r] = cbase + r < d.c;
        roff[r] = (unsigned)((rok[r] ? cbase + r : d.c - 1) * d.len);
    }
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const int j = n1_col<NF>(c, nf);
        const int ps = n1_pos(c, j, inside[nf]);
        own[nf] = inside[nf] && j >= c.H && j < c.H + tp && c.p0 - c.H + j < d.len;
        poff[nf] = (unsigned)ps;
        ooff[nf] = (unsigned)(cbase * (int)d.out_sc + ps * (int)d.out_sx);
        boff[nf] = (unsigned)(j * N1_LD + 16 * c.kq);
    }
    n1_load_tile<NF>(d.x, d.x_sn, d.x_sc, d.x_sx, d.cin, c, tg, FUSED ? f.cn : 1.f);
    n1_stage_vectors(d, c, nullptr, smod);
    n1_load_w(d.w, 1, c, wB);
    const int64_t plane = (int64_t)d.c * d.len;
    // what a block's VJP reads from the forward, in D layout on every column still inside the validity cone (halo columns: written by
    // the neighbours' forward)
    n1_f32x4 ez[NF], ea[NF];
    float emean[NF], erstd[NF];
    auto fetch_saved = [&](const int k, auto NFA_) {
        constexpr int NFA = decltype(NFA_)::value;
        const float* zs = d.z_save + (int64_t)k * d.save_stride + (int64_t)c.n * plane;
        const float* as = d.a_save + (int64_t)k * d.save_stride + (int64_t)c.n * plane;
        const float* ms = d.mean_save + (int64_t)k * d.stat_stride + (int64_t)c.n * d.len;
        const float* rs = d.rstd_save + (int64_t)k * d.stat_stride + (int64_t)c.n * d.len;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int nf = 0; nf < NFA; ++nf) {
                const unsigned o = roff[r] + poff[nf];
                ez[nf][r] = zs[o];
                ea[nf][r] = as[o];
            }
#pragma unroll
        for (int nf = 0; nf < NFA; ++nf) { emean[nf] = ms[poff[nf]]; erstd[nf] = rs[poff[nf]]; }
    };
    // reversed block index kk = nblocks - 1 - k: convolution indices 1 + 2 kk (conv2^T) and 2 + 2 kk (conv1^T).  As in the forward: the
    // multiplies / act' / LayerNorm-backward of reversed block kk leave the outer fragment out from kk = 3, the store of g (conv2^T's
    // input, columns [1 + 2 kk, ..)) from kk = 4.
    const int kl = d.nblocks - 1;
    if (d.nblocks > 0) {
        if (kl >= 3) fetch_saved(kl, FULL{});              // (kk = 0: everything)
        else fetch_saved(kl, FULL{});
    }
    __syncthreads();
    // ---- tail^T: g = conv^T(cotangent)
    n1_f32x4 g[NF];
    n1_mm<NF, NF>(wA, tg, boff, g);
    const bool silu = d.act == SDA_ACT_SILU;
    const float inv_c = 1.f / (float)d.c, inv_v = 1.f / (float)(d.unbiased ? d.c - 1 : d.c);
    int conv = 1;                                          // index of the convolution whose weights sit in wB
    auto block = [&](const int k, auto NFS_, auto NFC_, auto NFN_) {
        // NFS: fragments of the g store, NFC: of everything after it, NFN: of the NEXT block's saved-tensor fetch
        constexpr int NFS = decltype(NFS_)::value, NFC = decltype(NFC_)::value;
        const n1_f32x4 emod = *reinterpret_cast<const n1_f32x4*>(smod + k * 64 + cbase);
        n1_store_tile<NF, NFS>(g, inside, rok, c, tg);
        n1_load_w(d.w, conv + 1, c, wA);                   // conv1^T of this block
        __syncthreads();
        // ---- conv2^T, x act'(z) -> LDS
        n1_f32x4 q[NF];
        n1_mm<NF, NFC>(wB, tg, boff, q);
        auto dact = [&](auto SILU_) {
#pragma unroll
            for (int nf = 0; nf < NFC; ++nf)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    q[nf][r] *= decltype(SILU_)::value ? sda_dact(SDA_ACT_SILU, ez[nf][r]) : sda_dact(d.act, ez[nf][r]);
        };
        if (silu) dact(std::true_type{});
        else dact(std::false_type{});
        n1_store_tile<NF, NFC>(q, inside, rok, c, tq);
        n1_load_w(d.w, conv + 2, c, wB);                   // conv2^T of the block before, or head^T
        __syncthreads();
        // ---- conv1^T -> gh; LayerNorm backward: g <- rstd (gh - mean_c(gh) - xh mean'_c(gh xh)) + g
        n1_f32x4 gh[NF], xh[NF];
        n1_mm<NF, NFC>(wA, tq, boff, gh);
        float s1[NF], s2[NF];
#pragma unroll
        for (int nf = 0; nf < NFC; ++nf) {
            s1[nf] = 0.f; s2[nf] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xh[nf][r] = rok[r] ? (ea[nf][r] + emod[r] - emean[nf]) * erstd[nf] : 0.f;
                const float gv = rok[r] ? gh[nf][r] : 0.f;
                s1[nf] += gv; s2[nf] += gv * xh[nf][r];
            }
        }
        n1_colsum2<NF, NFC>(s1, s2, red, c);
#pragma unroll
        for (int nf = 0; nf < NFC; ++nf) {
            const float av = s1[nf] * inv_c, bv = s2[nf] * inv_v;
#pragma unroll
            for (int r = 0; r < 4; ++r) g[nf][r] += erstd[nf] * (gh[nf][r] - av - xh[nf][r] * bv);
        }
        if (k > 0) fetch_saved(k - 1, NFN_);
        __syncthreads();                                   // (red is reused by the next block's sums)
        conv += 2;
    };
    {
        int kk = 0;
        const int nb = d.nblocks;
        for (; kk < nb && (kk < 2 || whole); ++kk) block(nb - 1 - kk, FULL{}, FULL{}, FULL{});
        if (kk < nb) { block(nb - 1 - kk, FULL{}, FULL{}, NARROW{}); ++kk; }            // kk = 2: the next block multiplies narrow
        if (kk < nb) { block(nb - 1 - kk, FULL{}, NARROW{}, NARROW{}); ++kk; }          // kk = 3
        for (; kk < nb; ++kk) block(nb - 1 - kk, NARROW{}, NARROW{}, NARROW{});
    }
    // ---- head^T (index 1 + 2 nblocks) -> input gradient (own columns, through the output strides)
    n1_f32x4 o[NF];
    const int nbn = whole ? 0 : d.nblocks;
    if (nbn >= 4) {
        n1_store_tile<NF, NR>(g, inside, rok, c, tg);
        __syncthreads();
        n1_mm<NF, NR>(wB, tg, boff, o);
    } else if (nbn == 3) {
        n1_store_tile<NF, NF>(g, inside, rok, c, tg);
        __syncthreads();
        n1_mm<NF, NR>(wB, tg, boff, o);
    } else {
        n1_store_tile<NF, NF>(g, inside, rok, c, tg);
        __syncthreads();
        n1_mm<NF, NF>(wB, tg, boff, o);
    }
    const int nfo = nbn >= 3 ? NR : NF;
    if constexpr (!FUSED) {
        float* ob = d.out + (int64_t)c.n * d.out_sn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* obr = ob + (int64_t)r * d.out_sc;
            const bool cok = cbase + r < d.cout;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
                if (nf < nfo && own[nf] && cok) obr[ooff[nf]] = o[nf][r];
        }
    } else {
        const float mu = f.coef[0], sg = f.coef[1];
        const bool bare = f.cx0 == 0.f && f.cx1 == 0.f && f.cn == 1.f;
        const float cx = f.cx0 + f.cx1 * sg;
        const float kk = sg / mu;
        float pr = 0.f, pc1 = 0.f;
        if (f.mode == 1) { pr = f.step_coef[0]; pc1 = f.step_coef[1]; }
        const float* gb = d.x + (int64_t)c.n * d.x_sn;                 // ghat (unscaled), same layout as the outputs
        const float* eb = f.eps + (int64_t)c.n * d.out_sn;
        float* ob = d.out + (int64_t)c.n * d.out_sn;
        float* xb = f.xs + (int64_t)c.n * d.out_sn;
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool cok = cbase + r < d.cout;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                if (!(nf < nfo && own[nf] && cok)) continue;
                const unsigned oo = ooff[nf] + (unsigned)(r * (int)d.out_sc);
                const float gv = gb[(unsigned)((cbase + r) * (int)d.x_sc) + poff[nf] * (unsigned)d.x_sx];
                const float vj = bare ? o[nf][r] : (gv * cx) + o[nf][r];
                const float ov = eb[oo] - kk * (gv - sg * vj);
                if (f.mode == 1) xb[oo] = pr * xb[oo] + pc1 * ov;
                else {
                    ob[oo] = ov;
                    acc += ov * ov;
                }
            }
        }
        if (f.mode == 2) {
            acc = sda_wave_sum(acc);
            __syncthreads();                               // (red is free: the last block's sums have been consumed)
            if (c.lane == 0) red[c.wave] = acc;
            __syncthreads();
            if (c.tid == 0) f.partial[(int64_t)c.n * f.partial_stride + (blockIdx.x - c.n * ptiles)] = (red[0] + red[1]) + (red[2] + red[3]);
        }
    }
}

static int net1d_check(const sda_net1d_desc* d, bool bwd) {
    if (!d || d->n < 1 || d->len < 1 || d->c < 2 || d->c > N1_MAXC || d->cin < 1 || d->cin > N1_MAXC || d->cout < 1 ||
        d->cout > N1_MAXC || d->nblocks < 0 || d->nblocks > SDA_NET1D_MAXB)
        return SDA_E_UNSUPPORTED;
    if (!d->x || !d->out || !d->w) return SDA_E_BADARG;
    // offsets inside one image are formed in 32 bits
    if ((int64_t)d->len * 64 >= (1LL << 30)) return SDA_E_UNSUPPORTED;
    auto span = [&](int64_t sc, int64_t sx, int ch) { return (sc < 0 ? -sc : sc) * ch + (sx < 0 ? -sx : sx) * (int64_t)d->len; };
    if (d->x_sc < 0 || d->x_sx < 0 || d->out_sc < 0 || d->out_sx < 0 || span(d->x_sc, d->x_sx, d->cin) >= (1LL << 30) ||
        span(d->out_sc, d->out_sx, d->cout) >= (1LL << 30))
        return SDA_E_UNSUPPORTED;
    const bool saves = d->a_save && d->z_save && d->mean_save && d->rstd_save;
    if (bwd && d->nblocks > 0 && !saves) return SDA_E_BADARG;
    if (!bwd && (d->a_save || d->z_save || d->mean_save || d->rstd_save) && !saves) return SDA_E_BADARG;
    return SDA_OK;
}

// columns per tile (16 NF): the fewer sequences there are, the more workgroups per sequence -- a tile's time is ~ its column
// count, and an idle CU is worth nothing: 64 columns (36 own positions with the six blocks of the Lorenz nets) when that fills the
// chip, else 48 (20 own), else 32 (4 own: Lorenz-63, one sequence of 64 positions = 16 workgroups)
static int net1d_nf(const sda_net1d_desc* d) {
    static const int forced = getenv("SDA_NET1D_NF") ? atoi(getenv("SDA_NET1D_NF")) : 0;
    const int H = 2 * d->nblocks + 2;
    if (forced >= 2 && forced <= 4 && 16 * forced - 2 * H >= 4) return forced;
    const int tp4 = 64 - 2 * H, tp3 = 48 - 2 * H, tp2 = 32 - 2 * H;
    if (tp4 < 4) return 0;
    auto wgs = [&](int tp) { return (int64_t)d->n * ((d->len + tp - 1) / tp); };
    if (wgs(tp4) >= 128 || tp3 < 4) return 4;
    if (wgs(tp3) >= 128 || tp2 < 4) return 3;
    return 2;
}

// the tiling of a launch: columns per tile (16 nf), own positions per tile, tiles per sequence, and whether a tile is a WHOLE sequence
// (zero padding, <= 80 positions: no halo) -- chosen when that costs fewer (rounds of workgroups over the CUs) x (columns per tile)
#define N1_MAXNF 5
static int net1d_tiling(const sda_net1d_desc* d, int* tp, int* ptiles, int* whole) {
    static const int wforce = getenv("SDA_NET1D_WHOLE") ? atoi(getenv("SDA_NET1D_WHOLE")) : -1;      // A/B runs: 0 never, 1 whenever possible
    *whole = 0;
    const int nf = net1d_nf(d);
    int64_t cost_t = -1;
    if (nf) {
        *tp = 16 * nf - 2 * (2 * d->nblocks + 2);
        *ptiles = (d->len + *tp - 1) / *tp;
        cost_t = (((int64_t)d->n * *ptiles + 255) / 256) * nf;
    }
    if (!d->circular && d->len <= 16 * N1_MAXNF && wforce != 0) {
        int nfw = (d->len + 15) / 16;
        if (nfw < 2) nfw = 2;
        const int64_t cost_w = (((int64_t)d->n + 255) / 256) * nfw;
        if (cost_t < 0 || cost_w < cost_t || wforce == 1) {
            *whole = 1; *tp = 16 * nfw; *ptiles = 1;
            return nfw;
        }
    }
    return nf;
}

template <bool BWD, int NF, bool FUSED>
static void net1d_launch_nf(const sda_net1d_desc* d, const sda_net1d_fuse& f, dim3 grid, int ptiles, int tp, int whole, hipStream_t stream) {
    if (BWD) hipLaunchKernelGGL((net1d_bwd_kernel<NF, FUSED>), grid, dim3(256), 0, stream, *d, f, ptiles, tp, whole);
    else hipLaunchKernelGGL((net1d_fwd_kernel<NF, FUSED>), grid, dim3(256), 0, stream, *d, f, ptiles, tp, whole);
}

template <bool BWD, bool FUSED>
static int net1d_launch(const sda_net1d_desc* d, const sda_net1d_fuse* fu, hipStream_t stream) {
    const int rc = net1d_check(d, BWD);
    if (rc != SDA_OK) return rc;
    int tp, ptiles, whole;
    const int nf = net1d_tiling(d, &tp, &ptiles, &whole);
    if (!nf) return SDA_E_UNSUPPORTED;
    if ((int64_t)d->n * ptiles > 0x7fffffffLL) return SDA_E_UNSUPPORTED;
    sda_net1d_fuse f = {};
    if (FUSED) {
        f = *fu;
        if (!f.coef) return SDA_E_BADARG;
        // eps / ghat / out / x share ONE layout (the output strides); the fused epilogues address all of them with it
        if (d->x_sn != d->out_sn || d->x_sc != d->out_sc || d->x_sx != d->out_sx || d->cin != d->cout) return SDA_E_BADARG;
        if (!BWD) {
            if (!f.y || !f.ghat || f.p_step < 1 || f.c_step < 1 || f.p_start < 0 || f.c_start < 0 || f.p_stop > d->len ||
                f.c_stop > d->cout || f.p_stop <= f.p_start || f.c_stop <= f.c_start)
                return SDA_E_BADARG;
        } else {
            if (!f.eps || f.mode < 0 || f.mode > 2 || (f.mode == 1 && (!f.xs || !f.step_coef)) ||
                (f.mode == 2 && (!f.partial || f.partial_stride < ptiles)))
                return SDA_E_BADARG;
            if (f.mode != 1) f.xs = d->out;                  // (never dereferenced; keeps the pointer arithmetic defined)
        }
    }
    const dim3 grid((unsigned)(d->n * ptiles));
    switch (nf) {
        case 2: net1d_launch_nf<BWD, 2, FUSED>(d, f, grid, ptiles, tp, whole, stream); break;
        case 3: net1d_launch_nf<BWD, 3, FUSED>(d, f, grid, ptiles, tp, whole, stream); break;
        case 4: net1d_launch_nf<BWD, 4, FUSED>(d, f, grid, ptiles, tp, whole, stream); break;
        default: net1d_launch_nf<BWD, 5, FUSED>(d, f, grid, ptiles, tp, whole, stream); break;
    }
    return sda_launch_status();
}

extern "C" int sda_net1d_fwd(const sda_net1d_desc* d, void* stream) { return net1d_launch<false, false>(d, nullptr, (hipStream_t)stream); }
extern "C" int sda_net1d_bwd(const sda_net1d_desc* d, void* stream) { return net1d_launch<true, false>(d, nullptr, (hipStream_t)stream); }
extern "C" int sda_net1d_fwd_fused(const sda_net1d_desc* d, const sda_net1d_fuse* f, void* stream) {
    if (!f) return SDA_E_BADARG;
    return net1d_launch<false, true>(d, f, (hipStream_t)stream);
}
extern "C" int sda_net1d_bwd_fused(const sda_net1d_desc* d, const sda_net1d_fuse* f, void* stream) {
    if (!f) return SDA_E_BADARG;
    return net1d_launch<true, true>(d, f, (hipStream_t)stream);
}
// tiles per sequence of the launch that would serve `d` (the row length of the fused backward's partial sums), <= 0: unsupported
extern "C" int sda_net1d_tiles(const sda_net1d_desc* d) {
    if (!d || d->len < 1 || d->nblocks < 0 || d->nblocks > SDA_NET1D_MAXB) return SDA_E_UNSUPPORTED;
    int tp, ptiles, whole;
    return net1d_tiling(d, &tp, &ptiles, &whole) ? ptiles : SDA_E_UNSUPPORTED;
}
