// A whole residual MLP (the Lorenz LOCAL score kernel: ScoreNet / ResMLP, sda/nn.py:31-71, sda/score.py:38-63 -- the network of four of
// the five checkpoints of experiments/lorenz/eval.py:33-39) in ONE launch, and its input VJP in one more.  The per-layer path
// (linear.hip: sda_linear / sda_row_ln) was ~20 launches forward + ~20 backward per score evaluation, each a separate pass over
// (rows x 128) activations in HBM; here a workgroup takes 16 NF rows (trajectory windows) through EVERY layer:
//   * rows are independent (LayerNorm is over a row's features), so there is no halo and no inter-workgroup dependency at all;
//   * wave w owns output features 32 w .. 32 w + 31 of every GEMM (two 16-row MFMA fragments, v_mfma_f32_16x16x4_f32: exact fp32;
//     A = weights held in registers, B = the activation tile in LDS, D = [feature][row]).  The K index of a fragment step is a
//     free permutation: lane group kq supplies features (K / 4) kq + s at step s, so a lane's B values of a row are CONSECUTIVE
//     features (16-byte LDS reads) and its A values are consecutive floats of one weight row -- torch's [out][in] layout read as is
//     (zero-padded copies: widths <= 16 -> 16, else -> 128), the backward-data GEMMs read the transposed copy the same way;
//   * the weights of the next GEMM's first fragment are loaded while the second fragment multiplies, the second fragment's while the
//     epilogue / LayerNorm runs: one register set, no exposed round trip;
//   * the residual stream stays in registers in D layout between layers; LayerNorm: lane-local over its 8 features, shuffles across
//     the 4 lane groups, LDS across the 4 waves (two passes: mean, centred squares; v_rsq_f32);
//   * saved for the VJP (own rows, 16-byte stores): block inputs, pre-activations, mean / rstd -- as the per-layer path saved.
// Roofline: the Lorenz local net is 0.34 MFLOP per window and direction; at 62 464 windows (eval.py's batch) 21.5 GFLOP = 0.14 ms of
// fp32 MFMA time per direction -- MFMA-bound once the launches are gone.
#include "sda_common.hpp"
#include <stdlib.h>
#include <type_traits>

#define ML_LD 132                      // LDS row stride (floats): 16-byte reads of consecutive rows land 4 banks apart
#define ML_W 128                       // padded width
#define ML_RL 20                       // floats per row line of the reduction exchange (16 used; 20: the 16 rows of a fragment hit 16 bank groups)

typedef float ml_f32x4 __attribute__((ext_vector_type(4)));

#ifdef SDA_ML_TRACE                    // tooling (tools/mlp_trace.py): per-phase cycle sums of workgroup 0 / thread 0
__device__ long long ml_trace[16];
#define ML_T0() long long ml_tl = __builtin_readcyclecounter()
#define ML_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long n_ = __builtin_readcyclecounter(); ml_trace[k] += n_ - ml_tl; ml_tl = n_; } } while (0)
extern "C" int sda_ml_trace_read(long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ml_trace), sizeof(long long) * 16) != hipSuccess) return SDA_E_BADARG;
    if (reset) { long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(ml_trace), z, sizeof(z)); }
    return SDA_OK;
}
#else
#define ML_T0() do {} while (0)
#define ML_STAMP(k) do {} while (0)
#endif

struct MlCtx {
    int tid, lane, wave, kq, li;
    int64_t row0;
    int rows;
};

static inline int ml_pad(int f) { return f <= 16 ? 16 : ML_W; }
__device__ __forceinline__ int ml_padd(int f) { return f <= 16 ? 16 : ML_W; }

// weights of GEMM fragment mf of this wave: wreg[s] = Wp[row = 16 (2 wave + mf) + li][KS kq + s], Wp = the padded [M][K] matrix.  In
// memory the matrix is packed in LANE order -- [fragment slot 2 wave + mf][q][lane][4]: element e of lane (kq, li) = Wp[16 slot + li][KS kq +
// 4 q + e] -- so every load instruction of a wave is one contiguous KiB.  (Read from the row-major matrix each instruction touched 64
// different cache lines, 16 bytes of each, and the 64 KiB of a 128 x 128 layer do not survive in a 32 KiB L1 until the line's other
// seven pieces are asked for: ~8x the bytes through the CU's L2 port, the kernel's bound in its first version -- 421 us per forward of the
// eval.py batch against 137 us of fp32 MFMA time.)
template <int KS>
__device__ __forceinline__ void ml_load_w(const float* w, int K, int M, int mf, const MlCtx& c, float (&wreg)[32]) {
    const int slot = 2 * c.wave + mf;
    if (16 * slot < M) {                                   // (wave uniform)
        const ml_f32x4* src = reinterpret_cast<const ml_f32x4*>(w) + (int64_t)slot * (KS / 4) * 64 + c.lane;
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
            const ml_f32x4 v = src[q * 64];
            wreg[4 * q] = v[0]; wreg[4 * q + 1] = v[1]; wreg[4 * q + 2] = v[2]; wreg[4 * q + 3] = v[3];
        }
    }
}
__device__ __forceinline__ void ml_load_w_any(const float* w, int K, int M, int mf, const MlCtx& c, float (&wreg)[32]) {
    if (K == ML_W) ml_load_w<32>(w, K, M, mf, c, wreg);
    else ml_load_w<4>(w, K, M, mf, c, wreg);
}

// acc[nf] = sum_k wreg[k] * tile[row 16 nf + li][feature KS kq + k]   for one fragment (16 output features x 16 NF rows)
template <int NF, int KS>
__device__ __forceinline__ void ml_mm(const float (&wreg)[32], const float* tile, const MlCtx& c, ml_f32x4 (&acc)[NF]) {
    const float* brow = tile + c.li * ML_LD + KS * c.kq;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[nf] = ml_f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int CH = KS < 8 ? KS : 8;                    // K steps per operand chunk
    constexpr int NCH = KS / CH;
    // two operand sets: chunk j + 1 is read while chunk j multiplies (a single set exposed the LDS latency at every chunk boundary)
    ml_f32x4 bv[2][NF][CH / 4];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) bv[0][nf][q] = *reinterpret_cast<const ml_f32x4*>(brow + 16 * nf * ML_LD + 4 * q);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (j + 1 < NCH) {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int q = 0; q < CH / 4; ++q)
                    bv[(j + 1) & 1][nf][q] = *reinterpret_cast<const ml_f32x4*>(brow + 16 * nf * ML_LD + (j + 1) * CH + 4 * q);
        }
#pragma unroll
        for (int e = 0; e < CH; ++e)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
                acc[nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j * CH + e], bv[j & 1][nf][e >> 2][e & 3], acc[nf], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                 // (keeps two chunks of operands live, not all of them)
    }
}

// one GEMM on the tile: both fragments of this wave; the NEXT GEMM's weights are loaded fragment by fragment behind the multiplies
template <int NF>
__device__ __forceinline__ void ml_gemm(float (&w0)[32], float (&w1)[32], const float* wcur_unused, int K, int M, const float* tile,
                                        const MlCtx& c, ml_f32x4 (&acc)[2][NF], const float* wnext, int Kn, int Mn) {
    const bool own0 = 16 * (2 * c.wave) < M, own1 = 16 * (2 * c.wave + 1) < M;
    if (own0) { if (K == ML_W) ml_mm<NF, 32>(w0, tile, c, acc[0]); else ml_mm<NF, 4>(w0, tile, c, acc[0]); }
    if (wnext) ml_load_w_any(wnext, Kn, Mn, 0, c, w0);
    if (own1) { if (K == ML_W) ml_mm<NF, 32>(w1, tile, c, acc[1]); else ml_mm<NF, 4>(w1, tile, c, acc[1]); }
    if (wnext) ml_load_w_any(wnext, Kn, Mn, 1, c, w1);
    if (!own0) for (int nf = 0; nf < NF; ++nf) acc[0][nf] = ml_f32x4{0.f, 0.f, 0.f, 0.f};
    if (!own1) for (int nf = 0; nf < NF; ++nf) acc[1][nf] = ml_f32x4{0.f, 0.f, 0.f, 0.f};
}

// registers (D layout) -> tile rows; features >= width and rows beyond the batch are written as zeros
template <int NF>
__device__ __forceinline__ void ml_store_tile(const ml_f32x4 (&v)[2][NF], int width, const MlCtx& c, float* tile) {
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        const int fb = 32 * c.wave + 16 * mf + 4 * c.kq;
        if (fb >= ml_padd(width)) continue;                // (features beyond the padded width do not exist in the tile's K range)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            ml_f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (fb + r < width && c.row0 + 16 * nf + c.li < c.rows) ? v[mf][nf][r] : 0.f;
            *reinterpret_cast<ml_f32x4*>(tile + (16 * nf + c.li) * ML_LD + fb) = o;
        }
    }
}

// sum over the features of every row: lane-local (2 fragments x 4), then ONE exchange through LDS -- every (wave, lane group) writes its
// partial of a row into that row's 16-float line, and each lane adds up the line of its rows (4 x 16-byte reads, a fixed tree).  (Two
// cross-lane shuffles per row before the exchange were two more dependent LDS-crossbar round trips per reduction.)
template <int NF>
__device__ __forceinline__ void ml_rowsum(float (&s)[NF], float* red, const MlCtx& c) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) red[(16 * nf + c.li) * ML_RL + 4 * c.wave + c.kq] = s[nf];
    __syncthreads();
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const ml_f32x4* line = reinterpret_cast<const ml_f32x4*>(red + (16 * nf + c.li) * ML_RL);
        const ml_f32x4 p0 = line[0], p1 = line[1], p2 = line[2], p3 = line[3];
        const ml_f32x4 q = (p0 + p1) + (p2 + p3);
        s[nf] = (q[0] + q[1]) + (q[2] + q[3]);
    }
}

__device__ __forceinline__ void ml_ctx(MlCtx& c, const sda_mlp_desc& d, int nc) {
    c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6); c.kq = c.lane >> 4; c.li = c.lane & 15;
    c.row0 = (int64_t)blockIdx.x * nc; c.rows = d.rows;
}

// rows [row0, row0 + NC) x `width` features of a row-major source -> tile (zero padded), then -> registers in D layout
template <int NF>
__device__ __forceinline__ void ml_load_rows(const float* src, int64_t ld, int width, const MlCtx& c, float* tile, ml_f32x4 (&a)[2][NF]) {
    constexpr int NC = 16 * NF;
    const int wp = ml_padd(width), lw = wp == 16 ? 4 : 7;
    // (all loads first, unconditional from clamped addresses: a load under a condition is a branch with its own s_waitcnt -- 32 serial
    // round trips, 35 000 cycles per tile in the first version, tools/mlp_trace.py)
    constexpr int PER = NC * ML_W / 256;
    float v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = c.tid + 256 * k, r = i >> lw, f = i & (wp - 1);
        const int64_t gr = c.row0 + r;
        const bool ok = i < NC * wp && gr < c.rows && f < width;
        v[k] = src[ok ? gr * ld + f : 0];
        if (!ok) v[k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = c.tid + 256 * k;
        if (i < NC * wp) tile[(i >> lw) * ML_LD + (i & (wp - 1))] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        const int fb = 32 * c.wave + 16 * mf + 4 * c.kq;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
            a[mf][nf] = fb < wp ? *reinterpret_cast<const ml_f32x4*>(tile + (16 * nf + c.li) * ML_LD + fb) : ml_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();                                       // (the tile is overwritten by the first GEMM's input)
}

// ------------------------------------------------------------------------------------------------------------ forward
// (NF = 4: the 64-row tile needs ~300 registers per lane -- one workgroup per CU; at two per CU it spilled 128-266 of them.  NF <= 2
// fits 256 registers: two workgroups per CU, one's LayerNorm / epilogue phases under the other's multiplies.)
template <int NF>
__global__ __launch_bounds__(256, NF <= 2 ? 2 : 1) void mlp_fwd_kernel(const sda_mlp_desc d) {
    constexpr int NC = 16 * NF;
    __shared__ __attribute__((aligned(16))) float tA[NC * ML_LD];
    __shared__ __attribute__((aligned(16))) float tB[NC * ML_LD];
    __shared__ __attribute__((aligned(16))) float red[2 * ML_RL * NC];
    MlCtx c;
    ml_ctx(c, d, NC);
    float w0[32], w1[32];
    ml_load_w_any(d.w + d.w_off[0], ml_padd(d.in_f[0]), ml_padd(d.out_f[0]), 0, c, w0);
    ml_load_w_any(d.w + d.w_off[0], ml_padd(d.in_f[0]), ml_padd(d.out_f[0]), 1, c, w1);
    ml_f32x4 a[2][NF];
    ML_T0();
    ml_load_rows<NF>(d.x, d.x_ld, d.in_f[0], c, tA, a);
    ML_STAMP(0);                                           // input rows
    const bool silu = d.act == SDA_ACT_SILU;
    int rb = 0;                                            // residual-block counter (index into the saves)
    for (int g = 0; g < d.ngemm; ++g) {
        const int K = ml_padd(d.in_f[g]), M = ml_padd(d.out_f[g]);
        const bool last = g + 1 == d.ngemm;
        const float* wn = last ? nullptr : d.w + d.w_off[g + 1];
        const int Kn = last ? 16 : ml_padd(d.in_f[g + 1]), Mn = last ? 16 : ml_padd(d.out_f[g + 1]);
        const float* bg = d.bias + d.b_off[g];
        ml_f32x4 bias[2];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            const int fb = 32 * c.wave + 16 * mf + 4 * c.kq;
            bias[mf] = fb < M ? *reinterpret_cast<const ml_f32x4*>(bg + fb) : ml_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        ml_f32x4 acc[2][NF];
        if (d.kind[g] == 0) {
            // ---- Linear: a <- W a + b
            ml_store_tile<NF>(a, d.in_f[g], c, tA);
            __syncthreads();
            ML_STAMP(1);
            ml_gemm<NF>(w0, w1, nullptr, K, M, tA, c, acc, wn, Kn, Mn);
            ML_STAMP(2);
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) a[mf][nf] = acc[mf][nf] + bias[mf];
            __syncthreads();                               // (tA is rewritten by the next layer's input)
            ML_STAMP(4);
        } else if (d.kind[g] == 1) {
            // ---- residual block, first half: save a; u = LN(a); z = W1 u + b1 (saved); act(z) -> tB
            const int cw = d.in_f[g];
            const float inv_c = 1.f / (float)cw, inv_v = 1.f / (float)(d.unbiased ? cw - 1 : cw);
            if (d.a_save) {
                float* as = d.a_save + (int64_t)rb * d.save_stride;
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) {
                    const int fb = 32 * c.wave + 16 * mf + 4 * c.kq;
                    if (fb >= K) continue;
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) {
                        const int64_t gr = c.row0 + 16 * nf + c.li;
                        if (gr < c.rows) *reinterpret_cast<ml_f32x4*>(as + gr * d.save_ld + fb) = a[mf][nf];
                    }
                }
            }
            float s[NF];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                s[nf] = 0.f;
#pragma unroll
                for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[nf] += (32 * c.wave + 16 * mf + 4 * c.kq + r < cw) ? a[mf][nf][r] : 0.f;
            }
            ml_rowsum<NF>(s, red, c);
            float mean[NF], rstd[NF];
            ml_f32x4 u[2][NF];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                mean[nf] = s[nf] * inv_c;
                s[nf] = 0.f;
#pragma unroll
                for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dl = a[mf][nf][r] - mean[nf];
                        s[nf] += (32 * c.wave + 16 * mf + 4 * c.kq + r < cw) ? dl * dl : 0.f;
                    }
            }
            ml_rowsum<NF>(s, red + ML_RL * NC, c);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                rstd[nf] = __builtin_amdgcn_rsqf(s[nf] * inv_v + d.eps);
#pragma unroll
                for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) u[mf][nf][r] = (a[mf][nf][r] - mean[nf]) * rstd[nf];
            }
            if (d.mean_save && c.wave == 0 && c.kq == 0) {
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    const int64_t gr = c.row0 + 16 * nf + c.li;
                    if (gr < c.rows) {
                        d.mean_save[(int64_t)rb * d.stat_stride + gr] = mean[nf];
                        d.rstd_save[(int64_t)rb * d.stat_stride + gr] = rstd[nf];
                    }
                }
            }
            ML_STAMP(3);                                   // a_save, LayerNorm
            ml_store_tile<NF>(u, cw, c, tA);
            __syncthreads();
            ML_STAMP(1);
            ml_gemm<NF>(w0, w1, nullptr, K, M, tA, c, acc, wn, Kn, Mn);
            ML_STAMP(2);
            float* zs = d.z_save ? d.z_save + (int64_t)rb * d.save_stride : nullptr;
            auto epi = [&](auto SILU_) {
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) {
                    const int fb = 32 * c.wave + 16 * mf + 4 * c.kq;
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) {
                        const ml_f32x4 zv = acc[mf][nf] + bias[mf];
                        const int64_t gr = c.row0 + 16 * nf + c.li;
                        if (zs && fb < M && gr < c.rows) *reinterpret_cast<ml_f32x4*>(zs + gr * d.save_ld + fb) = zv;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[mf][nf][r] = decltype(SILU_)::value ? sda_act(SDA_ACT_SILU, zv[r]) : sda_act(d.act, zv[r]);
                    }
                }
            };
            if (silu) epi(std::true_type{});
            else epi(std::false_type{});
            ML_STAMP(4);                                   // z_save, activation
            ml_store_tile<NF>(acc, d.out_f[g], c, tB);
            __syncthreads();
            ML_STAMP(1);
        } else {
            // ---- residual block, second half: a += W2 act(z) + b2
            ml_gemm<NF>(w0, w1, nullptr, K, M, tB, c, acc, wn, Kn, Mn);
            ML_STAMP(2);
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) a[mf][nf] += acc[mf][nf] + bias[mf];
            ++rb;
        }
    }
    // ---- output rows (real width of the last layer)
    const int wo = d.out_f[d.ngemm - 1];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        const int fb = 32 * c.wave + 16 * mf + 4 * c.kq;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int64_t gr = c.row0 + 16 * nf + c.li;
            if (gr >= c.rows) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (fb + r < wo) d.out[gr * d.out_ld + fb + r] = a[mf][nf][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ input VJP
// d.w = the TRANSPOSED padded matrices ([in_pad][out_pad] per GEMM, same offsets table); x = cotangent rows (width out_f[last]), out =
// input-gradient rows (width in_f[0]); the GEMM list is walked backwards.
template <int NF>
__global__ __launch_bounds__(256, NF <= 2 ? 2 : 1) void mlp_bwd_kernel(const sda_mlp_desc d) {
    constexpr int NC = 16 * NF;
    __shared__ __attribute__((aligned(16))) float tA[NC * ML_LD];
    __shared__ __attribute__((aligned(16))) float tB[NC * ML_LD];
    __shared__ __attribute__((aligned(16))) float red[2 * ML_RL * NC];
    MlCtx c;
    ml_ctx(c, d, NC);
    float w0[32], w1[32];
    const int gl = d.ngemm - 1;
    // (backward GEMM of forward GEMM g: M = in_pad[g] output features, K = out_pad[g])
    ml_load_w_any(d.w + d.w_off[gl], ml_padd(d.out_f[gl]), ml_padd(d.in_f[gl]), 0, c, w0);
    ml_load_w_any(d.w + d.w_off[gl], ml_padd(d.out_f[gl]), ml_padd(d.in_f[gl]), 1, c, w1);
    ml_f32x4 gr_[2][NF];
    ml_load_rows<NF>(d.x, d.x_ld, d.out_f[gl], c, tA, gr_);
    const bool silu = d.act == SDA_ACT_SILU;
    int rb = 0;
    for (int g = 0; g < d.ngemm; ++g) rb += d.kind[g] == 2;
    for (int g = gl; g >= 0; --g) {
        const int K = ml_padd(d.out_f[g]), M = ml_padd(d.in_f[g]);
        const bool last = g == 0;
        const float* wn = last ? nullptr : d.w + d.w_off[g - 1];
        const int Kn = last ? 16 : ml_padd(d.out_f[g - 1]), Mn = last ? 16 : ml_padd(d.in_f[g - 1]);
        ml_f32x4 acc[2][NF];
        if (d.kind[g] == 0) {
            ml_store_tile<NF>(gr_, d.out_f[g], c, tA);
            __syncthreads();
            ml_gemm<NF>(w0, w1, nullptr, K, M, tA, c, acc, wn, Kn, Mn);
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) gr_[mf][nf] = acc[mf][nf];
            __syncthreads();
        } else if (d.kind[g] == 2) {
            // ---- second half of a block, backwards: q = W2^T g, x act'(z) -> tB
            --rb;
            ml_store_tile<NF>(gr_, d.out_f[g], c, tA);
            __syncthreads();
            ml_gemm<NF>(w0, w1, nullptr, K, M, tA, c, acc, wn, Kn, Mn);
            const float* zs = d.z_save + (int64_t)rb * d.save_stride;
            auto dact = [&](auto SILU_) {
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) {
                    const int fb = 32 * c.wave + 16 * mf + 4 * c.kq;
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) {
                        const int64_t grw = c.row0 + 16 * nf + c.li;
                        ml_f32x4 zv = {0.f, 0.f, 0.f, 0.f};
                        if (fb < M && grw < c.rows) zv = *reinterpret_cast<const ml_f32x4*>(zs + grw * d.save_ld + fb);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[mf][nf][r] *= decltype(SILU_)::value ? sda_dact(SDA_ACT_SILU, zv[r]) : sda_dact(d.act, zv[r]);
                    }
                }
            };
            if (silu) dact(std::true_type{});
            else dact(std::false_type{});
            ml_store_tile<NF>(acc, d.in_f[g], c, tB);
            __syncthreads();
        } else {
            // ---- first half, backwards: gh = W1^T q; g += LN^T(gh)
            const int cw = d.in_f[g];
            const float inv_c = 1.f / (float)cw, inv_v = 1.f / (float)(d.unbiased ? cw - 1 : cw);
            ml_gemm<NF>(w0, w1, nullptr, K, M, tB, c, acc, wn, Kn, Mn);
            const float* as = d.a_save + (int64_t)rb * d.save_stride;
            ml_f32x4 xh[2][NF];
            float s1[NF], s2[NF], rs[NF];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const int64_t grw = c.row0 + 16 * nf + c.li;
                const bool rowok = grw < c.rows;
                const float mean = rowok ? d.mean_save[(int64_t)rb * d.stat_stride + grw] : 0.f;
                rs[nf] = rowok ? d.rstd_save[(int64_t)rb * d.stat_stride + grw] : 0.f;
                s1[nf] = 0.f; s2[nf] = 0.f;
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) {
                    const int fb = 32 * c.wave + 16 * mf + 4 * c.kq;
                    ml_f32x4 av = {0.f, 0.f, 0.f, 0.f};
                    if (fb < M && rowok) av = *reinterpret_cast<const ml_f32x4*>(as + grw * d.save_ld + fb);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool fok = fb + r < cw;
                        xh[mf][nf][r] = fok ? (av[r] - mean) * rs[nf] : 0.f;
                        const float gv = fok ? acc[mf][nf][r] : 0.f;
                        s1[nf] += gv; s2[nf] += gv * xh[mf][nf][r];
                    }
                }
            }
            ml_rowsum<NF>(s1, red, c);
            ml_rowsum<NF>(s2, red + ML_RL * NC, c);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const float av = s1[nf] * inv_c, bv = s2[nf] * inv_v;
#pragma unroll
                for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gr_[mf][nf][r] += rs[nf] * (acc[mf][nf][r] - av - xh[mf][nf][r] * bv);
            }
            __syncthreads();                               // (red / the tiles are reused by the next layer)
        }
    }
    const int wo = d.in_f[0];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        const int fb = 32 * c.wave + 16 * mf + 4 * c.kq;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int64_t grw = c.row0 + 16 * nf + c.li;
            if (grw >= c.rows) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (fb + r < wo) d.out[grw * d.out_ld + fb + r] = gr_[mf][nf][r];
        }
    }
}

static int mlp_check(const sda_mlp_desc* d, bool bwd) {
    if (!d || d->rows < 1 || d->ngemm < 1 || d->ngemm > SDA_MLP_MAXG) return SDA_E_UNSUPPORTED;
    if (!d->x || !d->out || !d->w || (!bwd && !d->bias)) return SDA_E_BADARG;
    int nres = 0;
    for (int g = 0; g < d->ngemm; ++g) {
        if (d->in_f[g] < 1 || d->out_f[g] < 1 || d->in_f[g] > ML_W || d->out_f[g] > ML_W || d->kind[g] < 0 || d->kind[g] > 2) return SDA_E_UNSUPPORTED;
        if (g > 0 && d->in_f[g] != d->out_f[g - 1]) return SDA_E_BADARG;
        if (d->kind[g] == 1) {
            if (g + 1 >= d->ngemm || d->kind[g + 1] != 2 || d->in_f[g] != d->out_f[g] || d->out_f[g + 1] != d->in_f[g]) return SDA_E_BADARG;
            if (d->unbiased && d->in_f[g] < 2) return SDA_E_UNSUPPORTED;
            ++nres;
        }
        if (d->kind[g] == 2 && (g == 0 || d->kind[g - 1] != 1)) return SDA_E_BADARG;
        if ((d->w_off[g] & 3) || (d->b_off[g] & 3)) return SDA_E_BADARG;
    }
    if ((reinterpret_cast<uintptr_t>(d->w) & 15) || (!bwd && (reinterpret_cast<uintptr_t>(d->bias) & 15))) return SDA_E_BADARG;
    const bool saves = d->a_save && d->z_save && d->mean_save && d->rstd_save;
    if (nres > 0) {
        if (bwd && !saves) return SDA_E_BADARG;
        if (!bwd && (d->a_save || d->z_save || d->mean_save || d->rstd_save) && !saves) return SDA_E_BADARG;
        if (saves && (d->save_ld < ML_W || (d->save_ld & 3) || (reinterpret_cast<uintptr_t>(d->a_save) & 15) ||
                      (reinterpret_cast<uintptr_t>(d->z_save) & 15) || (d->save_stride & 3)))
            return SDA_E_BADARG;
    }
    return SDA_OK;
}

template <bool BWD>
static int mlp_launch(const sda_mlp_desc* d, hipStream_t stream) {
    const int rc = mlp_check(d, BWD);
    if (rc != SDA_OK) return rc;
    // rows per tile: 64 (one workgroup per CU) / 32 (two per CU) when that fills the chip, else 16-row tiles
    static const int forced = getenv("SDA_MLP_NF") ? atoi(getenv("SDA_MLP_NF")) : 0;
    int nf = d->rows >= 32 * 512 ? 2 : 1;
    if (forced == 1 || forced == 2 || forced == 4) nf = forced;
    const int nc = 16 * nf;
    const int64_t tiles = ((int64_t)d->rows + nc - 1) / nc;
    if (tiles > 0x7fffffffLL) return SDA_E_UNSUPPORTED;
    const dim3 grid((unsigned)tiles);
    if (BWD) {
        if (nf == 4) hipLaunchKernelGGL(mlp_bwd_kernel<4>, grid, dim3(256), 0, stream, *d);
        else if (nf == 2) hipLaunchKernelGGL(mlp_bwd_kernel<2>, grid, dim3(256), 0, stream, *d);
        else hipLaunchKernelGGL(mlp_bwd_kernel<1>, grid, dim3(256), 0, stream, *d);
    } else {
        if (nf == 4) hipLaunchKernelGGL(mlp_fwd_kernel<4>, grid, dim3(256), 0, stream, *d);
        else if (nf == 2) hipLaunchKernelGGL(mlp_fwd_kernel<2>, grid, dim3(256), 0, stream, *d);
        else hipLaunchKernelGGL(mlp_fwd_kernel<1>, grid, dim3(256), 0, stream, *d);
    }
    return sda_launch_status();
}

extern "C" int sda_mlp_fwd(const sda_mlp_desc* d, void* stream) { return mlp_launch<false>(d, (hipStream_t)stream); }
extern "C" int sda_mlp_bwd(const sda_mlp_desc* d, void* stream) { return mlp_launch<true>(d, (hipStream_t)stream); }
