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
//     in registers, straight from the packed [tap][k_pad][m_pad] layout, B from LDS).  The weights of convolution i + 1 are
//     loaded (L2 -> registers, 48 dwords per lane) while convolution i multiplies: two register sets, alternating.
//   * the residual stream lives in registers in MFMA D layout (channel 16 w + 4 kq + r, column 16 nf + li) between layers;
//     LayerNorm statistics are reduced lane-locally, across the four lane groups (shuffles), across the four waves (LDS).
//   * what the VJP needs (block inputs a, pre-activations z, mean / rstd per position) is written for the own columns only;
//     the backward kernel reads it for its halo columns too (written by the neighbours' forward).
//   * input / output are addressed through (image, channel, position) strides: the (B, L, C) <-> (B, C, L) transposes of
//     MCScoreWrapper (sda/score.py:104-110) cost nothing on either side.
#include "sda_common.hpp"
#include <type_traits>

#define N1_LD 80                       // LDS row stride: 80 mod 32 = 16 -> the two k rows of a 32-lane access group hit disjoint banks
#define N1_MAXC 64

typedef float n1_f32x4 __attribute__((ext_vector_type(4)));

struct N1Ctx {
    int tid, lane, wave, kq, li, co0, n, p0, H, len;
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

// all A fragments of one convolution for this wave: one batch of unconditional loads from clamped addresses
__device__ __forceinline__ void n1_load_w(const float* w, int k_pad, int m_pad, const N1Ctx& c, float (&wreg)[3][16]) {
    const bool on = c.co0 < m_pad;
    const float* wl = w + c.kq * m_pad + (on ? c.co0 : 0) + c.li;
    const int frag = 4 * m_pad, tapstride = k_pad * m_pad, last = ((k_pad >> 2) - 1) * frag;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int cb = 0; cb < 16; ++cb) {
            const int off = cb * frag < last ? cb * frag : last;           // (clamped: rows beyond k_pad multiply zero tile rows)
            wreg[tap][cb] = wl[tap * tapstride + off];
        }
}

// acc[nf] = sum_{tap, cb} A(tap, cb) B[4 cb + k][16 nf + li + tap]   (tile rows of N1_LD floats; tile column jj <-> conv
// column jj - 1).  All 16 K fragments, unconditionally (see block1d.hip: a runtime trip count costs more than the surplus MFMAs).
template <int NF>
__device__ __forceinline__ void n1_mm(const float (&wreg)[3][16], const float* tile, const N1Ctx& c, n1_f32x4 (&acc)[NF]) {
    const float* brow = tile + c.kq * N1_LD + c.li;
    float bv[2][NF][3];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        acc[nf] = n1_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) bv[0][nf][tap] = brow[16 * nf + tap];
    }
#pragma unroll
    for (int cb = 0; cb < 16; ++cb) {
        const int cn = cb + 1 < 16 ? cb + 1 : cb;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) bv[(cb + 1) & 1][nf][tap] = brow[4 * cn * N1_LD + 16 * nf + tap];
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
                acc[nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap][cb], bv[cb & 1][nf][tap], acc[nf], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * NF, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * NF, 0);
    }
}

// a strided (image, channel, position) tensor -> tile rows [0, 64) x conv columns [0, NC): thread (column j = tid & 63,
// channel group tid >> 6): channels sub + 4 i.  Rows >= channels and columns outside the sequence are zero.
template <int NF>
__device__ __forceinline__ void n1_load_tile(const float* src, int64_t sn, int64_t sc, int64_t sx, int channels, const N1Ctx& c,
                                             float* tile) {
    constexpr int NC = 16 * NF;
    const int j = c.lane, sub = c.wave;
    if (j < NC) {
        bool inside;
        const int ps = n1_pos(c, j, inside);
        const float* base = src + (int64_t)c.n * sn + (int64_t)ps * sx;
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = sub + 4 * i, cic = ci < channels ? ci : channels - 1;
            v[i] = base[(int64_t)cic * sc];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = sub + 4 * i;
            tile[ci * N1_LD + 1 + j] = (inside && ci < channels) ? v[i] : 0.f;
        }
    }
}

// registers in D layout -> tile rows of this wave (masked: columns outside the sequence and channels >= c are zero)
template <int NF>
__device__ __forceinline__ void n1_store_tile(const n1_f32x4 (&v)[NF], const bool (&inside)[NF], int c_real, const N1Ctx& c, float* tile) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = c.co0 + 4 * c.kq + r;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) tile[co * N1_LD + 1 + 16 * nf + c.li] = (inside[nf] && co < c_real) ? v[nf][r] : 0.f;
    }
}

// sum over the channels of every column: lane-local over r, across the 4 lane groups, across the 4 waves (LDS; one barrier)
template <int NF>
__device__ __forceinline__ void n1_colsum(float (&s)[NF], float* red, const N1Ctx& c) {
    constexpr int NC = 16 * NF;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        s[nf] += __shfl_xor(s[nf], 16, 64);
        s[nf] += __shfl_xor(s[nf], 32, 64);
        if (c.kq == 0) red[c.wave * NC + 16 * nf + c.li] = s[nf];
    }
    __syncthreads();
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const int m = 16 * nf + c.li;
        s[nf] = (red[m] + red[NC + m]) + (red[2 * NC + m] + red[3 * NC + m]);
    }
}

__device__ __forceinline__ void n1_ctx(N1Ctx& c, const sda_net1d_desc& d, int ptiles, int tp) {
    c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6; c.kq = c.lane >> 4; c.li = c.lane & 15;
    c.co0 = 16 * c.wave; c.n = blockIdx.x / ptiles; c.p0 = (blockIdx.x - c.n * ptiles) * tp;
    c.H = 2 * d.nblocks + 2; c.len = d.len; c.circular = d.circular != 0;
}

// ------------------------------------------------------------------------------------------------------------ forward
template <int NF>
__global__ __launch_bounds__(256) void net1d_fwd_kernel(const sda_net1d_desc d, int ptiles, int tp) {
    constexpr int NC = 16 * NF;
    __shared__ float tin[N1_MAXC * N1_LD];                 // input of the next convolution, tile column jj <-> conv column jj - 1
    __shared__ float tz[N1_MAXC * N1_LD];                  // act(z) between the two convolutions of a block
    __shared__ float red[2 * 4 * NC];
    N1Ctx c;
    n1_ctx(c, d, ptiles, tp);
    float wA[3][16], wB[3][16];
    n1_load_w(d.w_head, d.k_pad_head, d.m_pad, c, wA);
    // the two edge columns of both tiles are never written again: they stand for data beyond the tile (zeros: whatever they
    // were, the columns they reach are halo columns that have lost their meaning by the time they matter)
    if (c.tid < 2 * N1_MAXC) {
        const int row = c.tid >> 1, col = (c.tid & 1) ? NC + 1 : 0;
        tin[row * N1_LD + col] = 0.f;
        tz[row * N1_LD + col] = 0.f;
    }
    bool inside[NF], own[NF];
    int pos[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const int j = 16 * nf + c.li;
        pos[nf] = n1_pos(c, j, inside[nf]);
        own[nf] = inside[nf] && j >= c.H && j < c.H + tp && c.p0 - c.H + j < d.len;      // (the un-wrapped position is this tile's)
    }
    n1_load_tile<NF>(d.x, d.x_sn, d.x_sc, d.x_sx, d.cin, c, tin);
    if (d.nblocks > 0) n1_load_w(d.w1[0], d.k_pad, d.m_pad, c, wB);
    else n1_load_w(d.w_tail, d.k_pad, d.m_pad_tail, c, wB);
    __syncthreads();
    // ---- head convolution: a = conv(x) + b
    n1_f32x4 a[NF];
    n1_mm<NF>(wA, tin, c, a);
    {
        float bh[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = c.co0 + 4 * c.kq + r;
            bh[r] = (d.b_head && co < d.c) ? d.b_head[co] : 0.f;
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[nf][r] += bh[r];
    }
    const bool silu = d.act == SDA_ACT_SILU;
    const float inv_c = 1.f / (float)d.c, inv_v = 1.f / (float)(d.unbiased ? d.c - 1 : d.c);
    const int64_t plane = (int64_t)d.c * d.len;
    for (int k = 0; k < d.nblocks; ++k) {
        // ---- per-channel operands of the block (scalars per lane: 4 channels)
        float mo[4], b1[4], b2[4];
        const float* mp = d.mod[k] ? d.mod[k] + (int64_t)c.n * d.mod_sn : nullptr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = c.co0 + 4 * c.kq + r, coc = co < d.c ? co : d.c - 1;
            mo[r] = mp ? mp[coc] : 0.f;
            b1[r] = d.b1[k] ? d.b1[k][coc] : 0.f;
            b2[r] = d.b2[k] ? d.b2[k][coc] : 0.f;
        }
        // ---- the block input is what the VJP differentiates through: save the own columns
        if (d.a_save) {
            float* as = d.a_save + (int64_t)k * d.save_stride + (int64_t)c.n * plane;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = c.co0 + 4 * c.kq + r;
                if (co < d.c)
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf)
                        if (own[nf]) as[(int64_t)co * d.len + pos[nf]] = a[nf][r];
            }
        }
        // ---- LayerNorm over channels of u = a + mod (two passes over registers: mean, then centred sum of squares)
        n1_f32x4 u[NF];
        float s[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            s[nf] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool on = (c.co0 + 4 * c.kq + r) < d.c;
                u[nf][r] = on ? a[nf][r] + mo[r] : 0.f;
                s[nf] += u[nf][r];
            }
        }
        n1_colsum<NF>(s, red, c);
        float mean[NF], rstd[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            mean[nf] = s[nf] * inv_c;
            s[nf] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool on = (c.co0 + 4 * c.kq + r) < d.c;
                const float dl = u[nf][r] - mean[nf];
                s[nf] += on ? dl * dl : 0.f;
            }
        }
        n1_colsum<NF>(s, red + 4 * NC, c);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            rstd[nf] = 1.0f / sqrtf(s[nf] * inv_v + d.eps);
#pragma unroll
            for (int r = 0; r < 4; ++r) u[nf][r] = (u[nf][r] - mean[nf]) * rstd[nf];
        }
        if (d.mean_save && c.wave == 0 && c.kq == 0) {
            float* ms = d.mean_save + (int64_t)k * d.stat_stride + (int64_t)c.n * d.len;
            float* rs = d.rstd_save + (int64_t)k * d.stat_stride + (int64_t)c.n * d.len;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
                if (own[nf]) { ms[pos[nf]] = mean[nf]; rs[pos[nf]] = rstd[nf]; }
        }
        n1_store_tile<NF>(u, inside, d.c, c, tin);
        n1_load_w(d.w2[k], d.k_pad, d.m_pad, c, wA);                       // (set A is free: the previous conv2 / the head is done)
        __syncthreads();
        // ---- conv1: z = conv(LN) + b1 -> saved (own columns); act(z) -> LDS
        n1_f32x4 z[NF];
        n1_mm<NF>(wB, tin, c, z);
        float* zs = d.z_save ? d.z_save + (int64_t)k * d.save_stride + (int64_t)c.n * plane : nullptr;
        auto conv1_epilogue = [&](auto SILU_) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = c.co0 + 4 * c.kq + r;
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    const float zv = z[nf][r] + b1[r];
                    if (zs && co < d.c && own[nf]) zs[(int64_t)co * d.len + pos[nf]] = zv;
                    z[nf][r] = decltype(SILU_)::value ? sda_act(SDA_ACT_SILU, zv) : sda_act(d.act, zv);
                }
            }
        };
        if (silu) conv1_epilogue(std::true_type{});
        else conv1_epilogue(std::false_type{});
        n1_store_tile<NF>(z, inside, d.c, c, tz);
        if (k + 1 < d.nblocks) n1_load_w(d.w1[k + 1], d.k_pad, d.m_pad, c, wB);
        else n1_load_w(d.w_tail, d.k_pad, d.m_pad_tail, c, wB);
        __syncthreads();
        // ---- conv2 + b2 + residual
        n1_f32x4 y[NF];
        n1_mm<NF>(wA, tz, c, y);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[nf][r] += y[nf][r] + b2[r];
    }
    // ---- tail convolution -> out (own columns, through the output strides)
    n1_store_tile<NF>(a, inside, d.c, c, tin);
    __syncthreads();
    n1_f32x4 o[NF];
    n1_mm<NF>(wB, tin, c, o);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = c.co0 + 4 * c.kq + r;
        if (co >= d.cout) continue;
        const float bt = d.b_tail ? d.b_tail[co] : 0.f;
        float* op = d.out + (int64_t)c.n * d.out_sn + (int64_t)co * d.out_sc;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
            if (own[nf]) op[(int64_t)pos[nf] * d.out_sx] = o[nf][r] + bt;
    }
}

// ------------------------------------------------------------------------------------------------------------ input VJP
// w_* are the BACKWARD-DATA packings here (sda_pack_conv_weight with transpose = 1): "head" = tail^T (cout -> c, runs first),
// "tail" = head^T (c -> cin, runs last); w1 / w2 of block k are conv1^T / conv2^T.  x = incoming cotangent, out = input gradient.
template <int NF>
__global__ __launch_bounds__(256) void net1d_bwd_kernel(const sda_net1d_desc d, int ptiles, int tp) {
    constexpr int NC = 16 * NF;
    __shared__ float tg[N1_MAXC * N1_LD];
    __shared__ float tq[N1_MAXC * N1_LD];
    __shared__ float red[2 * 4 * NC];
    N1Ctx c;
    n1_ctx(c, d, ptiles, tp);
    float wA[3][16], wB[3][16];
    n1_load_w(d.w_head, d.k_pad_head, d.m_pad, c, wA);
    if (c.tid < 2 * N1_MAXC) {
        const int row = c.tid >> 1, col = (c.tid & 1) ? NC + 1 : 0;
        tg[row * N1_LD + col] = 0.f;
        tq[row * N1_LD + col] = 0.f;
    }
    bool inside[NF], own[NF];
    int pos[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const int j = 16 * nf + c.li;
        pos[nf] = n1_pos(c, j, inside[nf]);
        own[nf] = inside[nf] && j >= c.H && j < c.H + tp && c.p0 - c.H + j < d.len;
    }
    n1_load_tile<NF>(d.x, d.x_sn, d.x_sc, d.x_sx, d.cin, c, tg);
    const int kl = d.nblocks - 1;
    if (d.nblocks > 0) n1_load_w(d.w2[kl], d.k_pad, d.m_pad, c, wB);
    else n1_load_w(d.w_tail, d.k_pad, d.m_pad_tail, c, wB);
    const int64_t plane = (int64_t)d.c * d.len;
    // what a block's VJP reads from the forward, in D layout on every column (halo columns: written by the neighbours)
    n1_f32x4 ez[NF], ea[NF];
    float emean[NF], erstd[NF], emod[4];
    auto fetch_saved = [&](int k) {
        const float* zs = d.z_save + (int64_t)k * d.save_stride + (int64_t)c.n * plane;
        const float* as = d.a_save + (int64_t)k * d.save_stride + (int64_t)c.n * plane;
        const float* ms = d.mean_save + (int64_t)k * d.stat_stride + (int64_t)c.n * d.len;
        const float* rs = d.rstd_save + (int64_t)k * d.stat_stride + (int64_t)c.n * d.len;
        const float* mp = d.mod[k] ? d.mod[k] + (int64_t)c.n * d.mod_sn : nullptr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = c.co0 + 4 * c.kq + r, coc = co < d.c ? co : d.c - 1;
            emod[r] = mp ? mp[coc] : 0.f;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                ez[nf][r] = zs[(int64_t)coc * d.len + pos[nf]];
                ea[nf][r] = as[(int64_t)coc * d.len + pos[nf]];
            }
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) { emean[nf] = ms[pos[nf]]; erstd[nf] = rs[pos[nf]]; }
    };
    if (d.nblocks > 0) fetch_saved(kl);
    __syncthreads();
    // ---- tail^T: g = conv^T(cotangent)
    n1_f32x4 g[NF];
    n1_mm<NF>(wA, tg, c, g);
    const bool silu = d.act == SDA_ACT_SILU;
    const float inv_c = 1.f / (float)d.c, inv_v = 1.f / (float)(d.unbiased ? d.c - 1 : d.c);
    for (int k = kl; k >= 0; --k) {
        n1_store_tile<NF>(g, inside, d.c, c, tg);
        n1_load_w(d.w1[k], d.k_pad, d.m_pad, c, wA);
        __syncthreads();
        // ---- conv2^T, x act'(z) -> LDS
        n1_f32x4 q[NF];
        n1_mm<NF>(wB, tg, c, q);
        auto dact = [&](auto SILU_) {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    q[nf][r] *= decltype(SILU_)::value ? sda_dact(SDA_ACT_SILU, ez[nf][r]) : sda_dact(d.act, ez[nf][r]);
        };
        if (silu) dact(std::true_type{});
        else dact(std::false_type{});
        n1_store_tile<NF>(q, inside, d.c, c, tq);
        if (k > 0) n1_load_w(d.w2[k - 1], d.k_pad, d.m_pad, c, wB);
        else n1_load_w(d.w_tail, d.k_pad, d.m_pad_tail, c, wB);
        __syncthreads();
        // ---- conv1^T -> gh; LayerNorm backward: g <- rstd (gh - mean_c(gh) - xh mean'_c(gh xh)) + g
        n1_f32x4 gh[NF], xh[NF];
        n1_mm<NF>(wA, tq, c, gh);
        float s1[NF], s2[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            s1[nf] = 0.f; s2[nf] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool on = (c.co0 + 4 * c.kq + r) < d.c;
                xh[nf][r] = on ? (ea[nf][r] + emod[r] - emean[nf]) * erstd[nf] : 0.f;
                const float gv = on ? gh[nf][r] : 0.f;
                s1[nf] += gv; s2[nf] += gv * xh[nf][r];
            }
        }
        n1_colsum<NF>(s1, red, c);
        n1_colsum<NF>(s2, red + 4 * NC, c);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const float av = s1[nf] * inv_c, bv = s2[nf] * inv_v;
#pragma unroll
            for (int r = 0; r < 4; ++r) g[nf][r] += erstd[nf] * (gh[nf][r] - av - xh[nf][r] * bv);
        }
        if (k > 0) fetch_saved(k - 1);
        __syncthreads();                                   // (red is reused by the next block's sums)
    }
    // ---- head^T -> input gradient (own columns, through the output strides)
    n1_store_tile<NF>(g, inside, d.c, c, tg);
    __syncthreads();
    n1_f32x4 o[NF];
    n1_mm<NF>(wB, tg, c, o);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = c.co0 + 4 * c.kq + r;
        if (co >= d.cout) continue;
        float* op = d.out + (int64_t)c.n * d.out_sn + (int64_t)co * d.out_sc;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
            if (own[nf]) op[(int64_t)pos[nf] * d.out_sx] = o[nf][r];
    }
}

static int net1d_check(const sda_net1d_desc* d, bool bwd) {
    if (!d || d->n < 1 || d->len < 1 || d->c < 2 || d->c > N1_MAXC || d->cin < 1 || d->cin > N1_MAXC || d->cout < 1 ||
        d->cout > N1_MAXC || d->nblocks < 0 || d->nblocks > SDA_NET1D_MAXB)
        return SDA_E_UNSUPPORTED;
    if (d->k_pad > N1_MAXC || d->k_pad % 4 || d->k_pad < d->c || d->m_pad > N1_MAXC || d->m_pad % 16 || d->m_pad < d->c ||
        d->k_pad_head > N1_MAXC || d->k_pad_head % 4 || d->k_pad_head < d->cin || d->m_pad_tail > N1_MAXC || d->m_pad_tail % 16 ||
        d->m_pad_tail < d->cout)
        return SDA_E_UNSUPPORTED;
    if (!d->x || !d->out || !d->w_head || !d->w_tail) return SDA_E_BADARG;
    for (int k = 0; k < d->nblocks; ++k)
        if (!d->w1[k] || !d->w2[k]) return SDA_E_BADARG;
    const bool saves = d->a_save && d->z_save && d->mean_save && d->rstd_save;
    if (bwd && d->nblocks > 0 && !saves) return SDA_E_BADARG;
    if (!bwd && (d->a_save || d->z_save || d->mean_save || d->rstd_save) && !saves) return SDA_E_BADARG;
    return SDA_OK;
}

// columns per tile: 64 (36 own positions with the six blocks of the Lorenz nets) when that fills the chip, else 48 (20 own)
static int net1d_nf(const sda_net1d_desc* d) {
    static const int forced = getenv("SDA_NET1D_NF") ? atoi(getenv("SDA_NET1D_NF")) : 0;
    const int H = 2 * d->nblocks + 2;
    if ((forced == 3 || forced == 4) && 16 * forced - 2 * H >= 4) return forced;
    const int tp4 = 64 - 2 * H, tp3 = 48 - 2 * H;
    if (tp4 < 4) return 0;
    if (tp3 >= 8 && (int64_t)d->n * ((d->len + tp4 - 1) / tp4) < 128) return 3;
    return 4;
}

template <bool BWD>
static int net1d_launch(const sda_net1d_desc* d, hipStream_t stream) {
    const int rc = net1d_check(d, BWD);
    if (rc != SDA_OK) return rc;
    const int nf = net1d_nf(d);
    if (!nf) return SDA_E_UNSUPPORTED;
    const int tp = 16 * nf - 2 * (2 * d->nblocks + 2), ptiles = (d->len + tp - 1) / tp;
    if ((int64_t)d->n * ptiles > 0x7fffffffLL) return SDA_E_UNSUPPORTED;
    const dim3 grid((unsigned)(d->n * ptiles));
    if (nf == 3) {
        if (BWD) hipLaunchKernelGGL(net1d_bwd_kernel<3>, grid, dim3(256), 0, stream, *d, ptiles, tp);
        else hipLaunchKernelGGL(net1d_fwd_kernel<3>, grid, dim3(256), 0, stream, *d, ptiles, tp);
    } else {
        if (BWD) hipLaunchKernelGGL(net1d_bwd_kernel<4>, grid, dim3(256), 0, stream, *d, ptiles, tp);
        else hipLaunchKernelGGL(net1d_fwd_kernel<4>, grid, dim3(256), 0, stream, *d, ptiles, tp);
    }
    return sda_launch_status();
}

extern "C" int sda_net1d_fwd(const sda_net1d_desc* d, void* stream) { return net1d_launch<false>(d, (hipStream_t)stream); }
extern "C" int sda_net1d_bwd(const sda_net1d_desc* d, void* stream) { return net1d_launch<true>(d, (hipStream_t)stream); }
