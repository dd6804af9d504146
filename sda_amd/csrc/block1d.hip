// One modulated residual block of a 1-D U-Net (sda/nn.py:18-28 with spatial = 1: y = a + conv2(act(conv1(LN(a + mod))))) and its
// input VJP, each as ONE launch.  The 1-D nets of the Lorenz experiments are latency-bound: a block is three launches forward
// (LayerNorm statistics, two convolutions of ~12 us each, most of it launch + one global round trip) and three backward; here a
// workgroup takes (image, 64 positions), loads its input tile with a 2-position halo once, and keeps everything between the
// two convolutions in LDS / registers: LayerNorm statistics are reduced in the workgroup (all <= 64 channels of a position
// live in it), conv1 runs on the 64 positions + halo the second convolution needs (80 columns: five 16-column MFMA
// fragments), its activated output goes to LDS, conv2 reads it from there.  The VJP mirrors it: conv2^T on 80 columns, x act'(z)
// into LDS, conv1^T, then the LayerNorm backward with its two channel reductions across the workgroup.
//   wave w: channels 16 w .. 16 w + 15 of every convolution output (v_mfma_f32_16x16x4_f32, A = weights straight from the packed
//   [tap][k_pad][m_pad] layout into registers, B from LDS, row stride 112: conflict free).
// Column j of an LDS tile = position p0 - 2 + j (input tiles, 82 columns) or p0 - 1 + j (conv1 / conv2^T outputs, 80 columns).
#include "sda_common.hpp"
#include <type_traits>
#include <stdlib.h>

#define B1_LD 112                      // (112 mod 32 = 16: the two k rows of a 32-lane LDS access group hit disjoint banks)
#define B1_MAXC 64

typedef float b1_f32x4 __attribute__((ext_vector_type(4)));

#ifdef SDA_B1_TRACE                    // tooling: per-phase cycle sums of workgroup 0 / wave 0 (tools/block1d_trace.py)
__device__ long long b1_trace[16];
#define B1_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long n_ = __builtin_readcyclecounter(); b1_trace[k] += n_ - b1_tl; b1_tl = n_; } } while (0)
#define B1_T0() long long b1_tl = __builtin_readcyclecounter()
extern "C" int sda_b1_trace_read(long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(b1_trace), sizeof(long long) * 16) != hipSuccess) return SDA_E_BADARG;
    if (reset) { long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(b1_trace), z, sizeof(z)); }
    return SDA_OK;
}
#else
#define B1_STAMP(k) do {} while (0)
#define B1_T0() do {} while (0)
#endif

struct B1Ctx {
    int tid, lane, wave, kq, li, co0, n, p0, ncb;
    bool wave_on;
};

// all A fragments of one convolution for this wave (unconditional loads from clamped addresses: one batch, one round trip)
__device__ __forceinline__ void b1_load_w(const sda_block1d_desc& d, const float* w, const B1Ctx& c, float (&wreg)[3][16]) {
    // per-lane base + wave-uniform fragment offset (scalar arithmetic: the loads take an SGPR offset), 32-bit throughout
    const float* wl = w + c.kq * d.m_pad + (c.wave_on ? c.co0 : 0) + c.li;
    const int frag = 4 * d.m_pad, tapstride = d.k_pad * d.m_pad, last = (c.ncb - 1) * frag;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int cb = 0; cb < 16; ++cb) {
            const int off = cb * frag < last ? cb * frag : last;           // (clamped: fragments beyond ncb are never multiplied)
            wreg[tap][cb] = wl[tap * tapstride + off];     // (uniform: kernel arguments and constants only)
        }
}

// acc[nf] += sum_{tap, cb} A(tap, cb) B[4 cb + k][16 nf + li + tap]   (tile: LDS rows of B1_LD floats)
template <int NF>
__device__ __forceinline__ void b1_mm(const float (&wreg)[3][16], int ncb, const float* tile, const B1Ctx& c, b1_f32x4 (&acc)[NF]) {
    const float* brow = tile + c.kq * B1_LD + c.li;
    float bv[2][NF][3];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) bv[0][nf][tap] = brow[16 * nf + tap];
    // (all 16 K fragments, unconditionally: a runtime trip count splits the unrolled loop into branches with accumulator
    //  copies around each -- measured 11 000 cycles for 96 MFMAs; rows of the tile beyond the channel count are zero, so the
    //  surplus fragments of a narrower net multiply clamped-address weights by zeros)
    (void)ncb;
#pragma unroll
    for (int cb = 0; cb < 16; ++cb) {
        constexpr int dummy = 0; (void)dummy;
        const int cn = cb + 1 < 16 ? cb + 1 : cb;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) bv[(cb + 1) & 1][nf][tap] = brow[4 * cn * B1_LD + 16 * nf + tap];
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
                acc[nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap][cb], bv[cb & 1][nf][tap], acc[nf], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * NF, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * NF, 0);
    }
}

// position of tile column j (input tiles): wrapped for circular padding; `inside` = carries data
__device__ __forceinline__ int b1_pos(const sda_block1d_desc& d, int p, bool& inside) {
    if (d.circular) {
        p = p < 0 ? p + d.len : (p >= d.len ? p - d.len : p);
        p = p < 0 ? p + d.len : (p >= d.len ? p - d.len : p);     // (tiles wider than a short sequence wrap twice)
    }
    inside = p >= 0 && p < d.len;
    return inside ? p : 0;
}

// ------------------------------------------------------------------------------------------------------------ forward
template <int B1_TP>
__global__ __launch_bounds__(256) void block1d_fwd_kernel(const sda_block1d_desc d, int ptiles) {
    constexpr int NF2 = B1_TP / 16, NF1 = NF2 + 1, B1_COLS = 16 * NF1 + 2;      // conv2 / conv1 fragments, input columns
    constexpr int NPASS = B1_COLS > 64 ? 2 : 1;
    __shared__ float tin[B1_MAXC * B1_LD];                 // LN(a + mod), columns p0 - 2 ..
    __shared__ float tz[B1_MAXC * B1_LD];                  // act(z), columns p0 - 1 ..
    __shared__ float part[4 * 96];                         // per-column partial sums of the four channel groups
    B1Ctx c;
    c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6; c.kq = c.lane >> 4; c.li = c.lane & 15;
    c.co0 = 16 * c.wave; c.n = blockIdx.x / ptiles; c.p0 = (blockIdx.x - c.n * ptiles) * B1_TP; c.ncb = d.k_pad >> 2;
    c.wave_on = c.co0 < d.m_pad;
    B1_T0();
    float w1[3][16], w2[3][16];
    b1_load_w(d, d.w1, c, w1);
    b1_load_w(d, d.w2, c, w2);
    const float* an = d.a + (int64_t)c.n * d.c * d.len;
    const float* mp = d.mod ? d.mod + (int64_t)c.n * d.mod_sn : nullptr;
    // ---- the residual / bias operands of the epilogue, requested early (MFMA D layout: channel co0 + 4 kq + r, position p0 + 16 nf + li)
    float eb1[4], eb2[4], ea[4][NF2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = c.co0 + 4 * c.kq + r, coc = co < d.c ? co : d.c - 1;
        eb1[r] = d.b1 ? d.b1[coc] : 0.f;
        eb2[r] = d.b2 ? d.b2[coc] : 0.f;
#pragma unroll
        for (int nf = 0; nf < NF2; ++nf) {
            const int pos = c.p0 + 16 * nf + c.li, pc = pos < d.len ? pos : d.len - 1;
            ea[r][nf] = an[(int64_t)coc * d.len + pc];
        }
    }
    // ---- input columns: thread (column j, channel group sub): channels sub + 4 i.  Columns 0..63: j = lane, sub = wave;
    //      columns 64..81: threads 0..71, j = 64 + (tid >> 2), sub = tid & 3.
    float v[2][16];
    int colj[2], cols_sub[2];
    bool cin_[2];
    colj[0] = c.lane; cols_sub[0] = c.wave;
    colj[1] = 64 + (c.tid >> 2); cols_sub[1] = c.tid & 3;
    const bool second = B1_COLS > 64 && c.tid < 4 * (B1_COLS - 64);
    const float* mq = mp ? mp : d.a;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        bool inside;
        const int ps = b1_pos(d, c.p0 - 2 + colj[pass], inside);
        cin_[pass] = inside;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = cols_sub[pass] + 4 * i, cic = ci < d.c ? ci : d.c - 1;
            float t;
            if (pass == 0) {
                // (first pass: the channel is wave uniform -- per-lane base + scalar offsets, a scalar load for the modulation)
                const int cu = __builtin_amdgcn_readfirstlane(cic);
                t = (an + ps)[cu * d.len] + (mp ? mq[cu] : 0.f);
            } else {
                t = an[(int64_t)cic * d.len + ps] + (mp ? mq[cic] : 0.f);
            }
            v[pass][i] = (ci < d.c) ? t : 0.f;
        }
    }
    // ---- LayerNorm statistics per column (two passes over registers: mean, then centred sum of squares)
    auto reduce_cols = [&](auto F) {
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            if (pass == 0 || second) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) s += F(pass, i);
                part[cols_sub[pass] * 96 + colj[pass]] = s;
            }
        }
        __syncthreads();
    };
    B1_STAMP(0);                                           // address arithmetic + issue of every load
    reduce_cols([&](int pass, int i) { return v[pass][i]; });
    B1_STAMP(1);                                           // first use of the input loads: the global round trip + 1st reduction
    float mean[2], rstd[2];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int j = pass == 0 || second ? colj[pass] : 0;
        mean[pass] = (part[j] + part[96 + j] + part[192 + j] + part[288 + j]) / (float)d.c;
    }
    __syncthreads();
    reduce_cols([&](int pass, int i) {
        const float dl = v[pass][i] - mean[pass];
        return (cols_sub[pass] + 4 * i) < d.c ? dl * dl : 0.f;
    });
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int j = pass == 0 || second ? colj[pass] : 0;
        const float var = (part[j] + part[96 + j] + part[192 + j] + part[288 + j]) / (float)(d.unbiased ? d.c - 1 : d.c);
        rstd[pass] = 1.0f / sqrtf(var + d.eps);
    }
    // ---- normalised tile -> LDS; statistics of the tile's own 64 positions -> global (the VJP needs them)
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (pass == 0 || second) {
            const int j = colj[pass];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ci = cols_sub[pass] + 4 * i;
                tin[ci * B1_LD + j] = (cin_[pass] && ci < d.c) ? (v[pass][i] - mean[pass]) * rstd[pass] : 0.f;
            }
            const int pos = c.p0 - 2 + j;
            if (cols_sub[pass] == 0 && j >= 2 && j < 2 + B1_TP && pos < d.len && d.mean) {
                d.mean[(int64_t)c.n * d.len + pos] = mean[pass];
                d.rstd[(int64_t)c.n * d.len + pos] = rstd[pass];
            }
        }
    }
    __syncthreads();
    B1_STAMP(2);                                           // second reduction, normalised tile -> LDS
    // ---- conv1 on 80 columns (positions p0 - 1 ..): z = conv + b1 -> global (own 64 positions), act(z) -> LDS
    const bool silu = d.act == SDA_ACT_SILU;               // (the reference nets; other activations through one out-of-line switch)
    {
        b1_f32x4 acc[NF1];
#pragma unroll
        for (int nf = 0; nf < NF1; ++nf) acc[nf] = b1_f32x4{0.f, 0.f, 0.f, 0.f};
        if (c.wave_on) b1_mm<NF1>(w1, c.ncb, tin, c, acc);             // (waves beyond m_pad only zero their rows of the next tile)
        // per-column facts once (not per element): carries data / is one of the tile's own positions
        bool cin1[NF1], cown[NF1];
#pragma unroll
        for (int nf = 0; nf < NF1; ++nf) {
            const int col = 16 * nf + c.li;                           // position p0 - 1 + col
            bool inside;
            (void)b1_pos(d, c.p0 - 1 + col, inside);
            cin1[nf] = inside;
            cown[nf] = d.z != nullptr && col >= 1 && col <= B1_TP && c.p0 - 1 + col < d.len;
        }
        float* const zt = d.z ? d.z + (int64_t)c.n * d.c * d.len + (c.p0 - 1) : nullptr;
        auto epilogue = [&](auto SILU_) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = c.co0 + 4 * c.kq + r;
                const bool on = co < d.c;
                float* const zr = zt + (on ? co : 0) * d.len;
#pragma unroll
                for (int nf = 0; nf < NF1; ++nf) {
                    const int col = 16 * nf + c.li;
                    const float z = acc[nf][r] + eb1[r];
                    if (on && cown[nf]) zr[col] = z;
                    const float az = decltype(SILU_)::value ? sda_act(SDA_ACT_SILU, z) : sda_act(d.act, z);
                    tz[co * B1_LD + col] = (cin1[nf] && on) ? az : 0.f;
                }
            }
        };
        if (silu) epilogue(std::true_type{});
        else epilogue(std::false_type{});
    }
    __syncthreads();
    B1_STAMP(3);                                           // conv1 (waits for the weight loads), z store, act -> LDS
    // ---- conv2 on the 64 positions, + b2 + a
    if (c.wave_on) {
        b1_f32x4 acc[NF2];
#pragma unroll
        for (int nf = 0; nf < NF2; ++nf) acc[nf] = b1_f32x4{0.f, 0.f, 0.f, 0.f};
        b1_mm<NF2>(w2, c.ncb, tz, c, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = c.co0 + 4 * c.kq + r;
            if (co >= d.c) continue;
#pragma unroll
            for (int nf = 0; nf < NF2; ++nf) {
                const int pos = c.p0 + 16 * nf + c.li;
                if (pos < d.len) d.y[((int64_t)c.n * d.c + co) * d.len + pos] = acc[nf][r] + eb2[r] + ea[r][nf];
            }
        }
    }
    B1_STAMP(4);                                           // conv2 + store issue
}

// ------------------------------------------------------------------------------------------------------------ input VJP
// w1 / w2 are the BACKWARD-DATA packings here (sda_pack_conv_weight with transpose = 1).
template <int B1_TP>
__global__ __launch_bounds__(256) void block1d_bwd_kernel(const sda_block1d_desc d, int ptiles) {
    constexpr int NF2 = B1_TP / 16, NF1 = NF2 + 1, B1_COLS = 16 * NF1 + 2;
    constexpr int NPASS = B1_COLS > 64 ? 2 : 1;
    __shared__ float tg[B1_MAXC * B1_LD];                  // g, columns p0 - 2 ..
    __shared__ float tq[B1_MAXC * B1_LD];                  // conv2^T(g) * act'(z), columns p0 - 1 ..
    __shared__ float red[4 * B1_TP * 2];                   // per-wave partial channel sums of the LayerNorm backward
    B1Ctx c;
    c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6; c.kq = c.lane >> 4; c.li = c.lane & 15;
    c.co0 = 16 * c.wave; c.n = blockIdx.x / ptiles; c.p0 = (blockIdx.x - c.n * ptiles) * B1_TP; c.ncb = d.k_pad >> 2;
    c.wave_on = c.co0 < d.m_pad;
    float w2[3][16], w1[3][16];
    b1_load_w(d, d.w2, c, w2);
    b1_load_w(d, d.w1, c, w1);
    const int64_t img = (int64_t)c.n * d.c * d.len;
    const float* gn = d.g + img;
    const float* mp = d.mod ? d.mod + (int64_t)c.n * d.mod_sn : nullptr;
    // ---- operands in MFMA D layout, requested early: z on the 80 conv2^T columns; a, g, statistics on the 64 positions
    float ez[4][NF1];
    float ea[4][NF2], eg[4][NF2], emean[NF2], erstd[NF2], emod[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = c.co0 + 4 * c.kq + r, coc = co < d.c ? co : d.c - 1;
        emod[r] = mp ? mp[coc] : 0.f;
#pragma unroll
        for (int nf = 0; nf < NF1; ++nf) {
            bool inside;
            const int ps = b1_pos(d, c.p0 - 1 + 16 * nf + c.li, inside);
            ez[r][nf] = d.z[img + (int64_t)coc * d.len + ps];
        }
#pragma unroll
        for (int nf = 0; nf < NF2; ++nf) {
            const int pos = c.p0 + 16 * nf + c.li, pc = pos < d.len ? pos : d.len - 1;
            ea[r][nf] = d.a[img + (int64_t)coc * d.len + pc];
            eg[r][nf] = gn[(int64_t)coc * d.len + pc];
        }
    }
#pragma unroll
    for (int nf = 0; nf < NF2; ++nf) {
        const int pos = c.p0 + 16 * nf + c.li, pc = pos < d.len ? pos : d.len - 1;
        emean[nf] = d.mean[(int64_t)c.n * d.len + pc];
        erstd[nf] = d.rstd[(int64_t)c.n * d.len + pc];
    }
    // ---- g tile with halo -> LDS
    {
        int colj[2], sub[2];
        colj[0] = c.lane; sub[0] = c.wave;
        colj[1] = 64 + (c.tid >> 2); sub[1] = c.tid & 3;
        const bool second = B1_COLS > 64 && c.tid < 4 * (B1_COLS - 64);
        float v[2][16];
        bool ins[2];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            bool inside;
            const int ps = b1_pos(d, c.p0 - 2 + colj[pass], inside);
            ins[pass] = inside;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ci = sub[pass] + 4 * i, cic = ci < d.c ? ci : d.c - 1;
                if (pass == 0) v[pass][i] = (gn + ps)[__builtin_amdgcn_readfirstlane(cic) * d.len];
                else v[pass][i] = gn[(int64_t)cic * d.len + ps];
            }
        }
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass)
            if (pass == 0 || second)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ci = sub[pass] + 4 * i;
                    tg[ci * B1_LD + colj[pass]] = (ins[pass] && ci < d.c) ? v[pass][i] : 0.f;
                }
    }
    __syncthreads();
    // ---- conv2^T on 80 columns, x act'(z) -> LDS
    const bool silu = d.act == SDA_ACT_SILU;
    {
        b1_f32x4 acc[NF1];
#pragma unroll
        for (int nf = 0; nf < NF1; ++nf) acc[nf] = b1_f32x4{0.f, 0.f, 0.f, 0.f};
        if (c.wave_on) b1_mm<NF1>(w2, c.ncb, tg, c, acc);
        bool cin1[NF1];
#pragma unroll
        for (int nf = 0; nf < NF1; ++nf) {
            bool inside;
            (void)b1_pos(d, c.p0 - 1 + 16 * nf + c.li, inside);
            cin1[nf] = inside;
        }
        auto epilogue = [&](auto SILU_) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = c.co0 + 4 * c.kq + r;
                const bool on = co < d.c;
#pragma unroll
                for (int nf = 0; nf < NF1; ++nf) {
                    const float dz = decltype(SILU_)::value ? sda_dact(SDA_ACT_SILU, ez[r][nf]) : sda_dact(d.act, ez[r][nf]);
                    tq[co * B1_LD + 16 * nf + c.li] = (cin1[nf] && on) ? acc[nf][r] * dz : 0.f;
                }
            }
        };
        if (silu) epilogue(std::true_type{});
        else epilogue(std::false_type{});
    }
    __syncthreads();
    // ---- conv1^T on the 64 positions -> gh; LayerNorm backward: gx = rstd (gh - mean_c(gh) - xh mean'_c(gh xh)) + g
    b1_f32x4 gh[NF2];
#pragma unroll
    for (int nf = 0; nf < NF2; ++nf) gh[nf] = b1_f32x4{0.f, 0.f, 0.f, 0.f};
    if (c.wave_on) b1_mm<NF2>(w1, c.ncb, tq, c, gh);
    float xh[4][NF2];
#pragma unroll
    for (int nf = 0; nf < NF2; ++nf) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool on = c.wave_on && (c.co0 + 4 * c.kq + r) < d.c;
            xh[r][nf] = on ? (ea[r][nf] + emod[r] - emean[nf]) * erstd[nf] : 0.f;
            const float gv = on ? gh[nf][r] : 0.f;
            s1 += gv; s2 += gv * xh[r][nf];
        }
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (c.kq == 0) { red[(c.wave * B1_TP + 16 * nf + c.li) * 2] = s1; red[(c.wave * B1_TP + 16 * nf + c.li) * 2 + 1] = s2; }
    }
    __syncthreads();
    if (!c.wave_on) return;
    const float ia = 1.f / (float)d.c, ib = 1.f / (float)(d.unbiased ? d.c - 1 : d.c);
#pragma unroll
    for (int nf = 0; nf < NF2; ++nf) {
        const int m = 16 * nf + c.li;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s1 += red[(w * B1_TP + m) * 2]; s2 += red[(w * B1_TP + m) * 2 + 1]; }
        const float av = s1 * ia, bv = s2 * ib;
        const int pos = c.p0 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = c.co0 + 4 * c.kq + r;
            if (co < d.c && pos < d.len)
                d.gx[img + (int64_t)co * d.len + pos] = erstd[nf] * (gh[nf][r] - av - xh[r][nf] * bv) + eg[r][nf];
        }
    }
}

static int block1d_check(const sda_block1d_desc* d, bool bwd) {
    if (!d || d->n < 1 || d->c < 2 || d->c > B1_MAXC || d->len < 1 || d->k_pad > B1_MAXC || d->k_pad % 4 || d->k_pad < d->c ||
        d->m_pad > B1_MAXC || d->m_pad % 16 || d->m_pad < d->c || !d->a || !d->w1 || !d->w2)
        return SDA_E_UNSUPPORTED;
    if (bwd ? (!d->g || !d->gx || !d->z || !d->mean || !d->rstd) : (!d->y || ((d->mean == nullptr) != (d->rstd == nullptr))))
        return SDA_E_BADARG;
    if ((int64_t)d->n * ((d->len + 15) / 16) > 0x7fffffffLL) return SDA_E_UNSUPPORTED;
    return SDA_OK;
}

// 64-position tiles when they fill the chip; 32- or 16-position tiles (more workgroups, less MFMA work each, a larger share of
// halo columns) for the small batches these kernels exist for
static int block1d_tile(const sda_block1d_desc* d) {
    static const int forced = getenv("SDA_BLOCK1D_TP") ? atoi(getenv("SDA_BLOCK1D_TP")) : 0;
    if (forced == 16 || forced == 32 || forced == 64) return forced;
    // (measured, Lorenz-96 batch 64 x 128 positions: 0.85 / 0.68 / 0.67 ms per sampling step with 64 / 32 / 16; Lorenz-63, one
    //  image of 64 positions: 0.70 / 0.53 / 0.44)
    if ((int64_t)d->n * ((d->len + 15) / 16) <= 1024) return 16;
    if ((int64_t)d->n * ((d->len + 31) / 32) <= 1024) return 32;
    return 64;
}

extern "C" int sda_block1d_fwd(const sda_block1d_desc* d, void* stream) {
    const int rc = block1d_check(d, false);
    if (rc != SDA_OK) return rc;
    const int tp = block1d_tile(d), ptiles = (d->len + tp - 1) / tp;
    if (tp == 64) hipLaunchKernelGGL(block1d_fwd_kernel<64>, dim3((unsigned)(d->n * ptiles)), dim3(256), 0, (hipStream_t)stream, *d, ptiles);
    else if (tp == 16) hipLaunchKernelGGL(block1d_fwd_kernel<16>, dim3((unsigned)(d->n * ptiles)), dim3(256), 0, (hipStream_t)stream, *d, ptiles);
    else hipLaunchKernelGGL(block1d_fwd_kernel<32>, dim3((unsigned)(d->n * ptiles)), dim3(256), 0, (hipStream_t)stream, *d, ptiles);
    return sda_launch_status();
}

extern "C" int sda_block1d_bwd(const sda_block1d_desc* d, void* stream) {
    const int rc = block1d_check(d, true);
    if (rc != SDA_OK) return rc;
    const int tp = block1d_tile(d), ptiles = (d->len + tp - 1) / tp;
    if (tp == 64) hipLaunchKernelGGL(block1d_bwd_kernel<64>, dim3((unsigned)(d->n * ptiles)), dim3(256), 0, (hipStream_t)stream, *d, ptiles);
    else if (tp == 16) hipLaunchKernelGGL(block1d_bwd_kernel<16>, dim3((unsigned)(d->n * ptiles)), dim3(256), 0, (hipStream_t)stream, *d, ptiles);
    else hipLaunchKernelGGL(block1d_bwd_kernel<32>, dim3((unsigned)(d->n * ptiles)), dim3(256), 0, (hipStream_t)stream, *d, ptiles);
    return sda_launch_status();
}
