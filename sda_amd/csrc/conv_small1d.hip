// Small 1-D convolutions (k = 3, stride 1, <= 64 channels in and out: every layer of the Lorenz U-Nets, sda/nn.py:113-176 with
// spatial = 1, and their backward-data forms).  These launches are a few hundred kFLOP per image: the staged producer /
// consumer kernel of conv_igemm.hip spends its 21 us per launch on exposed global-load round trips (one per K-stage) and
// pipeline fill, not on arithmetic.  Here a workgroup takes (image, 64 positions, all couts), issues EVERY load it needs at
// once -- the whole weight slab as MFMA A fragments straight into registers, the input tile (+ halo) with the loader fusions
// (modulation, LayerNorm, activation, circular / zero padding) into LDS -- so one round trip is paid, then runs the whole K
// extent (3 taps x cin) on v_mfma_f32_16x16x4_f32 from LDS, and stores with the epilogue fusions (bias, x act'(z), residual).
//   wave w: couts 16 w .. 16 w + 15, four 16-position fragments;  A[cout][k] = W[tap][ci][cout] (the packed [tap][cin_pad][cout_pad]
//   layout of sda_pack_conv_weight, read as is),  B[k][pos] = V[ci][pos + tap - 1] from LDS (row stride 80: conflict free).
// Bound: latency (one global round trip + 192 dependent-free MFMAs per wave); algorithmic flops 2 n wo cout cin 3.
#include "sda_common.hpp"
#include <type_traits>

#define S1_LD 80                       // LDS row stride (66 used; 80 mod 32 = 16: the two k rows of a 32-lane group hit disjoint banks)
#define S1_MAXC 64

typedef float s1_f32x4 __attribute__((ext_vector_type(4)));

template <int S1_TP>                   // positions per workgroup: 64, 32 or 16
__global__ __launch_bounds__(256) void conv_small1d_kernel(const sda_conv_desc d, int ptiles) {
    constexpr int NF = S1_TP / 16;
    __shared__ float sin[S1_MAXC * S1_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x / ptiles, p0 = (blockIdx.x - n * ptiles) * S1_TP;
    const int ncb = d.cin_pad >> 2;                        // K fragments (4 channels each) per tap
    const int kq = lane >> 4, li = lane & 15;
    const int co0 = 16 * wave;
    const bool wave_on = co0 < d.cout_pad;
    // ---- all weight fragments of this wave: 3 x ncb dwords per lane, one batch of loads
    float wreg[3][16];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int cb = 0; cb < 16; ++cb)
        {
            // (unconditional loads from clamped, always valid addresses: a load under a condition is a branch with its own
            // s_waitcnt, i.e. one serial round trip per load; what lies beyond cin_pad / cout_pad is never multiplied)
            const int cbc = cb < ncb ? cb : ncb - 1;
            wreg[tap][cb] = d.w[((int64_t)tap * d.cin_pad + 4 * cbc + kq) * d.cout_pad + (wave_on ? co0 : 0) + li];
        }
    // ---- input tile with halo: positions p0 - 1 .. p0 + 64 (LDS column j = position - p0 + 1), loader fusions applied once
    // per element.  A thread owns one column and every fourth channel: its 16 loads are independent (one round trip), the
    // LayerNorm statistics of its position are loaded once; threads 0-127 also take the two halo columns.
    {
        const int64_t m = (int64_t)n + d.x_n_off;
        const float* xi = d.x + (m / d.n_inner) * d.x_sn_outer + (m % d.n_inner) * d.x_sn_inner;
        const float* mp = d.mod ? d.mod + (int64_t)n * d.mod_sn : nullptr;
        // AM: 0 no activation, 1 SiLU (the reference nets), 2 any other (rolled loop: the five-way switch is not unrolled 16 x)
        auto column = [&](int j, int c_first, int c_step, auto NC_, auto AM_) {
            constexpr int NC = decltype(NC_)::value, AM = decltype(AM_)::value;
            int pos = p0 + j - 1;
            if (d.circular) pos = pos < 0 ? pos + d.ws : (pos >= d.ws ? pos - d.ws : pos);
            const bool inside = pos >= 0 && pos < d.ws;
            const int ps = inside ? pos : 0;
            // (every load unconditional, from a clamped address: see the weights)
            const int64_t st = (int64_t)n * d.ws + ps;
            const float* lm = d.ln_mean ? d.ln_mean : d.x;                 // (d.x: a valid dummy; the value is then unused)
            const float* lr = d.ln_rstd ? d.ln_rstd : d.x;
            const float mean_l = lm[d.ln_mean ? st : 0], rstd_l = lr[d.ln_mean ? st : 0];
            const float mean = d.ln_mean ? mean_l : 0.f, rstd = d.ln_mean ? rstd_l : 1.f;
            const float* mq = mp ? mp : d.x;
            float v[NC], mv[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int ci = c_first + c_step * i;
                const int cic = ci < d.cx ? ci : d.cx - 1;
                v[i] = xi[(int64_t)cic * d.x_sc + (int64_t)ps * d.x_sx];
                mv[i] = mq[mp ? cic : 0];
            }
            if constexpr (AM != 2) {
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const int ci = c_first + c_step * i;
                    float t = v[i];
                    if (mp) t += mv[i];
                    t = (t - mean) * rstd;
                    if constexpr (AM == 1) t = sda_act(SDA_ACT_SILU, t);
                    if (ci < S1_MAXC) sin[ci * S1_LD + j] = (inside && ci < d.cx) ? t : 0.f;
                }
            } else {
#pragma unroll 1
                for (int i = 0; i < NC; ++i) {
                    const int ci = c_first + c_step * i;
                    float t = v[i];
                    if (mp) t += mv[i];
                    t = sda_act(d.act_in, (t - mean) * rstd);
                    if (ci < S1_MAXC) sin[ci * S1_LD + j] = (inside && ci < d.cx) ? t : 0.f;
                }
            }
        };
        auto both = [&](auto AM_) {
            if (lane < S1_TP) column(1 + lane, wave, 4, std::integral_constant<int, 16>{}, AM_);
            if (tid < 128) column(tid < 64 ? 0 : S1_TP + 1, tid & 63, 1, std::integral_constant<int, 1>{}, AM_);
        };
        if (d.act_in == SDA_ACT_NONE) both(std::integral_constant<int, 0>{});
        else if (d.act_in == SDA_ACT_SILU) both(std::integral_constant<int, 1>{});
        else both(std::integral_constant<int, 2>{});
    }
    __syncthreads();
    if (!wave_on) return;
    // ---- the epilogue's operands (bias, act' input, residual) are requested now: their round trip runs under the multiply
    float ebias[4], ez[4][NF], eres[4][NF];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = co0 + 4 * kq + r;
        const int coc = co < d.cout ? co : d.cout - 1;     // (clamped: unconditional loads, see above)
        ebias[r] = d.bias ? d.bias[coc] : 0.f;
        const int64_t row = ((int64_t)n * d.cout + coc) * d.wo;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int pos = p0 + 16 * nf + li;
            const int pc = pos < d.wo ? pos : d.wo - 1;
            ez[r][nf] = d.dact_z ? d.dact_z[row + pc] : 0.f;
            eres[r][nf] = d.res ? d.res[row + pc] : 0.f;
        }
    }
    // ---- multiply: acc[nf] = sum_{tap, cb} A(tap, cb) B(cb, nf, tap); the B values of K fragment cb + 1 are read from LDS
    // before the 12 MFMAs of fragment cb are issued
    s1_f32x4 acc[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[nf] = s1_f32x4{0.f, 0.f, 0.f, 0.f};
    const float* brow = sin + kq * S1_LD + li;             // + 4 cb rows, + 16 nf + tap columns
    float bv[2][NF][3];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) bv[0][nf][tap] = brow[16 * nf + tap];
    // (all 16 K fragments unconditionally: a runtime trip count turns the unrolled loop into branches with accumulator copies
    //  around each; the LDS rows beyond cin are zero, so surplus fragments multiply clamped-address weights by zeros)
#pragma unroll
    for (int cb = 0; cb < 16; ++cb) {
        const int cn = cb + 1 < 16 ? cb + 1 : cb;          // (the last fragment re-reads itself)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) bv[(cb + 1) & 1][nf][tap] = brow[4 * cn * S1_LD + 16 * nf + tap];
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
                acc[nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap][cb], bv[cb & 1][nf][tap], acc[nf], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * NF, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * NF, 0);
    }
    // ---- epilogue: lane holds couts co0 + 4 kq + r, position p0 + 16 nf + li
    auto dact_any = [&](float z) __attribute__((noinline)) { return sda_dact(d.act_d, z); };
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = co0 + 4 * kq + r;
        if (co >= d.cout) continue;
        const int64_t row = ((int64_t)n * d.cout + co) * d.wo;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int pos = p0 + 16 * nf + li;
            if (pos < d.wo) {
                float v = acc[nf][r] + ebias[r];
                if (d.dact_z) v *= d.act_d == SDA_ACT_SILU ? sda_dact(SDA_ACT_SILU, ez[r][nf]) : dact_any(ez[r][nf]);
                if (d.res) v += eres[r][nf];
                d.out[row + pos] = v;
            }
        }
    }
}

// eligibility + tile length (16 / 32 / 64 positions); 0 -> the launch is served by the general kernels, < 0 -> bad descriptor
static int small1d_plan(const sda_conv_desc* d) {
    static const bool off = getenv("SDA_CONV_SMALL1D") && atoi(getenv("SDA_CONV_SMALL1D")) == 0;
    if (off || !d) return 0;
    if (d->kh != 1 || d->kw != 3 || d->stride_h != 1 || d->stride_w != 1 || d->hs != 1 || d->ho != 1 || d->up_h != 1 || d->up_w != 1 ||
        d->zins_h != 1 || d->zins_w != 1 || d->cctx != 0 || d->explicit_pad || d->out_sn || d->out_sc || d->out_sy || d->out_sx)
        return 0;
    if (d->cin_pad > S1_MAXC || d->cin_pad % 4 || d->cout_pad > 64 || d->cout_pad % 16 || d->cout > d->cout_pad || d->cx > d->cin_pad ||
        d->wo != d->ws || d->wo < 1 || d->n < 1 || d->n_inner < 1 || !d->x || !d->w || !d->out)
        return 0;
    if ((d->ln_mean == nullptr) != (d->ln_rstd == nullptr)) return SDA_E_BADARG;
    // 64-position tiles for batches that fill the chip, smaller ones otherwise (see block1d.hip)
    const int tp = (int64_t)d->n * ((d->wo + 15) / 16) <= 1024 ? 16 : ((int64_t)d->n * ((d->wo + 31) / 32) <= 1024 ? 32 : 64);
    // this kernel is for launches that cannot fill the chip with the staged kernel's tiles; big batches stay there
    if ((int64_t)d->n * ((d->wo + tp - 1) / tp) > 4096) return 0;
    return tp;
}

// would sda_conv_igemm serve this launch with conv_small1d_kernel?  (sda_conv_igemm_path)
int sda_small1d_path(const sda_conv_desc* d) { return small1d_plan(d) > 0; }

// SDA_E_UNSUPPORTED -> the launch is served by the general kernels
int sda_small1d_try(const sda_conv_desc* d, hipStream_t stream) {
    const int tp = small1d_plan(d);
    if (tp == 0) return SDA_E_UNSUPPORTED;
    if (tp < 0) return tp;
    const int ptiles = (d->wo + tp - 1) / tp;
    const int64_t grid = (int64_t)d->n * ptiles;
    if (tp == 16) hipLaunchKernelGGL(conv_small1d_kernel<16>, dim3((unsigned)grid), dim3(256), 0, stream, *d, ptiles);
    else if (tp == 32) hipLaunchKernelGGL(conv_small1d_kernel<32>, dim3((unsigned)grid), dim3(256), 0, stream, *d, ptiles);
    else hipLaunchKernelGGL(conv_small1d_kernel<64>, dim3((unsigned)grid), dim3(256), 0, stream, *d, ptiles);
    return sda_launch_status();
}
