// The bookkeeping around the fused 1-D score evaluations of net1d.hip (~15 small launches around the
// network were 25 % of a Lorenz sampling step):
//   step1d_prologue_kernel   -- per predictor-corrector step: this step's row of the host-evaluated schedule table
//                               (sda/score.py:250-253), mu / sigma of both evaluation times (score.py:195-210), the time
//                               embedding (score.py:15-35) and every block's modulation vector (nn.py:132-135) for both
//                               times -- one launch instead of 2 x (2 sda_vp_schedule + sda_time_embed + sda_linear_small)
//                               + 3 table-bookkeeping launches.
//   pc_correct_keyed_kernel  -- the Langevin correction (score.py:257-261) with its row-keyed noise generated in the kernel.
// Both replay the arithmetic of the kernels they replace operation for operation (elementwise.hip, noise.hip), so the fused
// step is bit-identical to the unfused one wherever the summation order is the same.
#include "sda_common.hpp"

#ifdef SDA_S1_TRACE                    // tooling build: cycle stamps of thread 0 at the phase boundaries -> out_coef[9 .. 15] (as floats)
#define S1_STAMP(k) do { if (threadIdx.x == 0) out_coef[9 + (k)] = (float)(__builtin_readcyclecounter() - s1_t0); } while (0)
#define S1_T0() const long long s1_t0 = __builtin_readcyclecounter()
#else
#define S1_STAMP(k) do {} while (0)
#define S1_T0() do {} while (0)
#endif
#define S1_THREADS 1024
#define S1_MAX_FEAT 128
#define S1_MAX_HIDDEN 1024
#define S1_MAX_E 256
#define S1_STAGE_MAX 8                 // 16-byte loads per thread of the staged weight copy
#define S1_PAD 4                       // floats of row padding in the staged weight copies (16-byte aligned rows, quarter-waves conflict free)

// STAGED: all three weight matrices are copied into LDS first (coalesced 16-byte loads issued before anything depends on t -- the
// copies overlap the schedule / feature arithmetic), and every dot product then reads LDS.  The first version read its rows from
// global memory inside the dependent loops: 32-42 us per launch (profiles/r04a_lorenz*_kernel_stats.csv), most of what the fused
// step had saved.  Same operations in the same order either way.
template <bool STAGED>
__global__ __launch_bounds__(S1_THREADS) void step1d_prologue_kernel(
    const float* __restrict__ table, int row_len, int64_t* istep, const float* __restrict__ t_dev, int nt, int alpha_kind, float eta,
    float k, int sigma_kind, const float* __restrict__ freqs, int nf, const float* __restrict__ w0, const float* __restrict__ b0,
    int hidden, const float* __restrict__ w2, const float* __restrict__ b2, int e, const float* __restrict__ wp,
    const float* __restrict__ bp, int cp, float* __restrict__ out_coef, int64_t* out_step, float* __restrict__ mod) {
    extern __shared__ __attribute__((aligned(16))) float s1_dyn[];
    __shared__ float tv[2];
    __shared__ __attribute__((aligned(16))) float feat[2][S1_MAX_FEAT];
    __shared__ __attribute__((aligned(16))) float hid[2][S1_MAX_HIDDEN];
    __shared__ __attribute__((aligned(16))) float emb[2][S1_MAX_E];
    const int tid = threadIdx.x;
    const int nin = 2 * nf;
    S1_T0();
    // row strides of the staged copies
    const int ld0 = nin + S1_PAD, ld2 = hidden + S1_PAD, ldp = e + S1_PAD;
    float* s0 = s1_dyn;
    float* s2 = s0 + hidden * ld0;
    float* sp = s2 + e * ld2;
    // (load order = issue order: the step counter first, the weight / bias copies behind it, then the table row that depends on the
    // counter -- two round trips in all; written the other way round they were four, 7 500 cycles)
    int64_t step = 0;
    if (table) step = *reinterpret_cast<const volatile int64_t*>(istep);
    if (STAGED) {
        // (nin, hidden, e are multiples of 4 and the matrices 16-byte aligned: checked by the launcher.)  Every 16-byte load of the three
        // matrices is issued before the first LDS store (a load -> store loop pays one round trip per iteration: 6 600 cycles of the
        // first staged version, tools/step1d_trace.py); at most S1_STAGE_MAX per thread, more than that takes the unstaged kernel.
        // row lengths are powers of two (launcher): (row, column) of a flat 16-byte index are a shift and a mask -- an integer
        // division per load cost the 16 waves 20 000 cycles in the version that had them
        const int l0 = 31 - __builtin_clz(nin >> 2), l2 = 31 - __builtin_clz(hidden >> 2), lp = 31 - __builtin_clz(e >> 2);
        const int n0 = hidden << l0, n2 = e << l2, np = cp << lp;
        float4 rv[S1_STAGE_MAX];
        int dsti[S1_STAGE_MAX];
#pragma unroll
        for (int i = 0; i < S1_STAGE_MAX; ++i) {
            int idx = tid + i * S1_THREADS;
            const float* src = nullptr;
            dsti[i] = -1;
            if (idx < n0) { src = w0 + 4 * (int64_t)idx; dsti[i] = (idx >> l0) * ld0 + 4 * (idx & ((1 << l0) - 1)); }
            else if ((idx -= n0) < n2) { src = w2 + 4 * (int64_t)idx; dsti[i] = (int)(s2 - s0) + (idx >> l2) * ld2 + 4 * (idx & ((1 << l2) - 1)); }
            else if ((idx -= n2) < np) { src = wp + 4 * (int64_t)idx; dsti[i] = (int)(sp - s0) + (idx >> lp) * ldp + 4 * (idx & ((1 << lp) - 1)); }
            // (unconditional, from a clamped address: a load under a condition compiles to a branch with its own s_waitcnt vmcnt(0) --
            // eight serial round trips, 20 000 cycles)
            rv[i] = *reinterpret_cast<const float4*>(src ? src : w0);
        }
#pragma unroll
        for (int i = 0; i < S1_STAGE_MAX; ++i)
            if (dsti[i] >= 0) *reinterpret_cast<float4*>(s0 + dsti[i]) = rv[i];
    }
    // the bias vectors and frequencies too: a global load inside one of the dependent phases below costs a round trip each time it
    // is reached (the projection loop paid one per output: 24 x ~0.7 us of the first staged version's 17-20 us)
    float* sb0 = STAGED ? sp + cp * ldp : s1_dyn;
    float* sb2 = sb0 + hidden;
    float* sbp = sb2 + e;
    float* sfr = sbp + cp;
    for (int i = tid; i < hidden; i += S1_THREADS) sb0[i] = b0[i];
    for (int i = tid; i < e; i += S1_THREADS) sb2[i] = b2[i];
    for (int i = tid; i < cp; i += S1_THREADS) sbp[i] = bp ? bp[i] : 0.f;
    for (int i = tid; i < nf; i += S1_THREADS) sfr[i] = freqs[i];
    if (tid < nt) {
        float t;
        if (table) t = table[step * row_len + tid];       // {t, t - dt}
        else t = t_dev[tid];
        tv[tid] = t;
        // mu, sigma: vp_schedule_kernel's arithmetic
        float a;
        if (alpha_kind == 0) a = 1.0f - (1.0f - eta) * t;
        else if (alpha_kind == 1) { const float c = cosf(k * t); a = c * c; }
        else a = expf(k * (t * t));
        float sg;
        if (sigma_kind == 0) sg = sqrtf((1.0f - a * a) + eta * eta);
        else if (sigma_kind == 1) sg = (1.0f - a * a) + eta;
        else sg = (1.0f - a) + eta;
        out_coef[2 * tid] = a;
        out_coef[2 * tid + 1] = sg;
        out_coef[7 + tid] = t;
    }
    if (tid == 0) {
        if (table) {
            out_coef[4] = table[step * row_len + 2];      // r
            out_coef[5] = table[step * row_len + 3];      // c1
            out_coef[6] = table[step * row_len + 4];      // sigma(t - dt)
        }
        if (out_step) out_step[0] = step;
    }
    S1_STAMP(0);                                           // staging issued, schedule scalars
    __syncthreads();
    S1_STAMP(1);                                           // ... landed
    // time_embed_kernel's arithmetic, both times side by side
    for (int i = tid; i < nt * nf; i += S1_THREADS) {
        const int it = i / nf, j = i - it * nf;
        const float ang = sfr[j] * tv[it];
        feat[it][j] = cosf(ang);
        feat[it][nf + j] = sinf(ang);
    }
    __syncthreads();
    S1_STAMP(2);                                           // features
    // sda_dot8 with its eight interleaved partial sums on eight adjacent lanes (sub = tid & 7 accumulates j = sub, sub + 8, ..
    // in index order), combined by three xor-shuffles in sda_dot8's tree order: identical values, an eighth of the chain
    const int sub = tid & 7, slot = tid >> 3;
    auto dot8_split = [&](const float* w, const float* v, int n) {
        float a = 0.f;
        int j = sub;
        for (; j + 24 < n; j += 32) {                      // four terms per trip: their eight LDS reads are in flight together
            const float w0_ = w[j], w1_ = w[j + 8], w2_ = w[j + 16], w3_ = w[j + 24];
            const float v0_ = v[j], v1_ = v[j + 8], v2_ = v[j + 16], v3_ = v[j + 24];
            a += w0_ * v0_; a += w1_ * v1_; a += w2_ * v2_; a += w3_ * v3_;
        }
        for (; j < n; j += 8) a += w[j] * v[j];
        a += __shfl_xor(a, 1, SDA_WAVE);
        a += __shfl_xor(a, 2, SDA_WAVE);
        a += __shfl_xor(a, 4, SDA_WAVE);
        return a;
    };
    for (int i = tid; i < nt * hidden; i += S1_THREADS) {
        const int it = i / hidden, hh = i - it * hidden;
        float acc;
        if (STAGED) {
            // sda_dot8 on 16-byte LDS reads (rows and feat are 16-byte aligned, nin % 8 == 0: launcher): same sums in the same order
            const float* wr = s0 + hh * ld0;
            float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < nin; j += 8) {
                const float4 wa = *reinterpret_cast<const float4*>(wr + j), wb = *reinterpret_cast<const float4*>(wr + j + 4);
                const float4 fa = *reinterpret_cast<const float4*>(feat[it] + j), fb = *reinterpret_cast<const float4*>(feat[it] + j + 4);
                a[0] += wa.x * fa.x; a[1] += wa.y * fa.y; a[2] += wa.z * fa.z; a[3] += wa.w * fa.w;
                a[4] += wb.x * fb.x; a[5] += wb.y * fb.y; a[6] += wb.z * fb.z; a[7] += wb.w * fb.w;
            }
            acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        } else {
            acc = sda_dot8(w0 + (int64_t)hh * nin, feat[it], nin);
        }
        hid[it][hh] = sda_act(SDA_ACT_SILU, acc + sb0[hh]);
    }
    __syncthreads();
    S1_STAMP(3);                                           // hidden layer
    for (int i0 = 0; i0 < nt * e; i0 += S1_THREADS / 8) {
        const int i = i0 + slot;
        const bool live = i < nt * e;
        const int it = live ? i / e : 0, o = live ? i - it * e : 0;
        const float* wr = STAGED ? s2 + o * ld2 : w2 + (int64_t)o * hidden;
        const float acc = dot8_split(wr, hid[it], hidden);
        if (live && sub == 0) emb[it][o] = acc + sb2[o];
    }
    __syncthreads();
    S1_STAMP(4);                                           // embedding (256-term sums)
    // linear_small_kernel's arithmetic: one wavefront per output, lanes stride the input features, shuffle-reduce.  With e <= 32
    // input features the upper half-wave only ever adds zeros there, so a wave serves TWO outputs, one per half (lane 0's / lane
    // 32's reduction tree touches its own half only from the 16-lane step on: identical sums).
    const int lane = tid & 63, wave = tid >> 6;
    if (cp == 0) {
        // no projection (a ScoreNet takes the embedding itself as features, sda/score.py:53-63): mod <- emb [nt][e]
        for (int i = tid; i < nt * e; i += S1_THREADS) mod[i] = emb[i / e][i - (i / e) * e];
    } else if (e <= 32) {
        // linear_small_kernel with <= 32 input features: lane l < e holds p_l = emb[l] w[l] (lanes beyond: 0), and the shuffle tree
        // (offsets 32, 16, 8, 4, 2, 1; the first step adds zeros) leaves lane 0 with ((..(p_l + p_{l+16}) + ..)): ONE thread per output
        // replays that tree on its 32 products -- the wave-per-output form spent 750 cycles per output on five dependent cross-lane
        // round trips (18 000 of the kernel's 34 600 cycles)
        for (int g = tid; g < nt * cp; g += S1_THREADS) {
            const int it = g / cp, o = g - it * cp;
            const float* wr = STAGED ? sp + o * ldp : wp + (int64_t)o * e;
            float pr[32];
            if (STAGED && e == 32) {                      // (16-byte aligned padded rows)
#pragma unroll
                for (int l4 = 0; l4 < 8; ++l4) {
                    const float4 wv = *reinterpret_cast<const float4*>(wr + 4 * l4), ev = *reinterpret_cast<const float4*>(emb[it] + 4 * l4);
                    pr[4 * l4] = ev.x * wv.x; pr[4 * l4 + 1] = ev.y * wv.y; pr[4 * l4 + 2] = ev.z * wv.z; pr[4 * l4 + 3] = ev.w * wv.w;
                }
            } else {
#pragma unroll
                for (int l = 0; l < 32; ++l) pr[l] = l < e ? emb[it][l] * wr[l] : 0.f;
            }
#pragma unroll
            for (int off = 16; off > 0; off >>= 1)
#pragma unroll
                for (int l = 0; l < off; ++l) pr[l] += pr[l + off];
            mod[g] = pr[0] + sbp[o];
        }
    } else {
        for (int g = wave; g < nt * cp; g += S1_THREADS / 64) {
            const int it = g / cp, o = g - it * cp;
            const float* wr = STAGED ? sp + o * ldp : wp + (int64_t)o * e;
            float acc = 0.f;
            for (int i = lane; i < e; i += 64) acc += emb[it][i] * wr[i];
            acc = sda_wave_sum(acc);
            if (lane == 0) mod[g] = acc + sbp[o];
        }
    }
    S1_STAMP(5);                                           // projection
    if (table && tid == 0) istep[0] = step + 1;            // (after the read above: this workgroup is the step counter's only reader)
}

extern "C" int sda_step1d_prologue(const float* table, int row_len, int64_t* istep, const float* t_dev, int nt, int alpha_kind,
                                   float eta, float k, int sigma_kind, const float* freqs, int nf, const float* w0, const float* b0,
                                   int hidden, const float* w2, const float* b2, int e, const float* wp, const float* bp, int cp,
                                   float* out_coef, int64_t* out_step, float* mod, void* stream) {
    if (!freqs || !w0 || !b0 || !w2 || !b2 || !out_coef || !mod || nf <= 0 || hidden <= 0 || e <= 0 || cp < 0 || (cp > 0 && !wp))
        return SDA_E_BADARG;
    if (table ? (!istep || row_len < 5 || nt != 2) : (!t_dev || nt < 1 || nt > 2)) return SDA_E_BADARG;
    if (alpha_kind < 0 || alpha_kind > 2 || sigma_kind < 0 || sigma_kind > 2) return SDA_E_BADARG;
    if (2 * nf > S1_MAX_FEAT || hidden > S1_MAX_HIDDEN || e > S1_MAX_E) return SDA_E_UNSUPPORTED;
    const int nin = 2 * nf;
    const int64_t vecs = 4LL * (hidden + e + cp + nf);                                       // staged biases + frequencies (both variants)
    const int64_t lds = vecs + 4LL * ((int64_t)hidden * (nin + S1_PAD) + (int64_t)e * (hidden + S1_PAD) + (int64_t)cp * (e + S1_PAD));
    const bool aligned = !(nin & 3) && !(hidden & 3) && !(e & 3) &&
                         !(((uintptr_t)w0 | (uintptr_t)w2 | (uintptr_t)(cp ? wp : w0)) & 15);
    const int64_t quads = ((int64_t)hidden * nin + (int64_t)e * hidden + (int64_t)cp * e) / 4;
    auto pow2 = [](int v) { return v > 0 && !(v & (v - 1)); };
    if (aligned && lds <= 140 * 1024 && quads <= (int64_t)S1_STAGE_MAX * S1_THREADS && pow2(nin) && pow2(hidden) && pow2(e) && nin >= 8) {
        static bool raised[SDA_MAX_DEVICES];
        const int rc = sda_raise_dyn_lds(reinterpret_cast<const void*>(step1d_prologue_kernel<true>), 140 * 1024, raised);
        if (rc != SDA_OK) return rc;
        hipLaunchKernelGGL(step1d_prologue_kernel<true>, dim3(1), dim3(S1_THREADS), (size_t)lds, (hipStream_t)stream, table, row_len, istep,
                           t_dev, nt, alpha_kind, eta, k, sigma_kind, freqs, nf, w0, b0, hidden, w2, b2, e, wp, bp, cp, out_coef, out_step,
                           mod);
    } else {
        if (vecs > 48 * 1024) return SDA_E_UNSUPPORTED;
        hipLaunchKernelGGL(step1d_prologue_kernel<false>, dim3(1), dim3(S1_THREADS), (size_t)vecs, (hipStream_t)stream, table, row_len, istep, t_dev,
                           nt, alpha_kind, eta, k, sigma_kind, freqs, nf, w0, b0, hidden, w2, b2, e, wp, bp, cp, out_coef, out_step, mod);
    }
    return sda_launch_status();
}

// ---- Philox4x32-10 + Box-Muller exactly as noise.hip (the same counters give the same z as sda_randn_rows)
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

__device__ __forceinline__ void s1_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c0, p1 = (uint64_t)PHILOX_M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__device__ __forceinline__ void s1_box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.28318530717958647692f * u2, &s, &c);
    z0 = r * c; z1 = r * s;
}

// grid = (blocks, b): block row b updates sample b; a thread owns quads (4 consecutive elements) of the sample
__global__ __launch_bounds__(256) void pc_correct_keyed_kernel(float* __restrict__ x, const float* __restrict__ eps, int64_t per_sample,
                                                               const float* __restrict__ partial, int nchunk, float tau, float sigma,
                                                               const float* __restrict__ coef, uint32_t k0, uint32_t k1, int64_t row0,
                                                               const int64_t* __restrict__ draw_dev, int64_t mul, int64_t add) {
    const int b = blockIdx.y;
    if (coef) sigma = coef[0];
    const int64_t draw = draw_dev[0] * mul + add;
    float tot = 0.f;
    for (int j = 0; j < nchunk; ++j) tot += partial[(int64_t)b * nchunk + j];
    const float delta = tau / (tot / (float)per_sample);
    const float sq = sqrtf(2.0f * delta);
    const int64_t base = (int64_t)b * per_sample;
    const int64_t quads = (per_sample + 3) >> 2;
    const uint64_t grow = (uint64_t)(row0 + b);
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (int64_t)gridDim.x * 256) {
        uint32_t w[4];
        s1_philox((uint32_t)q, (uint32_t)grow, (uint32_t)draw,
                  (uint32_t)((uint64_t)draw >> 32) ^ ((uint32_t)((uint64_t)q >> 32) << 16) ^ ((uint32_t)(grow >> 32) << 24), k0, k1, w);
        float z[4];
        s1_box_muller(w[0], w[1], z[0], z[1]);
        s1_box_muller(w[2], w[3], z[2], z[3]);
        const int64_t left = per_sample - 4 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < left) {
                const int64_t o = base + 4 * q + e;
                x[o] = x[o] - (delta * eps[o] + sq * z[e]) * sigma;
            }
    }
}

extern "C" int sda_pc_correct_keyed(float* x, const float* eps, int b, int64_t per_sample, const float* partial, int nchunk, float tau,
                                    float sigma, const float* coef_dev, uint64_t seed, int64_t row0, const int64_t* draw_dev,
                                    int64_t draw_mul, int64_t draw_add, void* stream) {
    if (!x || !eps || !partial || !draw_dev || b <= 0 || per_sample <= 0 || nchunk <= 0 || b > 65535 || row0 < 0) return SDA_E_BADARG;
    int64_t blocks = (((per_sample + 3) >> 2) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pc_correct_keyed_kernel, dim3((unsigned)blocks, b), dim3(256), 0, (hipStream_t)stream, x, eps, per_sample, partial,
                       nchunk, tau, sigma, coef_dev, (uint32_t)seed, (uint32_t)(seed >> 32), row0, draw_dev, draw_mul, draw_add);
    return sda_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// The tail of a fused guided evaluation of a LOCAL score network (sda_mlp_fwd_win / sda_mlp_bwd_win): the overlapping-window sum
// of the input gradient (`unfold`'s adjoint, sda/score.py:146-153 -- the one place overlaps add), the estimator's affine part, the
// guided score eps - (sigma/mu)(ghat - sigma J^T ghat) (score.py:394-396) and, by mode, what sda_net1d_bwd_fused's epilogue does:
// 0 write it; 1 x <- r x + c1 . in place; 2 write it and the trajectory's sum of squares into partial[b] (one chunk per sample).
// One workgroup per trajectory (L x C elements: 195 for the Lorenz job).
__global__ __launch_bounds__(256) void mc_finish_kernel(const float* __restrict__ eps, const float* __restrict__ ghat,
                                                        const float* __restrict__ gwin, int nw, int k, int c, float cx0, float cx1,
                                                        const float* __restrict__ coef, int mode, float* __restrict__ out, float* xs,
                                                        const float* __restrict__ step_coef, float* __restrict__ partial) {
    __shared__ float wsum[4];
    const int b = blockIdx.x, L = nw + 2 * k, per = L * c;
    const float mu = coef[0], sg = coef[1];
    const float cx = cx0 + cx1 * sg, kk = sg / mu;
    const bool bare = cx0 == 0.f && cx1 == 0.f;
    float pr = 0.f, pc1 = 0.f;
    if (mode == 1) { pr = step_coef[0]; pc1 = step_coef[1]; }
    float acc = 0.f;
    for (int e = threadIdx.x; e < per; e += 256) {
        const int l = e / c, ch = e - l * c;
        float gx = 0.f;
        for (int j = 0; j <= 2 * k; ++j) {                 // (ascending j: sda_unfold_adjoint's order)
            const int i = l - j;
            if (i >= 0 && i < nw) gx += gwin[((int64_t)b * nw + i) * 16 + j * c + ch];
        }
        const int64_t o = (int64_t)b * per + e;
        const float gv = ghat[o];
        const float vj = bare ? gx : (gv * cx) + gx;
        const float ov = eps[o] - kk * (gv - sg * vj);
        if (mode == 1) xs[o] = pr * xs[o] + pc1 * ov;
        else { out[o] = ov; acc += ov * ov; }
    }
    if (mode == 2) {
        acc = sda_wave_sum(acc);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) partial[b] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    }
}

extern "C" int sda_mc_finish(const float* eps, const float* ghat, const float* gwin, int b, int nw, int k, int c, float cx0, float cx1,
                             const float* coef, int mode, float* out, float* xs, const float* step_coef, float* partial, void* stream) {
    if (!eps || !ghat || !gwin || !coef || b <= 0 || nw <= 0 || k < 0 || c <= 0 || (2 * k + 1) * c > 16 || mode < 0 || mode > 2)
        return SDA_E_BADARG;
    if ((mode != 1 && !out) || (mode == 1 && (!xs || !step_coef)) || (mode == 2 && !partial)) return SDA_E_BADARG;
    hipLaunchKernelGGL(mc_finish_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, eps, ghat, gwin, nw, k, c, cx0, cx1, coef, mode, out,
                       xs, step_coef, partial);
    return sda_launch_status();
}
