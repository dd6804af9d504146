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

#define S1_THREADS 1024
#define S1_MAX_FEAT 128
#define S1_MAX_HIDDEN 1024
#define S1_MAX_E 256

__global__ __launch_bounds__(S1_THREADS) void step1d_prologue_kernel(
    const float* __restrict__ table, int row_len, int64_t* istep, const float* __restrict__ t_dev, int nt, int alpha_kind, float eta,
    float k, int sigma_kind, const float* __restrict__ freqs, int nf, const float* __restrict__ w0, const float* __restrict__ b0,
    int hidden, const float* __restrict__ w2, const float* __restrict__ b2, int e, const float* __restrict__ wp,
    const float* __restrict__ bp, int cp, float* __restrict__ out_coef, int64_t* out_step, float* __restrict__ mod) {
    __shared__ float tv[2];
    __shared__ float feat[2][S1_MAX_FEAT];
    __shared__ float hid[2][S1_MAX_HIDDEN];
    __shared__ float emb[2][S1_MAX_E];
    const int tid = threadIdx.x;
    int64_t step = 0;
    if (table) step = istep[0];
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
    __syncthreads();
    // time_embed_kernel's arithmetic, both times side by side
    for (int i = tid; i < nt * nf; i += S1_THREADS) {
        const int it = i / nf, j = i - it * nf;
        const float ang = freqs[j] * tv[it];
        feat[it][j] = cosf(ang);
        feat[it][nf + j] = sinf(ang);
    }
    __syncthreads();
    const int nin = 2 * nf;
    for (int i = tid; i < nt * hidden; i += S1_THREADS) {
        const int it = i / hidden, hh = i - it * hidden;
        float acc = 0.f;
        const float* wr = w0 + (int64_t)hh * nin;
        for (int j = 0; j < nin; ++j) acc += wr[j] * feat[it][j];
        hid[it][hh] = sda_act(SDA_ACT_SILU, acc + b0[hh]);
    }
    __syncthreads();
    for (int i = tid; i < nt * e; i += S1_THREADS) {
        const int it = i / e, o = i - it * e;
        float acc = 0.f;
        const float* wr = w2 + (int64_t)o * hidden;
        for (int j = 0; j < hidden; ++j) acc += wr[j] * hid[it][j];
        emb[it][o] = acc + b2[o];
    }
    __syncthreads();
    // linear_small_kernel's arithmetic: one wavefront per output, lanes stride the input features, shuffle-reduce
    const int lane = tid & 63, wave = tid >> 6;
    for (int g = wave; g < nt * cp; g += S1_THREADS / 64) {
        const int it = g / cp, o = g - it * cp;
        const float* wr = wp + (int64_t)o * e;
        float acc = 0.f;
        for (int i = lane; i < e; i += 64) acc += emb[it][i] * wr[i];
        acc = sda_wave_sum(acc);
        if (lane == 0) mod[g] = acc + (bp ? bp[o] : 0.f);
    }
    if (table && tid == 0) istep[0] = step + 1;            // (after the read above: this workgroup is the step counter's only reader)
}

extern "C" int sda_step1d_prologue(const float* table, int row_len, int64_t* istep, const float* t_dev, int nt, int alpha_kind,
                                   float eta, float k, int sigma_kind, const float* freqs, int nf, const float* w0, const float* b0,
                                   int hidden, const float* w2, const float* b2, int e, const float* wp, const float* bp, int cp,
                                   float* out_coef, int64_t* out_step, float* mod, void* stream) {
    if (!freqs || !w0 || !b0 || !w2 || !b2 || !wp || !out_coef || !mod || nf <= 0 || hidden <= 0 || e <= 0 || cp <= 0)
        return SDA_E_BADARG;
    if (table ? (!istep || row_len < 5 || nt != 2) : (!t_dev || nt < 1 || nt > 2)) return SDA_E_BADARG;
    if (alpha_kind < 0 || alpha_kind > 2 || sigma_kind < 0 || sigma_kind > 2) return SDA_E_BADARG;
    if (2 * nf > S1_MAX_FEAT || hidden > S1_MAX_HIDDEN || e > S1_MAX_E) return SDA_E_UNSUPPORTED;
    hipLaunchKernelGGL(step1d_prologue_kernel, dim3(1), dim3(S1_THREADS), 0, (hipStream_t)stream, table, row_len, istep, t_dev, nt,
                       alpha_kind, eta, k, sigma_kind, freqs, nf, w0, b0, hidden, w2, b2, e, wp, bp, cp, out_coef, out_step, mod);
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
