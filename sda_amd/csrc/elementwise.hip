// HBM-bound glue kernels of the sampling loop: time embedding, fold / adjoints, predictor-corrector updates,
// guidance elementwise pieces.  All are streaming kernels (one read + one write of each operand); the per-sample
// reduction uses 64-lane wavefront shuffles.
#include "sda_common.hpp"

extern "C" int sda_abi_version(void) { return SDA_ABI_VERSION; }

// ------------------------------------------------------------------------------------------ TimeEmbedding (score.py:15-35)
// one workgroup per time value; feat -> hidden (SiLU) -> emb, everything staged in LDS
#define TE_THREADS 256
#define TE_MAX_FEAT 128
#define TE_MAX_HIDDEN 1024

__global__ __launch_bounds__(TE_THREADS) void time_embed_kernel(const float* __restrict__ t, const float* __restrict__ freqs,
                                                                int nf, const float* __restrict__ w0,
                                                                const float* __restrict__ b0, int hidden,
                                                                const float* __restrict__ w2,
                                                                const float* __restrict__ b2, int e,
                                                                float* __restrict__ emb) {
    __shared__ float feat[TE_MAX_FEAT];
    __shared__ float hid[TE_MAX_HIDDEN];
    const int it = blockIdx.x;
    const float tv = t[it];
    for (int j = threadIdx.x; j < nf; j += TE_THREADS) {
        const float ang = freqs[j] * tv;
        feat[j] = cosf(ang);
        feat[nf + j] = sinf(ang);
    }
    __syncthreads();
    const int nin = 2 * nf;
    for (int hh = threadIdx.x; hh < hidden; hh += TE_THREADS) {
        const float acc = sda_dot8(w0 + (int64_t)hh * nin, feat, nin);
        hid[hh] = sda_act(SDA_ACT_SILU, acc + b0[hh]);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < e; o += TE_THREADS) {
        const float acc = sda_dot8(w2 + (int64_t)o * hidden, hid, hidden);
        emb[(int64_t)it * e + o] = acc + b2[o];
    }
}

extern "C" int sda_time_embed(const float* t, int nt, const float* freqs, int nf, const float* w0, const float* b0,
                              int hidden, const float* w2, const float* b2, int e, float* emb, void* stream) {
    if (!t || !freqs || !w0 || !b0 || !w2 || !b2 || !emb || nt <= 0 || nf <= 0 || hidden <= 0 || e <= 0) return SDA_E_BADARG;
    if (2 * nf > TE_MAX_FEAT || hidden > TE_MAX_HIDDEN) return SDA_E_UNSUPPORTED;
    hipLaunchKernelGGL(time_embed_kernel, dim3(nt), dim3(TE_THREADS), 0, (hipStream_t)stream, t, freqs, nf, w0, b0, hidden,
                       w2, b2, e, emb);
    return sda_launch_status();
}

// y[r][o] = b[o] + sum_i x[r][i] w[o][i]  -- the blocks' `project` Linears, all concatenated (nn.py:132-135).
// One wavefront per output: lanes stride the input features, shuffle-reduce.
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x, int rows, int in_f,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           int out_f, float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (gw >= (int64_t)rows * out_f) return;
    const int r = (int)(gw / out_f), o = (int)(gw - (int64_t)r * out_f);
    const float* xr = x + (int64_t)r * in_f;
    const float* wr = w + (int64_t)o * in_f;
    float acc = 0.f;
    for (int i = lane; i < in_f; i += 64) acc += xr[i] * wr[i];
    acc = sda_wave_sum(acc);
    if (lane == 0) y[gw] = acc + (b ? b[o] : 0.f);
}

extern "C" int sda_linear_small(const float* x, int rows, int in_f, const float* w, const float* b, int out_f, float* y,
                                void* stream) {
    if (!x || !w || !y || rows <= 0 || in_f <= 0 || out_f <= 0) return SDA_E_BADARG;
    const int64_t waves = (int64_t)rows * out_f;
    const int64_t blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffLL) return SDA_E_UNSUPPORTED;
    hipLaunchKernelGGL(linear_small_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, rows, in_f, w, b,
                       out_f, y);
    return sda_launch_status();
}

// ------------------------------------------------------------------------------------------ fold and adjoints (score.py:146-164)
static inline unsigned grid_for(int64_t total, int threads, int64_t cap = 65536) {
    int64_t b = (total + threads - 1) / threads;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// out[b][l][c][p] = s[b][win(l)][slot(l)*c + c][p]
__global__ void fold_kernel(const float* __restrict__ s, int nb, int nw, int k, int c, int hw, float* __restrict__ out) {
    const int L = nw + 2 * k, wlen = 2 * k + 1;
    const int64_t chw = (int64_t)c * hw;
    const int64_t total = (int64_t)nb * L * chw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t rem = i % chw;
        const int64_t bl = i / chw;
        const int l = (int)(bl % L);
        const int64_t b = bl / L;
        int win, slot;
        if (l < k) { win = 0; slot = l; }
        else if (l >= nw + k) { win = nw - 1; slot = l - (nw - 1); }
        else { win = l - k; slot = k; }
        out[i] = s[((b * nw + win) * wlen + slot) * chw + rem];
    }
}

extern "C" int sda_fold(const float* s, int b, int nw, int k, int c, int hw, float* out, void* stream) {
    if (!s || !out || b <= 0 || nw <= 0 || k < 0 || c <= 0 || hw <= 0) return SDA_E_BADARG;
    const int64_t total = (int64_t)b * (nw + 2 * k) * c * hw;
    hipLaunchKernelGGL(fold_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, s, b, nw, k, c, hw, out);
    return sda_launch_status();
}

// g_s[b][i][j][c][p] = fold reads (i,j) ? g_out[b][i+j][c][p] : 0
__global__ void fold_adjoint_kernel(const float* __restrict__ g_out, int nb, int nw, int k, int c, int hw,
                                    float* __restrict__ g_s) {
    const int L = nw + 2 * k, wlen = 2 * k + 1;
    const int64_t chw = (int64_t)c * hw;
    const int64_t total = (int64_t)nb * nw * wlen * chw;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t rem = idx % chw;
        int64_t r = idx / chw;
        const int j = (int)(r % wlen); r /= wlen;
        const int i = (int)(r % nw);
        const int64_t b = r / nw;
        const bool sel = (j == k) || (i == 0 && j < k) || (i == nw - 1 && j > k);
        g_s[idx] = sel ? g_out[(b * L + i + j) * chw + rem] : 0.f;
    }
}

extern "C" int sda_fold_adjoint(const float* g_out, int b, int nw, int k, int c, int hw, float* g_s, void* stream) {
    if (!g_out || !g_s || b <= 0 || nw <= 0 || k < 0 || c <= 0 || hw <= 0) return SDA_E_BADARG;
    const int64_t total = (int64_t)b * nw * (2 * k + 1) * c * hw;
    hipLaunchKernelGGL(fold_adjoint_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, g_out, b, nw, k,
                       c, hw, g_s);
    return sda_launch_status();
}

// g_x[b][l][c][p] = sum_{j, i = l - j in [0,nw)} g_win[b][i][j*c + c][p]     (window stride = win_c_total channels)
__global__ void unfold_adjoint_kernel(const float* __restrict__ g_win, int nb, int nw, int k, int c, int hw,
                                      int64_t win_c_total, float* __restrict__ g_x) {
    const int L = nw + 2 * k, wlen = 2 * k + 1;
    const int64_t chw = (int64_t)c * hw;
    const int64_t total = (int64_t)nb * L * chw;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t rem = idx % chw;
        const int64_t bl = idx / chw;
        const int l = (int)(bl % L);
        const int64_t b = bl / L;
        float acc = 0.f;
        for (int j = 0; j < wlen; ++j) {
            const int i = l - j;
            if (i >= 0 && i < nw) acc += g_win[(b * nw + i) * win_c_total * hw + (int64_t)j * chw + rem];
        }
        g_x[idx] = acc;
    }
}

extern "C" int sda_unfold_adjoint(const float* g_win, int b, int nw, int k, int c, int hw, int64_t win_c_total, float* g_x,
                                  void* stream) {
    if (!g_win || !g_x || b <= 0 || nw <= 0 || k < 0 || c <= 0 || hw <= 0 || win_c_total < (int64_t)(2 * k + 1) * c)
        return SDA_E_BADARG;
    const int64_t total = (int64_t)b * (nw + 2 * k) * c * hw;
    hipLaunchKernelGGL(unfold_adjoint_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, g_win, b, nw,
                       k, c, hw, win_c_total, g_x);
    return sda_launch_status();
}

// ------------------------------------------------------------------------------------------ VP schedule (score.py:195-210, 279-302)
// out = {mu(t), sigma(t)} for a device-resident scalar t: one launch instead of the ~10 scalar torch kernels that
// alpha(t), alpha(t)**2, 1 - ..., sqrt make of it (they dominate the launch count of the 1-D nets' guided step).
//   alpha_kind 0 'lin': 1 - (1 - eta) t      1 'cos': cos(k t)^2, k = acos(sqrt(eta))      2 'exp': exp(k t^2), k = log(eta)
//   sigma_kind 0 VPSDE: sqrt(1 - a^2 + eta^2)    1 SubVPSDE: 1 - a^2 + eta    2 SubSubVPSDE: 1 - a + eta
__global__ void vp_schedule_kernel(const float* __restrict__ t, int alpha_kind, float eta, float k, int sigma_kind,
                                   float* __restrict__ out) {
    const float tv = t[0];
    float a;
    if (alpha_kind == 0) a = 1.0f - (1.0f - eta) * tv;
    else if (alpha_kind == 1) { const float c = cosf(k * tv); a = c * c; }
    else a = expf(k * (tv * tv));
    float sg;
    if (sigma_kind == 0) sg = sqrtf((1.0f - a * a) + eta * eta);
    else if (sigma_kind == 1) sg = (1.0f - a * a) + eta;
    else sg = (1.0f - a) + eta;
    out[0] = a;
    out[1] = sg;
}

extern "C" int sda_vp_schedule(const float* t, int alpha_kind, float eta, float k, int sigma_kind, float* out2, void* stream) {
    if (!t || !out2 || alpha_kind < 0 || alpha_kind > 2 || sigma_kind < 0 || sigma_kind > 2) return SDA_E_BADARG;
    hipLaunchKernelGGL(vp_schedule_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, t, alpha_kind, eta, k, sigma_kind, out2);
    return sda_launch_status();
}

// ------------------------------------------------------------------------------------------ predictor / corrector (score.py:250-261)
__global__ void pc_predict_kernel(float* __restrict__ x, const float* __restrict__ eps, int64_t numel, float r, float c1,
                                  const float* __restrict__ coef) {
    if (coef) { r = coef[0]; c1 = coef[1]; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
        x[i] = r * x[i] + c1 * eps[i];
}

extern "C" int sda_pc_predict(float* x, const float* eps, int64_t numel, float r, float c1, const float* coef_dev,
                              void* stream) {
    if (!x || !eps || numel <= 0) return SDA_E_BADARG;
    hipLaunchKernelGGL(pc_predict_kernel, dim3(grid_for(numel, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, eps, numel,
                       r, c1, coef_dev);
    return sda_launch_status();
}

// partial[b][j] = sum of eps^2 over chunk j of sample b  (deterministic two-stage reduction; grid = (nchunk, b))
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ eps, int64_t per_sample,
                                                            float* __restrict__ partial, int nchunk) {
    __shared__ float wsum[4];
    const int j = blockIdx.x, b = blockIdx.y;
    const int64_t chunk = (per_sample + nchunk - 1) / nchunk;
    const int64_t lo = (int64_t)j * chunk;
    int64_t hi = lo + chunk;
    if (hi > per_sample) hi = per_sample;
    const float* e = eps + (int64_t)b * per_sample;
    float acc = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) { const float v = e[i]; acc += v * v; }
    acc = sda_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)b * nchunk + j] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

extern "C" int sda_sumsq_partial(const float* eps, int b, int64_t per_sample, float* partial, int nchunk, void* stream) {
    if (!eps || !partial || b <= 0 || per_sample <= 0 || nchunk <= 0 || nchunk > 1024 || b > 65535) return SDA_E_BADARG;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nchunk, b), dim3(256), 0, (hipStream_t)stream, eps, per_sample, partial,
                       nchunk);
    return sda_launch_status();
}

// delta = tau / mean(eps^2);  x -= (delta*eps + sqrt(2 delta) z) * sigma           (grid = (blocks, b))
__global__ __launch_bounds__(256) void pc_correct_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                                         const float* __restrict__ z, int64_t per_sample,
                                                         const float* __restrict__ partial, int nchunk, float tau,
                                                         float sigma, const float* __restrict__ coef) {
    const int b = blockIdx.y;
    if (coef) sigma = coef[0];
    float tot = 0.f;
    for (int j = 0; j < nchunk; ++j) tot += partial[(int64_t)b * nchunk + j];
    const float delta = tau / (tot / (float)per_sample);
    const float sq = sqrtf(2.0f * delta);
    const int64_t base = (int64_t)b * per_sample;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_sample; i += (int64_t)gridDim.x * 256) {
        const int64_t o = base + i;
        x[o] = x[o] - (delta * eps[o] + sq * z[o]) * sigma;
    }
}

extern "C" int sda_pc_correct(float* x, const float* eps, const float* z, int b, int64_t per_sample, const float* partial,
                              int nchunk, float tau, float sigma, const float* coef_dev, void* stream) {
    if (!x || !eps || !z || !partial || b <= 0 || per_sample <= 0 || nchunk <= 0 || b > 65535) return SDA_E_BADARG;
    hipLaunchKernelGGL(pc_correct_kernel, dim3(grid_for(per_sample, 256, 2048), b), dim3(256), 0, (hipStream_t)stream, x,
                       eps, z, per_sample, partial, nchunk, tau, sigma, coef_dev);
    return sda_launch_status();
}

// ------------------------------------------------------------------------------------------ guidance glue (score.py:387,396)
__global__ void denoise_kernel(const float* __restrict__ x, const float* __restrict__ eps, int64_t numel, float mu,
                               float sigma, const float* __restrict__ coef, float* __restrict__ xhat) {
    if (coef) { mu = coef[0]; sigma = coef[1]; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
        xhat[i] = (x[i] - sigma * eps[i]) / mu;
}

extern "C" int sda_denoise(const float* x, const float* eps, int64_t numel, float mu, float sigma,
                           const float* coef_dev, float* xhat, void* stream) {
    if (!x || !eps || !xhat || numel <= 0) return SDA_E_BADARG;
    hipLaunchKernelGGL(denoise_kernel, dim3(grid_for(numel, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, eps, numel,
                       mu, sigma, coef_dev, xhat);
    return sda_launch_status();
}

// s = d log_p / d x = ghat/mu - (sigma/mu) * J_eps^T ghat ;  out = eps - sigma * s      (vjp = J_eps^T ghat)
__global__ void guided_combine_kernel(const float* __restrict__ eps, const float* __restrict__ ghat,
                                      const float* __restrict__ vjp, int64_t numel, float mu, float sigma,
                                      const float* __restrict__ coef, float* __restrict__ out) {
    if (coef) { mu = coef[0]; sigma = coef[1]; }
    const float k = sigma / mu;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = eps[i] - k * (ghat[i] - (vjp ? sigma * vjp[i] : 0.f));
}

extern "C" int sda_guided_combine(const float* eps, const float* ghat, const float* vjp, int64_t numel, float mu,
                                  float sigma, const float* coef_dev, float* out, void* stream) {
    if (!eps || !ghat || !out || numel <= 0) return SDA_E_BADARG;
    hipLaunchKernelGGL(guided_combine_kernel, dim3(grid_for(numel, 256, 8192)), dim3(256), 0, (hipStream_t)stream, eps, ghat,
                       vjp, numel, mu, sigma, coef_dev, out);
    return sda_launch_status();
}

// (y - A x_hat) / (std^2 + gamma (sigma / mu)^2): the cotangent of the Gaussian likelihood (sda/score.py:389-392) for scalar
// std / gamma, in one launch instead of seven tiny elementwise ones; y broadcasts over the leading (batch) axis when it is
// shorter than ax (y_numel divides numel).
__global__ void gauss_cotangent_kernel(const float* __restrict__ y, int64_t y_numel, const float* __restrict__ ax, int64_t numel,
                                       float std, float gamma, float mu, float sigma, const float* __restrict__ coef,
                                       float* __restrict__ out) {
    if (coef) { mu = coef[0]; sigma = coef[1]; }
    // (the reference's operation order, rounded step by step: no fused multiply-add, a division per element)
    const float r = __fdiv_rn(sigma, mu);
    const float var = __fadd_rn(__fmul_rn(std, std), __fmul_rn(gamma, __fmul_rn(r, r)));
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = __fdiv_rn(y[y_numel == numel ? i : i % y_numel] - ax[i], var);
}

extern "C" int sda_gauss_cotangent(const float* y, int64_t y_numel, const float* ax, int64_t numel, float std, float gamma,
                                   float mu, float sigma, const float* coef_dev, float* out, void* stream) {
    if (!y || !ax || !out || numel <= 0 || y_numel <= 0 || numel % y_numel) return SDA_E_BADARG;
    hipLaunchKernelGGL(gauss_cotangent_kernel, dim3(grid_for(numel, 256, 8192)), dim3(256), 0, (hipStream_t)stream, y, y_numel, ax,
                       numel, std, gamma, mu, sigma, coef_dev, out);
    return sda_launch_status();
}
