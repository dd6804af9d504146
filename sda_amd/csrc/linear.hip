// Row-major fully-connected layers for the Lorenz *local* score kernel (ScoreNet / ResMLP: sda/nn.py:31-71,
// sda/score.py:38-63) on the fp32 matrix cores, plus zuko LayerNorm over the last axis with 64-lane shuffles.
//
//   sda_linear : Y[r][o] = (act_out?)( sum_i act_in(X[r][i]) * Wop[i][o] + b[o] ) * act'(Z[r][o])? + R[r][o]?
//                Wop = W^T (forward, W is torch's [out][in]) or W (backward-data: gX = gY W).
//   sda_row_ln / sda_row_ln_bwd : per-row standardisation (unbiased variance, eps) and its input gradient.
//
// GEMM tiling: workgroup = 4 wavefronts = 128 rows x 32*NT columns; wave w owns rows [32w, 32w+32);
// A = X[row = lane&31][k = lane>>5], B = Wop[k = lane>>5][col = lane&31], both from LDS tiles padded to 17 floats per
// row (stride 17 -> the 32 lanes of a half-wave hit 32 different banks).  These layers are tiny (widths 128-256) and
// launch-latency bound in the sampler; the kernel is written for correctness and coalescing, not for the MFMA roofline.
#include "sda_common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct LinearParams {
    const float* x; const float* w; const float* b; const float* z; const float* res; float* y;
    int rows, in_f, out_f;
    int trans_w;        // 0: Wop[i][o] = w[o*in_f + i]   1: Wop[i][o] = w[i*out_f + o]
    int act_in, act_out, act_d;
};

#define LIN_BK 16
#define LIN_LD 17

template <int NT>
__global__ __launch_bounds__(256) void linear_kernel(const LinearParams p) {
    __shared__ float sX[128 * LIN_LD];
    __shared__ float sW[32 * NT * LIN_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, khalf = lane >> 5;
    const int row0 = blockIdx.x * 128, col0 = blockIdx.y * 32 * NT;
    f32x16 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    for (int k0 = 0; k0 < p.in_f; k0 += LIN_BK) {
        for (int idx = tid; idx < 128 * LIN_BK; idx += 256) {
            const int r = idx / LIN_BK, kk = idx - r * LIN_BK;
            const int gr = row0 + r, gk = k0 + kk;
            float v = 0.f;
            if (gr < p.rows && gk < p.in_f) {
                v = p.x[(int64_t)gr * p.in_f + gk];
                if (p.act_in) v = sda_act(p.act_in, v);
            }
            sX[r * LIN_LD + kk] = v;
        }
        for (int idx = tid; idx < 32 * NT * LIN_BK; idx += 256) {
            int j, kk;
            if (p.trans_w) { kk = idx / (32 * NT); j = idx - kk * (32 * NT); }
            else { j = idx / LIN_BK; kk = idx - j * LIN_BK; }
            const int gc = col0 + j, gk = k0 + kk;
            float v = 0.f;
            if (gc < p.out_f && gk < p.in_f)
                v = p.trans_w ? p.w[(int64_t)gk * p.out_f + gc] : p.w[(int64_t)gc * p.in_f + gk];
            sW[j * LIN_LD + kk] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k2 = 0; k2 < LIN_BK / 2; ++k2) {
            const float a = sX[(wave * 32 + l31) * LIN_LD + 2 * k2 + khalf];
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const float b = sW[(q * 32 + l31) * LIN_LD + 2 * k2 + khalf];
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int gc = col0 + q * 32 + l31;
        const float bias = (p.b && gc < p.out_f) ? p.b[gc] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (gr < p.rows && gc < p.out_f) {
                const int64_t off = (int64_t)gr * p.out_f + gc;
                float v = acc[q][r] + bias;
                if (p.act_out) v = sda_act(p.act_out, v);
                if (p.z) v *= sda_dact(p.act_d, p.z[off]);
                if (p.res) v += p.res[off];
                p.y[off] = v;
            }
        }
    }
}

extern "C" int sda_linear(const float* x, int rows, int in_f, const float* w, const float* b, int out_f, int trans_w,
                          int act_in, int act_out, const float* dact_z, int act_d, const float* res, float* y,
                          void* stream) {
    if (!x || !w || !y || rows <= 0 || in_f <= 0 || out_f <= 0) return SDA_E_BADARG;
    LinearParams p{x, w, b, dact_z, res, y, rows, in_f, out_f, trans_w, act_in, act_out, act_d};
    const int nt = out_f > 96 ? 4 : (out_f > 64 ? 3 : (out_f > 32 ? 2 : 1));
    dim3 grid((rows + 127) / 128, (out_f + 32 * nt - 1) / (32 * nt)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (nt) {
        case 1: hipLaunchKernelGGL(linear_kernel<1>, grid, block, 0, s, p); break;
        case 2: hipLaunchKernelGGL(linear_kernel<2>, grid, block, 0, s, p); break;
        case 3: hipLaunchKernelGGL(linear_kernel<3>, grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL(linear_kernel<4>, grid, block, 0, s, p); break;
    }
    return sda_launch_status();
}

// ---------------------------------------------------------------- LayerNorm over the last axis: one wavefront per row
#define RLN_MAXF 16   // up to 16 * 64 = 1024 features held in registers

__global__ __launch_bounds__(256) void row_ln_kernel(const float* __restrict__ x, int rows, int f, float eps, int unbiased,
                                                     float* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (row >= rows) return;
    const float* xr = x + row * f;
    float v[RLN_MAXF];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RLN_MAXF; ++i) {
        const int k = lane + i * 64;
        v[i] = k < f ? xr[k] : 0.f;
        s += v[i];
    }
    s = sda_wave_sum(s);
    const float m = __shfl(s, 0, 64) / (float)f;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < RLN_MAXF; ++i) {
        const int k = lane + i * 64;
        const float dlt = k < f ? v[i] - m : 0.f;
        q += dlt * dlt;
    }
    q = sda_wave_sum(q);
    const float var = __shfl(q, 0, 64) / (float)(unbiased ? f - 1 : f);
    const float r = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < RLN_MAXF; ++i) {
        const int k = lane + i * 64;
        if (k < f) y[row * f + k] = (v[i] - m) * r;
    }
    if (lane == 0) {
        if (mean) mean[row] = m;
        if (rstd) rstd[row] = r;
    }
}

extern "C" int sda_row_ln(const float* x, int rows, int f, float eps, int unbiased, float* y, float* mean, float* rstd,
                          void* stream) {
    if (!x || !y || rows <= 0 || f <= 0) return SDA_E_BADARG;
    if (f > RLN_MAXF * 64 || (unbiased && f < 2)) return SDA_E_UNSUPPORTED;
    const int64_t blocks = ((int64_t)rows + 3) / 4;
    if (blocks > 0x7fffffffLL) return SDA_E_UNSUPPORTED;
    hipLaunchKernelGGL(row_ln_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, rows, f, eps, unbiased,
                       y, mean, rstd);
    return sda_launch_status();
}

// gx = (res ? res : 0) + rstd * (gh - mean(gh) - h * sum(gh*h)/(f-1|f)),  h = (x - mean) * rstd
__global__ __launch_bounds__(256) void row_ln_bwd_kernel(const float* __restrict__ gh, const float* __restrict__ x, int rows,
                                                         int f, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, int unbiased,
                                                         const float* __restrict__ res, float* __restrict__ gx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (row >= rows) return;
    const float m = mean[row], r = rstd[row];
    float g[RLN_MAXF], h[RLN_MAXF];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < RLN_MAXF; ++i) {
        const int k = lane + i * 64;
        g[i] = k < f ? gh[row * f + k] : 0.f;
        h[i] = k < f ? (x[row * f + k] - m) * r : 0.f;
        s1 += g[i];
        s2 += g[i] * h[i];
    }
    s1 = __shfl(sda_wave_sum(s1), 0, 64) / (float)f;
    s2 = __shfl(sda_wave_sum(s2), 0, 64) / (float)(unbiased ? f - 1 : f);
#pragma unroll
    for (int i = 0; i < RLN_MAXF; ++i) {
        const int k = lane + i * 64;
        if (k < f) {
            float v = r * (g[i] - s1 - h[i] * s2);
            if (res) v += res[row * f + k];
            gx[row * f + k] = v;
        }
    }
}

extern "C" int sda_row_ln_bwd(const float* gh, const float* x, int rows, int f, const float* mean, const float* rstd,
                              int unbiased, const float* res, float* gx, void* stream) {
    if (!gh || !x || !mean || !rstd || !gx || rows <= 0 || f <= 0) return SDA_E_BADARG;
    if (f > RLN_MAXF * 64) return SDA_E_UNSUPPORTED;
    const int64_t blocks = ((int64_t)rows + 3) / 4;
    if (blocks > 0x7fffffffLL) return SDA_E_UNSUPPORTED;
    hipLaunchKernelGGL(row_ln_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gh, x, rows, f, mean,
                       rstd, unbiased, res, gx);
    return sda_launch_status();
}
