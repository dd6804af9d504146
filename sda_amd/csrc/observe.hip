// Linear observation operators of the reference's experiments and their adjoints:
//   strided sub-sampling x[..., a::s, ...]   (experiments/lorenz/eval.py:75, kolmogorov/figures.ipynb#cell30-39)
//   KolmogorovFlow.coarsen  (block mean)     (sda/mcs.py:340-347)
//   KolmogorovFlow.vorticity (periodic central differences)   (sda/mcs.py:361-375)
// With an adjoint available, GaussianScore needs no autograd through A: d log p / d x_hat = A^T((y - A x_hat)/var).
// All kernels are streaming (HBM-bound): one read of the source, one write of the destination.
#include "sda_common.hpp"

static inline unsigned obs_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b > 65536) b = 65536;
    if (b < 1) b = 1;
    return (unsigned)b;
}

struct Slice5 { int size[5]; int start[5]; int step[5]; int osize[5]; };

// out[o0..o4] = x[start + o*step]   over a 5-D view (leading dims padded with size 1)
__global__ void strided_gather_kernel(const float* __restrict__ x, Slice5 s, float* __restrict__ out, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i, src = 0, mul = 1;
        int idx[5];
#pragma unroll
        for (int d = 4; d >= 0; --d) { idx[d] = (int)(r % s.osize[d]); r /= s.osize[d]; }
#pragma unroll
        for (int d = 4; d >= 0; --d) { src += (int64_t)(s.start[d] + idx[d] * s.step[d]) * mul; mul *= s.size[d]; }
        out[i] = x[src];
    }
}

// gx = 0 except gx[start + o*step] = r[o]
__global__ void strided_scatter_kernel(const float* __restrict__ r, Slice5 s, float* __restrict__ gx, int64_t total_x) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_x; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t q = i, dst = 0, mul = 1;
        bool hit = true;
        int o[5];
#pragma unroll
        for (int d = 4; d >= 0; --d) {
            const int c = (int)(q % s.size[d]); q /= s.size[d];
            const int rel = c - s.start[d];
            hit = hit && rel >= 0 && (rel % s.step[d]) == 0 && (rel / s.step[d]) < s.osize[d];
            o[d] = rel / s.step[d];
        }
        if (hit) {
#pragma unroll
            for (int d = 4; d >= 0; --d) { dst += (int64_t)o[d] * mul; mul *= s.osize[d]; }
        }
        gx[i] = hit ? r[dst] : 0.f;
    }
}

// stop (optional): exclusive end per dim (a crop x[..., a:b:s]); NULL = to the end of the dim
static int fill_slice(Slice5* s, const int* size, const int* start, const int* step, const int* stop = nullptr) {
    for (int d = 0; d < 5; ++d) {
        if (size[d] <= 0 || step[d] <= 0 || start[d] < 0 || start[d] >= size[d]) return SDA_E_BADARG;
        const int end = stop ? stop[d] : size[d];
        if (end <= start[d] || end > size[d]) return SDA_E_BADARG;
        s->size[d] = size[d]; s->start[d] = start[d]; s->step[d] = step[d];
        s->osize[d] = (end - start[d] + step[d] - 1) / step[d];
    }
    return SDA_OK;
}

extern "C" int sda_obs_subsample(const float* x, const int* size5, const int* start5, const int* step5, const int* stop5,
                                 float* out, void* stream) {
    if (!x || !out || !size5 || !start5 || !step5) return SDA_E_BADARG;
    Slice5 s;
    int rc = fill_slice(&s, size5, start5, step5, stop5);
    if (rc) return rc;
    int64_t total = 1;
    for (int d = 0; d < 5; ++d) total *= s.osize[d];
    hipLaunchKernelGGL(strided_gather_kernel, dim3(obs_grid(total)), dim3(256), 0, (hipStream_t)stream, x, s, out, total);
    return sda_launch_status();
}

extern "C" int sda_obs_subsample_adjoint(const float* r, const int* size5, const int* start5, const int* step5, const int* stop5,
                                         float* gx, void* stream) {
    if (!r || !gx || !size5 || !start5 || !step5) return SDA_E_BADARG;
    Slice5 s;
    int rc = fill_slice(&s, size5, start5, step5, stop5);
    if (rc) return rc;
    int64_t total = 1;
    for (int d = 0; d < 5; ++d) total *= s.size[d];
    hipLaunchKernelGGL(strided_scatter_kernel, dim3(obs_grid(total)), dim3(256), 0, (hipStream_t)stream, r, s, gx, total);
    return sda_launch_status();
}

// Gaussian-likelihood guidance through a subsampling observation in ONE launch (sda/score.py:387-394 with A = x[..., ::s]):
//   g[i] = (y[o(i)] - (x[i] - sigma eps[i]) / mu) / (std^2 + gamma (sigma/mu)^2)   on the observed lattice, 0 elsewhere
// = A^T((y - A x_hat) / var), i.e. sda_denoise + sda_obs_subsample + sda_gauss_cotangent + sda_obs_subsample_adjoint (four
// launches of ~5 us each on the latency-bound Lorenz workloads).  y broadcasts over the leading axis (y_numel divides |A x|).
__global__ void subsample_guidance_kernel(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ y,
                                          int64_t y_numel, Slice5 s, int64_t total_x, float std, float gamma, float mu, float sigma,
                                          const float* __restrict__ coef, float* __restrict__ g) {
    if (coef) { mu = coef[0]; sigma = coef[1]; }
    const float r = __fdiv_rn(sigma, mu);
    const float var = __fadd_rn(__fmul_rn(std, std), __fmul_rn(gamma, __fmul_rn(r, r)));
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_x; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t q = i, dst = 0, mul = 1;
        bool hit = true;
        int o[5];
#pragma unroll
        for (int d = 4; d >= 0; --d) {
            const int c = (int)(q % s.size[d]); q /= s.size[d];
            const int rel = c - s.start[d];
            hit = hit && rel >= 0 && (rel % s.step[d]) == 0 && (rel / s.step[d]) < s.osize[d];
            o[d] = rel / s.step[d];
        }
        float v = 0.f;
        if (hit) {
#pragma unroll
            for (int d = 4; d >= 0; --d) { dst += (int64_t)o[d] * mul; mul *= s.osize[d]; }
            const float xh = (x[i] - sigma * eps[i]) / mu;
            v = __fdiv_rn(y[dst % y_numel] - xh, var);
        }
        g[i] = v;
    }
}

extern "C" int sda_obs_subsample_guidance(const float* x, const float* eps, const float* y, int64_t y_numel, const int* size5,
                                          const int* start5, const int* step5, const int* stop5, float std, float gamma,
                                          float mu, float sigma, const float* coef_dev, float* g, void* stream) {
    if (!x || !eps || !y || !g || !size5 || !start5 || !step5 || y_numel <= 0) return SDA_E_BADARG;
    Slice5 s;
    int rc = fill_slice(&s, size5, start5, step5, stop5);    // (slices with a stop: fewer observed positions along that axis)
    if (rc) return rc;
    int64_t total = 1, ototal = 1;
    for (int d = 0; d < 5; ++d) { total *= s.size[d]; ototal *= s.osize[d]; }
    if (ototal % y_numel) return SDA_E_BADARG;
    hipLaunchKernelGGL(subsample_guidance_kernel, dim3(obs_grid(total)), dim3(256), 0, (hipStream_t)stream, x, eps, y, y_numel, s,
                       total, std, gamma, mu, sigma, coef_dev, g);
    return sda_launch_status();
}

// coarsen: out[n][y][x] = mean over the f x f cell   (planes = product of leading dims)
__global__ void coarsen_kernel(const float* __restrict__ x, int64_t planes, int h, int w, int f, float* __restrict__ out) {
    const int ho = h / f, wo = w / f;
    const int64_t total = planes * ho * wo;
    const float inv = 1.0f / (float)(f * f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % wo);
        const int oy = (int)((i / wo) % ho);
        const int64_t p = i / ((int64_t)wo * ho);
        const float* src = x + p * h * w + (int64_t)oy * f * w + ox * f;
        float acc = 0.f;
        for (int a = 0; a < f; ++a)
            for (int b = 0; b < f; ++b) acc += src[a * w + b];
        out[i] = acc * inv;
    }
}

__global__ void coarsen_adjoint_kernel(const float* __restrict__ r, int64_t planes, int h, int w, int f,
                                       float* __restrict__ gx) {
    const int ho = h / f, wo = w / f;
    const int64_t total = planes * h * w;
    const float inv = 1.0f / (float)(f * f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % w);
        const int yy = (int)((i / w) % h);
        const int64_t p = i / ((int64_t)w * h);
        gx[i] = r[(p * ho + yy / f) * wo + xx / f] * inv;
    }
}

extern "C" int sda_obs_coarsen(const float* x, int64_t planes, int h, int w, int f, float* out, void* stream) {
    if (!x || !out || planes <= 0 || h <= 0 || w <= 0 || f <= 0 || h % f || w % f) return SDA_E_BADARG;
    hipLaunchKernelGGL(coarsen_kernel, dim3(obs_grid(planes * (h / f) * (w / f))), dim3(256), 0, (hipStream_t)stream, x,
                       planes, h, w, f, out);
    return sda_launch_status();
}

extern "C" int sda_obs_coarsen_adjoint(const float* r, int64_t planes, int h, int w, int f, float* gx, void* stream) {
    if (!r || !gx || planes <= 0 || h <= 0 || w <= 0 || f <= 0 || h % f || w % f) return SDA_E_BADARG;
    hipLaunchKernelGGL(coarsen_adjoint_kernel, dim3(obs_grid(planes * h * w)), dim3(256), 0, (hipStream_t)stream, r, planes,
                       h, w, f, gx);
    return sda_launch_status();
}

// vorticity: x [pairs][2][h][w] (u, v) -> w[pairs][h][w] = (u[x+1]-u[x-1])/2 - (v[y+1]-v[y-1])/2, periodic
__global__ void vorticity_kernel(const float* __restrict__ x, int64_t pairs, int h, int w, float* __restrict__ out) {
    const int64_t total = pairs * h * w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % w);
        const int yy = (int)((i / w) % h);
        const int64_t p = i / ((int64_t)w * h);
        const float* u = x + p * 2 * h * w;
        const float* v = u + (int64_t)h * w;
        const int xp = xx + 1 == w ? 0 : xx + 1, xm = xx == 0 ? w - 1 : xx - 1;
        const int yp = yy + 1 == h ? 0 : yy + 1, ym = yy == 0 ? h - 1 : yy - 1;
        out[i] = (u[yy * w + xp] - u[yy * w + xm]) * 0.5f - (v[yp * w + xx] - v[ym * w + xx]) * 0.5f;
    }
}

// adjoint: gu[y][x] = (r[y][x-1] - r[y][x+1])/2 ;  gv[y][x] = (r[y+1][x] - r[y-1][x])/2
__global__ void vorticity_adjoint_kernel(const float* __restrict__ r, int64_t pairs, int h, int w, float* __restrict__ gx) {
    const int64_t total = pairs * h * w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % w);
        const int yy = (int)((i / w) % h);
        const int64_t p = i / ((int64_t)w * h);
        const float* rp = r + p * h * w;
        const int xp = xx + 1 == w ? 0 : xx + 1, xm = xx == 0 ? w - 1 : xx - 1;
        const int yp = yy + 1 == h ? 0 : yy + 1, ym = yy == 0 ? h - 1 : yy - 1;
        float* gu = gx + p * 2 * h * w;
        float* gv = gu + (int64_t)h * w;
        gu[yy * w + xx] = (rp[yy * w + xm] - rp[yy * w + xp]) * 0.5f;
        gv[yy * w + xx] = (rp[yp * w + xx] - rp[ym * w + xx]) * 0.5f;
    }
}

extern "C" int sda_obs_vorticity(const float* x, int64_t pairs, int h, int w, float* out, void* stream) {
    if (!x || !out || pairs <= 0 || h < 3 || w < 3) return SDA_E_BADARG;
    hipLaunchKernelGGL(vorticity_kernel, dim3(obs_grid(pairs * h * w)), dim3(256), 0, (hipStream_t)stream, x, pairs, h, w, out);
    return sda_launch_status();
}

extern "C" int sda_obs_vorticity_adjoint(const float* r, int64_t pairs, int h, int w, float* gx, void* stream) {
    if (!r || !gx || pairs <= 0 || h < 3 || w < 3) return SDA_E_BADARG;
    hipLaunchKernelGGL(vorticity_adjoint_kernel, dim3(obs_grid(pairs * h * w)), dim3(256), 0, (hipStream_t)stream, r, pairs,
                       h, w, gx);
    return sda_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// The non-linear / masked / coupled observations of the reference's experiments, each with the VJP of its
// linearisation so that GaussianScore (sda/score.py:389-394) needs no autograd through A:
//   pointwise: w / (1 + |w|) (kolmogorov/figures.ipynb#cell23, the saturating sensor), tanh, square, |w|
//   mask:      A(x) * mask  (figures.ipynb#cell4; the mask broadcasts over the leading dims)
//   timediff:  x[.., i, ..] - x[.., j, ..] along one axis (the loop closure of figures.ipynb#cell43)
enum { OBS_PW_SATURATE = 1, OBS_PW_TANH = 2, OBS_PW_SQUARE = 3, OBS_PW_ABS = 4 };

__device__ __forceinline__ float obs_pw(int kind, float v) {
    switch (kind) {
        case OBS_PW_SATURATE: return __fdiv_rn(v, 1.f + fabsf(v));
        case OBS_PW_TANH: return tanhf(v);
        case OBS_PW_SQUARE: return v * v;
        case OBS_PW_ABS: return fabsf(v);
        default: return v;
    }
}

__device__ __forceinline__ float obs_dpw(int kind, float v) {
    switch (kind) {
        case OBS_PW_SATURATE: { const float d = 1.f + fabsf(v); return __fdiv_rn(1.f, d * d); }     // d/dv v/(1+|v|) = 1/(1+|v|)^2
        case OBS_PW_TANH: { const float t = tanhf(v); return 1.f - t * t; }
        case OBS_PW_SQUARE: return 2.f * v;
        case OBS_PW_ABS: return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
        default: return 1.f;
    }
}

// r == nullptr: out = f(x);  else: out = r * f'(x)
__global__ void pointwise_kernel(const float* __restrict__ x, const float* __restrict__ r, int64_t n, int kind, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = r ? r[i] * obs_dpw(kind, x[i]) : obs_pw(kind, x[i]);
}

extern "C" int sda_obs_pointwise(const float* x, int64_t n, int kind, float* out, void* stream) {
    if (!x || !out || n <= 0 || kind < OBS_PW_SATURATE || kind > OBS_PW_ABS) return SDA_E_BADARG;
    hipLaunchKernelGGL(pointwise_kernel, dim3(obs_grid(n)), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, n, kind, out);
    return sda_launch_status();
}

extern "C" int sda_obs_pointwise_vjp(const float* x, const float* r, int64_t n, int kind, float* gx, void* stream) {
    if (!x || !r || !gx || n <= 0 || kind < OBS_PW_SATURATE || kind > OBS_PW_ABS) return SDA_E_BADARG;
    hipLaunchKernelGGL(pointwise_kernel, dim3(obs_grid(n)), dim3(256), 0, (hipStream_t)stream, x, r, n, kind, gx);
    return sda_launch_status();
}

__global__ void mask_kernel(const float* __restrict__ x, int64_t n, const float* __restrict__ m, int64_t m_numel, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = x[i] * m[i % m_numel];
}

// out = x * mask, the mask broadcast over the leading dims (m_numel divides n); self-adjoint
extern "C" int sda_obs_mask(const float* x, int64_t n, const float* m, int64_t m_numel, float* out, void* stream) {
    if (!x || !m || !out || n <= 0 || m_numel <= 0 || n % m_numel) return SDA_E_BADARG;
    hipLaunchKernelGGL(mask_kernel, dim3(obs_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, m, m_numel, out);
    return sda_launch_status();
}

// x [outer][len][inner] -> out [outer][inner] = x[:, i] - x[:, j]
__global__ void timediff_kernel(const float* __restrict__ x, int64_t outer, int len, int64_t inner, int i, int j, float* __restrict__ out) {
    const int64_t total = outer * inner;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = e / inner, q = e - o * inner;
        const float* b = x + o * len * inner + q;
        out[e] = b[(int64_t)i * inner] - b[(int64_t)j * inner];
    }
}

// gx [outer][len][inner] = +r at index i, -r at index j, 0 elsewhere (i == j: all zero)
__global__ void timediff_adjoint_kernel(const float* __restrict__ r, int64_t outer, int len, int64_t inner, int i, int j,
                                        float* __restrict__ gx) {
    const int64_t total = outer * len * inner;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = e % inner, l = (e / inner) % len, o = e / (inner * len);
        const float v = r[o * inner + q];
        gx[e] = (l == i ? v : 0.f) - (l == j ? v : 0.f);
    }
}

extern "C" int sda_obs_timediff(const float* x, int64_t outer, int len, int64_t inner, int i, int j, float* out, void* stream) {
    if (!x || !out || outer <= 0 || len <= 0 || inner <= 0 || i < 0 || j < 0 || i >= len || j >= len) return SDA_E_BADARG;
    hipLaunchKernelGGL(timediff_kernel, dim3(obs_grid(outer * inner)), dim3(256), 0, (hipStream_t)stream, x, outer, len, inner, i, j, out);
    return sda_launch_status();
}

extern "C" int sda_obs_timediff_adjoint(const float* r, int64_t outer, int len, int64_t inner, int i, int j, float* gx, void* stream) {
    if (!r || !gx || outer <= 0 || len <= 0 || inner <= 0 || i < 0 || j < 0 || i >= len || j >= len) return SDA_E_BADARG;
    hipLaunchKernelGGL(timediff_adjoint_kernel, dim3(obs_grid(outer * len * inner)), dim3(256), 0, (hipStream_t)stream, r, outer, len,
                       inner, i, j, gx);
    return sda_launch_status();
}
