// Channel LayerNorm (zuko.nn.LayerNorm(dim=-(spatial+1)); call sites sda/nn.py:137,163) on PLANAR tensors.
//
// x: [n][c][hw].  One thread owns one pixel; consecutive lanes own consecutive pixels, so every channel-plane
// access of a wavefront is one contiguous 256-byte segment.  These kernels are HBM-bound: algorithmic bytes are
//   ln_stats : read x once (second pass re-reads through L2/MALL)                   = c*hw*4 B / image
//   ln_bwd   : read gh (x4 when pooling) + x twice (2nd pass cached), write gx
// The normalisation itself is never materialised in the network path: sda_conv_igemm's loader applies
// (x + mod - mean) * rstd while it stages the halo tile.  sda_ln_apply exists for tests / unfused callers.
#include "sda_common.hpp"

// streaming accesses of the quad kernels: every byte is touched once.  On the 96-channel level (SPLIT = 8: 128-byte runs per
// channel plane, 3 GB tensors) the non-temporal hint is worth 7 % on ln_stats (5.9 -> 6.3 TB/s) and 2-8 % on ln_bwd (5.3-5.6 ->
// 5.8); on the 192 / 384-channel levels (SPLIT = 16: 64-byte runs) it costs 4-6 % -- measured with tools/ln_bench.py at 120
// windows -- so it is a per-instantiation choice.
typedef float ln_f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ln_ld4(const float* p) {
    if constexpr (NT) {
        const ln_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const ln_f32x4*>(p));
        return make_float4(v[0], v[1], v[2], v[3]);
    } else {
        return *reinterpret_cast<const float4*>(p);
    }
}
template <bool NT>
__device__ __forceinline__ void ln_st4(float* p, float4 v) {
    if constexpr (NT) __builtin_nontemporal_store(ln_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<ln_f32x4*>(p));
    else *reinterpret_cast<float4*>(p) = v;
}

#include <stdlib.h>

#define LN_THREADS 256

__global__ __launch_bounds__(LN_THREADS) void ln_stats_kernel(const float* __restrict__ x, int64_t npix, int c, int hw,
                                                              const float* __restrict__ mod, int64_t mod_sn, float eps,
                                                              int unbiased, float* __restrict__ mean,
                                                              float* __restrict__ rstd) {
    const int64_t idx = (int64_t)blockIdx.x * LN_THREADS + threadIdx.x;
    if (idx >= npix) return;
    const int64_t n = idx / hw;
    const int p = (int)(idx - n * hw);
    const float* xp = x + n * (int64_t)c * hw + p;
    const float* mp = mod ? mod + n * mod_sn : nullptr;
    // pass 1: mean
    float s = 0.f;
    for (int k = 0; k < c; ++k) {
        float v = xp[(int64_t)k * hw];
        if (mp) v += mp[k];
        s += v;
    }
    const float m = s / (float)c;
    // pass 2: centred second moment (two-pass, as torch.var_mean does; no E[x^2]-m^2 cancellation)
    float q = 0.f;
    for (int k = 0; k < c; ++k) {
        float v = xp[(int64_t)k * hw];
        if (mp) v += mp[k];
        const float dlt = v - m;
        q += dlt * dlt;
    }
    const float var = q / (float)(unbiased ? c - 1 : c);
    mean[idx] = m;
    rstd[idx] = 1.0f / sqrtf(var + eps);
}

// c <= LN_REG_C (the 96-channel level of the Kolmogorov net carries most of the LayerNorm traffic): the pixel's channel
// vector stays in registers between the two passes, so x is read from HBM exactly once (the generic kernel re-reads it;
// a 256-pixel workgroup's 96 KB slice does not survive in L2 next to 255 other workgroups').
#define LN_REG_C 96
// SPLIT adjacent lanes share a pixel (channels k = sub + SPLIT j each), so c <= SPLIT * LN_REG_C: 192- and 384-channel levels too.
template <int SPLIT>
__global__ __launch_bounds__(LN_THREADS) void ln_stats_reg_kernel(const float* __restrict__ x, int64_t npix, int c, int hw,
                                                                  const float* __restrict__ mod, int64_t mod_sn, float eps,
                                                                  int unbiased, float* __restrict__ mean,
                                                                  float* __restrict__ rstd) {
    const int64_t gt = (int64_t)blockIdx.x * LN_THREADS + threadIdx.x;
    const int64_t idx = gt / SPLIT;
    const int sub = (int)(gt - idx * SPLIT);
    const bool live = idx < npix;                       // (whole lane groups: no lane of a live group exits before the shuffles)
    const int64_t ii = live ? idx : 0;
    const int64_t n = ii / hw;
    const int p = (int)(ii - n * hw);
    const float* xp = x + n * (int64_t)c * hw + p;
    const float* mp = mod ? mod + n * mod_sn : nullptr;
    float v[LN_REG_C];
#pragma unroll
    for (int j = 0; j < LN_REG_C; ++j) {
        const int k = sub + SPLIT * j;
        v[j] = k < c ? xp[(int64_t)k * hw] : 0.f;
    }
    if (mp) {
#pragma unroll
        for (int j = 0; j < LN_REG_C; ++j) {
            const int k = sub + SPLIT * j;
            if (k < c) v[j] += mp[k];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_REG_C; ++j) s += v[j];
#pragma unroll
    for (int o = 1; o < SPLIT; o <<= 1) s += __shfl_xor(s, o, 64);
    const float m = s / (float)c;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_REG_C; ++j) {
        const float dlt = v[j] - m;
        q += (sub + SPLIT * j) < c ? dlt * dlt : 0.f;
    }
#pragma unroll
    for (int o = 1; o < SPLIT; o <<= 1) q += __shfl_xor(q, o, 64);
    if (live && sub == 0) {
        mean[idx] = m;
        rstd[idx] = 1.0f / sqrtf(q / (float)(unbiased ? c - 1 : c) + eps);
    }
}

// hw % 4 == 0, c <= SPLIT * CPL: a lane owns FOUR consecutive pixels (16-byte loads) and every SPLIT-th channel, the SPLIT lanes
// of a quad sit 64 / SPLIT apart (see ln_bwd_quad_kernel); the channel vectors stay in registers between the mean and the
// centred pass.
// NW > 1: NW wavefronts of a workgroup share a quad group, each with its own slice of SPLIT * CPL channels (c <= NW * SPLIT * CPL), and
// combine their partial sums through LDS -- the 384-channel level keeps the 128-byte runs and the register footprint of the 96-channel
// layout (8 lanes x 12 channels, four wavefronts) instead of 24 channels per lane on 64-byte runs.
template <int SPLIT, int CPL, int NW = 1>
__global__ __launch_bounds__(LN_THREADS) void ln_stats_quad_kernel(const float* __restrict__ x, int64_t nquad, int c, int hw,
                                                                   const float* __restrict__ mod, int64_t mod_sn, float eps,
                                                                   int unbiased, float* __restrict__ mean,
                                                                   float* __restrict__ rstd) {
    constexpr int QW = 64 / SPLIT;
    constexpr int WPB = LN_THREADS / 64;
    static_assert(WPB % NW == 0, "cooperating wavefronts must tile the workgroup");
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6, cw = wib % NW;
    const int64_t wave = NW > 1 ? (int64_t)blockIdx.x * (WPB / NW) + wib / NW : ((int64_t)blockIdx.x * LN_THREADS + threadIdx.x) >> 6;
    const int64_t quad = wave * QW + (lane & (QW - 1));
    const int sub = cw * (SPLIT * CPL) + lane / QW;          // first channel of this lane
    __shared__ float4 part[2][NW > 1 ? WPB : 1][QW];
    const bool live = quad < nquad;
    const int64_t pix = (live ? quad : 0) * 4;
    const int64_t n = pix / hw;
    const int p = (int)(pix - n * hw);
    const float* xp = x + n * (int64_t)c * hw + p;
    const float* mp = mod ? mod + n * mod_sn : nullptr;
    float4 v[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int k = sub + SPLIT * j;
        v[j] = ln_ld4<SPLIT == 8>(xp + (int64_t)(k < c ? k : 0) * hw);
        if (k >= c) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (mp) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int k = sub + SPLIT * j;
            const float mv = k < c ? mp[k] : 0.f;
            v[j].x += mv; v[j].y += mv; v[j].z += mv; v[j].w += mv;
        }
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < CPL; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
#pragma unroll
    for (int o = QW; o < 64; o <<= 1) {
        s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64); s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
    }
    if constexpr (NW > 1) {
        if (lane < QW) part[0][wib][lane] = s;
        __syncthreads();
        s = part[0][wib - cw][lane & (QW - 1)];
#pragma unroll
        for (int i = 1; i < NW; ++i) { const float4 t = part[0][wib - cw + i][lane & (QW - 1)]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    }
    const float ic = 1.f / (float)c;
    const float4 m = make_float4(s.x * ic, s.y * ic, s.z * ic, s.w * ic);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        if (sub + SPLIT * j < c) {
            const float a = v[j].x - m.x, b = v[j].y - m.y, cc = v[j].z - m.z, dd = v[j].w - m.w;
            q.x += a * a; q.y += b * b; q.z += cc * cc; q.w += dd * dd;
        }
    }
#pragma unroll
    for (int o = QW; o < 64; o <<= 1) {
        q.x += __shfl_xor(q.x, o, 64); q.y += __shfl_xor(q.y, o, 64); q.z += __shfl_xor(q.z, o, 64); q.w += __shfl_xor(q.w, o, 64);
    }
    if constexpr (NW > 1) {
        if (lane < QW) part[1][wib][lane] = q;
        __syncthreads();
        q = part[1][wib - cw][lane & (QW - 1)];
#pragma unroll
        for (int i = 1; i < NW; ++i) { const float4 t = part[1][wib - cw + i][lane & (QW - 1)]; q.x += t.x; q.y += t.y; q.z += t.z; q.w += t.w; }
    }
    if (live && sub == 0) {
        const float iv = 1.f / (float)(unbiased ? c - 1 : c);
        *reinterpret_cast<float4*>(mean + pix) = m;
        *reinterpret_cast<float4*>(rstd + pix) = make_float4(1.0f / sqrtf(q.x * iv + eps), 1.0f / sqrtf(q.y * iv + eps),
                                                             1.0f / sqrtf(q.z * iv + eps), 1.0f / sqrtf(q.w * iv + eps));
    }
}

// Few pixels (the 1-D Lorenz nets: n*hw in the hundreds): one thread per pixel would leave the chip idle and serialise a
// dependent load per channel, so one WAVEFRONT takes a pixel, its 64 lanes stride the channel axis and reduce with shuffles.
__global__ __launch_bounds__(LN_THREADS) void ln_stats_wave_kernel(const float* __restrict__ x, int64_t npix, int c, int hw,
                                                                   const float* __restrict__ mod, int64_t mod_sn, float eps,
                                                                   int unbiased, float* __restrict__ mean,
                                                                   float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63;
    const int64_t idx = ((int64_t)blockIdx.x * LN_THREADS + threadIdx.x) >> 6;
    if (idx >= npix) return;
    const int64_t n = idx / hw;
    const int p = (int)(idx - n * hw);
    const float* xp = x + n * (int64_t)c * hw + p;
    const float* mp = mod ? mod + n * mod_sn : nullptr;
    float s = 0.f;
    for (int k = lane; k < c; k += 64) s += xp[(int64_t)k * hw] + (mp ? mp[k] : 0.f);
    const float m = __shfl(sda_wave_sum(s), 0, 64) / (float)c;
    float q = 0.f;
    for (int k = lane; k < c; k += 64) {
        const float dlt = xp[(int64_t)k * hw] + (mp ? mp[k] : 0.f) - m;
        q += dlt * dlt;
    }
    const float var = __shfl(sda_wave_sum(q), 0, 64) / (float)(unbiased ? c - 1 : c);
    if (lane == 0) { mean[idx] = m; rstd[idx] = 1.0f / sqrtf(var + eps); }
}

// below this many pixels the wave-per-pixel kernels are used (measured on the Lorenz-96 net, 8192 pixels x 64 channels:
// statistics 6.4 us wave-per-pixel vs 9.5 us register kernel; backward 15 us wave-per-pixel vs 6.2 us quad kernel)
#define LN_SMALL_PIXELS 16384
#define LN_BWD_SMALL_PIXELS 4096

extern "C" int sda_ln_stats(const float* x, int n, int c, int hw, const float* mod, int64_t mod_sn, float eps,
                            int unbiased, float* mean, float* rstd, void* stream) {
    if (!x || !mean || !rstd || n <= 0 || c <= 0 || hw <= 0) return SDA_E_BADARG;
    if (unbiased && c < 2) return SDA_E_UNSUPPORTED;
    const int64_t npix = (int64_t)n * hw;
    if (npix < LN_SMALL_PIXELS) {
        hipLaunchKernelGGL(ln_stats_wave_kernel, dim3((unsigned)((npix + 3) / 4)), dim3(LN_THREADS), 0, (hipStream_t)stream, x,
                           npix, c, hw, mod, mod_sn, eps, unbiased, mean, rstd);
        return sda_launch_status();
    }
    const int64_t blocks = (npix + LN_THREADS - 1) / LN_THREADS;
    if (blocks > 0x7fffffffLL) return SDA_E_UNSUPPORTED;
    static const int quad_mode = getenv("SDA_LN_STATS_QUAD") ? atoi(getenv("SDA_LN_STATS_QUAD")) : 1;
    if (quad_mode && hw % 4 == 0 && c > 48 && c <= 384 &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd)) & 15) == 0) {
        const int64_t nquad = npix / 4;
        hipStream_t st = (hipStream_t)stream;
        dim3 bl(LN_THREADS);
        // round 6: exact-fit layouts for the 64 / 128 / 256-channel levels of the reference's default widths (the 96 / 192 / 384 layouts
        // served them with a third of every lane's channel slots empty -- loads of channel 0 that are thrown away).  SDA_LN_FIT=0: A/B
        static const bool fit = !(getenv("SDA_LN_FIT") && atoi(getenv("SDA_LN_FIT")) == 0);
        if (fit && c <= 64) hipLaunchKernelGGL((ln_stats_quad_kernel<8, 8>), dim3((unsigned)((nquad + 31) / 32)), bl, 0, st, x, nquad, c, hw, mod, mod_sn, eps, unbiased, mean, rstd);
        else if (fit && c > 96 && c <= 128 && quad_mode != 3 && quad_mode != 4) hipLaunchKernelGGL((ln_stats_quad_kernel<8, 16>), dim3((unsigned)((nquad + 31) / 32)), bl, 0, st, x, nquad, c, hw, mod, mod_sn, eps, unbiased, mean, rstd);
        else if (fit && c > 192 && c <= 256 && quad_mode != 3) hipLaunchKernelGGL((ln_stats_quad_kernel<8, 8, 4>), dim3((unsigned)((nquad + 7) / 8)), bl, 0, st, x, nquad, c, hw, mod, mod_sn, eps, unbiased, mean, rstd);
        else if (c <= 96) hipLaunchKernelGGL((ln_stats_quad_kernel<8, 12>), dim3((unsigned)((nquad + 31) / 32)), bl, 0, st, x, nquad, c, hw, mod, mod_sn, eps, unbiased, mean, rstd);
        // 192 channels as 8 lanes x 24 channels (128-byte runs, non-temporal): 0.298 -> 0.233 ms at 120 windows, 5.1 -> 6.5 TB/s
        // (SDA_LN_STATS_QUAD=3: the 16 x 12 layout, for A/B); 384 channels as 8 x 48 (297 registers) lose: 4.65 vs 5.1 TB/s
        else if (c <= 192 && quad_mode == 4) hipLaunchKernelGGL((ln_stats_quad_kernel<8, 12, 2>), dim3((unsigned)((nquad + 15) / 16)), bl, 0, st, x, nquad, c, hw, mod, mod_sn, eps, unbiased, mean, rstd);
        // 384 channels: four cooperating wavefronts on the 96-channel layout, partial sums through LDS: 0.148 -> 0.132 ms, 5.1 -> 5.7 TB/s
        // (for 192 channels two cooperating wavefronts measure 6.3 TB/s against 8 x 24's 6.5; ln_bwd gains nothing from either: 5.2 / 5.75)
        else if (c > 192 && quad_mode != 3) hipLaunchKernelGGL((ln_stats_quad_kernel<8, 12, 4>), dim3((unsigned)((nquad + 7) / 8)), bl, 0, st, x, nquad, c, hw, mod, mod_sn, eps, unbiased, mean, rstd);
        else if (c <= 192 && quad_mode != 3) hipLaunchKernelGGL((ln_stats_quad_kernel<8, 24>), dim3((unsigned)((nquad + 31) / 32)), bl, 0, st, x, nquad, c, hw, mod, mod_sn, eps, unbiased, mean, rstd);
        else if (c <= 192) hipLaunchKernelGGL((ln_stats_quad_kernel<16, 12>), dim3((unsigned)((nquad + 15) / 16)), bl, 0, st, x, nquad, c, hw, mod, mod_sn, eps, unbiased, mean, rstd);
        else hipLaunchKernelGGL((ln_stats_quad_kernel<16, 24>), dim3((unsigned)((nquad + 15) / 16)), bl, 0, st, x, nquad, c, hw, mod, mod_sn, eps, unbiased, mean, rstd);
        return sda_launch_status();
    }
    if (c > LN_REG_C / 2 && c <= 4 * LN_REG_C && blocks * 4 <= 0x7fffffffLL) {
        const int split = c <= LN_REG_C ? 1 : (c <= 2 * LN_REG_C ? 2 : 4);
        dim3 gr((unsigned)((npix * split + LN_THREADS - 1) / LN_THREADS)), bl(LN_THREADS);
        hipStream_t st = (hipStream_t)stream;
        if (split == 1) hipLaunchKernelGGL(ln_stats_reg_kernel<1>, gr, bl, 0, st, x, npix, c, hw, mod, mod_sn, eps, unbiased, mean, rstd);
        else if (split == 2) hipLaunchKernelGGL(ln_stats_reg_kernel<2>, gr, bl, 0, st, x, npix, c, hw, mod, mod_sn, eps, unbiased, mean, rstd);
        else hipLaunchKernelGGL(ln_stats_reg_kernel<4>, gr, bl, 0, st, x, npix, c, hw, mod, mod_sn, eps, unbiased, mean, rstd);
        return sda_launch_status();
    }
    hipLaunchKernelGGL(ln_stats_kernel, dim3((unsigned)blocks), dim3(LN_THREADS), 0, (hipStream_t)stream, x, npix, c, hw,
                       mod, mod_sn, eps, unbiased, mean, rstd);
    return sda_launch_status();
}

__global__ __launch_bounds__(LN_THREADS) void ln_apply_kernel(const float* __restrict__ x, int64_t npix, int c, int hw,
                                                              const float* __restrict__ mod, int64_t mod_sn,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, float* __restrict__ y) {
    const int64_t idx = (int64_t)blockIdx.x * LN_THREADS + threadIdx.x;
    if (idx >= npix) return;
    const int64_t n = idx / hw;
    const int p = (int)(idx - n * hw);
    const int64_t base = n * (int64_t)c * hw + p;
    const float* mp = mod ? mod + n * mod_sn : nullptr;
    const float m = mean[idx], r = rstd[idx];
    for (int k = 0; k < c; ++k) {
        float v = x[base + (int64_t)k * hw];
        if (mp) v += mp[k];
        y[base + (int64_t)k * hw] = (v - m) * r;
    }
}

extern "C" int sda_ln_apply(const float* x, int n, int c, int hw, const float* mod, int64_t mod_sn, const float* mean,
                            const float* rstd, float* y, void* stream) {
    if (!x || !mean || !rstd || !y || n <= 0 || c <= 0 || hw <= 0) return SDA_E_BADARG;
    const int64_t npix = (int64_t)n * hw;
    const int64_t blocks = (npix + LN_THREADS - 1) / LN_THREADS;
    if (blocks > 0x7fffffffLL) return SDA_E_UNSUPPORTED;
    hipLaunchKernelGGL(ln_apply_kernel, dim3((unsigned)blocks), dim3(LN_THREADS), 0, (hipStream_t)stream, x, npix, c, hw,
                       mod, mod_sn, mean, rstd, y);
    return sda_launch_status();
}

// Backward of h = (u - mean(u)) * rstd, u = x + mod, w.r.t. x (channel axis, per pixel):
//   gx_j = rstd * ( gh_j - mean_c(gh) - h_j * sum_c(gh * h) / (c-1 | c) )      (+ res_j)
// POOL: gh lives at (2h x 2w) [or (h x 2w) for 1-D nets] and is summed over each 2x2 (1x2) cell first
// (= backward of nn.Upsample(nearest), sda/nn.py:164).
template <int POOL_H, int POOL_W>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const float* __restrict__ gh, const float* __restrict__ x,
                                                            int64_t npix, int c, int h, int w,
                                                            const float* __restrict__ mod, int64_t mod_sn,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int unbiased,
                                                            const float* __restrict__ res, float* __restrict__ gx) {
    const int64_t idx = (int64_t)blockIdx.x * LN_THREADS + threadIdx.x;
    if (idx >= npix) return;
    const int hw = h * w;
    const int64_t n = idx / hw;
    const int p = (int)(idx - n * hw);
    const int py = p / w, px = p - py * w;
    const int64_t xbase = n * (int64_t)c * hw + p;
    const int gw = w * POOL_W;
    const int64_t ghw = (int64_t)hw * POOL_H * POOL_W;
    const int64_t gbase = n * (int64_t)c * ghw + (int64_t)(py * POOL_H) * gw + px * POOL_W;
    const float* mp = mod ? mod + n * mod_sn : nullptr;
    const float m = mean[idx], r = rstd[idx];

    auto load_g = [&](int k) -> float {
        const float* g = gh + gbase + (int64_t)k * ghw;
        float v = g[0];
        if (POOL_W == 2) v += g[1];
        if (POOL_H == 2) { v += g[gw]; if (POOL_W == 2) v += g[gw + 1]; }
        return v;
    };

    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < c; ++k) {
        float u = x[xbase + (int64_t)k * hw];
        if (mp) u += mp[k];
        const float hh = (u - m) * r;
        const float g = load_g(k);
        s1 += g;
        s2 += g * hh;
    }
    const float a = s1 / (float)c;
    const float b = s2 / (float)(unbiased ? c - 1 : c);
    for (int k = 0; k < c; ++k) {
        float u = x[xbase + (int64_t)k * hw];
        if (mp) u += mp[k];
        const float hh = (u - m) * r;
        float v = r * (load_g(k) - a - hh * b);
        if (res) v += res[xbase + (int64_t)k * hw];
        gx[xbase + (int64_t)k * hw] = v;
    }
}

// No pooling, c <= SPLIT * LN_BWD_CPL: SPLIT adjacent lanes share a pixel (channels k = sub + SPLIT j each) and keep their
// gh and normalised activations in LN_BWD_CPL registers apiece between the reduction and the update -- gh, x, res are read
// once and gx written once (4 streams instead of the generic kernel's 6) at full occupancy (a one-lane-per-pixel register
// kernel needs 214-256 VGPRs and measured 1.5-2x SLOWER than the generic loop).
template <int SPLIT, int LN_BWD_CPL, int POOL = 1>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_split_kernel(const float* __restrict__ gh, const float* __restrict__ x,
                                                                  int64_t npix, int c, int hw, int w,
                                                                  const float* __restrict__ mod, int64_t mod_sn,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, int unbiased,
                                                                  const float* __restrict__ res, float* __restrict__ gx) {
    const int64_t gt = (int64_t)blockIdx.x * LN_THREADS + threadIdx.x;
    const int64_t idx = gt / SPLIT;
    const int sub = (int)(gt - idx * SPLIT);
    const bool live = idx < npix;
    const int64_t ii = live ? idx : 0;
    const int64_t n = ii / hw;
    const int p = (int)(ii - n * hw);
    const int64_t base = n * (int64_t)c * hw + p;
    const float* mp = mod ? mod + n * mod_sn : nullptr;
    const float m = mean[ii], r = rstd[ii];
    const int py = p / w, px = p - py * w;
    const int64_t gbase = n * (int64_t)c * (4 * (int64_t)hw) + (int64_t)(2 * py) * (2 * w) + 2 * px;     // (POOL == 2 only)
    float g[LN_BWD_CPL], hh[LN_BWD_CPL];
#pragma unroll
    for (int j = 0; j < LN_BWD_CPL; ++j) {
        const int k = sub + SPLIT * j;
        if (POOL == 1) {
            g[j] = k < c ? gh[base + (int64_t)k * hw] : 0.f;
        } else {                                        // gh at 2x resolution: sum of the 2x2 cell (backward of nearest upsample)
            const float* gp = gh + gbase + (int64_t)k * (4 * (int64_t)hw);
            g[j] = k < c ? (gp[0] + gp[1]) + (gp[2 * w] + gp[2 * w + 1]) : 0.f;
        }
        hh[j] = k < c ? x[base + (int64_t)k * hw] : 0.f;
    }
    if (mp) {
#pragma unroll
        for (int j = 0; j < LN_BWD_CPL; ++j) {
            const int k = sub + SPLIT * j;
            if (k < c) hh[j] += mp[k];
        }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_BWD_CPL; ++j) {
        hh[j] = (sub + SPLIT * j) < c ? (hh[j] - m) * r : 0.f;
        s1 += g[j];
        s2 += g[j] * hh[j];
    }
#pragma unroll
    for (int o = 1; o < SPLIT; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    const float a = s1 / (float)c;
    const float b = s2 / (float)(unbiased ? c - 1 : c);
    if (!live) return;
#pragma unroll
    for (int j = 0; j < LN_BWD_CPL; ++j) {
        const int k = sub + SPLIT * j;
        if (k < c) {
            float v = r * (g[j] - a - hh[j] * b);
            if (res) v += res[base + (int64_t)k * hw];
            gx[base + (int64_t)k * hw] = v;
        }
    }
}

// No pooling, hw % 4 == 0, c <= SPLIT * CPL: a lane owns FOUR consecutive pixels (16-byte loads and stores) and every
// SPLIT-th channel; a wavefront covers 64 / SPLIT pixel quads, so each channel plane is touched in contiguous runs of
// (64 / SPLIT) * 16 bytes -- a whole 128-byte line for SPLIT = 8 (the dword version above moves 64-byte runs with four times
// the instructions).  The SPLIT lanes of a quad sit 64 / SPLIT apart and combine their partial sums with cross-lane adds.
template <int SPLIT, int CPL, int POOL = 1, int NW = 1>   // POOL == 2: gh at twice the resolution, summed over 2 x 2 cells (w = row length); NW: see ln_stats_quad_kernel
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_quad_kernel(const float* __restrict__ gh, const float* __restrict__ x,
                                                                 int64_t nquad, int c, int hw, int w, const float* __restrict__ mod,
                                                                 int64_t mod_sn, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, int unbiased,
                                                                 const float* __restrict__ res, float* __restrict__ gx,
                                                                 float* __restrict__ amax) {
    constexpr int QW = 64 / SPLIT;                          // quads per wavefront
    constexpr int WPB = LN_THREADS / 64;
    static_assert(WPB % NW == 0, "cooperating wavefronts must tile the workgroup");
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6, cw = wib % NW;
    const int64_t wave = NW > 1 ? (int64_t)blockIdx.x * (WPB / NW) + wib / NW : ((int64_t)blockIdx.x * LN_THREADS + threadIdx.x) >> 6;
    const int64_t quad = wave * QW + (lane & (QW - 1));
    const int sub = cw * (SPLIT * CPL) + lane / QW;          // first channel of this lane
    __shared__ float4 part[2][NW > 1 ? WPB : 1][QW];
    const bool live = quad < nquad;
    const int64_t pix = (live ? quad : 0) * 4;
    const int64_t n = pix / hw;
    const int p = (int)(pix - n * hw);
    const int64_t base = n * (int64_t)c * hw + p;
    const float* mp = mod ? mod + n * mod_sn : nullptr;
    const float4 m4 = *reinterpret_cast<const float4*>(mean + pix), r4 = *reinterpret_cast<const float4*>(rstd + pix);
    float4 g[CPL], hh[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int k = sub + SPLIT * j;
        const bool on = k < c;
        const int64_t off = base + (int64_t)(on ? k : 0) * hw;
        if (POOL == 1) {
            g[j] = ln_ld4<SPLIT == 8>(gh + off);
        } else {
            const int py = p / w, px = p - py * w;
            const float* gp = gh + (n * (int64_t)c + (on ? k : 0)) * (4 * (int64_t)hw) + (int64_t)(2 * py) * (2 * w) + 2 * px;
            const float4 a0 = *reinterpret_cast<const float4*>(gp), a1 = *reinterpret_cast<const float4*>(gp + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(gp + 2 * w), b1 = *reinterpret_cast<const float4*>(gp + 2 * w + 4);
            g[j] = make_float4((a0.x + a0.y) + (b0.x + b0.y), (a0.z + a0.w) + (b0.z + b0.w), (a1.x + a1.y) + (b1.x + b1.y),
                               (a1.z + a1.w) + (b1.z + b1.w));
        }
        hh[j] = ln_ld4<SPLIT == 8>(x + off);
        if (!on) { g[j] = make_float4(0.f, 0.f, 0.f, 0.f); hh[j] = make_float4(m4.x, m4.y, m4.z, m4.w); }
    }
    // the residual stream's loads go out with the other two streams' (a third more bytes in flight per wavefront), not behind
    // the cross-lane reduction: 96 channels at 256^2 5.57 -> 5.86 TB/s, 384 channels at 64^2 4.05 -> 5.12 (tools/ln_bench.py, 120 windows)
    float4 rr_[CPL];
    if (res) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int k = sub + SPLIT * j;
            rr_[j] = ln_ld4<SPLIT == 8>(res + base + (int64_t)(k < c ? k : 0) * hw);
        }
    }
    if (mp) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int k = sub + SPLIT * j;
            const float mv = k < c ? mp[k] : 0.f;
            hh[j].x += mv; hh[j].y += mv; hh[j].z += mv; hh[j].w += mv;
        }
    }
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        hh[j].x = (hh[j].x - m4.x) * r4.x; hh[j].y = (hh[j].y - m4.y) * r4.y;
        hh[j].z = (hh[j].z - m4.z) * r4.z; hh[j].w = (hh[j].w - m4.w) * r4.w;
        s1.x += g[j].x; s1.y += g[j].y; s1.z += g[j].z; s1.w += g[j].w;
        s2.x += g[j].x * hh[j].x; s2.y += g[j].y * hh[j].y; s2.z += g[j].z * hh[j].z; s2.w += g[j].w * hh[j].w;
    }
#pragma unroll
    for (int o = QW; o < 64; o <<= 1) {
        s1.x += __shfl_xor(s1.x, o, 64); s1.y += __shfl_xor(s1.y, o, 64); s1.z += __shfl_xor(s1.z, o, 64); s1.w += __shfl_xor(s1.w, o, 64);
        s2.x += __shfl_xor(s2.x, o, 64); s2.y += __shfl_xor(s2.y, o, 64); s2.z += __shfl_xor(s2.z, o, 64); s2.w += __shfl_xor(s2.w, o, 64);
    }
    if constexpr (NW > 1) {
        if (lane < QW) { part[0][wib][lane] = s1; part[1][wib][lane] = s2; }
        __syncthreads();
        s1 = part[0][wib - cw][lane & (QW - 1)]; s2 = part[1][wib - cw][lane & (QW - 1)];
#pragma unroll
        for (int i = 1; i < NW; ++i) {
            const float4 t1 = part[0][wib - cw + i][lane & (QW - 1)], t2 = part[1][wib - cw + i][lane & (QW - 1)];
            s1.x += t1.x; s1.y += t1.y; s1.z += t1.z; s1.w += t1.w;
            s2.x += t2.x; s2.y += t2.y; s2.z += t2.z; s2.w += t2.w;
        }
    }
    if (!live && !amax) return;
    float am = 0.f;
    const float ia = 1.f / (float)c, ib = 1.f / (float)(unbiased ? c - 1 : c);
    const float4 a = make_float4(s1.x * ia, s1.y * ia, s1.z * ia, s1.w * ia);
    const float4 b = make_float4(s2.x * ib, s2.y * ib, s2.z * ib, s2.w * ib);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int k = sub + SPLIT * j;
        if (k < c && live) {
            const int64_t off = base + (int64_t)k * hw;
            float4 v = make_float4(r4.x * (g[j].x - a.x - hh[j].x * b.x), r4.y * (g[j].y - a.y - hh[j].y * b.y),
                                   r4.z * (g[j].z - a.z - hh[j].z * b.z), r4.w * (g[j].w - a.w - hh[j].w * b.w));
            if (res) {
                const float4 rr = rr_[j];
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            ln_st4<SPLIT == 8>(gx + off, v);
            am = fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    }
    // optional: max |gx| of the launch (as uint bits; zeroed by the launcher) -- the input scale of the f16 x 2 convolution that reads gx
    // next (csrc/conv_h2.hip), without a pass over the tensor.  One atomic per wavefront, and only while it would still raise the value.
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
        if (lane == 0 && am > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(am));
    }
}

template <int POOL_H, int POOL_W>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_wave_kernel(const float* __restrict__ gh, const float* __restrict__ x,
                                                                 int64_t npix, int c, int h, int w,
                                                                 const float* __restrict__ mod, int64_t mod_sn,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, int unbiased,
                                                                 const float* __restrict__ res, float* __restrict__ gx) {
    const int lane = threadIdx.x & 63;
    const int64_t idx = ((int64_t)blockIdx.x * LN_THREADS + threadIdx.x) >> 6;
    if (idx >= npix) return;
    const int hw = h * w;
    const int64_t n = idx / hw;
    const int p = (int)(idx - n * hw);
    const int py = p / w, px = p - py * w;
    const int64_t xbase = n * (int64_t)c * hw + p;
    const int gw = w * POOL_W;
    const int64_t ghw = (int64_t)hw * POOL_H * POOL_W;
    const int64_t gbase = n * (int64_t)c * ghw + (int64_t)(py * POOL_H) * gw + px * POOL_W;
    const float* mp = mod ? mod + n * mod_sn : nullptr;
    const float m = mean[idx], r = rstd[idx];
    auto load_g = [&](int k) -> float {
        const float* g = gh + gbase + (int64_t)k * ghw;
        float v = g[0];
        if (POOL_W == 2) v += g[1];
        if (POOL_H == 2) { v += g[gw]; if (POOL_W == 2) v += g[gw + 1]; }
        return v;
    };
    float s1 = 0.f, s2 = 0.f;
    for (int k = lane; k < c; k += 64) {
        const float hh = (x[xbase + (int64_t)k * hw] + (mp ? mp[k] : 0.f) - m) * r;
        const float g = load_g(k);
        s1 += g;
        s2 += g * hh;
    }
    const float a = __shfl(sda_wave_sum(s1), 0, 64) / (float)c;
    const float b = __shfl(sda_wave_sum(s2), 0, 64) / (float)(unbiased ? c - 1 : c);
    for (int k = lane; k < c; k += 64) {
        const float hh = (x[xbase + (int64_t)k * hw] + (mp ? mp[k] : 0.f) - m) * r;
        float v = r * (load_g(k) - a - hh * b);
        if (res) v += res[xbase + (int64_t)k * hw];
        gx[xbase + (int64_t)k * hw] = v;
    }
}

__global__ void ln_amax_zero_kernel(float* __restrict__ amax) { amax[0] = 0.f; }

static int ln_bwd_launch(const float* gh, const float* x, int n, int c, int h, int w, const float* mod, int64_t mod_sn,
                         const float* mean, const float* rstd, int unbiased, int pool_h, int pool_w, const float* res,
                         float* gx, float* amax, bool* amax_served, void* stream);

extern "C" int sda_ln_bwd(const float* gh, const float* x, int n, int c, int h, int w, const float* mod, int64_t mod_sn,
                          const float* mean, const float* rstd, int unbiased, int pool_h, int pool_w, const float* res,
                          float* gx, void* stream) {
    bool served = false;
    return ln_bwd_launch(gh, x, n, c, h, w, mod, mod_sn, mean, rstd, unbiased, pool_h, pool_w, res, gx, nullptr, &served, stream);
}

// sda_ln_bwd that also reports amax[0] = max |gx| (device scalar): in the kernel's own epilogue on the 16-byte layouts of the U-Net
// levels, by an sda_absmax pass over gx otherwise
extern "C" int sda_ln_bwd_amax(const float* gh, const float* x, int n, int c, int h, int w, const float* mod, int64_t mod_sn,
                               const float* mean, const float* rstd, int unbiased, int pool_h, int pool_w, const float* res,
                               float* gx, float* amax, void* stream) {
    if (!amax) return SDA_E_BADARG;
    bool served = false;
    const int rc = ln_bwd_launch(gh, x, n, c, h, w, mod, mod_sn, mean, rstd, unbiased, pool_h, pool_w, res, gx, amax, &served, stream);
    if (rc != SDA_OK || served) return rc;
    if (reinterpret_cast<uintptr_t>(gx) & 15) return SDA_E_UNSUPPORTED;
    return sda_absmax(gx, (int64_t)n * c * h * w, amax, stream);
}

static int ln_bwd_launch(const float* gh, const float* x, int n, int c, int h, int w, const float* mod, int64_t mod_sn,
                         const float* mean, const float* rstd, int unbiased, int pool_h, int pool_w, const float* res,
                         float* gx, float* amax, bool* amax_served, void* stream) {
    if (!gh || !x || !mean || !rstd || !gx || n <= 0 || c <= 0 || h <= 0 || w <= 0) return SDA_E_BADARG;
    // (zeroed by a kernel, not a 4-byte memset: see sda_absmax)
    if (amax) hipLaunchKernelGGL(ln_amax_zero_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, amax);
    const int shape = pool_h * 10 + pool_w;          // 11: no pooling, 12: 1-D nets (length axis only), 22: 2-D nets
    if (shape != 11 && shape != 12 && shape != 22) return SDA_E_UNSUPPORTED;
    const int64_t npix = (int64_t)n * h * w;
    const int64_t blocks = (npix + LN_THREADS - 1) / LN_THREADS;
    if (blocks > 0x7fffffffLL) return SDA_E_UNSUPPORTED;
    dim3 grid((unsigned)blocks), block(LN_THREADS);
    hipStream_t s = (hipStream_t)stream;
    if (npix < LN_BWD_SMALL_PIXELS) {
        dim3 gs((unsigned)((npix + 3) / 4));
        if (shape == 11)
            hipLaunchKernelGGL((ln_bwd_wave_kernel<1, 1>), gs, block, 0, s, gh, x, npix, c, h, w, mod, mod_sn, mean, rstd,
                               unbiased, res, gx);
        else if (shape == 12)
            hipLaunchKernelGGL((ln_bwd_wave_kernel<1, 2>), gs, block, 0, s, gh, x, npix, c, h, w, mod, mod_sn, mean, rstd,
                               unbiased, res, gx);
        else
            hipLaunchKernelGGL((ln_bwd_wave_kernel<2, 2>), gs, block, 0, s, gh, x, npix, c, h, w, mod, mod_sn, mean, rstd,
                               unbiased, res, gx);
        return sda_launch_status();
    }
    const bool quad_ok = shape == 11 && (h * w) % 4 == 0 && c > 48 && c <= 384 &&
                         ((reinterpret_cast<uintptr_t>(gh) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gx) |
                           reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd) |
                           reinterpret_cast<uintptr_t>(res)) & 15) == 0;
    static const int quad_mode = getenv("SDA_LN_BWD_QUAD") ? atoi(getenv("SDA_LN_BWD_QUAD")) : 1;
    if (quad_ok && quad_mode) {
        const int hw = h * w;
        const int64_t nquad = npix / 4;
        // (lanes per quad x channels per lane: c = 96: 8 x 12, 128-byte runs; c = 192 / 384: 16 x 12 / 16 x 24, 64-byte runs)
        static const bool fit = !(getenv("SDA_LN_FIT") && atoi(getenv("SDA_LN_FIT")) == 0);      // (exact-fit layouts: see sda_ln_stats)
        if (fit && c <= 64) {
            dim3 gr((unsigned)((nquad + 31) / 32));
            hipLaunchKernelGGL((ln_bwd_quad_kernel<8, 8>), gr, block, 0, s, gh, x, nquad, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx, amax); *amax_served = amax != nullptr;
        } else if (fit && c > 96 && c <= 128 && quad_mode != 3 && quad_mode != 4) {
            dim3 gr((unsigned)((nquad + 31) / 32));
            hipLaunchKernelGGL((ln_bwd_quad_kernel<8, 16>), gr, block, 0, s, gh, x, nquad, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx, amax); *amax_served = amax != nullptr;
        } else if (fit && c > 192 && c <= 256 && quad_mode != 4) {
            dim3 gr((unsigned)((nquad + 15) / 16));
            hipLaunchKernelGGL((ln_bwd_quad_kernel<16, 16>), gr, block, 0, s, gh, x, nquad, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx, amax); *amax_served = amax != nullptr;
        } else if (c <= 96) {
            dim3 gr((unsigned)((nquad + 31) / 32));         // 8 quads per wavefront, 4 wavefronts per workgroup
            hipLaunchKernelGGL((ln_bwd_quad_kernel<8, 12>), gr, block, 0, s, gh, x, nquad, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx, amax); *amax_served = amax != nullptr;
        } else if (c <= 192 && quad_mode == 4) {            // (A/B: two cooperating wavefronts, 8 lanes x 12 channels each)
            dim3 gr((unsigned)((nquad + 15) / 16));
            hipLaunchKernelGGL((ln_bwd_quad_kernel<8, 12, 1, 2>), gr, block, 0, s, gh, x, nquad, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx, amax); *amax_served = amax != nullptr;
        } else if (c > 192 && quad_mode == 4) {             // (A/B: four cooperating wavefronts)
            dim3 gr((unsigned)((nquad + 7) / 8));
            hipLaunchKernelGGL((ln_bwd_quad_kernel<8, 12, 1, 4>), gr, block, 0, s, gh, x, nquad, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx, amax); *amax_served = amax != nullptr;
        } else if (c <= 192 && quad_mode != 3) {
            // 8 lanes x 24 channels: 128-byte runs like the 96-channel level (one wavefront per SIMD -- 376 registers -- but with the
            // residual loads ahead of the reduction it beats 16 x 12's 64-byte runs: 1.21 -> 1.07 ms at 120 windows, 5.0 -> 5.65 TB/s;
            // before that change it lost, 4.48 vs 4.87.  SDA_LN_BWD_QUAD=3 keeps 16 x 12 for A/B.  384 channels as 32 x 12: 3.8 vs 5.0 TB/s)
            dim3 gr((unsigned)((nquad + 31) / 32));
            hipLaunchKernelGGL((ln_bwd_quad_kernel<8, 24>), gr, block, 0, s, gh, x, nquad, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx, amax); *amax_served = amax != nullptr;
        } else if (c <= 192) {
            dim3 gr((unsigned)((nquad + 15) / 16));
            hipLaunchKernelGGL((ln_bwd_quad_kernel<16, 12>), gr, block, 0, s, gh, x, nquad, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx, amax); *amax_served = amax != nullptr;
        } else {
            dim3 gr((unsigned)((nquad + 15) / 16));
            hipLaunchKernelGGL((ln_bwd_quad_kernel<16, 24>), gr, block, 0, s, gh, x, nquad, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx, amax); *amax_served = amax != nullptr;
        }
    } else if (shape == 11 && c > 48 && c <= 384 && blocks * 8 <= 0x7fffffffLL) {
        // lanes per pixel x channels per lane, picked per width from measurements on the Kolmogorov net's three levels
        // (c = 96: 4 x 24 1.57 ms vs 2 x 48 2.43, 8 x 12 1.85;  c = 192: 8 x 24 0.86 vs 4 x 48 1.13;  c = 384: 8 x 48 0.59 vs 16 x 24 0.82)
        const int hw = h * w;
        const int split = c <= 96 ? 4 : 8;
        dim3 gr((unsigned)((npix * split + LN_THREADS - 1) / LN_THREADS));
        if (c <= 96) hipLaunchKernelGGL((ln_bwd_split_kernel<4, 24>), gr, block, 0, s, gh, x, npix, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx);
        else if (c <= 192) hipLaunchKernelGGL((ln_bwd_split_kernel<8, 24>), gr, block, 0, s, gh, x, npix, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx);
        else hipLaunchKernelGGL((ln_bwd_split_kernel<8, 48>), gr, block, 0, s, gh, x, npix, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx);
    } else if (shape == 22 && quad_mode && w % 4 == 0 && c > 48 && c <= 384 &&
               ((reinterpret_cast<uintptr_t>(gh) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gx) |
                 reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd) | reinterpret_cast<uintptr_t>(res)) & 15) == 0) {
        // (the up-sampling tails: gh at twice the resolution; 2 w floats per row keep every 16-byte load aligned)
        const int hw = h * w;
        const int64_t nquad = npix / 4;
        if (c <= 96) {
            dim3 gr((unsigned)((nquad + 31) / 32));
            hipLaunchKernelGGL((ln_bwd_quad_kernel<8, 12, 2>), gr, block, 0, s, gh, x, nquad, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx, amax); *amax_served = amax != nullptr;
        } else if (c <= 192) {
            dim3 gr((unsigned)((nquad + 15) / 16));
            hipLaunchKernelGGL((ln_bwd_quad_kernel<16, 12, 2>), gr, block, 0, s, gh, x, nquad, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx, amax); *amax_served = amax != nullptr;
        } else {
            dim3 gr((unsigned)((nquad + 15) / 16));
            hipLaunchKernelGGL((ln_bwd_quad_kernel<16, 24, 2>), gr, block, 0, s, gh, x, nquad, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx, amax); *amax_served = amax != nullptr;
        }
    } else if (shape == 22 && c > 48 && c <= 384 && blocks * 8 <= 0x7fffffffLL) {
        const int hw = h * w;
        const int split = c <= 96 ? 4 : 8;
        dim3 gr((unsigned)((npix * split + LN_THREADS - 1) / LN_THREADS));
        if (c <= 96) hipLaunchKernelGGL((ln_bwd_split_kernel<4, 24, 2>), gr, block, 0, s, gh, x, npix, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx);
        else if (c <= 192) hipLaunchKernelGGL((ln_bwd_split_kernel<8, 24, 2>), gr, block, 0, s, gh, x, npix, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx);
        else hipLaunchKernelGGL((ln_bwd_split_kernel<8, 48, 2>), gr, block, 0, s, gh, x, npix, c, hw, w, mod, mod_sn, mean, rstd, unbiased, res, gx);
    } else if (shape == 11) {
        hipLaunchKernelGGL((ln_bwd_kernel<1, 1>), grid, block, 0, s, gh, x, npix, c, h, w, mod, mod_sn, mean, rstd,
                           unbiased, res, gx);
    } else if (shape == 12) {
        hipLaunchKernelGGL((ln_bwd_kernel<1, 2>), grid, block, 0, s, gh, x, npix, c, h, w, mod, mod_sn, mean, rstd,
                           unbiased, res, gx);
    } else {
        hipLaunchKernelGGL((ln_bwd_kernel<2, 2>), grid, block, 0, s, gh, x, npix, c, h, w, mod, mod_sn, mean, rstd,
                           unbiased, res, gx);
    }
    return sda_launch_status();
}
