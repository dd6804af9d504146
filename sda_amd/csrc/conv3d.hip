// 3-D convolutions for `UNet(spatial=3)` (reference: sda/nn.py:114-118 selects nn.Conv3d, nn.py:148-206 builds the same
// heads / blocks / tails around it; ATen does the arithmetic there).  No reference experiment uses a 3-D U-Net, so this is
// the path's general-purpose member, not a tuned one: ONE gather kernel -- an implicit GEMM on the fp32 matrix cores whose B
// operand is read straight from the planar (N, C, D, H, W) tensor -- serves every launch of the forward pass and of the
// input VJP through three per-axis index maps:
//     v = o * stride + tap - pad                (virtual coordinate of the tap under output position o)
//     circular: v mod V        zeros: 0 <= v < V, else the tap contributes nothing
//     up  > 1 : source = v / up                 (nearest-neighbour up-sampling in the loader: the tails, nn.py:161-169)
//     dil > 1 : source = v / dil  iff dil | v   (zero insertion: the transposed stride-`dil` convolution of a head's VJP)
// The transposed convolutions take the flipped, (cin <-> cout)-swapped weights (packed by sda_pack_conv3d_weight).
// Loader: optional activation of the gathered values.  Epilogue: + bias, then either act(.) or x act'(z), then + res.  LayerNorm is applied by sda_ln_apply before the launch and
// differentiated by sda_ln_bwd after it (both treat the three spatial axes as one plane).
//
// Tile: a wavefront owns 16 consecutive output positions x 64 output channels (four 16x16 accumulators);
// v_mfma_f32_16x16x4_f32 with A = packed weights [16 couts x 4 cins] (one coalesced 256-byte read per fragment, L2-resident)
// and B = [4 cins x 16 positions] gathered one float per lane.  Loop order: taps outside (the index maps are evaluated once
// per tap and position), input-channel quads inside.
#include "sda_common.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef C3_U
#define C3_U 2
#endif

struct AxisMap { int in, out, k, pad, stride, up, dil, vext; };

__device__ __forceinline__ int c3_src(const AxisMap& a, int o, int tap, int circular) {
    int v = o * a.stride + tap - a.pad;
    if (circular) { v %= a.vext; if (v < 0) v += a.vext; }
    else if (v < 0 || v >= a.vext) return -1;
    if (a.dil > 1) { if (v % a.dil) return -1; v /= a.dil; }
    else if (a.up > 1) v /= a.up;
    return v < a.in ? v : -1;
}

struct Conv3dArgs {
    const float* x; const float* w; const float* bias; const float* z; const float* res; float* out;
    AxisMap ad, ah, aw;
    int n, cin, cout, ciq, mblocks, circular, act, act_in;
};

__global__ __launch_bounds__(256) void conv3d_kernel(Conv3dArgs g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kg = lane >> 4;
    const int64_t P = (int64_t)g.ad.out * g.ah.out * g.aw.out;
    const int64_t pos = ((int64_t)blockIdx.x * 4 + wave) * 16 + col;
    const bool live = pos < P;
    const int img = blockIdx.z, mb0 = blockIdx.y * 4;
    int ow = 0, oh = 0, od = 0;
    if (live) { ow = (int)(pos % g.aw.out); const int64_t r = pos / g.aw.out; oh = (int)(r % g.ah.out); od = (int)(r / g.ah.out); }
    const int64_t plane = (int64_t)g.ad.in * g.ah.in * g.aw.in;
    const float* xi = g.x + (int64_t)img * g.cin * plane;
    f32x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nmb = min(4, g.mblocks - mb0);
    int tap = 0;
    for (int kd = 0; kd < g.ad.k; ++kd) {
        const int sd = c3_src(g.ad, od, kd, g.circular);
        for (int kh = 0; kh < g.ah.k; ++kh) {
            const int sh = c3_src(g.ah, oh, kh, g.circular);
            for (int kw = 0; kw < g.aw.k; ++kw, ++tap) {
                const int sw = c3_src(g.aw, ow, kw, g.circular);
                const bool ok = live && sd >= 0 && sh >= 0 && sw >= 0;
                // a tap no position of the wavefront sees (zero-inserted source: 7 of 8 taps of a stride-2 layer's VJP;
                // padding at a volume face) multiplies zeros only: skip its whole channel loop
                if (!__builtin_amdgcn_ballot_w64(ok)) continue;
                const float* xs = xi + ((int64_t)sd * g.ah.in + sh) * g.aw.in + sw;
                const float* wt = g.w + ((int64_t)tap * g.ciq * g.mblocks + mb0) * 64 + lane;
                // C3_U input-channel quads per iteration, all their loads issued before the MFMAs (the loop is latency-bound: one
                // wavefront per SIMD or two, every operand a fresh global / L2 round trip)
                for (int q = 0; q < g.ciq; q += C3_U) {
                    float b[C3_U], a[C3_U][4];
#pragma unroll
                    for (int u = 0; u < C3_U; ++u) {
                        const bool have = q + u < g.ciq;
                        const int ci = (q + u) * 4 + kg;
                        b[u] = (ok && have && ci < g.cin) ? xs[(int64_t)ci * plane] : 0.f;
                        const float* wq = wt + (int64_t)(have ? q + u : q) * g.mblocks * 64;
#pragma unroll
                        for (int m = 0; m < 4; ++m) a[u][m] = m < nmb ? wq[m * 64] : 0.f;
                    }
                    if (g.act_in) {
#pragma unroll
                        for (int u = 0; u < C3_U; ++u) b[u] = sda_act(g.act_in, b[u]);       // (act(0) = 0 for every activation)
                    }
#pragma unroll
                    for (int u = 0; u < C3_U; ++u) {
                        if (q + u < g.ciq) {
#pragma unroll
                            for (int m = 0; m < 4; ++m)
                                if (m < nmb) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][m], b[u], acc[m], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    if (!live) return;
    // D fragment: lane holds couts 4 * kg + r (r = 0..3) of its m-block at position `col`
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (m >= nmb) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = (mb0 + m) * 16 + kg * 4 + r;
            if (co >= g.cout) continue;
            const int64_t o = ((int64_t)img * g.cout + co) * P + pos;
            float v = acc[m][r] + (g.bias ? g.bias[co] : 0.f);
            if (g.z) v *= sda_dact(g.act, g.z[o]);
            else v = sda_act(g.act, v);
            if (g.res) v += g.res[o];
            g.out[o] = v;
        }
    }
}

static int c3_axis(AxisMap* a, int in, int out, int k, int pad, int stride, int up, int dil, int circular) {
    if (in <= 0 || out <= 0 || k <= 0 || stride <= 0 || up <= 0 || dil <= 0 || (up > 1 && dil > 1)) return SDA_E_BADARG;
    a->in = in; a->out = out; a->k = k; a->pad = pad; a->stride = stride; a->up = up; a->dil = dil;
    // extent of the virtual signal the taps slide over: up-sampled, or zero-inserted (circular: a whole period)
    a->vext = dil > 1 ? (circular ? in * dil : (in - 1) * dil + 1) : in * up;
    return SDA_OK;
}

extern "C" int sda_conv3d(const sda_conv3d_desc* d, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!d || !d->x || !d->w || !d->out || d->n <= 0 || d->cin <= 0 || d->cout <= 0) return SDA_E_BADARG;
    if (d->z && d->act == SDA_ACT_NONE) return SDA_E_BADARG;
    Conv3dArgs g;
    g.x = d->x; g.w = d->w; g.bias = d->bias; g.z = d->z; g.res = d->res; g.out = d->out;
    int rc;
    if ((rc = c3_axis(&g.ad, d->in_size[0], d->out_size[0], d->k[0], d->pad[0], d->stride[0], d->up[0], d->dil[0], d->circular))) return rc;
    if ((rc = c3_axis(&g.ah, d->in_size[1], d->out_size[1], d->k[1], d->pad[1], d->stride[1], d->up[1], d->dil[1], d->circular))) return rc;
    if ((rc = c3_axis(&g.aw, d->in_size[2], d->out_size[2], d->k[2], d->pad[2], d->stride[2], d->up[2], d->dil[2], d->circular))) return rc;
    g.n = d->n; g.cin = d->cin; g.cout = d->cout; g.ciq = (d->cin + 3) / 4; g.mblocks = (d->cout + 15) / 16;
    g.circular = d->circular ? 1 : 0; g.act = d->act; g.act_in = d->act_in;
    const int64_t P = (int64_t)g.ad.out * g.ah.out * g.aw.out;
    const int64_t gx = (P + 63) / 64;
    if (gx > 0x7fffffff || d->n > 65535 || (g.mblocks + 3) / 4 > 65535) return SDA_E_UNSUPPORTED;
    if ((int64_t)d->cin * g.ad.in * g.ah.in * g.aw.in >= ((int64_t)1 << 40)) return SDA_E_UNSUPPORTED;
    hipLaunchKernelGGL(conv3d_kernel, dim3((unsigned)gx, (unsigned)((g.mblocks + 3) / 4), (unsigned)d->n), dim3(256), 0, stream, g);
    return sda_launch_status();
}

// w (cout, cin, kd, kh, kw) -> the kernel's A fragments [tap][cin quad][cout block of 16][lane]: lane l holds the weight of
// cout = 16 * block + l % 16, cin = 4 * quad + l / 16 (zero beyond the real channels).  transpose != 0 packs the operator of the
// input VJP instead: taps flipped on every axis, cin and cout exchanged.
__global__ void pack3d_kernel(const float* __restrict__ w, int cout, int cin, int kd, int kh, int kw, int transpose,
                              float* __restrict__ dst, int64_t total) {
    const int M = transpose ? cin : cout, K = transpose ? cout : cin;
    const int ciq = (K + 3) / 4, mblocks = (M + 15) / 16, taps = kd * kh * kw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        int64_t r = i >> 6;
        const int mb = (int)(r % mblocks); r /= mblocks;
        const int q = (int)(r % ciq);
        int tap = (int)(r / ciq);
        const int m = mb * 16 + (lane & 15), k = q * 4 + (lane >> 4);
        float v = 0.f;
        if (m < M && k < K) {
            if (transpose) tap = taps - 1 - tap;
            const int co = transpose ? k : m, ci = transpose ? m : k;
            v = w[((int64_t)co * cin + ci) * taps + tap];
        }
        dst[i] = v;
    }
}

extern "C" int64_t sda_conv3d_packed_floats(int cout, int cin, int kd, int kh, int kw, int transpose) {
    const int M = transpose ? cin : cout, K = transpose ? cout : cin;
    return (int64_t)kd * kh * kw * ((K + 3) / 4) * ((M + 15) / 16) * 64;
}

extern "C" int sda_pack_conv3d_weight(const float* w, int cout, int cin, int kd, int kh, int kw, int transpose, float* dst,
                                      void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w || !dst || cout <= 0 || cin <= 0 || kd <= 0 || kh <= 0 || kw <= 0) return SDA_E_BADARG;
    const int64_t total = sda_conv3d_packed_floats(cout, cin, kd, kh, kw, transpose);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(pack3d_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, cout, cin, kd, kh, kw, transpose, dst, total);
    return sda_launch_status();
}

// Adjoint of nearest-neighbour up-sampling by (fd, fh, fw): out[n, c, d, h, w] = sum of the fd x fh x fw cell of g.
__global__ void pool3d_kernel(const float* __restrict__ g, int d, int h, int w, int fd, int fh, int fw, float* __restrict__ out,
                              int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w);
        int64_t r = i / w;
        const int y = (int)(r % h); r /= h;
        const int zc = (int)(r % d);
        const int64_t nc = r / d;
        const float* src = g + ((nc * (d * fd) + (int64_t)zc * fd) * (h * fh) + (int64_t)y * fh) * (w * fw) + (int64_t)x * fw;
        float s = 0.f;
        for (int a = 0; a < fd; ++a)
            for (int b = 0; b < fh; ++b)
                for (int c = 0; c < fw; ++c) s += src[((int64_t)a * (h * fh) + b) * (w * fw) + c];
        out[i] = s;
    }
}

extern "C" int sda_pool3d_sum(const float* g, int64_t nc, int d, int h, int w, int fd, int fh, int fw, float* out,
                              void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!g || !out || nc <= 0 || d <= 0 || h <= 0 || w <= 0 || fd <= 0 || fh <= 0 || fw <= 0) return SDA_E_BADARG;
    const int64_t total = nc * d * h * w;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(pool3d_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, g, d, h, w, fd, fh, fw, out, total);
    return sda_launch_status();
}
