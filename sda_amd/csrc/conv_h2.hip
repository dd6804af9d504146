// conv_h2_kernel -- 3 x 3 stride-1 convolution with the fp32 multiply EMULATED on the f16 matrix cores ("h2": every fp32 value as two
// halves), fp32 accumulation.  OPT-IN (sda_conv_desc.w_h2; the default product path multiplies in fp32 on v_mfma_f32_16x16x4_f32).
//
// Replaces, like csrc/conv_wino4.hip, the block convolutions of the U-Net and their backward-data (sda/nn.py:131-142; the gradient
// torch.autograd propagates through them at sda/score.py:394): 36 of the 42 convolutions of the reference's Kolmogorov net.
//
// Arithmetic.  x = hi + lo with hi = f16(s x), lo = f16(s x - hi) (the difference is exact in fp32; s = one power of two per tensor
// that puts max |x| at 2^11, so that hi never overflows and the low piece's fp16 subnormal spacing, 2^-24, is 2^-35 of the tensor's
// scale): 22 significand bits in 4 bytes -- the operand bytes of fp32.  A product of two halves is exact in fp32, and
//     x w  ~  hi_x hi_w + hi_x lo_w + lo_x hi_w          (dropped: lo_x lo_w, relative 2^-22)
// is three v_mfma_f32_16x16x32_f16 per 16 couts x 16 pixels x 32 channels where the fp32 pipe needs eight v_mfma_f32_16x16x4_f32: at the
// measured issue rates (17 against 32 cycles per instruction and SIMD, MI355X_MICROARCH.md) 51 cycles against 256.  That is 5x per
// multiply -- more than Winograd F(2x2,3x3) saves (2.25x), so this kernel is a DIRECT convolution: no transforms, no transform-domain
// round-off, no helper-wave arithmetic.  Measured against float64 (tools/f16_split_numerics.py): the same 3e-7 relative error as the
// fp32 Winograd kernel.
//
// Structure (one workgroup = 4 waves, one per SIMD, up to 512 registers each; tile = 96 couts x 16 x 16 pixels of one image):
//   * K loop over chunks of 32 input channels.  The chunk's 18 x 18 halo tile lives in LDS as [pixel][hi: 32 halves | lo: 32 halves]
//     (+ 32 B pad: 160 B per pixel makes the ds_read_b128 of 16 consecutive pixels conflict-free), double buffered, ONE barrier per
//     chunk.  Every wave fills its quarter of the next chunk's tile while it multiplies the current one: wave w owns channels 8 w .. 8 w + 7
//     -- global loads (padding / wrap resolved once per tile), the loader fusions of the reference's blocks (time modulation +
//     LayerNorm, or the activation; sda/nn.py:28,137-139), the split, two 16-byte LDS stores per pixel.  The f16 MFMA leaves three
//     issue slots per instruction free, and the loader needs ~330 of a chunk's ~1 900.
//   * per tap and chunk a wave multiplies 6 cout fragments x 4 pixel rows x 3 products = 72 MFMAs.  B: eight ds_read_b128 (the tap is
//     a pixel offset into the halo tile).  A: twelve global_load_dwordx4 straight from the packed weights (sda_pack_conv_weight_h2:
//     fragments in lane order, 12 KiB per tap and chunk, the same addresses in all four waves -- L1 hits for three of them), one tap
//     ahead.
//   * epilogue: x 1 / (s_x s_w), + bias, x act'(z) or + residual, 64-byte row segments; optionally max |out| (one atomic per wave)
//     so that the NEXT h2 launch knows its input scale without a pass over the tensor.
// Roofline: f16 matrix pipe (2.5 PFLOP/s dense / 3 products); algorithmic bytes: x once per 96-cout tile x 1.27 (halo), out once.
#include "sda_common.hpp"
#include <stdlib.h>

typedef _Float16 h2_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2_h2 __attribute__((ext_vector_type(2)));
typedef float h2_f4 __attribute__((ext_vector_type(4)));

#define H2_TS 16                       // tile side (pixels)
#define H2_HS (H2_TS + 2)              // halo side
#define H2_NPX (H2_HS * H2_HS)         // 324 halo pixels
#define H2_PXB 160                     // bytes per halo pixel in LDS: 64 hi + 64 lo + 32 pad
#define H2_TILE (H2_NPX * H2_PXB)      // 51 840 B
#define H2_CK 32                       // channels per chunk
#define H2_BM 96                       // couts per workgroup
#define H2_RND ((H2_NPX + 63) / 64)    // 6 loader rounds per wave and chunk
#define H2_TARGET_EXP 11               // max |s x| in [2^10, 2^11]

// power-of-two scale that puts `amax` into [2^(T-1), 2^T]  (amax = 0 / denormal: 1)
__host__ __device__ __forceinline__ float h2_scale_of(float amax) {
    union { float f; uint32_t u; } v;
    v.f = amax;
    const int e = (int)((v.u >> 23) & 0xff);          // biased exponent: amax in [2^(e-127), 2^(e-126))
    if (e == 0 || e == 0xff) return 1.0f;
    int se = 127 + H2_TARGET_EXP - (e - 126);
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    v.u = (uint32_t)se << 23;
    return v.f;
}

struct h2_args {
    const void* w;              // packed halves (sda_pack_conv_weight_h2)
    float w_scale;              // s_w
    const float* x_amax;        // device: max |loader output| (NULL: x_amax_static)
    float x_amax_static;
    float* out_amax;            // device (optional): atomically maxed with |out| as uint bits
    int tiles_x, tiles_y, n_ct, nchunk;
};

template <int LOADER>           // 0 plain, 1 activation (SiLU), 2 LayerNorm (+ optional modulation)
__global__ __launch_bounds__(256) void conv_h2_kernel(const sda_conv_desc d, const h2_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- tile of this workgroup.  Consecutive logical tiles (all cout tiles of a pixel tile, then the row of pixel tiles) go to ONE
    // XCD (blockIdx % 8 is the XCD): they share halo lines and weight fragments in that XCD's L2.
    int t = blockIdx.x;
    {
        const int total = gridDim.x;
        if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);
    }
    const int ct = t % a.n_ct; t /= a.n_ct;
    const int bx = t % a.tiles_x; t /= a.tiles_x;
    const int by = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int oy0 = by * H2_TS, ox0 = bx * H2_TS, co0 = ct * H2_BM;
    const float sx = h2_scale_of(a.x_amax ? a.x_amax[0] : a.x_amax_static);

    // ---- loader plan of this lane (the same for every chunk): halo pixel p = lane + 64 r, channels 8 wave .. + 7 of the chunk
    const float* ximg = d.x + (int64_t)n * d.x_sn_outer + (int64_t)(8 * wave) * d.x_sc;
    int goff[H2_RND];
    unsigned valid = 0;
    float mean[H2_RND], rstd[H2_RND];
#pragma unroll
    for (int r = 0; r < H2_RND; ++r) {
        const int p = lane + 64 * r;
        const int hy = p / H2_HS, hx = p - hy * H2_HS;
        int y = oy0 + hy - 1, x = ox0 + hx - 1;
        bool ok = p < H2_NPX;
        if (d.circular) {
            y = y < 0 ? y + d.hs : (y >= d.hs ? y - d.hs : y);
            x = x < 0 ? x + d.ws : (x >= d.ws ? x - d.ws : x);
        } else {
            ok = ok && y >= 0 && y < d.hs && x >= 0 && x < d.ws;
        }
        y = ok ? y : 0;
        x = ok ? x : 0;
        goff[r] = y * (int)d.x_sy + x * (int)d.x_sx;
        valid |= ok ? (1u << r) : 0u;
        if (LOADER == 2) {
            const int64_t sp = (int64_t)n * d.hs * d.ws + (int64_t)y * d.ws + x;
            mean[r] = d.ln_mean[sp];
            rstd[r] = d.ln_rstd[sp];
        }
    }
    const float* modp = (LOADER == 2 && d.mod) ? d.mod + (int64_t)n * d.mod_sn + 8 * wave : nullptr;
    const int lds_wr = lane * H2_PXB + wave * 16;                   // + 64 r * H2_PXB; + 64 for the low piece

    float raw[2][8];                                                // two rounds in flight
    auto load_round = [&](int chunk, int r, float (&v)[8]) {
        const float* src = ximg + (int64_t)chunk * H2_CK * d.x_sc + goff[r];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = src[(int64_t)i * d.x_sc];
    };
    auto store_round = [&](int chunk, int r, const float (&v)[8], unsigned char* buf) {
        h2_h8 hi, lo;
        const bool ok = (valid >> r) & 1u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float u = v[i];
            if (LOADER == 2) {
                if (modp) u += modp[chunk * H2_CK + i];
                u = (u - mean[r]) * rstd[r];
            }
            if (LOADER == 1) u = sda_act(d.act_in, u);
            u = ok ? u * sx : 0.f;
            const _Float16 h = (_Float16)u;
            hi[i] = h;
            lo[i] = (_Float16)(u - (float)h);
        }
        if (lane + 64 * r < H2_NPX) {
            *reinterpret_cast<h2_h8*>(buf + lds_wr + 64 * r * H2_PXB) = hi;
            *reinterpret_cast<h2_h8*>(buf + lds_wr + 64 * r * H2_PXB + 64) = lo;
        }
    };

    // ---- consumer addressing.  B fragment j of tap (dy, dx): pixels (4 wave + j + dy) * 18 + dx + (lane & 15), channels 8 (lane >> 4) ..
    const int b_rd = ((4 * wave) * H2_HS + (lane & 15)) * H2_PXB + (lane >> 4) * 16;
    // A: [cout tile][chunk][tap][m][piece][lane] x 16 B
    const h2_h8* wq = reinterpret_cast<const h2_h8*>(a.w) + (int64_t)ct * a.nchunk * (9 * 6 * 2 * 64) + lane;

    h2_f4 acc[6][4];
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = h2_f4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: chunk 0 into buffer 0
#pragma unroll
    for (int r = 0; r < H2_RND; ++r) {
        load_round(0, r, raw[0]);
        store_round(0, r, raw[0], smem);
    }
    h2_h8 A[2][6][2], B[2][4][2];
    auto load_A = [&](int chunk, int tap, h2_h8 (&dst)[6][2]) {
        const h2_h8* p = wq + (int64_t)(chunk * 9 + tap) * (6 * 2 * 64);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            dst[m][0] = p[(m * 2 + 0) * 64];
            dst[m][1] = p[(m * 2 + 1) * 64];
        }
    };
    auto load_B = [&](const unsigned char* buf, int tap, h2_h8 (&dst)[4][2]) {
        const int dy = tap / 3, dx = tap - 3 * dy;
        const unsigned char* p = buf + b_rd + (dy * H2_HS + dx) * H2_PXB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dst[j][0] = *reinterpret_cast<const h2_h8*>(p + j * H2_HS * H2_PXB);
            dst[j][1] = *reinterpret_cast<const h2_h8*>(p + j * H2_HS * H2_PXB + 64);
        }
    };
    load_A(0, 0, A[0]);
    __syncthreads();

    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
        const unsigned char* cur = smem + (chunk & 1) * H2_TILE;
        unsigned char* nxt = smem + ((chunk + 1) & 1) * H2_TILE;
        const int cn = chunk + 1 < a.nchunk ? chunk + 1 : chunk;       // (last chunk: reloads itself into the idle buffer; branch-free)
        load_B(cur, 0, B[0]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int s = tap & 1;
            // operands of the next tap (the first tap of the next chunk: A only -- its B tile is complete after the barrier)
            if (tap < 8) {
                load_A(chunk, tap + 1, A[s ^ 1]);
                load_B(cur, tap + 1, B[s ^ 1]);
            } else {
                load_A(cn, 0, A[s ^ 1]);
            }
            // this wave's share of the next chunk's tile: round r is requested in tap r and stored two taps later
            if (tap >= 2 && tap < H2_RND + 2) store_round(cn, tap - 2, raw[tap & 1], nxt);      // (consumes raw[tap & 1] first)
            if (tap < H2_RND) load_round(cn, tap, raw[tap & 1]);
            // small products first (fp32 accumulation)
#pragma unroll
            for (int m = 0; m < 6; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][m][0], B[s][j][1], acc[m][j], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 6; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][m][1], B[s][j][0], acc[m][j], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 6; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][m][0], B[s][j][0], acc[m][j], 0, 0, 0);
        }
        __syncthreads();
        // 9 taps: the operand set of the next chunk's tap 0 is A[1]; keep the parity bookkeeping static
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            A[0][m][0] = A[1][m][0];
            A[0][m][1] = A[1][m][1];
        }
    }

    // ---- epilogue.  acc[m][j][r]: cout co0 + 16 m + 4 (lane >> 4) + r, pixel (oy0 + 4 wave + j, ox0 + (lane & 15))
    const float inv = 1.0f / (sx * a.w_scale);
    const int64_t osn = (int64_t)d.cout * d.ho * d.wo, osc = (int64_t)d.ho * d.wo;
    const int64_t obase = (int64_t)n * osn + (int64_t)(co0 + 4 * (lane >> 4)) * osc + (int64_t)(oy0 + 4 * wave) * d.wo + ox0 + (lane & 15);
    float amax = 0.f;
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        float bias[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[r] = d.bias ? d.bias[co0 + 16 * m + 4 * (lane >> 4) + r] : 0.f;
        float opnd[4][4];
        const float* op = d.dact_z ? d.dact_z : d.res;
        if (op) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) opnd[j][r] = op[obase + (int64_t)(16 * m + r) * osc + (int64_t)j * d.wo];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[m][j][r] * inv + bias[r];
                if (d.dact_z) v *= sda_dact(d.act_d, opnd[j][r]);
                else if (d.res) v += opnd[j][r];
                amax = fmaxf(amax, fabsf(v));
                d.out[obase + (int64_t)(16 * m + r) * osc + (int64_t)j * d.wo] = v;
            }
    }
    if (a.out_amax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_down(amax, off, SDA_WAVE));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(a.out_amax), __float_as_uint(amax));
    }
}

// ---------------------------------------------------------------------------------------------------------------- weight packing
// dst (16-byte units): ((((ct * nchunk + chunk) * 9 + tap) * 6 + m) * 2 + piece) * 64 + lane  ->  8 halves:
//   forward  (transpose = 0): W[co = 96 ct + 16 m + (lane & 15)][ci = 32 chunk + 8 (lane >> 4) + i][tap]
//   backward (transpose = 1): the operator of the input VJP -- its "cout" is the forward cin and vice versa, taps flipped:
//                             W[co = 32 chunk + 8 (lane >> 4) + i][ci = 96 ct + 16 m + (lane & 15)][8 - tap]
__global__ void pack_h2_kernel(const float* __restrict__ w, int cout, int cin, int transpose, float scale, h2_h8* __restrict__ dst,
                               int64_t units) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    const int lane = (int)(u & 63);
    int64_t t = u >> 6;
    const int piece = (int)(t & 1); t >>= 1;
    const int m = (int)(t % 6); t /= 6;
    const int tap = (int)(t % 9); t /= 9;
    const int M = transpose ? cin : cout, K = transpose ? cout : cin;     // operator rows / contraction
    const int nchunk = K / H2_CK;
    const int chunk = (int)(t % nchunk);
    const int ct = (int)(t / nchunk);
    const int row = H2_BM * ct + 16 * m + (lane & 15);
    h2_h8 out;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = H2_CK * chunk + 8 * (lane >> 4) + i;
        float v = 0.f;
        if (row < M) {
            const int co = transpose ? k : row, ci = transpose ? row : k, tp = transpose ? 8 - tap : tap;
            v = w[((int64_t)co * cin + ci) * 9 + tp] * scale;
        }
        const _Float16 h = (_Float16)v;
        out[i] = piece ? (_Float16)(v - (float)h) : h;
    }
    dst[u] = out;
}

extern "C" int64_t sda_conv_h2_packed_bytes(int cout, int cin, int transpose) {
    const int M = transpose ? cin : cout, K = transpose ? cout : cin;
    if (M <= 0 || K <= 0 || M % H2_BM || K % H2_CK) return 0;
    return (int64_t)(M / H2_BM) * (K / H2_CK) * 9 * 6 * 2 * 64 * 16;
}

extern "C" float sda_conv_h2_scale(float amax) { return h2_scale_of(amax); }

extern "C" int sda_pack_conv_weight_h2(const float* w, int cout, int cin, int transpose, float w_amax, void* dst, void* stream) {
    const int64_t bytes = sda_conv_h2_packed_bytes(cout, cin, transpose);
    if (!w || !dst || bytes == 0) return SDA_E_BADARG;
    const int64_t units = bytes / 16;
    hipLaunchKernelGGL(pack_h2_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cout, cin, transpose,
                       h2_scale_of(w_amax), reinterpret_cast<h2_h8*>(dst), units);
    return sda_launch_status();
}

// max |x| over a contiguous tensor into amax[0] (as uint bits; the caller zeroes it): for inputs whose producer did not report it
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n4, int64_t n, float* __restrict__ amax) {
    float m = 0.f;
    const h2_f4* x4 = reinterpret_cast<const h2_f4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const h2_f4 v = __builtin_nontemporal_load(x4 + i);
        m = fmaxf(fmaxf(fmaxf(m, fabsf(v[0])), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) m = fmaxf(m, fabsf(x[4 * n4 + threadIdx.x]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, SDA_WAVE));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(m));
}

extern "C" int sda_absmax(const float* x, int64_t numel, float* amax, void* stream) {
    if (!x || !amax || numel <= 0 || (reinterpret_cast<uintptr_t>(x) & 15)) return SDA_E_BADARG;
    hipError_t e = hipMemsetAsync(amax, 0, sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    const int64_t n4 = numel / 4;
    const int64_t want = (n4 + 255) / 256;
    const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n4, numel, amax);
    return sda_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------- launcher
static bool h2_ok(const sda_conv_desc* d) {
    if (!d || !d->x || !d->out || !d->w_h2) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride_h != 1 || d->stride_w != 1 || d->explicit_pad || d->up_h != 1 || d->up_w != 1 ||
        d->zins_h != 1 || d->zins_w != 1 || d->pool_h > 1 || d->pool_w > 1)
        return false;
    if (d->cctx != 0 || d->n_inner != 1 || d->x_n_off != 0) return false;
    if (d->cx % H2_CK || d->cout % H2_BM || d->ho != d->hs || d->wo != d->ws || d->ho % H2_TS || d->wo % H2_TS) return false;
    if (d->out_sn || d->out_sc || d->out_sy || d->out_sx) return false;
    if ((d->ln_mean == nullptr) != (d->ln_rstd == nullptr)) return false;
    if (d->mod && !d->ln_mean) return false;
    if (d->ln_mean && d->act_in != SDA_ACT_NONE) return false;
    if (d->dact_z && d->res) return false;
    if (d->x_sc < 0 || d->x_sy < 0 || d->x_sx < 0 ||
        (int64_t)d->cx * d->x_sc + (int64_t)d->hs * d->x_sy + (int64_t)d->ws * d->x_sx >= (1LL << 31))
        return false;
    const int64_t tiles = (int64_t)d->n * (d->ho / H2_TS) * (d->wo / H2_TS) * (d->cout / H2_BM);
    return tiles >= 1 && tiles <= 0x7fffffffLL;
}

extern "C" int sda_conv_h2_supported(const sda_conv_desc* d) { return h2_ok(d) ? 1 : 0; }

extern "C" int sda_conv_h2(const sda_conv_desc* d, void* stream) {
    if (!h2_ok(d)) return SDA_E_UNSUPPORTED;
    h2_args a;
    a.w = d->w_h2;
    a.w_scale = d->w_h2_scale;
    a.x_amax = d->x_amax;
    a.x_amax_static = d->x_amax_static;
    a.out_amax = d->out_amax;
    a.tiles_x = d->wo / H2_TS;
    a.tiles_y = d->ho / H2_TS;
    a.n_ct = d->cout / H2_BM;
    a.nchunk = d->cx / H2_CK;
    if (!(a.w_scale > 0.f) || (!a.x_amax && !(a.x_amax_static > 0.f))) return SDA_E_BADARG;
    const int lds = 2 * H2_TILE;
    const unsigned grid = (unsigned)((int64_t)d->n * a.tiles_x * a.tiles_y * a.n_ct);
    int rc;
    if (d->ln_mean) {
        static bool set2[SDA_MAX_DEVICES];
        if ((rc = sda_raise_dyn_lds(reinterpret_cast<const void*>(conv_h2_kernel<2>), lds, set2)) != SDA_OK) return rc;
        hipLaunchKernelGGL(conv_h2_kernel<2>, dim3(grid), dim3(256), (size_t)lds, (hipStream_t)stream, *d, a);
    } else if (d->act_in != SDA_ACT_NONE) {
        static bool set1[SDA_MAX_DEVICES];
        if ((rc = sda_raise_dyn_lds(reinterpret_cast<const void*>(conv_h2_kernel<1>), lds, set1)) != SDA_OK) return rc;
        hipLaunchKernelGGL(conv_h2_kernel<1>, dim3(grid), dim3(256), (size_t)lds, (hipStream_t)stream, *d, a);
    } else {
        static bool set0[SDA_MAX_DEVICES];
        if ((rc = sda_raise_dyn_lds(reinterpret_cast<const void*>(conv_h2_kernel<0>), lds, set0)) != SDA_OK) return rc;
        hipLaunchKernelGGL(conv_h2_kernel<0>, dim3(grid), dim3(256), (size_t)lds, (hipStream_t)stream, *d, a);
    }
    return sda_launch_status();
}
