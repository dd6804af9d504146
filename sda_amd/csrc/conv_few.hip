// 3 x 3 convolutions with at most 16 output channels: the tail of the U-Net (sda/nn.py:166-176, hidden -> state channels: 96 -> 10
// for the Kolmogorov nets) and the backward-data form of its head (96 -> 10 + forcing, the forcing gradient dropped).  The
// general kernels put output channels on the 32-wide M side of v_mfma_f32_32x32x2_f32: 10 of 32 rows carry data (0.17 of the
// matrix peak, 2 % of a configs[3] step for 0.5 % of its flops).  Here M is 16 wide (v_mfma_f32_16x16x4_f32), the launch is
// plain enough to need no loader fusions, and nothing is specialised away from the memory system:
//   * workgroup = 4 waves, tile = 8 rows x 32 columns of one image, wave w owns rows 2 w, 2 w + 1 (four 16-pixel fragments, 16
//     accumulator registers); two workgroups per CU hide each other's global round trips (no producer / consumer split);
//   * K runs in stages of 16 input channels: the 10 x 34 halo tile of each channel and the [9][16][16] weight slab go global ->
//     registers (issued before the stage's multiply) -> LDS (after it); one barrier per stage, two LDS buffers;
//   * A (weights, 16 couts x 4 channels) fragments are read once per stage and reused by the four pixel fragments; B comes from
//     the halo tile, plane stride 368 floats (= 16 mod 32: the two channel rows a 32-lane group reads sit in disjoint banks).
// Roofline: fp32 matrix pipe at 10/16 useful rows; HBM floor (input read once) is ~5x below.
#include "sda_common.hpp"
#include <stdlib.h>

#define CF_CK 16
#define CF_TR 8
#define CF_TW 32
#define CF_HR (CF_TR + 2)
#define CF_HC (CF_TW + 2)
#define CF_PLANE 368                   // >= CF_HR * 36, = 16 mod 32
#define CF_ROW 36
#define CF_NPOS (CF_HR * CF_HC)        // 340 halo positions per channel
#define CF_NLD ((CF_CK * CF_NPOS + 255) / 256)     // 22 input loads per thread and stage
#define CF_WLD ((9 * CF_CK * 16 + 255) / 256)      // 9 weight loads per thread and stage
#define CF_BUF (CF_CK * CF_PLANE + 9 * CF_CK * 16) // floats per stage buffer: tile + weights

typedef float cf_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void conv_few_kernel(const sda_conv_desc d, int tiles_x, int tiles_y) {
    __shared__ float smem[2 * CF_BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, li = lane & 15;
    int t = blockIdx.x;
    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y;
    const int n = t / tiles_y;
    const int oy0 = by * CF_TR, ox0 = bx * CF_TW;
    const int ng = n + d.x_n_off;
    const float* ximg = d.x + (int64_t)(ng / d.n_inner) * d.x_sn_outer + (int64_t)(ng % d.n_inner) * d.x_sn_inner;
    // ---- per-thread load plan (tile invariant across stages): element e = tid + 256 i -> (local channel, halo position)
    unsigned goff[CF_NLD];             // element offset from the stage's first channel
    int loff[CF_NLD];                  // LDS offset inside the stage's tile, -1 = no element
    unsigned live = 0;                 // bit i: the position carries data (inside the image or circular)
#pragma unroll
    for (int i = 0; i < CF_NLD; ++i) {
        const int e = tid + 256 * i;
        const int ch = e / CF_NPOS, pos = e - ch * CF_NPOS;
        const int hy = pos / CF_HC, hx = pos - hy * CF_HC;
        int y = oy0 - 1 + hy, x = ox0 - 1 + hx;
        bool ok = e < CF_CK * CF_NPOS;
        if (d.circular) {
            y = y < 0 ? y + d.hs : (y >= d.hs ? y - d.hs : y);
            x = x < 0 ? x + d.ws : (x >= d.ws ? x - d.ws : x);
        } else {
            ok = ok && y >= 0 && y < d.hs && x >= 0 && x < d.ws;
        }
        const int yc = ok ? y : 0, xc = ok ? x : 0, cc = e < CF_CK * CF_NPOS ? ch : 0;
        goff[i] = (unsigned)(cc * (int)d.x_sc + yc * (int)d.x_sy + xc * (int)d.x_sx);
        loff[i] = e < CF_CK * CF_NPOS ? ch * CF_PLANE + hy * CF_ROW + hx : -1;
        live |= ok ? (1u << i) : 0u;
    }
    // weights: slab element f = tid + 256 i -> (tap, channel, cout): packed layout [tap][cin_pad][cout_pad]
    unsigned woff[CF_WLD];
#pragma unroll
    for (int i = 0; i < CF_WLD; ++i) {
        const int f = tid + 256 * i;
        const int tap = f / (CF_CK * 16), r = f - tap * (CF_CK * 16), ch = r >> 4, co = r & 15;
        woff[i] = (unsigned)((tap * d.cin_pad + ch) * d.cout_pad + co);
    }
    float vin[CF_NLD], vw[CF_WLD];
    const int nstage = d.cin_pad / CF_CK;
    auto load = [&](int st) {
        const float* xs = ximg + (int64_t)(st * CF_CK) * d.x_sc;
        const float* ws = d.w + (int64_t)(st * CF_CK) * d.cout_pad;
#pragma unroll
        for (int i = 0; i < CF_NLD; ++i) vin[i] = xs[goff[i]];
#pragma unroll
        for (int i = 0; i < CF_WLD; ++i) vw[i] = ws[woff[i]];
    };
    auto commit = [&](float* buf) {
        // (every staged channel exists: the launcher requires cx == cin_pad)
#pragma unroll
        for (int i = 0; i < CF_NLD; ++i)
            if (loff[i] >= 0) buf[loff[i]] = ((live >> i) & 1u) ? vin[i] : 0.f;
#pragma unroll
        for (int i = 0; i < CF_WLD; ++i) buf[CF_CK * CF_PLANE + tid + 256 * i] = vw[i];
    };
    cf_f32x4 acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[f] = cf_f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment f of this wave: row 2 wave + (f >> 1), columns 16 (f & 1) ..
    const int brow = (2 * wave) * CF_ROW + li + kq * CF_PLANE;
    load(0);
    commit(smem);
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        const float* buf = smem + (st & 1) * CF_BUF;
        if (st + 1 < nstage) load(st + 1);
        const float* wl = buf + CF_CK * CF_PLANE + kq * 16 + li;          // A[m = li][k = kq] of (tap, k-step): + (tap * 16 + 4 ks) * 16
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - 3 * dy;
            float a[4], b[4][4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[ks] = wl[(tap * CF_CK + 4 * ks) * 16];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    b[ks][f] = buf[brow + 4 * ks * CF_PLANE + ((f >> 1) + dy) * CF_ROW + 16 * (f & 1) + dx];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], b[ks][f], acc[f], 0, 0, 0);
        }
        if (st + 1 < nstage) commit(smem + ((st + 1) & 1) * CF_BUF);
        __syncthreads();
    }
    // ---- epilogue: D[m = 4 kq + r][n = li] -> out[n][co][oy][ox .. ox + 15], + bias (+ residual)
    const int64_t hw = (int64_t)d.ho * d.wo;
    float* on = d.out + (int64_t)n * d.cout * hw;
    const float* rn = d.res ? d.res + (int64_t)n * d.cout * hw : nullptr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = 4 * kq + r;
        if (co >= d.cout) continue;
        const float bias = d.bias ? d.bias[co] : 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int64_t o = (int64_t)co * hw + (int64_t)(oy0 + 2 * wave + (f >> 1)) * d.wo + ox0 + 16 * (f & 1) + li;
            float v = acc[f][r] + bias;
            if (rn) v += rn[o];
            on[o] = v;
        }
    }
}

// eligibility: plain 3 x 3, stride 1, <= 16 output channels, no loader fusions / context / zero insertion / strided output,
// image a multiple of the 8 x 32 tile, input channels padded to a multiple of 16 inside the tensor's own channel count
static bool few_ok(const sda_conv_desc* d) {
    static const bool off = getenv("SDA_CONV_FEW") && atoi(getenv("SDA_CONV_FEW")) == 0;
    if (off || !d || !d->x || !d->w || !d->out) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride_h != 1 || d->stride_w != 1 || d->up_h != 1 || d->up_w != 1 || d->zins_h != 1 ||
        d->zins_w != 1 || d->explicit_pad || d->out_sn || d->out_sc || d->out_sy || d->out_sx)
        return false;
    if (d->cout < 1 || d->cout > 16 || d->cout_pad < 16 || d->cctx != 0 || d->mod || d->ln_mean || d->ln_rstd || d->act_in != SDA_ACT_NONE ||
        d->dact_z)
        return false;
    if (d->cin_pad % CF_CK || d->cx != d->cin_pad || d->n_inner < 1) return false;          // (every staged channel exists)
    if (d->ho != d->hs || d->wo != d->ws || (d->ho % CF_TR) || (d->wo % CF_TW)) return false;
    if (d->x_sc < 0 || d->x_sy < 0 || d->x_sx < 0 ||
        (int64_t)d->cx * d->x_sc + (int64_t)d->hs * d->x_sy + (int64_t)d->ws * d->x_sx >= (1LL << 31))
        return false;
    if ((int64_t)9 * d->cin_pad * d->cout_pad >= (1LL << 31)) return false;
    const int64_t tiles = (int64_t)d->n * (d->ho / CF_TR) * (d->wo / CF_TW);
    return tiles >= 1 && tiles <= 0x7fffffffLL;
}

int sda_few_path(const sda_conv_desc* d) { return few_ok(d) ? 1 : 0; }

// SDA_E_UNSUPPORTED -> the launch is served by the general kernels
int sda_few_try(const sda_conv_desc* d, hipStream_t stream) {
    if (!few_ok(d)) return SDA_E_UNSUPPORTED;
    const int tx = d->wo / CF_TW, ty = d->ho / CF_TR;
    hipLaunchKernelGGL(conv_few_kernel, dim3((unsigned)((int64_t)d->n * tx * ty)), dim3(256), 0, stream, *d, tx, ty);
    return sda_launch_status();
}
