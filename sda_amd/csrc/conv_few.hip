// 3 x 3 convolutions with at most 16 output channels: the tail of the U-Net (sda/nn.py:166-176, hidden -> state channels: 96 -> 10
// for the Kolmogorov nets) and the backward-data form of its head (96 -> 10 + forcing, the forcing gradient dropped).  The
// general kernels put output channels on the 32-wide M side of v_mfma_f32_32x32x2_f32: 10 of 32 rows carry data (0.17 of the
// matrix peak, 2 % of a configs[3] step for 0.5 % of its flops).  Here M is 16 wide (v_mfma_f32_16x16x4_f32), the launch is
// plain enough to need no loader fusions, and nothing is specialised away from the memory system:
//   * workgroup = 4 consumer + 4 producer waves (two workgroups per CU), tile = 8 rows x 32 columns of one image; consumer wave w owns
//     rows 2 w, 2 w + 1 (four 16-pixel fragments, 16 accumulator registers).  (Round 3 first ran this with four do-everything waves:
//     0.36 -- the vector-ALU half of a wave's stage, load addresses and the commit's selects, cannot issue while the sibling
//     workgroup's wave on the same SIMD multiplies, so the two workgroups serialised; the split keeps the multiply streams free of it);
//   * K runs in stages of 16 input channels: the 10 x 34 halo tile of each channel and the [9][16][16] weight slab go global ->
//     producer registers -> LDS one stage ahead; one barrier per stage, two LDS buffers;
//   * A (weights, 16 couts x 4 channels): one 8-byte LDS read per tap and half stage (the slab is staged [tap][kq][cout][k-step]);
//     B: every tile value a lane multiplies in the stage is read once (4 rows x 6 columns x 4 k-steps = 96 reads, not 9 taps x 16)
//     and reused by the taps that meet it -- halo tile plane stride 368 floats (= 16 mod 32: the two channel rows a 32-lane group
//     reads sit in disjoint banks).
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
#define CF_BUF (CF_CK * CF_PLANE + 9 * CF_CK * 16) // floats per stage buffer: tile + weights

typedef float cf_f32x4 __attribute__((ext_vector_type(4)));
typedef float cf_f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512, 4) void conv_few_kernel(const sda_conv_desc d, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) float smem[2 * CF_BUF];
    const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, li = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8), and a halo row (34 floats starting one float before a
    // 128-byte line) touches three lines: with tile = blockIdx, horizontally adjacent tiles always ran on DIFFERENT XCDs and each
    // L2 fetched all three lines -- 9.5 GB per launch against 3.35 GB algorithmic (profiles/r04_kolmogorov256_few_traffic.json).
    // Each XCD now walks a contiguous range of tiles: neighbours in x and y meet in the same L2.
    int t;
    {
        const int total = (int)gridDim.x, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
        const int tq = total >> 3, tr = total & 7;
        t = xcd * tq + (xcd < tr ? xcd : tr) + slot;
    }
    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y;
    const int n = t / tiles_y;
    const int oy0 = by * CF_TR, ox0 = bx * CF_TW;
    const int nstage = d.cin_pad / CF_CK;
    if (wave >= 4) {
        // ================================================================ producers: global -> registers -> LDS, one stage ahead
        const int ptid = tid - 256;
        const int ng = n + d.x_n_off;
        const float* ximg = d.x + (int64_t)(ng / d.n_inner) * d.x_sn_outer + (int64_t)(ng % d.n_inner) * d.x_sn_inner;
        // per-thread load plan (tile invariant across stages): halo positions ptid and ptid + 256 (of 340) of EVERY channel of a stage
        // -- the channel offset is a scalar, so a load needs no vector address arithmetic and the plan is four registers
        unsigned goff[2];                  // element offset inside a channel plane
        int loff[2];                       // LDS offset inside a channel's tile plane, -1 = no position
        bool live[2];                      // the position carries data (inside the image or circular)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pos = ptid + 256 * j;
            const bool valid = pos < CF_NPOS;
            const int hy = pos / CF_HC, hx = pos - hy * CF_HC;
            int y = oy0 - 1 + hy, x = ox0 - 1 + hx;
            bool ok = valid;
            if (d.circular) {
                y = y < 0 ? y + d.hs : (y >= d.hs ? y - d.hs : y);
                x = x < 0 ? x + d.ws : (x >= d.ws ? x - d.ws : x);
            } else {
                ok = ok && y >= 0 && y < d.hs && x >= 0 && x < d.ws;
            }
            goff[j] = (unsigned)((ok ? y : 0) * (int)d.x_sy + (ok ? x : 0) * (int)d.x_sx);
            loff[j] = valid ? hy * CF_ROW + hx : -1;
            live[j] = ok;
        }
        // weights: thread ptid = (channel, cout) of every tap of the packed layout [tap][cin_pad][cout_pad]; LDS slot
        // [tap][kq = ch & 3][cout][ks = ch >> 2]: a consumer lane reads the k-steps of a tap as one value
        const int wch = ptid >> 4, wco = ptid & 15;
        const unsigned woff = (unsigned)(wch * d.cout_pad + wco);
        const int wdst = CF_CK * CF_PLANE + ((wch & 3) * 16 + wco) * 4 + (wch >> 2);
        const int64_t wtap = (int64_t)d.cin_pad * d.cout_pad;
        auto produce = [&](int st, float* buf) {
            // (every staged channel exists: the launcher requires cx == cin_pad)
            const float* xs = ximg + (int64_t)(st * CF_CK) * d.x_sc;
            const float* ws = d.w + (int64_t)(st * CF_CK) * d.cout_pad;
            float vin[CF_CK][2], vw[9];
#pragma unroll
            for (int ch = 0; ch < CF_CK; ++ch) {
                vin[ch][0] = (xs + (int64_t)ch * d.x_sc)[goff[0]];
                vin[ch][1] = loff[1] >= 0 ? (xs + (int64_t)ch * d.x_sc)[goff[1]] : 0.f;
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) vw[tap] = (ws + tap * wtap)[woff];
            if (d.circular) {
                // every position carries data: plain stores -- no vector-ALU instruction in a producer's stage (one would wait for a
                // gap in the consumers' MFMA streams on its SIMD)
#pragma unroll
                for (int ch = 0; ch < CF_CK; ++ch) {
                    buf[ch * CF_PLANE + loff[0]] = vin[ch][0];
                    if (loff[1] >= 0) buf[ch * CF_PLANE + loff[1]] = vin[ch][1];
                }
            } else {
#pragma unroll
                for (int ch = 0; ch < CF_CK; ++ch) {
                    buf[ch * CF_PLANE + loff[0]] = live[0] ? vin[ch][0] : 0.f;
                    if (loff[1] >= 0) buf[ch * CF_PLANE + loff[1]] = live[1] ? vin[ch][1] : 0.f;
                }
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) buf[wdst + tap * 256] = vw[tap];
        };
        produce(0, smem);
        __syncthreads();
        for (int st = 0; st < nstage; ++st) {
            if (st + 1 < nstage) produce(st + 1, smem + ((st + 1) & 1) * CF_BUF);
            __syncthreads();
        }
        return;
    }
    // ==================================================================== consumers: LDS reads + MFMA only in the loop
    cf_f32x4 acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[f] = cf_f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment f of this wave: row 2 wave + (f >> 1), columns 16 (f & 1) ..
    const int brow = (2 * wave) * CF_ROW + li + kq * CF_PLANE;
    __syncthreads();                                       // stage 0 has landed
    for (int st = 0; st < nstage; ++st) {
        const float* buf = smem + (st & 1) * CF_BUF;
        const float* wl = buf + CF_CK * CF_PLANE + (kq * 16 + li) * 4;
        // two halves of two k-steps each.  A[m = li][k = kq] of the half's k-steps for all nine taps: one 8-byte read per tap;
        // B: every tile value this lane multiplies is read ONCE per half -- rows 0 .. 3 of the wave's window, columns li + dx and
        // li + 16 + dx (dx = 0 .. 2) -- and reused by the (up to two) taps that meet it
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            cf_f32x2 a[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) a[tap] = *reinterpret_cast<const cf_f32x2*>(wl + tap * 256 + 2 * half);
            float bb[2][4][6];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 6; ++c)
                        bb[k2][r][c] = buf[brow + 4 * (2 * half + k2) * CF_PLANE + r * CF_ROW + 16 * (c / 3) + (c % 3)];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int f = 0; f < 4; ++f)
                        acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tap][k2], bb[k2][(f >> 1) + dy][3 * (f & 1) + dx], acc[f], 0, 0, 0);
            }
        }
        __syncthreads();                                   // the producers may refill this buffer; the next stage is ready
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
    hipLaunchKernelGGL(conv_few_kernel, dim3((unsigned)((int64_t)d->n * tx * ty)), dim3(512), 0, stream, *d, tx, ty);
    return sda_launch_status();
}
