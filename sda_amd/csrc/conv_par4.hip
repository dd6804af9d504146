// Backward-data of a stride-2 3 x 3 convolution (the level heads of the U-Net, sda/nn.py:152-159; the gradient torch.autograd
// propagates through them at sda/score.py:394) as ONE launch.  g_x[2 i + p] only receives tap 1 (p = 0, from g[i]) or taps 0, 2
// (p = 1, from g[i + 1], g[i]) per axis, so the four output parity classes are a 1 x 1, a 1 x 2, a 2 x 1 and a 2 x 2 stride-1
// convolution over g -- together the 9 taps of one 3 x 3 layer.  Round 2 ran them as four launches of the general kernel, each
// re-reading g and writing every other pixel of every other row with 4-byte stores (0.40-0.52 of the matrix peak, the 1 x 1 class
// latency-bound at 0.25-0.36).  Here a workgroup stages the g tile ONCE per K-stage, keeps the four class accumulators (192
// registers: 32 macro-pixels x 96 couts x 4 classes per wave; 128 on the 64-cout tile, template parameter MT), and writes
// (px = 0, 1) pairs as 8-byte stores -- contiguous rows:
//   * workgroup = 4 consumer waves + 4 producer waves (a wave that issues LDS-DMA itself gets a compiler-inserted vmcnt(0) in front
//     of every LDS read: the loads must come from other waves), tile = 8 x 16 macro-pixels (16 x 32 output pixels) of one image x
//     96 couts; consumer wave w owns macro-rows 2 w, 2 w + 1;
//   * K-stage = 8 channels of g: the 9 x 17 window tile through the producers' registers (padding / wrap applied once per
//     element), the [9 taps][8][96] weight slab by LDS-DMA (global_load_lds dwordx4); two stage buffers, one barrier per stage;
//   * v_mfma_f32_32x32x2_f32, M = couts, N = macro-pixels: per 2-channel step 4 B reads (the window offsets) feed 27 MFMAs.
// Weights: the four classes' sda_pack_conv_weight packings (transpose = 1) back to back in class order (0,0), (0,1), (1,0), (1,1):
// [9][cin_pad][cout_pad].  Roofline: fp32 matrix pipe; algorithmic bytes: g once, the skip once, g_x once.
#include "sda_common.hpp"
#include <stdlib.h>

#define P4_CK 8
#define P4_TR 8
#define P4_TW 16
#define P4_HR (P4_TR + 1)
#define P4_HC (P4_TW + 1)
#define P4_NPOS (P4_HR * P4_HC)        // 153 window positions per channel
#define P4_PLANE 176                   // >= 153, = 16 mod 32
// cout tile = 32 MT: 96 (MT = 3: the reference's training widths (96, 192, 384)) or 64 (MT = 2, round 6: its default widths
// (64, 128, 256), experiments/kolmogorov/utils.py:52 -- 128 accumulator registers instead of 192) or 32 (MT = 1: the remaining
// multiples of 32, sda/nn.py:99); 128 would need 256
#define P4_BM_OF(MT) (32 * (MT))
#define P4_WSZ_OF(MT) (9 * P4_CK * P4_BM_OF(MT))   // 6912 | 4608 floats
#define P4_BUF_OF(MT) (P4_WSZ_OF(MT) + P4_CK * P4_PLANE)
#define P4_NLD ((P4_CK * P4_NPOS + 255) / 256)     // 5 input loads per thread and stage
#define P4_NW_OF(MT) ((P4_WSZ_OF(MT) / 4 + 255) / 256)   // 7 | 5 16-byte weight chunks per thread and stage

typedef float p4_f32x16 __attribute__((ext_vector_type(16)));
typedef float p4_f32x2 __attribute__((ext_vector_type(2)));

// tap -> (class, window row, window column), classes in the order of the weight pack
__device__ constexpr int P4_CL[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
__device__ constexpr int P4_DY[9] = {0, 0, 0, 0, 1, 0, 0, 1, 1};
__device__ constexpr int P4_DX[9] = {0, 0, 1, 0, 0, 0, 1, 0, 1};

template <int MT>
__global__ __launch_bounds__(512, 2) void conv_par4_kernel(const sda_conv_desc d, int tiles_x, int tiles_y, int n_ct) {
    constexpr int P4_BM = P4_BM_OF(MT), P4_WSZ = P4_WSZ_OF(MT), P4_BUF = P4_BUF_OF(MT), P4_NW = P4_NW_OF(MT);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, khalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int t = blockIdx.x;
    const int ct = t % n_ct; t /= n_ct;
    const int bx = t % tiles_x; t /= tiles_x;
    const int by = t % tiles_y;
    const int n = t / tiles_y;
    const int oy0 = by * P4_TR, ox0 = bx * P4_TW, co0 = ct * P4_BM;
    const int nstage = d.cin_pad / P4_CK;
    if (wave >= 4) {
        // ================================================================ producers: global -> LDS, one stage ahead
        const int ptid = tid - 256;
        const float* gimg = d.x + (int64_t)n * d.x_sn_outer;
        // per-thread load plan (the same for every stage): element e = ptid + 256 i -> (local channel, window position)
        unsigned goff[P4_NLD];
        int loff[P4_NLD];
        unsigned live = 0;
#pragma unroll
        for (int i = 0; i < P4_NLD; ++i) {
            const int e = ptid + 256 * i;
            const bool in = e < P4_CK * P4_NPOS;
            const int ch = in ? e / P4_NPOS : 0, pos = in ? e - ch * P4_NPOS : 0;
            const int hy = pos / P4_HC, hx = pos - hy * P4_HC;
            int y = oy0 + hy, x = ox0 + hx;                               // (window origin = the macro-pixel itself: pad 0)
            bool ok = in;
            if (d.circular) {
                y = y >= d.hs ? y - d.hs : y;
                x = x >= d.ws ? x - d.ws : x;
            } else {
                ok = ok && y < d.hs && x < d.ws;
            }
            goff[i] = (unsigned)(ch * (int)d.x_sc + (ok ? y : 0) * (int)d.x_sy + (ok ? x : 0) * (int)d.x_sx);
            loff[i] = in ? P4_WSZ + ch * P4_PLANE + hy * P4_HC + hx : -1;
            live |= ok ? (1u << i) : 0u;
        }
        // weights: 16-byte chunk f = ptid + 256 i of the stage slab [tap][ck][96]: row = f / 24, c4 = f % 24
        unsigned woff[P4_NW];
#pragma unroll
        for (int i = 0; i < P4_NW; ++i) {
            const int f = ptid + 256 * i;
            const int row = f / (P4_BM / 4), c4 = f - row * (P4_BM / 4);
            const int tap = row / P4_CK, ck = row - tap * P4_CK;
            woff[i] = (unsigned)((tap * d.cin_pad + ck) * d.cout_pad + co0 + 4 * c4);
        }
        auto produce = [&](int st, float* buf) {
            const float* ws = d.w + (int64_t)(st * P4_CK) * d.cout_pad;
#pragma unroll
            for (int i = 0; i < P4_NW; ++i) {
                const int f = ptid + 256 * i;
                if (i + 1 < P4_NW || f < P4_WSZ / 4)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ws + woff[i]),
                                                     (__attribute__((address_space(3))) void*)(buf + (f - lane) * 4), 16, 0, 0);
            }
            const float* xs = gimg + (int64_t)(st * P4_CK) * d.x_sc;
            float vin[P4_NLD];
#pragma unroll
            for (int i = 0; i < P4_NLD; ++i) vin[i] = xs[goff[i]];
#pragma unroll
            for (int i = 0; i < P4_NLD; ++i)
                if (loff[i] >= 0) buf[loff[i]] = ((live >> i) & 1u) ? vin[i] : 0.f;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the LDS-DMA of the weights has landed)
        };
        produce(0, smem);
        __syncthreads();
        for (int st = 0; st < nstage; ++st) {
            if (st + 1 < nstage) produce(st + 1, smem + ((st + 1) & 1) * P4_BUF);
            __syncthreads();
        }
        return;
    }
    // ==================================================================== consumers: LDS reads + MFMA only in the loop
    p4_f32x16 acc[4][MT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][m][r] = 0.f;
    // B: channel 2 k2 + khalf at macro-pixel l31 of this wave: row 2 wave + (l31 >> 4), column l31 & 15
    const int bbase = P4_WSZ + khalf * P4_PLANE + (2 * wave + (l31 >> 4)) * P4_HC + (l31 & 15);
    // A: W[tap][channel 2 k2 + khalf][cout 32 m + l31]
    const int abase = khalf * P4_BM + l31;
    __syncthreads();                                       // stage 0 has landed
    for (int st = 0; st < nstage; ++st) {
        const float* buf = smem + (st & 1) * P4_BUF;
#pragma unroll
        for (int k2 = 0; k2 < P4_CK / 2; ++k2) {
            float b[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) b[o] = buf[bbase + 2 * k2 * P4_PLANE + (o >> 1) * P4_HC + (o & 1)];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                float a[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = buf[abase + (tap * P4_CK + 2 * k2) * P4_BM + 32 * m];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[P4_CL[tap]][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[2 * P4_DY[tap] + P4_DX[tap]], acc[P4_CL[tap]][m], 0, 0, 0);
            }
        }
        __syncthreads();                                   // the producers may refill this buffer; the next stage is ready
    }
    // ---- epilogue.  D[m][r]: cout co0 + 32 m + 4 khalf + (r & 3) + 8 (r >> 2), macro-pixel l31.  Class (py, px) of macro-pixel
    // (oy, ox) is output pixel (2 oy + py, 2 ox + px): out + py * out_sy / 2 + px (the descriptor carries the class-(0,0) view)
    const int oy = oy0 + 2 * wave + (l31 >> 4), ox = ox0 + (l31 & 15);
    const int64_t row_sz = d.out_sy / 2;
    const int64_t pbase = (int64_t)n * d.out_sn + (int64_t)oy * d.out_sy + (int64_t)ox * d.out_sx;
    // (the skip operand of eight couts x two rows is requested in one batch, then added and stored: a load per store would wait
    //  for its round trip 96 times per tile)
    const int64_t cbase = pbase + (int64_t)(co0 + 4 * khalf) * d.out_sc;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            p4_f32x2 rv[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 8 * rh + j;
                const int64_t o = cbase + (int64_t)(32 * m + (r & 3) + 8 * (r >> 2)) * d.out_sc;
#pragma unroll
                for (int py = 0; py < 2; ++py)
                    rv[2 * j + py] = d.res ? *reinterpret_cast<const p4_f32x2*>(d.res + o + py * row_sz) : p4_f32x2{0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 8 * rh + j;
                const int64_t o = cbase + (int64_t)(32 * m + (r & 3) + 8 * (r >> 2)) * d.out_sc;
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    const p4_f32x2 v = p4_f32x2{acc[2 * py][m][r], acc[2 * py + 1][m][r]} + rv[2 * j + py];
                    *reinterpret_cast<p4_f32x2*>(d.out + o + py * row_sz) = v;
                }
            }
        }
}

static bool par4_ok(const sda_conv_desc* d) {
    static const bool off = getenv("SDA_CONV_PAR4") && atoi(getenv("SDA_CONV_PAR4")) == 0;
    if (off || !d || !d->x || !d->w || !d->out) return false;
    if (d->kh != 2 || d->kw != 2 || !d->explicit_pad || d->pad_h != 0 || d->pad_w != 0 || d->stride_h != 1 || d->stride_w != 1 ||
        d->up_h != 1 || d->up_w != 1 || d->zins_h != 1 || d->zins_w != 1)
        return false;
    if (d->cctx != 0 || d->mod || d->ln_mean || d->ln_rstd || d->act_in != SDA_ACT_NONE || d->dact_z || d->bias || d->n_inner != 1 ||
        d->x_n_off != 0)
        return false;
    if ((d->cout % 32) || d->cout_pad != d->cout || d->cin_pad % P4_CK || d->cx != d->cin_pad) return false;
    if (d->ho != d->hs || d->wo != d->ws || (d->ho % P4_TR) || (d->wo % P4_TW)) return false;
    // the class-(0,0) view of a planar [n][cout][2 ho][2 wo] tensor: pixel stride 2, row stride 2 rows
    if (d->out_sx != 2 || d->out_sy != 4 * (int64_t)d->wo || d->out_sc != 4 * (int64_t)d->ho * d->wo ||
        d->out_sn != d->out_sc * d->cout)
        return false;
    if ((reinterpret_cast<uintptr_t>(d->out) & 7) || (d->res && (reinterpret_cast<uintptr_t>(d->res) & 7)) ||
        (reinterpret_cast<uintptr_t>(d->w) & 15) || (d->cout_pad & 3))
        return false;
    if (d->x_sc < 0 || d->x_sy < 0 || d->x_sx < 0 ||
        (int64_t)d->cx * d->x_sc + (int64_t)d->hs * d->x_sy + (int64_t)d->ws * d->x_sx >= (1LL << 31))
        return false;
    if ((int64_t)9 * d->cin_pad * d->cout_pad >= (1LL << 31)) return false;
    const int64_t tiles = (int64_t)d->n * (d->ho / P4_TR) * (d->wo / P4_TW) * (d->cout / P4_BM_OF(d->cout % 96 == 0 ? 3 : (d->cout % 64 == 0 ? 2 : 1)));
    return tiles >= 1 && tiles <= 0x7fffffffLL;
}

template <int MT>
static int par4_launch(const sda_conv_desc* d, hipStream_t stream) {
    static bool attr_set[SDA_MAX_DEVICES];
    const int lds = 2 * P4_BUF_OF(MT) * 4;
    const int rc = sda_raise_dyn_lds(reinterpret_cast<const void*>(conv_par4_kernel<MT>), lds, attr_set);
    if (rc != SDA_OK) return rc;
    const int tx = d->wo / P4_TW, ty = d->ho / P4_TR, n_ct = d->cout / P4_BM_OF(MT);
    hipLaunchKernelGGL(conv_par4_kernel<MT>, dim3((unsigned)((int64_t)d->n * tx * ty * n_ct)), dim3(512), (size_t)lds, stream, *d, tx, ty, n_ct);
    return sda_launch_status();
}

// One launch for the four parity classes of a stride-2 3 x 3 convolution's backward-data (see the header of this file).
// d: the class-(0,0) launch of the split formulation -- kh = kw = 2, explicit_pad with pad 0, out / res = the class-(0,0) views --
// except that d->w holds all four classes' packings back to back.  SDA_E_UNSUPPORTED -> run the four launches.
extern "C" int sda_conv_parity4(const sda_conv_desc* d, void* stream) {
    if (!par4_ok(d)) return SDA_E_UNSUPPORTED;
    return d->cout % 96 == 0 ? par4_launch<3>(d, (hipStream_t)stream)
                             : (d->cout % 64 == 0 ? par4_launch<2>(d, (hipStream_t)stream) : par4_launch<1>(d, (hipStream_t)stream));
}
