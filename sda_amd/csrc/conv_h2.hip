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
// is three v_mfma_f32_32x32x16_f16 (32 cycles each from one wave) per 32 couts x 32 pixels x 16 channels where the fp32 pipe needs sixteen
// v_mfma_f32_16x16x4_f32 (32 cycles each): 96 cycles against 512, 5.3x per multiply -- more than Winograd F(2x2,3x3) saves (2.25x), so this
// kernel is a DIRECT convolution: no transforms, no transform-domain round-off, no helper-wave arithmetic.  (The 16x16x32 shape issues
// only every 25 cycles from a single wave -- profiles/r04_bf16x6_prototype.txt, and this kernel's first version: 28 per MFMA.)
// Measured against float64 (tools/f16_split_numerics.py, tools/h2_check.py): the error class of the fp32 Winograd kernel
// (max |err| / max |ref| 1e-7 .. 1.9e-6 over the layer shapes, the fp32 kernels 2e-7 .. 1.9e-6).
//
// Structure (round 5, seventh version; one PERSISTENT workgroup per CU = 4 consumer + 4 producer waves, <= 256 registers each; tile = 96
// couts (64 for widths that are multiples of 64 only: template parameter MB, round 6) x 16 x 16 pixels of one image; csrc/conv_par4.hip is the same pattern on the fp32 pipe):
//   * K loop in stages of 16 input channels (one K step of the MFMA).  A stage in LDS = the cout tile's weight slab for all nine taps
//     ([tap][cout fragment][piece][lane] x 16 B = 54 KiB, the packing's own order) + the 18 x 18 halo tile as [pixel][hi: 16 halves |
//     lo: 16 halves | 16 B pad] (26 KiB); two stage buffers = 162 432 B of the CU's 163 840; ONE barrier per stage.
//   * producers (one stage ahead of the consumers): the slab by LDS-DMA (inline asm global_load_lds_dwordx4, 54 pieces of 1 KiB over
//     the four waves; hipcc neither counts it nor drains vmcnt around it), the tile through registers: wave = 8 channels x half of the
//     halo pixels, lane = pixel -- global loads (padding / wrap resolved once per tile; scalar channel bases + one 32-bit lane offset),
//     the loader fusions of the reference's blocks (time modulation + LayerNorm, or the activation; sda/nn.py:28,137-139), the split,
//     two ds_write_b128 per pixel.  A stage's values are REQUESTED TWO STAGES AHEAD (three register sets; an HBM round trip under load
//     is longer than one stage's 2.2 us of multiplies) and waited for with a counted vmcnt: the loads are issued after the stage's DMA,
//     so "at most the 24 newest outstanding" = the DMA has landed and so have the next stage's values.
//   * consumers: per tap 6 + 4 ds_read_b128 (one tap ahead, interleaved 1 : 1 with the first MFMAs) feed 3 cout fragments x 2 pixel
//     fragments x 3 products = 18 v_mfma_f32_32x32x16_f16; nothing else in the loop (no vector ALU, no vector memory).  The B fragment's
//     MFMA column -> pixel map follows ds_read_b128's lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32: one LDS cycle
//     each): a group reads 16 consecutive pixels of one halo row at 80 B = 20 banks apart -- all 64 banks once, no row padding
//     (SQ_LDS_BANK_CONFLICT = 0, profiles/r05_h2_pmc_v7_384ch.txt).
//   * persistent schedule: workgroup b (XCD b % 8) takes, pass after pass, a run of consecutive logical tiles together with its XCD's
//     other 31 workgroups (halo lines and weight slabs meet in one L2); the producers run across tile boundaries, so the next tile's
//     first stage is in LDS when the consumers come back from their epilogue.
//   * epilogue (consumers): x 1 / (s_x s_w), + bias, x act'(z) or + residual, 64-byte row segments; optionally max |out| (one atomic
//     per wave and LAUNCH) so that the NEXT h2 launch knows its input scale without a pass over the tensor.
// Roofline: f16 matrix pipe (2.5 PFLOP/s dense / 3 products).  What that pipe SUSTAINS on this data is less: with full-mantissa random
// halves in the operands the power-managed clock falls to 1.64 GHz (tools/h2_power_probe.hip: 1 592 TFLOP/s = 0.64 of the nominal
// peak with this kernel's LDS operand traffic, 2 411 on zeros); the kernel's counters show the matrix pipe busy 0.83 of its cycles
// at 1.45 GHz (profiles/r05_h2_pmc_v7_384ch.txt).  Algorithmic bytes: x once per 96-cout tile x 1.27 (halo), out once.
#include "sda_common.hpp"
#include <stdlib.h>
#include <type_traits>

typedef _Float16 h2_h8 __attribute__((ext_vector_type(8)));
typedef float h2_f4 __attribute__((ext_vector_type(4)));
typedef float h2_f16v __attribute__((ext_vector_type(16)));

#define H2_TS 16                       // tile side (pixels)
#define H2_HS (H2_TS + 2)              // halo side
#define H2_NPX (H2_HS * H2_HS)         // 324 halo pixels
#define H2_CK 16                       // channels per stage (one K step of the 32x32x16 MFMA)
#define H2_KQ 32                       // the layer's contraction must be a multiple of this: an EVEN number of stages per tile (two stage buffers;
                                       // the consumers pick the buffer from the tile-local stage index).  Round 5 had 96 here: its producers unrolled
                                       // (3 value sets x 2 buffers =) six stages and restarted the ring with every tile; the ring now runs ACROSS tiles
#define H2_PXB 80                      // bytes per halo pixel in LDS: 32 hi + 32 lo + 16 pad (20 banks: any 16 consecutive pixels cover all 64)
#define H2_BTILE (H2_NPX * H2_PXB)     // 25 920 B
#define H2_ASLAB (9 * 3 * 2 * 1024)    // 55 296 B: a stage's weights, [tap][cout fragment][piece][lane] x 16 B
#define H2_STAGE (H2_ASLAB + H2_BTILE) // 81 216 B
#define H2_LDS (2 * H2_STAGE)          // 162 432 B of the CU's 163 840
#define H2_BM 96                       // couts per workgroup, at most (template parameter MB = 32-cout fragments per tile: 3, or 2 for widths that
                                       // are multiples of 64 only -- the reference's default (64, 128, 256), kolmogorov/utils.py:52; round 6)
#define H2_BM_OF(MB) (32 * (MB))
#define H2_MB_OF(rows) ((rows) % 96 == 0 ? 3 : ((rows) % 64 == 0 ? 2 : 0))
#define H2_PRND 3                      // loader rounds per producer wave and stage (pixels lane + 64 (2 r + half))
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
    int tiles_x, tiles_y, n_ct, nchunk, ntiles;
    int c16;                    // (MODE 2) 16-channel chunks per parity class: nchunk = 4 c16 stages
    int stagger;                // (tooling) start delay of workgroup slot s: (s & 3) * stagger * 8 128 cycles
};

// barrier of the 4 + 4 waves.  Not __syncthreads(): that also drains vmcnt, and a consumer's epilogue stores (gfx9 counts stores in
// vmcnt) would be waited for at the next tile's first stage.  Consumers wait for their LDS reads only; producers also for their
// loads / LDS-DMA (hipcc does not count LDS-DMA: the wait is ours).
#define H2_BARRIER_CONSUMER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define H2_BARRIER_PRODUCER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ABL (tooling builds, -DSDA_H2_ABLATE + $SDA_H2_ABL; results WRONG): 1 no weight DMA, 2 no activation loader, 4 no MFMAs,
// 8 no epilogue stores -- what each costs, measured by leaving it out (tools/h2_check.py)
// MODE 0: the 3 x 3 convolution, 9 taps per stage.  MODE 1 = UP-SAMPLED source, 4 taps per stage (sda_conv_desc.up_h = up_w = 2, the tails sda/nn.py:161-169):
// output pixel (2 i + py, 2 j + px) of a 3 x 3 convolution over the nearest-up-sampled image sees only the 2 x 2 source pixels
// (i - 1 + py + a, j - 1 + px + b), so each output parity class is a 2 x 2-tap convolution of the LOW-resolution tile with the taps that
// fall on one source pixel pre-summed (sda_pack_conv_weight_h2_up): 4 / 9 of the multiplies.  A tile is then (class, cout tile, 16 x
// 16 low-resolution pixels); the class only moves the consumers' tile origin by (py, px) and interleaves the stores.
// MODE 2 = the TRANSPOSE of that: the VJP of such a tail, summed over the 2 x 2 up-sampling cells (sda_conv_desc.pool_h = pool_w = 2) --
//     gx[ci][u][v] = sum over classes, taps, co of  wsum[class][co][ci][a][b]  g[co][2 (u + 1 - py - a) + py][2 (v + 1 - px - b) + px]:
// a 2 x 2-tap convolution over the four PARITY PLANES of the fine-resolution gradient as 4 x as many input channels.  A stage is then
// (class, 16 channels): the producers sample the class's plane (stride-2 address plan + a class offset on the scalar base), the
// consumers shift their tile origin per stage; the output is the low-resolution tensor itself.
// MODE 3 = the VJP of a STRIDE-2 3 x 3 convolution (the level heads, sda/nn.py:152-159; sda_conv_desc.zins_h = zins_w = 2): output pixel
// 2 m + p receives tap 1 from g[m] (p = 0) or taps 0, 2 from g[m + 1], g[m] (p = 1) per axis -- the four output parity classes are 1 x 1,
// 1 x 2, 2 x 1 and 2 x 2-tap convolutions of g (what csrc/conv_par4.hip runs on the fp32 pipe): MODE 1's structure with a per-class tap
// count.  MODE 4 = that head's FORWARD (stride_h = stride_w = 2): the transpose -- input parity planes as in MODE 2, 1 / 2 / 2 / 4 taps
// per stage class.  Both issue exactly the 9 taps of the layer.
// MB: 32-cout fragments per tile (3 = 96 couts, 2 = 64).  The LDS layout keeps the 96-cout slab's place for the weights (the halo tile sits
// behind it either way); a tap's fragments are MB x 2 KiB, the DMA copies nt x 2 MB pieces, a tap multiplies MB x 2 x 3 MFMAs.
template <int LOADER, int MODE = 0, int ABL = 0, int MB = 3>         // LOADER: 0 plain, 1 activation (SiLU), 2 LayerNorm (+ optional modulation)
__global__ __launch_bounds__(512) void conv_h2_kernel(const sda_conv_desc d, const h2_args a) {
    static_assert(MB == 2 || MB == 3, "cout tile = 64 or 96");
    constexpr bool PLANES = MODE == 2 || MODE == 4;                 // the producers sample input parity planes; the class changes per STAGE
    constexpr bool SCATTER = MODE == 1 || MODE == 3;                // a tile writes one output parity class; the class changes per TILE
    constexpr bool VTAPS = MODE == 3 || MODE == 4;                  // (1 + py) x (1 + px) taps per class instead of 4
    constexpr int NT = MODE == 0 ? 9 : 4;                           // (most) taps of a stage
    constexpr int TAPB = MB * 2 * 1024;                             // bytes of one tap's weight fragments
    const int PH = PLANES ? d.hs >> 1 : d.hs, PW = PLANES ? d.ws >> 1 : d.ws;            // the grid the tiles walk (PLANES: a parity plane)
    // taps of class c = 2 py + px, and the taps of the classes before it
    auto cls_nt = [](int c) { return VTAPS ? (1 + (c >> 1)) * (1 + (c & 1)) : NT; };
    auto cls_cum = [](int c) { return VTAPS ? ((0x5310 >> (4 * c)) & 15) : NT * c; };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- persistent schedule.  Workgroup b lives on XCD b % 8; in pass `it` the G / 8 workgroups of an XCD take G / 8 CONSECUTIVE
    // logical tiles (all cout tiles of a pixel tile, then the row of pixel tiles): they share halo lines and the weight slabs in that
    // XCD's L2 while they are hot.
    const int G = gridDim.x;
    const bool xwalk = (G & 7) == 0;
    const int per = G >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    auto tile_of = [&](int it) { return xwalk ? (it * 8 + xcd) * per + slot : it * G + (int)blockIdx.x; };
    // (the producer wrote it with device-scope atomics: read it at device scope too -- a scalar-cache line left over from the previous
    //  replay of a captured step is not good enough)
    const float sx = h2_scale_of(a.x_amax ? __hip_atomic_load(a.x_amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.x_amax_static);

#ifdef SDA_H2_ABLATE
    for (int i = 0; i < (slot & 3) * a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    if (wave >= 4) {
        // ================================================================ producers: global -> (loader fusions, split) -> LDS
        const int pw = wave - 4;
        const int ch8 = (pw & 1) * 8;                               // this wave's eight channels of a stage
        const int half = pw >> 1;                                   // ... and its half of the pixel rounds
        struct plan_t {
            const float* ximg;                                      // image + channel ch8
            const float* modp;
            int ct;                                                 // cout tile (SCATTER: class * cout tiles + cout tile)
            unsigned goff[H2_PRND];                                 // BYTE offsets of the pixels (scalar base + 32-bit lane offset loads)
            float mean[H2_PRND], rstd[H2_PRND];
            unsigned valid;
        };
        int lds_wr[H2_PRND];
#pragma unroll
        for (int r = 0; r < H2_PRND; ++r) {
            const int p = lane + 64 * (2 * r + half);
            lds_wr[r] = p < H2_NPX ? H2_ASLAB + p * H2_PXB + (pw & 1) * 16 : -1;      // (+ 32: the low piece)
        }
        auto make_plan = [&](int L, plan_t& P) {
            int t = L;
            const int ct = t % a.n_ct; t /= a.n_ct;
            const int bx = t % a.tiles_x; t /= a.tiles_x;
            const int by = t % a.tiles_y;
            const int n = t / a.tiles_y;
            const int oy0 = by * H2_TS, ox0 = bx * H2_TS;
            P.ximg = d.x + (int64_t)n * d.x_sn_outer + (int64_t)ch8 * d.x_sc;
            P.modp = (LOADER == 2 && d.mod) ? d.mod + (int64_t)n * d.mod_sn + ch8 : nullptr;
            P.ct = ct;
            P.valid = 0;
#pragma unroll
            for (int r = 0; r < H2_PRND; ++r) {
                const int p = lane + 64 * (2 * r + half);
                const int hy = p / H2_HS, hx = p - hy * H2_HS;
                int y = oy0 + hy - 1, x = ox0 + hx - 1;
                bool ok = p < H2_NPX;
                if (d.circular) {
                    y = y < 0 ? y + PH : (y >= PH ? y - PH : y);
                    x = x < 0 ? x + PW : (x >= PW ? x - PW : x);
                } else {
                    ok = ok && y >= 0 && y < PH && x >= 0 && x < PW;
                }
                y = ok ? y : 0;
                x = ok ? x : 0;
                P.goff[r] = 4u * (unsigned)((PLANES ? 2 : 1) * (y * (int)d.x_sy + x * (int)d.x_sx));
                P.valid |= ok ? (1u << r) : 0u;
                if (LOADER == 2) {
                    const int64_t sp = (int64_t)n * d.hs * d.ws + (int64_t)y * d.ws + x;
                    P.mean[r] = d.ln_mean[sp];
                    P.rstd[r] = d.ln_rstd[sp] * sx;                 // (the power-of-two scale folded in: exact)
                } else {
                    P.mean[r] = 0.f;
                    P.rstd[r] = sx;
                }
            }
        };
        auto load_from = [&](const float* ximg, const unsigned (&goff)[H2_PRND], int chunk, float (&v)[H2_PRND][8]) {
            if (ABL & 2) return;
            // (MODE 2: stage = class * c16 + chunk; the class's parity plane starts (py, px) pixels into the image)
            const int cls = PLANES ? chunk / a.c16 : 0, ck = PLANES ? chunk - cls * a.c16 : chunk;
            const char* src = reinterpret_cast<const char*>(ximg + (int64_t)ck * H2_CK * d.x_sc + (cls >> 1) * d.x_sy + (cls & 1) * d.x_sx);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const char* ch = src + (int64_t)i * d.x_sc * 4;     // (wave-uniform: an SGPR pair)
#pragma unroll
                for (int r = 0; r < H2_PRND; ++r) v[r][i] = *reinterpret_cast<const float*>(ch + goff[r]);
            }
        };
        auto load_stage = [&](const plan_t& P, int chunk, float (&v)[H2_PRND][8]) { load_from(P.ximg, P.goff, chunk, v); };
        // LDS-DMA by inline asm: hipcc does not count it (the waits are ours, below), and -- unlike the builtin -- it does not make hipcc
        // drain vmcnt in front of every LDS access and register-load use of the loop.
        // a stage's weight slab: [MODE 0] cout tile, chunk; [SCATTER] class (of the tile), cout tile, chunk; [MODE 2] cout tile, stage;
        // [MODE 4] class (of the stage), cout tile, chunk -- classes outermost, each with its own tap count
        auto slab_of = [&](const plan_t& P, int chunk, int& nt) -> const unsigned char* {
            const unsigned char* w0 = reinterpret_cast<const unsigned char*>(a.w);
            if (MODE == 0 || MODE == 2) { nt = NT; return w0 + ((int64_t)P.ct * a.nchunk + chunk) * (NT * TAPB); }
            if (SCATTER) {
                const int nct1 = a.n_ct >> 2, cls = P.ct / nct1, ct = P.ct - cls * nct1;
                nt = cls_nt(cls);
                return w0 + ((int64_t)cls_cum(cls) * nct1 * a.nchunk + ((int64_t)ct * a.nchunk + chunk) * nt) * TAPB;
            }
            const int cls = chunk / a.c16, ck = chunk - cls * a.c16;                       // MODE 4
            nt = cls_nt(cls);
            return w0 + ((int64_t)cls_cum(cls) * a.n_ct * a.c16 + ((int64_t)P.ct * a.c16 + ck) * nt) * TAPB;
        };
        auto dma_weights = [&](const plan_t& P, int chunk, int bufsel) {
            if (ABL & 1) return;
            int nt;
            const unsigned char* src = slab_of(P, chunk, nt) + lane * 16;
            const unsigned lds0 = __builtin_amdgcn_readfirstlane(
                (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)(smem + bufsel * H2_STAGE)));
            // nt * 2 MB pieces of 1 KiB (one wave-instruction each): wave pw takes pieces pw, pw + 4, ...
            const int npc = nt * 2 * MB;
#pragma unroll
            for (int k = 0; k < (NT * 2 * MB + 3) / 4; ++k) {
                const int piece = pw + 4 * k;
                if (piece < npc) {
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src + piece * 1024), "s"(lds0 + piece * 1024) : "memory");
                }
            }
        };
        auto convert_stage = [&](const plan_t& P, int chunk, const float (&v)[H2_PRND][8], unsigned char* buf) {
            if (ABL & 2) return;
            float mod[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) mod[i] = (LOADER == 2 && P.modp) ? P.modp[chunk * H2_CK + i] : 0.f;
#pragma unroll
            for (int r = 0; r < H2_PRND; ++r) {
                h2_h8 hi, lo;
                const bool ok = (P.valid >> r) & 1u;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float u = v[r][i];
                    if (LOADER == 2) u = ((u + mod[i]) - P.mean[r]) * P.rstd[r];
                    else if (LOADER == 1) u = (u * sda_sigmoid(u)) * sx;         // (SiLU: the launcher admits no other activation here)
                    else u = u * sx;
                    u = ok ? u : 0.f;
                    const _Float16 h = (_Float16)u;
                    hi[i] = h;
                    lo[i] = (_Float16)(u - (float)h);
                }
                if (lds_wr[r] >= 0) {
                    *reinterpret_cast<h2_h8*>(buf + lds_wr[r]) = hi;
                    *reinterpret_cast<h2_h8*>(buf + lds_wr[r] + 32) = lo;
                }
            }
        };
        // values a finished wait has landed: hide them from hipcc's own load bookkeeping (beside LDS-DMA it waits vmcnt(0) in front of
        // the first use of any register load -- here that would drain the NEXT stage's loads, issued at the top of the iteration)
        auto launder = [&](float (&v)[H2_PRND][8]) {
#pragma unroll
            for (int r = 0; r < H2_PRND; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(v[r][i]));
        };
        plan_t P0, P1;
        int it = 0;
        if (tile_of(0) >= a.ntiles) return;
        make_plan(tile_of(0), P0);
        // Three register sets of loader values: a stage's values are requested TWO stages ahead (an HBM round trip under load is 2-3 us,
        // a stage's multiplies 2.2 us), in program order AFTER the stage's weight DMA -- vmcnt counts in order, so "at most the 24 newest
        // outstanding" = this stage's DMA has landed and so have the values the NEXT iteration converts.
        float raw[3][H2_PRND][8];
        load_stage(P0, 0, raw[0]);
        load_stage(P0, 1, raw[1]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        launder(raw[0]);
        launder(raw[1]);
        bool more = false;
        P1 = P0;
        // The ring of (3 value sets x 2 stage buffers) runs ACROSS tiles: a tile has an even number of stages (H2_KQ), so its first stage
        // always lands in buffer 0 -- where the consumers look for it -- but on any of the three value sets.  `chunk` is the tile-local
        // stage; at a tile's end the plan of the next tile (made one tile ahead) takes over.  Returns false behind the last stage.
        int chunk = 0;
        {
            const int Ln = tile_of(1);
            more = Ln < a.ntiles;
            if (more) make_plan(Ln, P1);
        }
        auto stage = [&](auto set_c, auto buf_c) -> bool {
            constexpr int S = decltype(set_c)::value, Bf = decltype(buf_c)::value;
            if (chunk == a.nchunk) {                                // (wave-uniform) the tile is complete: the next one's first stage
                if (!more) return false;
                P0 = P1;
                ++it;
                const int Ln = tile_of(it + 1);
                more = Ln < a.ntiles;
                if (more) make_plan(Ln, P1); else P1 = P0;
                chunk = 0;
            }
            unsigned char* buf = smem + Bf * H2_STAGE;
            dma_weights(P0, chunk, Bf);
            asm volatile("" ::: "memory");                          // (no register load may be hoisted above the DMA: the count below)
            // ALWAYS 24 loads (past the last tile: a re-read of this tile, P1 = P0) -- behind a branch, hipcc's own wait in front of the
            // launder below assumes the path without them and drains the loads just issued
            const int c2 = chunk + 2;
            const bool own = c2 < a.nchunk;
            unsigned goff[H2_PRND];
#pragma unroll
            for (int r = 0; r < H2_PRND; ++r) goff[r] = own ? P0.goff[r] : P1.goff[r];
            load_from(own ? P0.ximg : P1.ximg, goff, own ? c2 : c2 - a.nchunk, raw[(S + 2) % 3]);
            convert_stage(P0, chunk, raw[S], buf);
            // in order: ... this stage's DMA | the 24 loads just issued.  (hipcc counts only its own loads: its wait in front of the
            // launder below is the same vmcnt(24), or stricter.)
            if (ABL & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            launder(raw[(S + 1) % 3]);
            H2_BARRIER_PRODUCER_LDS();
            ++chunk;
            return true;
        };
        using c0 = std::integral_constant<int, 0>;
        using c1 = std::integral_constant<int, 1>;
        using c2t = std::integral_constant<int, 2>;
        for (;;) {
            if (!stage(c0{}, c0{})) break;
            if (!stage(c1{}, c1{})) break;
            if (!stage(c2t{}, c0{})) break;
            if (!stage(c0{}, c1{})) break;
            if (!stage(c1{}, c0{})) break;
            if (!stage(c2t{}, c1{})) break;
        }
        return;
    }

    // ==================================================================== consumers: LDS reads + MFMAs only in the loop
    // B fragment: MFMA column n = lane & 31 is a pixel of a 2 x 16 patch.  ds_read_b128 serves the lane groups {0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31} (+ 32) in one LDS cycle each: a group reads 16 CONSECUTIVE pixels of one row (stride 80 B = 20 banks: all 64
    // banks once), so the columns of a row go to one group's lanes in order.
    const int n31 = lane & 31;
    const int prow = ((n31 >= 4 && n31 < 12) || (n31 >= 16 && n31 < 20) || n31 >= 28) ? 1 : 0;
    const int pcol = prow ? (n31 < 12 ? n31 - 4 : (n31 < 20 ? n31 - 8 : n31 - 16)) : (n31 < 4 ? n31 : (n31 < 16 ? n31 - 8 : n31 - 12));
    const int b_rd = H2_ASLAB + ((4 * wave + prow) * H2_HS + pcol) * H2_PXB + (lane >> 5) * 16;
    const int a_rd = lane * 16;
    float amax = 0.f;
    for (int it = 0;; ++it) {
        int t = tile_of(it);
        if (t >= a.ntiles) break;
        const int cta = t % a.n_ct; t /= a.n_ct;
        const int bx = t % a.tiles_x; t /= a.tiles_x;
        const int by = t % a.tiles_y;
        const int n = t / a.tiles_y;
        // SCATTER: cta = class * (cout tiles) + cout tile; class (py, px) = the output parity this tile writes
        const int nct1 = SCATTER ? a.n_ct >> 2 : a.n_ct;
        const int cls = SCATTER ? cta / nct1 : 0, ct = SCATTER ? cta - cls * nct1 : cta;
        const int cy = cls >> 1, cx = cls & 1;
        const int oy0 = by * H2_TS, ox0 = bx * H2_TS, co0 = ct * H2_BM_OF(MB);

        h2_f16v acc[MB][2];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][f][r] = 0.f;

        for (int chunk = 0; chunk < a.nchunk; ++chunk) {
            H2_BARRIER_CONSUMER();                                  // the stage is published (and the other buffer released)
            const unsigned char* st = smem + (chunk & 1) * H2_STAGE;
            // the stage's tap geometry: tap (ta, tb) reads halo pixel (oyb + ta oys, oxb + tb oxs) + the tile pixel
            //   MODE 0: the 3 x 3 window.   MODE 1 (class of the tile): (py + ta, px + tb).   MODE 2 (class of the stage): (2 - py - ta, ..).
            //   MODE 3 (tile): 1 + py taps, (1 + py - ta): tap 0 = forward tap 0 from g[m + 1], tap 1 = forward tap 2 from g[m]; one tap: tap 1, g[m].
            //   MODE 4 (stage): 1 + py taps, py ? (ta) : (1): plane 0 gives tap 1 at row i; plane 1 gives tap 0 at row i - 1 and tap 2 at row i.
            const int scls = PLANES ? chunk / a.c16 : cls;
            const int spy = scls >> 1, spx = scls & 1;
            const int nx = VTAPS ? 1 + spx : (NT == 9 ? 3 : 2);
            const int oyb = MODE == 1 ? spy : MODE == 2 ? 2 - spy : MODE == 3 ? 1 + spy : MODE == 4 ? 1 - spy : 0;
            const int oxb = MODE == 1 ? spx : MODE == 2 ? 2 - spx : MODE == 3 ? 1 + spx : MODE == 4 ? 1 - spx : 0;
            const int ost = (MODE == 2 || MODE == 3) ? -1 : 1;
            const int nt = cls_nt(scls);                            // (wave-uniform; VTAPS: 1, 2 or 4 of the body's 4 taps run)
            h2_h8 A[2][MB][2], B[2][2][2];                          // [set][fragment][piece]
            auto stage_body = [&](auto ntc_) {
                constexpr int NTc = decltype(ntc_)::value;
                auto load_AB = [&](int tap, h2_h8 (&Ad)[MB][2], h2_h8 (&Bd)[2][2]) {
                    const int ta = NTc == 9 ? tap / 3 : VTAPS ? (nx == 1 ? tap : tap >> 1) : tap >> 1;
                    const int tb = NTc == 9 ? tap - 3 * ta : VTAPS ? (nx == 1 ? 0 : tap & 1) : tap & 1;
                    const unsigned char* pa = st + a_rd + tap * TAPB;
                    const unsigned char* pb = st + b_rd + ((oyb + ta * ost) * H2_HS + (oxb + tb * ost)) * H2_PXB;
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        Bd[f][0] = *reinterpret_cast<const h2_h8*>(pb + 2 * f * H2_HS * H2_PXB);
                        Bd[f][1] = *reinterpret_cast<const h2_h8*>(pb + 2 * f * H2_HS * H2_PXB + 32);
                    }
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        Ad[m][0] = *reinterpret_cast<const h2_h8*>(pa + (2 * m + 0) * 1024);
                        Ad[m][1] = *reinterpret_cast<const h2_h8*>(pa + (2 * m + 1) * 1024);
                    }
                };
                load_AB(0, A[0], B[0]);
#pragma unroll
                for (int tap = 0; tap < NTc; ++tap) {
                    const int s = tap & 1;
                    if (VTAPS && tap >= nt) break;                  // (wave-uniform: this class has 1, 2 or 4 taps)
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap < NTc - 1 && (!VTAPS || tap + 1 < nt)) load_AB(tap + 1, A[s ^ 1], B[s ^ 1]);
                    if (ABL & 4) {                                  // (keep the operands alive without multiplying)
#pragma unroll
                        for (int m = 0; m < MB; ++m) acc[m][0][0] += (float)A[s][m][0][0] + (float)A[s][m][1][0];
#pragma unroll
                        for (int f = 0; f < 2; ++f) acc[0][f][1] += (float)B[s][f][0][0] + (float)B[s][f][1][0];
                    } else {
                        // small products first (fp32 accumulation)
#pragma unroll
                        for (int m = 0; m < MB; ++m)
#pragma unroll
                            for (int f = 0; f < 2; ++f)
                                acc[m][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][m][0], B[s][f][1], acc[m][f], 0, 0, 0);
#pragma unroll
                        for (int m = 0; m < MB; ++m)
#pragma unroll
                            for (int f = 0; f < 2; ++f)
                                acc[m][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][m][1], B[s][f][0], acc[m][f], 0, 0, 0);
#pragma unroll
                        for (int m = 0; m < MB; ++m)
#pragma unroll
                            for (int f = 0; f < 2; ++f)
                                acc[m][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][m][0], B[s][f][0], acc[m][f], 0, 0, 0);
                        // the tap's issue order: the next tap's 2 MB + 4 operand reads (ten at MB = 3) between the first MFMAs
                        if (tap < NTc - 1) {
#pragma unroll
                            for (int k = 0; k < 2 * MB + 4; ++k) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            }
                            __builtin_amdgcn_sched_group_barrier(0x008, 6 * MB - (2 * MB + 4), 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            stage_body(std::integral_constant<int, NT>{});
        }

        // ---- epilogue.  acc[m][f][r]: cout co0 + 32 m + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), pixel row oy0 + 4 wave + 2 f + prow,
        // column ox0 + pcol
        const float inv = 1.0f / (sx * a.w_scale);
        const int OH = MODE == 2 ? d.ho >> 1 : d.ho, OW = MODE == 2 ? d.wo >> 1 : d.wo;      // (MODE 2: `out` is the pooled tensor)
        const int64_t osn = (int64_t)d.cout * OH * OW, osc = (int64_t)OH * OW;
        // (SCATTER: this tile writes the pixels (2 y + cy, 2 x + cx) of the fine grid: every other pixel of every other row)
        const int64_t obase = SCATTER
            ? (int64_t)n * osn + (int64_t)(co0 + 4 * (lane >> 5)) * osc + (int64_t)(2 * (oy0 + 4 * wave + prow) + cy) * OW + 2 * (ox0 + pcol) + cx
            : (int64_t)n * osn + (int64_t)(co0 + 4 * (lane >> 5)) * osc + (int64_t)(oy0 + 4 * wave + prow) * OW + ox0 + pcol;
        constexpr int FROW = SCATTER ? 4 : 2;                     // output rows between a wave's two pixel fragments
        // (the epilogue's mode is decided ONCE, by uniform branches around straight-line copies: tested per element the compiler emits a
        //  branch per store.)  A cout fragment's operands are requested together, before its first store.
        auto epilogue = [&](auto mode, auto with_bias) {
            constexpr int EPI = decltype(mode)::value;             // 0 none, 1 x act'(z), 2 + res
            constexpr bool BIAS = decltype(with_bias)::value;
            const float* op = EPI == 1 ? d.dact_z : d.res;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                float bias[16], opnd[2][16];
#pragma unroll
                for (int r = 0; r < 16; ++r) bias[r] = BIAS ? d.bias[co0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] : 0.f;
                if (EPI != 0) {
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            opnd[f][r] = op[obase + (int64_t)(32 * m + (r & 3) + 8 * (r >> 2)) * osc + (int64_t)(FROW * f) * OW];
                }
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[m][f][r] * inv + bias[r];
                        if (EPI == 1) v *= sda_dact(SDA_ACT_SILU, opnd[f][r]);
                        if (EPI == 2) v += opnd[f][r];
                        amax = fmaxf(amax, fabsf(v));
                        if (!(ABL & 8) || v == 1.2345e30f) d.out[obase + (int64_t)(32 * m + (r & 3) + 8 * (r >> 2)) * osc + (int64_t)(FROW * f) * OW] = v;
                    }
            }
        };
        using h2_c0 = std::integral_constant<int, 0>;
        using h2_c1 = std::integral_constant<int, 1>;
        using h2_c2 = std::integral_constant<int, 2>;
        if (d.dact_z) {
            if (d.bias) epilogue(h2_c1{}, std::true_type{}); else epilogue(h2_c1{}, std::false_type{});
        } else if (d.res) {
            if (d.bias) epilogue(h2_c2{}, std::true_type{}); else epilogue(h2_c2{}, std::false_type{});
        } else {
            if (d.bias) epilogue(h2_c0{}, std::true_type{}); else epilogue(h2_c0{}, std::false_type{});
        }
    }
    if (a.out_amax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_down(amax, off, SDA_WAVE));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(a.out_amax), __float_as_uint(amax));
    }
}

// ---------------------------------------------------------------------------------------------------------------- weight packing
// dst (16-byte units): ((((ct * nchunk + chunk) * 9 + tap) * 3 + m) * 2 + piece) * 64 + lane  ->  8 halves  (chunk: 16 channels):
//   forward  (transpose = 0): W[co = 96 ct + 32 m + (lane & 31)][ci = 16 chunk + 8 (lane >> 5) + i][tap]
//   backward (transpose = 1): the operator of the input VJP -- its "cout" is the forward cin and vice versa, taps flipped:
//                             W[co = 16 chunk + 8 (lane >> 5) + i][ci = 96 ct + 32 m + (lane & 31)][8 - tap]
// A (cout tile, chunk) slab is 54 KiB, contiguous: the LDS image of a stage, copied by LDS-DMA.
__global__ void pack_h2_kernel(const float* __restrict__ w, int cout, int cin, int transpose, float scale, h2_h8* __restrict__ dst,
                               int64_t units, int ntap, int mb) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    const int lane = (int)(u & 63);
    int64_t t = u >> 6;
    const int piece = (int)(t & 1); t >>= 1;
    const int m = (int)(t % mb); t /= mb;                 // (mb = 32-cout fragments per tile: 3, or 2 for rows % 96 != 0)
    const int tap = (int)(t % ntap); t /= ntap;
    const int M = transpose ? cin : cout, K = transpose ? cout : cin;     // operator rows / contraction
    const int nchunk = K / H2_CK;
    const int chunk = (int)(t % nchunk);
    const int ct = (int)(t / nchunk);
    const int row = 32 * mb * ct + 32 * m + (lane & 31);
    h2_h8 out;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = H2_CK * chunk + 8 * (lane >> 5) + i;
        float v = 0.f;
        if (row < M) {
            const int co = transpose ? k : row, ci = transpose ? row : k, tp = transpose ? ntap - 1 - tap : tap;
            v = w[((int64_t)co * cin + ci) * ntap + tp] * scale;
        }
        const _Float16 h = (_Float16)v;
        out[i] = piece ? (_Float16)(v - (float)h) : h;
    }
    dst[u] = out;
}

extern "C" int64_t sda_conv_h2_packed_bytes(int cout, int cin, int transpose) {
    const int M = transpose ? cin : cout, K = transpose ? cout : cin;
    const int mb = M > 0 ? H2_MB_OF(M) : 0;
    if (M <= 0 || K <= 0 || !mb || K % H2_KQ) return 0;
    return (int64_t)(M / H2_BM_OF(mb)) * (K / H2_CK) * (9 * mb * 2 * 1024);
}

extern "C" float sda_conv_h2_scale(float amax) { return h2_scale_of(amax); }

extern "C" int sda_pack_conv_weight_h2(const float* w, int cout, int cin, int transpose, float w_amax, void* dst, void* stream) {
    const int64_t bytes = sda_conv_h2_packed_bytes(cout, cin, transpose);
    if (!w || !dst || bytes == 0) return SDA_E_BADARG;
    const int64_t units = bytes / 16;
    hipLaunchKernelGGL(pack_h2_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cout, cin, transpose,
                       h2_scale_of(w_amax), reinterpret_cast<h2_h8*>(dst), units, 9, H2_MB_OF(transpose ? cin : cout));
    return sda_launch_status();
}

// The up-sampled form (conv_h2_kernel<.., 4>): wsum is [4 classes][cout][cin][4 taps] -- class (py, px), tap (a, b): the sum of the 3 x 3
// taps (dy, dx) that read source pixel (i - 1 + py + a, j - 1 + px + b) of output pixel (2 i + py, 2 j + px), summed in fp32 by the
// caller -- packed like a convolution with 4 cout rows per class: [class * (cout / 96) + cout tile][chunk][tap][m][piece][lane].
extern "C" int64_t sda_conv_h2_up_packed_bytes(int cout, int cin) {
    const int mb = cout > 0 ? H2_MB_OF(cout) : 0;
    if (cout <= 0 || cin <= 0 || !mb || cin % H2_KQ) return 0;
    return (int64_t)4 * (cout / H2_BM_OF(mb)) * (cin / H2_CK) * (4 * mb * 2 * 1024);
}

extern "C" int sda_pack_conv_weight_h2_up(const float* wsum, int cout, int cin, float w_amax, void* dst, void* stream) {
    const int64_t bytes = sda_conv_h2_up_packed_bytes(cout, cin);
    if (!wsum || !dst || bytes == 0) return SDA_E_BADARG;
    const int64_t units = bytes / 16;
    hipLaunchKernelGGL(pack_h2_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wsum, 4 * cout, cin, 0,
                       h2_scale_of(w_amax), reinterpret_cast<h2_h8*>(dst), units, 4, H2_MB_OF(cout));
    return sda_launch_status();
}

// Generic: w is [rows][k][ntap] (fp32), packed [row tile][k chunk][tap][m][piece][lane]; rows % 96 == 0, k % 96 == 0, ntap 4 or 9.  The
// pooled tail VJP (conv_h2_kernel<.., 2>) takes rows = the forward cin, k = 4 classes x the forward cout (class-major), 4 taps:
// w[ci][class * cout + co][2 a + b] = wsum[class][co][ci][2 a + b] of sda_pack_conv_weight_h2_up.
extern "C" int64_t sda_conv_h2_rows_packed_bytes(int rows, int k, int ntap) {
    const int mb = rows > 0 ? H2_MB_OF(rows) : 0;
    if (rows <= 0 || k <= 0 || !mb || k % H2_KQ || (ntap != 1 && ntap != 2 && ntap != 4 && ntap != 9)) return 0;
    return (int64_t)(rows / H2_BM_OF(mb)) * (k / H2_CK) * ((int64_t)ntap * mb * 2 * 1024);
}

extern "C" int sda_pack_conv_weight_h2_rows(const float* w, int rows, int k, int ntap, float w_amax, void* dst, void* stream) {
    const int64_t bytes = sda_conv_h2_rows_packed_bytes(rows, k, ntap);
    if (!w || !dst || bytes == 0) return SDA_E_BADARG;
    const int64_t units = bytes / 16;
    hipLaunchKernelGGL(pack_h2_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, rows, k, 0,
                       h2_scale_of(w_amax), reinterpret_cast<h2_h8*>(dst), units, ntap, H2_MB_OF(rows));
    return sda_launch_status();
}

// max |x| over a contiguous tensor into amax[0] (as uint bits; the caller zeroes it): for inputs whose producer did not report it
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n4, int64_t n, float* __restrict__ amax) {
    float m = 0.f;
    const h2_f4* x4 = reinterpret_cast<const h2_f4*>(x);
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {                 // four 16-byte requests in flight per lane
        h2_f4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(x4 + i + k * stride);
#pragma unroll
        for (int k = 0; k < 4; ++k) m = fmaxf(fmaxf(fmaxf(m, fabsf(v[k][0])), fabsf(v[k][1])), fmaxf(fabsf(v[k][2]), fabsf(v[k][3])));
    }
    for (; i < n4; i += stride) {
        const h2_f4 v = __builtin_nontemporal_load(x4 + i);
        m = fmaxf(fmaxf(fmaxf(m, fabsf(v[0])), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) m = fmaxf(m, fabsf(x[4 * n4 + threadIdx.x]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, SDA_WAVE));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(m));
}

__global__ void absmax_zero_kernel(float* __restrict__ amax) { amax[0] = 0.f; }

extern "C" int sda_absmax(const float* x, int64_t numel, float* amax, void* stream) {
    if (!x || !amax || numel <= 0 || (reinterpret_cast<uintptr_t>(x) & 15)) return SDA_E_BADARG;
    // (zeroed by a KERNEL: a 4-byte hipMemsetAsync captured into a hipGraph was observed to land after the reduction that follows it
    //  on replays that start from an idle device -- amax 0, a scale 2^10 too large, inf in the halves)
    hipLaunchKernelGGL(absmax_zero_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, amax);
    const int64_t n4 = numel / 4;
    const int64_t want = (n4 + 255) / 256;
    const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n4, numel, amax);
    return sda_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------- launcher
static bool h2_ok(const sda_conv_desc* d) {
    if (!d || !d->x || !d->out || !d->w_h2) return false;
    const bool up = d->up_h == 2 && d->up_w == 2;                  // MODE 1 (w_h2 = sda_pack_conv_weight_h2_up's packing)
    const bool pool = d->pool_h == 2 && d->pool_w == 2;            // MODE 2 (sda_pack_conv_weight_h2_rows of the class-major transpose)
    const bool zins = d->zins_h == 2 && d->zins_w == 2;            // MODE 3 (the four classes' sda_pack_conv_weight_h2_rows packings, back to back)
    const bool s2 = d->stride_h == 2 && d->stride_w == 2;          // MODE 4 (the same, forward orientation)
    if (d->kh != 3 || d->kw != 3 || d->explicit_pad || (int)up + (int)pool + (int)zins + (int)s2 > 1) return false;
    if (!(s2 || (d->stride_h == 1 && d->stride_w == 1)) || !(up || (d->up_h == 1 && d->up_w == 1)) ||
        !(zins || (d->zins_h == 1 && d->zins_w == 1)) || !(pool || (d->pool_h <= 1 && d->pool_w <= 1)))
        return false;
    if (d->cctx != 0 || d->n_inner != 1 || d->x_n_off != 0) return false;
    const int ts = (pool || s2) ? 2 * H2_TS : H2_TS;               // source pixels per tile side
    const int mb = d->cout > 0 ? H2_MB_OF(d->cout) : 0;
    if (d->cx % H2_KQ || !mb || d->hs % ts || d->ws % ts) return false;
    if (s2 ? (2 * d->ho != d->hs || 2 * d->wo != d->ws) : (d->ho != ((up || zins) ? 2 : 1) * d->hs || d->wo != ((up || zins) ? 2 : 1) * d->ws)) return false;
    if (up && (d->dact_z || d->act_in != SDA_ACT_NONE)) return false;
    if (pool && (d->dact_z || d->res || d->bias || d->ln_mean || d->act_in != SDA_ACT_NONE)) return false;
    if ((zins || s2) && (d->dact_z || d->ln_mean || d->act_in != SDA_ACT_NONE)) return false;
    if (d->out_sn || d->out_sc || d->out_sy || d->out_sx) return false;
    if ((d->ln_mean == nullptr) != (d->ln_rstd == nullptr)) return false;
    if (d->mod && !d->ln_mean) return false;
    if (d->ln_mean && d->act_in != SDA_ACT_NONE) return false;
    if (d->act_in != SDA_ACT_NONE && d->act_in != SDA_ACT_SILU) return false;
    if (d->dact_z && d->act_d != SDA_ACT_SILU) return false;
    if (d->dact_z && d->res) return false;
    if (d->x_sc < 0 || d->x_sy < 0 || d->x_sx < 0 ||
        (int64_t)d->cx * d->x_sc + (int64_t)d->hs * d->x_sy + (int64_t)d->ws * d->x_sx >= (1LL << 31))
        return false;
    const int64_t tiles = (int64_t)d->n * (d->hs / ts) * (d->ws / ts) * (d->cout / H2_BM_OF(mb)) * ((up || zins) ? 4 : 1);
    return tiles >= 1 && tiles <= 0x3fffffffLL;
}

extern "C" int sda_conv_h2_supported(const sda_conv_desc* d) { return h2_ok(d) ? 1 : 0; }

// one (loader, mode) kernel at the launch's cout tile (96 or 64 couts: MB = 3 / 2)
template <int LOADER, int MODE>
static int h2_launch(const sda_conv_desc* d, const h2_args& a, unsigned grid, int lds, int mb, hipStream_t stream) {
    static bool set3[SDA_MAX_DEVICES], set2[SDA_MAX_DEVICES];
    int rc;
    if (mb == 3) {
        if ((rc = sda_raise_dyn_lds(reinterpret_cast<const void*>(conv_h2_kernel<LOADER, MODE, 0, 3>), lds, set3)) != SDA_OK) return rc;
        hipLaunchKernelGGL((conv_h2_kernel<LOADER, MODE, 0, 3>), dim3(grid), dim3(512), (size_t)lds, stream, *d, a);
    } else {
        if ((rc = sda_raise_dyn_lds(reinterpret_cast<const void*>(conv_h2_kernel<LOADER, MODE, 0, 2>), lds, set2)) != SDA_OK) return rc;
        hipLaunchKernelGGL((conv_h2_kernel<LOADER, MODE, 0, 2>), dim3(grid), dim3(512), (size_t)lds, stream, *d, a);
    }
    return sda_launch_status();
}

extern "C" int sda_conv_h2(const sda_conv_desc* d, void* stream) {
    if (!h2_ok(d)) return SDA_E_UNSUPPORTED;
    h2_args a;
    a.w = d->w_h2;
    a.w_scale = d->w_h2_scale;
    a.x_amax = d->x_amax;
    a.x_amax_static = d->x_amax_static;
    a.out_amax = d->out_amax;
    const bool up = d->up_h == 2, pool = d->pool_h == 2, zins = d->zins_h == 2, s2 = d->stride_h == 2;
    const bool planes = pool || s2;
    a.tiles_x = d->ws / (planes ? 2 * H2_TS : H2_TS);              // (tiles of the grid the kernel walks: the source grid, or one of its parity planes)
    a.tiles_y = d->hs / (planes ? 2 * H2_TS : H2_TS);
    const int mb = H2_MB_OF(d->cout);
    a.n_ct = (d->cout / H2_BM_OF(mb)) * ((up || zins) ? 4 : 1);
    a.c16 = d->cx / H2_CK;
    a.nchunk = a.c16 * (planes ? 4 : 1);
    a.stagger = 0;
#ifdef SDA_H2_ABLATE
    { static const int stg = getenv("SDA_H2_STAGGER") ? atoi(getenv("SDA_H2_STAGGER")) : 0; a.stagger = stg; }
#endif
    if (!(a.w_scale > 0.f) || (!a.x_amax && !(a.x_amax_static > 0.f))) return SDA_E_BADARG;
    const int lds = H2_LDS;
    a.ntiles = (int)((int64_t)d->n * a.tiles_x * a.tiles_y * a.n_ct);
    // persistent workgroups, one per CU (the stage buffers take the whole LDS); a multiple of 8 keeps the XCD-contiguous tile walk
    int cus = sda_cu_count();
    if (cus <= 0) cus = 256;
    unsigned grid = (unsigned)(a.ntiles < cus ? a.ntiles : cus);
    if (grid >= 8) grid &= ~7u;
    int rc = SDA_OK;
#ifdef SDA_H2_ABLATE
    {
        static const int abl_env = getenv("SDA_H2_ABL") ? atoi(getenv("SDA_H2_ABL")) : 0;
        const int abl = mb == 3 ? abl_env : 0;               // (the ablation variants exist for the 96-cout tile)
        static bool seta[16][SDA_MAX_DEVICES];
        const void* fn = nullptr;
#define H2_ABL_CASE(v) case v: fn = reinterpret_cast<const void*>(conv_h2_kernel<0, 0, v>); \
            if (!d->ln_mean && d->act_in == SDA_ACT_NONE) { if ((rc = sda_raise_dyn_lds(fn, lds, seta[v])) != SDA_OK) return rc; \
                hipLaunchKernelGGL((conv_h2_kernel<0, 0, v>), dim3(grid), dim3(512), (size_t)lds, (hipStream_t)stream, *d, a); return sda_launch_status(); } break;
        switch (abl) {
            H2_ABL_CASE(1) H2_ABL_CASE(2) H2_ABL_CASE(3) H2_ABL_CASE(4) H2_ABL_CASE(7) H2_ABL_CASE(8) H2_ABL_CASE(11) H2_ABL_CASE(15)
            default: break;
        }
    }
#endif
    // (the stride-2 heads and their VJP, the tails' VJP: plain loader; the tails: LayerNorm or plain; the block convolutions: all three)
    if (zins) return h2_launch<0, 3>(d, a, grid, lds, mb, (hipStream_t)stream);
    if (s2) return h2_launch<0, 4>(d, a, grid, lds, mb, (hipStream_t)stream);
    if (pool) return h2_launch<0, 2>(d, a, grid, lds, mb, (hipStream_t)stream);
    if (up) return d->ln_mean ? h2_launch<2, 1>(d, a, grid, lds, mb, (hipStream_t)stream) : h2_launch<0, 1>(d, a, grid, lds, mb, (hipStream_t)stream);
    if (d->ln_mean) return h2_launch<2, 0>(d, a, grid, lds, mb, (hipStream_t)stream);
    if (d->act_in != SDA_ACT_NONE) return h2_launch<1, 0>(d, a, grid, lds, mb, (hipStream_t)stream);
    return h2_launch<0, 0>(d, a, grid, lds, mb, (hipStream_t)stream);
}
