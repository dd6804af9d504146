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
// only every 25 cycles from a single wave -- profiles/r04_bf16x6_prototype.txt, and this kernel's first version: 28 per MFMA.)  Measured against float64 (tools/f16_split_numerics.py): the same 3e-7 relative error as the
// fp32 Winograd kernel.
//
// Structure (one workgroup = 4 waves, one per SIMD, up to 512 registers each; tile = 96 couts x 16 x 16 pixels of one image):
//   * K loop over chunks of 32 input channels.  The chunk's 18 x 18 halo tile lives in LDS as [pixel][hi: 32 halves | lo: 32 halves]
//     (+ 32 B pad per pixel and 16 B per halo row: the ds_read_b128 of a 2 x 16-pixel fragment is conflict-free), double buffered, ONE barrier per
//     chunk.  Every wave fills its quarter of the next chunk's tile while it multiplies the current one: wave w owns channels 8 w .. 8 w + 7
//     -- global loads (padding / wrap resolved once per tile), the loader fusions of the reference's blocks (time modulation +
//     LayerNorm, or the activation; sda/nn.py:28,137-139), the split, two 16-byte LDS stores per pixel.  The f16 MFMA leaves three
//     issue slots per instruction free, and the loader needs ~330 of a chunk's ~1 900.
//   * per tap and chunk a wave multiplies 2 K steps x 3 cout fragments x 2 pixel fragments (two tile rows each) x 3 products = 36
//     MFMAs.  B: eight ds_read_b128 (the tap is a pixel offset into the halo tile).  A: the tap's 12 KiB of packed weights
//     (sda_pack_conv_weight_h2: fragments in lane order) go through a two-slot LDS ring -- every wave fetches a quarter (three
//     global_load_dwordx4, three taps ahead) and stores it two taps ahead, one barrier per tap, twelve ds_read_b128 per wave one tap
//     ahead.  (First version: every wave fetched all twelve fragments itself -- 4 x the L1 traffic, and the kernel ran as slowly with
//     its MFMAs removed: profiles/r05_h2_ablation.txt.)
//   * epilogue: x 1 / (s_x s_w), + bias, x act'(z) or + residual, 64-byte row segments; optionally max |out| (one atomic per wave)
//     so that the NEXT h2 launch knows its input scale without a pass over the tensor.
// Roofline: f16 matrix pipe (2.5 PFLOP/s dense / 3 products); algorithmic bytes: x once per 96-cout tile x 1.27 (halo), out once.
#include "sda_common.hpp"
#include <stdlib.h>
#include <type_traits>

typedef _Float16 h2_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2_h2 __attribute__((ext_vector_type(2)));
typedef float h2_f4 __attribute__((ext_vector_type(4)));
typedef float h2_f16v __attribute__((ext_vector_type(16)));

#define H2_TS 16                       // tile side (pixels)
#define H2_HS (H2_TS + 2)              // halo side
#define H2_NPX (H2_HS * H2_HS)         // 324 halo pixels
#define H2_PXB 160                     // bytes per halo pixel in LDS: 64 hi + 64 lo + 32 pad
#define H2_ROWB (H2_HS * H2_PXB + 16)   // 2 896 B per halo row (row pad: the two rows of a 32-pixel fragment land on different bank quads)
#define H2_DUMMY (H2_HS * H2_ROWB)      // 52 128: where loader lanes without a pixel (slots 324 .. 383) store -- never read
#define H2_TILE (H2_DUMMY + 64 * H2_PXB) // 62 368 B
#define H2_ASLOT (12 * 64 * 16)         // 12 288 B: one tap's weight fragments
#define H2_LDS (2 * H2_TILE + 2 * H2_ASLOT)
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

// ABL (tooling builds, -DSDA_H2_ABLATE + $SDA_H2_ABL; results WRONG): 1 no weight loads in the loop, 2 no loader in the loop, 4 no MFMAs,
// 8 no epilogue stores -- what each costs, measured by leaving it out (tools/h2_check.py --ablate)
template <int LOADER, int ABL = 0>           // LOADER: 0 plain, 1 activation (SiLU), 2 LayerNorm (+ optional modulation)
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
    // (the producer wrote it with device-scope atomics: read it at device scope too -- a scalar-cache line left over from the previous
    //  replay of a captured step is not good enough)
    const float sx = h2_scale_of(a.x_amax ? __hip_atomic_load(a.x_amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.x_amax_static);

    // ---- loader plan of this lane (the same for every chunk): halo pixel p = lane + 64 r, channels 8 wave .. + 7 of the chunk
    const float* ximg = d.x + (int64_t)n * d.x_sn_outer + (int64_t)(8 * wave) * d.x_sc;
    int goff[H2_RND], lds_wr[H2_RND];
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
        lds_wr[r] = (p < H2_NPX ? hy * H2_ROWB + hx * H2_PXB : H2_DUMMY + lane * H2_PXB) + wave * 16;     // (+ 64: the low piece)
        valid |= ok ? (1u << r) : 0u;
        if (LOADER == 2) {
            const int64_t sp = (int64_t)n * d.hs * d.ws + (int64_t)y * d.ws + x;
            mean[r] = d.ln_mean[sp];
            rstd[r] = d.ln_rstd[sp];
        }
    }
    const float* modp = (LOADER == 2 && d.mod) ? d.mod + (int64_t)n * d.mod_sn + 8 * wave : nullptr;

    float raw[2][H2_RND][8];                                        // the loader's values of two chunks: requested a whole chunk ahead
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
            if (LOADER == 1) u = u * sda_sigmoid(u);                 // (SiLU: the launcher admits no other activation here)
            u = ok ? u * sx : 0.f;
            const _Float16 h = (_Float16)u;
            hi[i] = h;
            lo[i] = (_Float16)(u - (float)h);
        }
        *reinterpret_cast<h2_h8*>(buf + lds_wr[r]) = hi;            // (no branch: a conditional store would split the tap's
        *reinterpret_cast<h2_h8*>(buf + lds_wr[r] + 64) = lo;       //  scheduling region)
    };

    // ---- consumer addressing.  B fragment f of tap (dy, dx), K step ks: pixel rows 4 wave + 2 f + ((lane & 31) >> 4) + dy, columns
    // dx + (lane & 15), channels 16 ks + 8 (lane >> 5) ..
    const int b_rd = (4 * wave + ((lane & 31) >> 4)) * H2_ROWB + (lane & 15) * H2_PXB + (lane >> 5) * 16;
    // A: [cout tile][chunk][tap][ks][m][piece][lane] x 16 B
    const h2_h8* wq = reinterpret_cast<const h2_h8*>(a.w) + (int64_t)ct * a.nchunk * (9 * 2 * 3 * 2 * 64) + lane;

    h2_f16v acc[3][2];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][f][r] = 0.f;

    // ---- prologue: chunk 0 into buffer 0 (all its requests in flight together), chunk 1 requested
#pragma unroll
    for (int r = 0; r < H2_RND; ++r) load_round(0, r, raw[0][r]);
    const int c1 = a.nchunk > 1 ? 1 : 0;
#pragma unroll
    for (int r = 0; r < H2_RND; ++r) load_round(c1, r, raw[1][r]);
    h2_h8 A[2][2][3][2], B[2][2][2][2];                             // [set][K step][fragment][piece]
    // weights: global tap g = 9 chunk + tap; its 12 fragments are contiguous in the packing.  This wave's quarter: fragments 3 wave + j.
    unsigned char* aring = smem + 2 * H2_TILE;
    const int gtaps = 9 * a.nchunk;
    h2_h8 aq[3];
    auto fetch_A = [&](int g) {                                     // global -> registers (quarter)
        const h2_h8* p = wq + (int64_t)(g < gtaps ? g : gtaps - 1) * (12 * 64) + 3 * wave * 64;
#pragma unroll
        for (int j = 0; j < 3; ++j) aq[j] = p[j * 64];
    };
    auto stash_A = [&](int slot) {                                  // registers -> ring slot (quarter)
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<h2_h8*>(aring + slot * H2_ASLOT + (3 * wave + j) * 1024 + lane * 16) = aq[j];
    };
    auto load_A = [&](int slot, h2_h8 (&dst)[2][3][2]) {           // ring slot -> operand registers (all twelve)
        const unsigned char* p = aring + slot * H2_ASLOT + lane * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                dst[ks][m][0] = *reinterpret_cast<const h2_h8*>(p + ((ks * 3 + m) * 2 + 0) * 1024);
                dst[ks][m][1] = *reinterpret_cast<const h2_h8*>(p + ((ks * 3 + m) * 2 + 1) * 1024);
            }
    };
    auto load_B = [&](const unsigned char* buf, int tap, h2_h8 (&dst)[2][2][2]) {
        const int dy = tap / 3, dx = tap - 3 * dy;
        const unsigned char* p = buf + b_rd + dy * H2_ROWB + dx * H2_PXB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                dst[ks][f][0] = *reinterpret_cast<const h2_h8*>(p + 2 * f * H2_ROWB + ks * 32);
                dst[ks][f][1] = *reinterpret_cast<const h2_h8*>(p + 2 * f * H2_ROWB + ks * 32 + 64);
            }
    };
    {
        // (every request of the prologue is in flight before the first use: one round trip, not one per operand)
        h2_h8 aq1[3];
        fetch_A(0);
        const h2_h8* p1 = wq + (int64_t)(gtaps > 1 ? 1 : 0) * (12 * 64) + 3 * wave * 64;
#pragma unroll
        for (int j = 0; j < 3; ++j) aq1[j] = p1[j * 64];
        stash_A(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<h2_h8*>(aring + H2_ASLOT + (3 * wave + j) * 1024 + lane * 16) = aq1[j];
    }
#pragma unroll
    for (int r = 0; r < H2_RND; ++r) store_round(0, r, raw[0][r], smem);
    fetch_A(2);
    __syncthreads();
    load_A(0, A[0]);
    __syncthreads();                                               // (slot 0 is read: tap 0 may overwrite it with tap 2's weights)

    // One chunk = nine taps; P = the operand set tap 0 multiplies with (the sets alternate per tap, so a chunk that starts on set 0
    // hands over on set 1: the chunk loop below runs in pairs and every index stays a compile-time constant -- no register moves).
    // The instruction order of a tap is pinned (sched_group_barrier): left alone, the scheduler sinks the next tap's loads to their
    // first use and every tap starts with an exposed L2 round trip (measured: 3 000 cycles per tap instead of 1 250).
    auto chunk_body = [&](auto parity, int chunk) {
        constexpr int P = decltype(parity)::value;
        const unsigned char* cur = smem + (chunk & 1) * H2_TILE;
        unsigned char* nxt = smem + ((chunk + 1) & 1) * H2_TILE;
        // (past the last chunk the loader re-requests / re-stores that chunk into the idle buffer: branch-free)
        const int cn = chunk + 1 < a.nchunk ? chunk + 1 : chunk, cnn = chunk + 2 < a.nchunk ? chunk + 2 : a.nchunk - 1;
        load_B(cur, 0, B[P]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int s = (P + tap) & 1;
            __builtin_amdgcn_sched_barrier(0);
            // weights: tap g + 2's quarter (fetched during the previous tap) into the slot tap g's weights were read from, tap g + 3's
            // requested, tap g + 1's twelve fragments read; B of the next tap (the first tap of the next chunk: after the barrier)
            if (!(ABL & 1)) {
                stash_A(s);
                fetch_A(9 * chunk + tap + 3);
                load_A(s ^ 1, A[s ^ 1]);
            }
            if (tap < 8) load_B(cur, tap + 1, B[s ^ 1]);
            // this wave's share of the tiles ahead: tap r stores round r of the NEXT chunk (requested a whole chunk ago -- an HBM round trip
            // under load is 2-3 us, three taps) and requests round r of the chunk after that into the set this chunk's values came from
            if (tap < H2_RND && !(ABL & 2)) {
                store_round(cn, tap, raw[P ^ 1][tap], nxt);
                load_round(cnn, tap, raw[P][tap]);
            }
            // small products first (fp32 accumulation)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ABL & 4) {                                     // (keep the operands alive without multiplying)
#pragma unroll
                    for (int m = 0; m < 3; ++m) acc[m][0][0] += (float)A[s][ks][m][0][0] + (float)A[s][ks][m][1][0];
#pragma unroll
                    for (int f = 0; f < 2; ++f) acc[0][f][1] += (float)B[s][ks][f][0][0] + (float)B[s][ks][f][1][0];
                    continue;
                }
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        acc[m][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][ks][m][0], B[s][ks][f][1], acc[m][f], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        acc[m][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][ks][m][1], B[s][ks][f][0], acc[m][f], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        acc[m][f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][ks][m][0], B[s][ks][f][0], acc[m][f], 0, 0, 0);
            }
            // ---- the tap's issue order: 36 MFMAs (32 cycles each) with everything else threaded between them: the ring stores first (the
            // barrier at the tap's end publishes them), the next tap's A and B reads, the requests, the loader's arithmetic riding along
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            if (tap < H2_RND) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 36, 0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                       // the ring slot (and, after tap 8, the next chunk's tile) is published
        }
    };
    {
        int chunk = 0;
        for (; chunk + 1 < a.nchunk; chunk += 2) {
            chunk_body(std::integral_constant<int, 0>{}, chunk);
            chunk_body(std::integral_constant<int, 1>{}, chunk + 1);
        }
        if (chunk < a.nchunk) chunk_body(std::integral_constant<int, 0>{}, chunk);
    }

    // ---- epilogue.  acc[m][f][r]: cout co0 + 32 m + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), pixel row oy0 + 4 wave + 2 f + ((lane & 31) >> 4),
    // column ox0 + (lane & 15)
    const float inv = 1.0f / (sx * a.w_scale);
    const int64_t osn = (int64_t)d.cout * d.ho * d.wo, osc = (int64_t)d.ho * d.wo;
    const int64_t obase = (int64_t)n * osn + (int64_t)(co0 + 4 * (lane >> 5)) * osc + (int64_t)(oy0 + 4 * wave + ((lane & 31) >> 4)) * d.wo + ox0 + (lane & 15);
    float amax = 0.f;
    // (the epilogue's mode is decided ONCE, by uniform branches around three straight-line copies: tested per element the
    //  compiler emits a branch per store.)  Every operand of the tile is requested before the first store: one round trip.
    auto epilogue = [&](auto mode, auto with_bias) {
        constexpr int EPI = decltype(mode)::value;                 // 0 none, 1 x act'(z), 2 + res
        constexpr bool BIAS = decltype(with_bias)::value;
        float bias[3][16], opnd[3][2][16];
        const float* op = EPI == 1 ? d.dact_z : d.res;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) bias[m][r] = BIAS ? d.bias[co0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] : 0.f;
            if (EPI != 0) {
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        opnd[m][f][r] = op[obase + (int64_t)(32 * m + (r & 3) + 8 * (r >> 2)) * osc + (int64_t)(2 * f) * d.wo];
            }
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[m][f][r] * inv + bias[m][r];
                    if (EPI == 1) v *= sda_dact(SDA_ACT_SILU, opnd[m][f][r]);
                    if (EPI == 2) v += opnd[m][f][r];
                    amax = fmaxf(amax, fabsf(v));
                    if (!(ABL & 8) || v == 1.2345e30f) d.out[obase + (int64_t)(32 * m + (r & 3) + 8 * (r >> 2)) * osc + (int64_t)(2 * f) * d.wo] = v;
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
    if (a.out_amax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_down(amax, off, SDA_WAVE));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(a.out_amax), __float_as_uint(amax));
    }
}

// ---------------------------------------------------------------------------------------------------------------- weight packing
// dst (16-byte units): (((((ct * nchunk + chunk) * 9 + tap) * 2 + ks) * 3 + m) * 2 + piece) * 64 + lane  ->  8 halves:
//   forward  (transpose = 0): W[co = 96 ct + 32 m + (lane & 31)][ci = 32 chunk + 16 ks + 8 (lane >> 5) + i][tap]
//   backward (transpose = 1): the operator of the input VJP -- its "cout" is the forward cin and vice versa, taps flipped:
//                             W[co = 32 chunk + 16 ks + 8 (lane >> 5) + i][ci = 96 ct + 32 m + (lane & 31)][8 - tap]
__global__ void pack_h2_kernel(const float* __restrict__ w, int cout, int cin, int transpose, float scale, h2_h8* __restrict__ dst,
                               int64_t units) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units) return;
    const int lane = (int)(u & 63);
    int64_t t = u >> 6;
    const int piece = (int)(t & 1); t >>= 1;
    const int m = (int)(t % 3); t /= 3;
    const int ks = (int)(t & 1); t >>= 1;
    const int tap = (int)(t % 9); t /= 9;
    const int M = transpose ? cin : cout, K = transpose ? cout : cin;     // operator rows / contraction
    const int nchunk = K / H2_CK;
    const int chunk = (int)(t % nchunk);
    const int ct = (int)(t / nchunk);
    const int row = H2_BM * ct + 32 * m + (lane & 31);
    h2_h8 out;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = H2_CK * chunk + 16 * ks + 8 * (lane >> 5) + i;
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
    return (int64_t)(M / H2_BM) * (K / H2_CK) * 9 * 2 * 3 * 2 * 64 * 16;
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
    if (d->kh != 3 || d->kw != 3 || d->stride_h != 1 || d->stride_w != 1 || d->explicit_pad || d->up_h != 1 || d->up_w != 1 ||
        d->zins_h != 1 || d->zins_w != 1 || d->pool_h > 1 || d->pool_w > 1)
        return false;
    if (d->cctx != 0 || d->n_inner != 1 || d->x_n_off != 0) return false;
    if (d->cx % H2_CK || d->cout % H2_BM || d->ho != d->hs || d->wo != d->ws || d->ho % H2_TS || d->wo % H2_TS) return false;
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
    const int lds = H2_LDS;
    const unsigned grid = (unsigned)((int64_t)d->n * a.tiles_x * a.tiles_y * a.n_ct);
    int rc = SDA_OK;
#ifdef SDA_H2_ABLATE
    {
        static const int abl = getenv("SDA_H2_ABL") ? atoi(getenv("SDA_H2_ABL")) : 0;
        static bool seta[16][SDA_MAX_DEVICES];
        const void* fn = nullptr;
#define H2_ABL_CASE(v) case v: fn = reinterpret_cast<const void*>(conv_h2_kernel<0, v>); \
            if (!d->ln_mean && d->act_in == SDA_ACT_NONE) { if ((rc = sda_raise_dyn_lds(fn, lds, seta[v])) != SDA_OK) return rc; \
                hipLaunchKernelGGL((conv_h2_kernel<0, v>), dim3(grid), dim3(256), (size_t)lds, (hipStream_t)stream, *d, a); return sda_launch_status(); } break;
        switch (abl) {
            H2_ABL_CASE(1) H2_ABL_CASE(2) H2_ABL_CASE(3) H2_ABL_CASE(4) H2_ABL_CASE(7) H2_ABL_CASE(8) H2_ABL_CASE(15)
            default: break;
        }
    }
#endif
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
