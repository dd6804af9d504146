// A whole residual MLP (the Lorenz LOCAL score kernel: ScoreNet / ResMLP, sda/nn.py:31-71, sda/score.py:38-63 -- the network of four of
// the five checkpoints of experiments/lorenz/eval.py:33-39) in ONE launch, and its input VJP in one more.  The per-layer path
// (linear.hip: sda_linear / sda_row_ln) was ~20 launches forward + ~20 backward per score evaluation, each a separate pass over
// (rows x 128) activations in HBM.  Rows are independent (LayerNorm is over a row's features), so:
//   * a wave owns 16 ROWS and ALL features of them ("row private"): v_mfma_f32_16x16x4_f32 with D = [feature 16][row 16], eight D
//     fragments = 128 features of the wave's rows in 32 registers.  The K index of a fragment step is a free permutation as long as
//     both operands agree: with k(kq, s) = 16 (s >> 2) + 4 kq + (s & 3) the B operand of step s IS register (s >> 2)[s & 3] of the
//     previous layer's D fragments -- a layer's output feeds the next layer's multiply as it stands: no activation exchange through
//     LDS, no tile stores, no barrier between a GEMM and the next one's input.  (The first version gave a wave 32 output features of a
//     64-row tile: every layer boundary was registers -> LDS -> barrier -> LDS reads, LayerNorm crossed the four waves through LDS,
//     the narrow last layers ran on one wave -- 0.43 of the matrix peak, and a second workgroup per CU does not hide vector-ALU phases
//     under an fp32 MFMA stream: it owns the SIMD's VALU.)
//   * the weights are the A operands: per GEMM one slab [fragment m][k quad sq][lane][4] (element e of lane (kq, li) =
//     W[16 m + li][16 sq + 4 kq + e]; + the bias) staged in LDS -- 64.5 KiB for a 128 x 128 layer, two buffers: the next GEMM's slab is
//     copied (16-byte loads -> 16-byte LDS stores, no vector ALU) in pieces between the current one's multiplies; lanes read A
//     fragments lane-linearly (conflict free), one 16-byte read per four MFMAs; ONE workgroup barrier per GEMM (slab hand-off);
//   * LayerNorm is wave private: a row's 128 features sit in 4 lanes x 32 registers -- lane-local sums + two shuffles;
//   * saved for the VJP (16-byte stores): block inputs, pre-activations, mean / rstd -- as the per-layer path saved.
// Widths are padded: outputs to 16 or 128 features, contraction lengths to 16 / 64 / 128.
// Roofline: the Lorenz local net is 0.34 MFLOP per window and direction; at 62 464 windows (eval.py's batch) 21.5 GFLOP = 0.14 ms of
// fp32 MFMA time per direction.
#include "sda_common.hpp"
#include <stdlib.h>
#include <type_traits>

#define ML_SLAB (8 * 8 * 256)          // floats of the largest slab (128 x 128)
#define ML_PIECE 4096                  // floats per staging piece (1024 float4: four per thread)
#define ML_BIAS 4096                   // floats of the bias region behind the two slab buffers (every GEMM's padded bias)

typedef float ml_f32x4 __attribute__((ext_vector_type(4)));

#ifdef SDA_ML_TRACE                    // tooling (tools/mlp_trace.py): per-phase cycle sums of workgroup 0 / wave 0, kept in scalar registers
__device__ long long ml_trace[16];     // and written once at the kernel's end (a stamp that touches memory drains the loads in flight)
#define ML_T0() long long ml_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long ml_tl = __builtin_readcyclecounter()
#define ML_STAMP(k) do { const long long n_ = __builtin_readcyclecounter(); ml_acc_[k] += n_ - ml_tl; ml_tl = n_; } while (0)
#define ML_DUMP() do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int k_ = 0; k_ < 8; ++k_) ml_trace[k_] += ml_acc_[k_]; } while (0)
extern "C" int sda_ml_trace_read(long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ml_trace), sizeof(long long) * 16) != hipSuccess) return SDA_E_BADARG;
    if (reset) { long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(ml_trace), z, sizeof(z)); }
    return SDA_OK;
}
#else
#define ML_T0() do {} while (0)
#define ML_STAMP(k) do {} while (0)
#define ML_DUMP() do {} while (0)
#endif

// padded sizes: an output width -> 16 or 128 features (1 or 8 D fragments); a contraction length -> 16 / 64 / 128 (1 / 4 / 8 K quads)
__host__ __device__ __forceinline__ int ml_mf(int out_f) { return out_f <= 16 ? 1 : 8; }
__host__ __device__ __forceinline__ int ml_kq(int in_f) { return in_f <= 16 ? 1 : (in_f <= 64 ? 4 : 8); }
// floats of a GEMM's slab in MEMORY: the matrix, zero padded to whole staging pieces
__host__ __device__ __forceinline__ int ml_slab_floats(int in_f, int out_f) { return (ml_mf(out_f) * ml_kq(in_f) * 256 + ML_PIECE - 1) / ML_PIECE * ML_PIECE; }

struct MlCtx {
    int tid, lane, wave, kq, li;
    int64_t row;                       // this lane's row (li of the wave's 16)
    bool rowok;
};

// copies pieces [first, first + n) x 256 float4 of the next slab global -> LDS (16-byte loads, then 16-byte stores: no vector ALU);
// issue() and commit() bracket the multiplies the copy hides behind
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ml_rsrc(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), (short)0, 0x7fffffff, 0x00020000);
}
struct MlStage {
    __amdgpu_buffer_rsrc_t src; unsigned toff;             // the slab as a buffer resource + the thread's byte offset: the piece and register
                                                           // offsets go in the scalar offset operand (a per-thread 64-bit pointer cost 8 VALU per quad)
    ml_f32x4* dst;                                         // (already offset by the thread id)
    int npieces;                                           // whole pieces of 1024 float4 (slabs are padded to that in memory)
    // A piece's four registers are LOCAL to the multiply that stages it (`sv[piece]` in ml_mm), never members that live across multiplies:
    // the loads sit under a run-time condition (this quad has a piece or not), and a register that carries an older value into that
    // condition comes out of it as a phi -- 16 v_mov_b64 per K quad on a SIMD whose vector ALU the MFMAs own.  Undefined on the other path,
    // it is just the load's destination.  (No bounds checks or per-load index arithmetic either; issuing ALL of a slab's loads up front and
    // committing four quads later measured slower: 16 loads in flight per lane.)
    __device__ __forceinline__ void issue(ml_f32x4 (&v)[4], int piece) const {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[i] = __builtin_bit_cast(ml_f32x4, __builtin_amdgcn_raw_buffer_load_b128(src, toff, piece * 16384 + 4096 * i, 0));
    }
    // The LDS stores are inline asm: a conditional LDS instruction the compiler can see makes it lose count of what is outstanding -- it
    // then waits lgkmcnt(0) in front of every quad's MFMAs.  Hidden from it, the count it keeps (the eight A reads) stays exact: hidden
    // stores only add to what is outstanding, so its waits are at worst early; the slab is read only behind the hand-off barrier, whose
    // s_waitcnt lgkmcnt(0) covers the stores.
    __device__ __forceinline__ void commit(const ml_f32x4 (&v)[4], int piece) const {
        const unsigned a = (unsigned)(uintptr_t)(dst + piece * 1024);      // (LDS byte address: the low 32 bits of the generic pointer)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(a), "v"(v[i]), "n"(4096 * i) : "memory");
    }
};

template <int I, int N, class F>
__device__ __forceinline__ void ml_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ml_static_for<I + 1, N>(f);
    }
}

// acc[m] = sum_s A(m, s) h[s >> 2][s & 3] over the KQ K quads; A from the LDS slab `wl` ([m][sq][lane][4]).  The next slab is staged in
// pieces of 1024 float4 between the K quads (`npieces` in all; more pieces than quads: the rest follow the last one).
#ifndef SDA_ML_VAR
#define SDA_ML_VAR 0      // (tooling: 1 = no slab commits, 2 = no A reads, 4 = no slab loads -- timing only, results are wrong)
#endif
template <int MF, int KQ>
__device__ __forceinline__ void ml_mm(const float* wl, const ml_f32x4 (&h)[8], ml_f32x4 (&acc)[8], const ml_f32x4 (&cinit)[8], MlStage& st,
                                      const MlCtx& c, float* sp, const ml_f32x4 (&sreg)[8]) {
    const ml_f32x4* wa = reinterpret_cast<const ml_f32x4*>(wl) + c.lane;
    const int npieces = st.npieces;
    ml_f32x4 A[2][MF];
    ml_f32x4 sv[KQ][4];                                    // (staging registers of piece sq: live from quad sq to quad sq + 1 only)
#pragma unroll
    for (int m = 0; m < MF; ++m) A[0][m] = wa[(m * KQ) * 64];
    ml_static_for<0, KQ>([&](auto SQ_) {
        constexpr int sq = decltype(SQ_)::value;
        // The NEXT quad's A fragments are requested first (a whole quad of MFMAs, 1024 cycles, to arrive; left to itself the scheduler
        // sinks them behind the 27th MFMA and the next quad opens on their latency).  The staging -- commit the piece loaded a quad ago,
        // load this quad's -- sits in the MIDDLE of the quad's MFMAs: the LDS counter retires in order and the compiler does not see the
        // commit's stores, so its wait for these A fragments at the top of the next quad also covers the stores -- half a quad old by then.
        if (sq + 1 < KQ && !(SDA_ML_VAR & 2)) {
#pragma unroll
            for (int m = 0; m < MF; ++m) A[(sq + 1) & 1][m] = wa[(m * KQ + sq + 1) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        auto mfmas = [&](int r) {
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                // (the accumulators start from `cinit` -- the bias -- instead of zero: no add in the epilogue)
                const ml_f32x4 cin = (sq == 0 && r == 0) ? cinit[m] : acc[m];
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[sq & 1][m][r], h[sq][r], cin, 0, 0, 0);
            }
        };
        mfmas(0); mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
#if !(SDA_ML_VAR & 1)
        if constexpr (sq >= 1) { if (sq - 1 < npieces) st.commit(sv[sq - 1], sq - 1); }
#endif
#if !(SDA_ML_VAR & 4)
        if (sq < npieces) st.issue(sv[sq], sq);
#endif
        // one 16-byte store of a saved stream (the block input or the pre-activation, for the VJP) per quad: in a burst in the epilogue
        // the 16 stores per lane queue behind the CU's 64 B/clk store path with nothing else for the wave to do (~11 000 cycles of a
        // tile's 172 000, and the waves reach the hand-off barrier apart); one per 32 MFMAs never queues
        if constexpr (MF == 8 && KQ == 8) { if (sp) *reinterpret_cast<ml_f32x4*>(sp + 16 * sq) = sreg[sq]; }
        __builtin_amdgcn_sched_barrier(0);
        mfmas(2); mfmas(3);
        __builtin_amdgcn_sched_barrier(0);
    });
    if (KQ - 1 < npieces) st.commit(sv[KQ - 1], KQ - 1);
    for (int p = KQ; p < npieces; ++p) { ml_f32x4 t[4]; st.issue(t, p); st.commit(t, p); }
#pragma unroll
    for (int m = MF; m < 8; ++m) acc[m] = ml_f32x4{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ void ml_gemm(const float* wl, int in_f, int out_f, const ml_f32x4 (&h)[8], ml_f32x4 (&acc)[8],
                                        const ml_f32x4 (&cinit)[8], MlStage& st, const MlCtx& c, float* sp, const ml_f32x4 (&sreg)[8]) {
    const int mf = ml_mf(out_f), kq = ml_kq(in_f);
    if (mf == 8) {
        if (kq == 8) ml_mm<8, 8>(wl, h, acc, cinit, st, c, sp, sreg);      // (the only shape that takes a save stream: callers pass
        else if (kq == 4) ml_mm<8, 4>(wl, h, acc, cinit, st, c, nullptr, sreg);   //  sp only for 128 -> 128)
        else ml_mm<8, 1>(wl, h, acc, cinit, st, c, nullptr, sreg);
    } else {
        if (kq == 8) ml_mm<1, 8>(wl, h, acc, cinit, st, c, nullptr, sreg);
        else if (kq == 4) ml_mm<1, 4>(wl, h, acc, cinit, st, c, nullptr, sreg);
        else ml_mm<1, 1>(wl, h, acc, cinit, st, c, nullptr, sreg);
    }
}

// a GEMM's descriptor entries.  They are read from the kernel-argument segment by a run-time index -- scalar loads, ~300 cycles each time
// the loop needs them right away; the loops keep the current and the next GEMM's in registers and fetch the one after next's while a GEMM
// multiplies (14 GEMMs per tile: ~10 000 of a tile's 168 000 cycles were this set-up)
struct MlMeta { int kind, in_f, out_f, b_off, w_off; };
__device__ __forceinline__ MlMeta ml_meta(const sda_mlp_desc& d, int g) {
    g = g < 0 ? 0 : (g >= d.ngemm ? d.ngemm - 1 : g);
    return MlMeta{d.kind[g], d.in_f[g], d.out_f[g], d.b_off[g], d.w_off[g]};
}

__device__ __forceinline__ void ml_ctx(MlCtx& c, const sda_mlp_desc& d) {
    c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6); c.kq = c.lane >> 4; c.li = c.lane & 15;
    c.row = (int64_t)blockIdx.x * 64 + 16 * c.wave + c.li;
    c.rowok = c.row < d.rows;
}

// sum over the features of a row: the row's values sit in the 4 lanes (kq) with this li -- two shuffles
__device__ __forceinline__ float ml_rowsum(float s) {
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    return s;
}

// the wave's rows x `width` features of a row-major source -> D-layout registers h[m][r] = x[row][16 m + 4 kq + r] (zero beyond)
__device__ __forceinline__ void ml_load_rows(const float* src, int64_t ld, int width, const MlCtx& c, ml_f32x4 (&h)[8]) {
    const float* xr = src + (c.rowok ? c.row : 0) * ld;
    const int nm = width <= 16 ? 1 : (width <= 64 ? 4 : 8);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        h[m] = ml_f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < nm) {                                      // (wave uniform)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * m + 4 * c.kq + r;
                const float v = xr[f < width ? f : 0];
                h[m][r] = (c.rowok && f < width) ? v : 0.f;
            }
        }
    }
}

__device__ __forceinline__ void ml_store_rows(float* dst, int64_t ld, int width, const MlCtx& c, const ml_f32x4 (&v)[8]) {
    if (!c.rowok) return;
    float* o = dst + c.row * ld;
    const int nm = width <= 16 ? 1 : 8;
#pragma unroll
    for (int m = 0; m < 8; ++m)
        if (m < nm) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * m + 4 * c.kq + r;
                if (f < width) o[f] = v[m][r];
            }
        }
}

// ---- window mode (MCScoreNet over a ScoreNet kernel, sda/score.py:134-164): the rows are the windows of B trajectories; the gather
// (`unfold`), the concatenation with the time embedding (score.py:57-62), `fold` and the Gaussian-likelihood glue of GaussianScore
// (score.py:387-392) are the loader and the epilogue of the launch -- see sda_mlp_fwd_win / sda_mlp_bwd_win in sda_hip.h.
struct MlWinRow { int b, i; bool first, lastw; };
__device__ __forceinline__ MlWinRow ml_win_row(const sda_mlp_win& w, const MlCtx& c) {
    MlWinRow r;
    const int64_t row = c.rowok ? c.row : 0;
    r.b = (int)(row / w.nw); r.i = (int)(row - (int64_t)r.b * w.nw);
    r.first = r.i == 0; r.lastw = r.i == w.nw - 1;
    return r;
}
// does `fold` read slot j of this window?  (the centre always; the leading slots of a trajectory's first window, the trailing ones of its last)
__device__ __forceinline__ bool ml_win_sel(const MlWinRow& r, int j, int k) { return j == k || (r.first && j < k) || (r.lastw && j > k); }

// ------------------------------------------------------------------------------------------------------------ forward
template <bool WIN>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(const sda_mlp_desc d, const sda_mlp_win w) {
    extern __shared__ __attribute__((aligned(16))) float ml_lds[];         // two slab buffers
    MlCtx c;
    ml_ctx(c, d);
    ML_T0();
    MlStage st;
    // slab 0 -> buffer 0
    st.src = ml_rsrc(d.w + d.w_off[0]); st.toff = 16u * c.tid;
    st.dst = reinterpret_cast<ml_f32x4*>(ml_lds) + c.tid;
    st.npieces = ml_slab_floats(d.in_f[0], d.out_f[0]) / ML_PIECE;
    for (int p = 0; p < st.npieces; ++p) { ml_f32x4 t[4]; st.issue(t, p); st.commit(t, p); }
    // every GEMM's bias -> LDS, once (read per GEMM as the C operand of its first MFMAs: from global memory each first MFMA waited a
    // full L2 round trip -- 14 x ~2 000 cycles of a tile's 120 000 in the GEMM phase)
    float* const bl = ml_lds + 2 * ML_SLAB;
    {
        const int nb = d.b_off[d.ngemm - 1] + 16 * ml_mf(d.out_f[d.ngemm - 1]);
        for (int i = c.tid; i < nb; i += 256) bl[i] = d.bias[i];
    }
    ml_f32x4 h[8], a[8], acc[8], zs[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) zs[m] = ml_f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (WIN) {
        // row (b, i): features [0, WC) = x[b][i .. i + 2k][:] -- WC consecutive floats of the trajectory --, then the time embedding
        const MlWinRow wr = ml_win_row(w, c);
        const int wc = (w.len - w.nw + 1) * w.c;
        const float* xr = w.x + ((int64_t)wr.b * w.len + wr.i) * w.c;
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * m + 4 * c.kq + r;
                const bool isx = f < wc, ise = !isx && f < wc + w.emb_n;
                const float xv = xr[isx ? f : 0], ev = w.emb[ise ? f - wc : 0];
                a[m][r] = !c.rowok ? 0.f : (isx ? xv : (ise ? ev : 0.f));
            }
    } else {
        ml_load_rows(d.x, d.x_ld, d.in_f[0], c, a);
    }
    __syncthreads();
    ML_STAMP(0);                                           // first slab + input rows
    const bool silu = d.act == SDA_ACT_SILU;
    int rb = 0;                                            // residual-block counter (index into the saves)
    MlMeta mc = ml_meta(d, 0), mn = ml_meta(d, 1);
    for (int g = 0; g < d.ngemm; ++g) {
        const MlMeta mm = ml_meta(d, g + 2);
        const float* wl = ml_lds + (g & 1) * ML_SLAB;
        const bool last = g + 1 == d.ngemm;
        st.src = ml_rsrc(d.w + (last ? 0 : mn.w_off)); st.toff = 16u * c.tid;
        st.dst = reinterpret_cast<ml_f32x4*>(ml_lds + ((g + 1) & 1) * ML_SLAB) + c.tid;
        st.npieces = last ? 0 : ml_slab_floats(mn.in_f, mn.out_f) / ML_PIECE;
        const int mf = ml_mf(mc.out_f), cw = mc.in_f;
        // the bias is the C operand of the GEMM's first MFMAs: loaded here, long before it is needed
        ml_f32x4 bias[8];
        {
            const float* bg = bl + mc.b_off + 4 * c.kq;
#pragma unroll
            for (int m = 0; m < 8; ++m) bias[m] = m < mf ? *reinterpret_cast<const ml_f32x4*>(bg + 16 * m) : ml_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (mc.kind == 1) {
            // ---- residual block, first half: save a; h = LN(a)
            const float inv_c = 1.f / (float)cw, inv_v = 1.f / (float)(d.unbiased ? cw - 1 : cw);
            float mean, rstd;
            // FULL: the width fills its fragments (128 of 128): no per-value masks -- vector-ALU instructions are what this kernel's time
            // outside the MFMAs is made of (2.4 per MFMA in the first version, rocprofv3 SQ_INSTS_VALU)
            auto ln = [&](auto FULL_) {
                constexpr bool FULL = decltype(FULL_)::value;
                float s = 0.f;
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s += (FULL || 16 * m + 4 * c.kq + r < cw) ? a[m][r] : 0.f;
                mean = ml_rowsum(s) * inv_c;
                s = 0.f;
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dl = a[m][r] - mean;
                        h[m][r] = dl;
                        s += (FULL || 16 * m + 4 * c.kq + r < cw) ? dl * dl : 0.f;
                    }
                rstd = __builtin_amdgcn_rsqf(ml_rowsum(s) * inv_v + d.eps);
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[m][r] = (FULL || 16 * m + 4 * c.kq + r < cw) ? h[m][r] * rstd : 0.f;
            };
            if (cw == 128) ln(std::true_type{});
            else ln(std::false_type{});
            if (d.mean_save && c.kq == 0 && c.rowok) {
                d.mean_save[(int64_t)rb * d.stat_stride + c.row] = mean;
                d.rstd_save[(int64_t)rb * d.stat_stride + c.row] = rstd;
            }
            ML_STAMP(3);                                   // a_save, LayerNorm
        }
        ML_STAMP(1);                                       // GEMM set-up (stage descriptor, bias fragments)
        // the save streams ride the 128 -> 128 multiplies (one store per K quad, see ml_mm): the block input under the block's first
        // multiply, the pre-activation -- a copy, the accumulators are rewritten -- under its second; other widths store in the epilogue
        const bool ride = d.z_save && cw == 128 && c.rowok;
        if (mc.kind == 0) ml_gemm(wl, mc.in_f, mc.out_f, a, acc, bias, st, c, nullptr, a);
        else if (mc.kind == 1)
            ml_gemm(wl, mc.in_f, mc.out_f, h, acc, bias, st, c,
                    ride ? d.a_save + (int64_t)rb * d.save_stride + c.row * d.save_ld + 4 * c.kq : nullptr, a);
        else
            ml_gemm(wl, mc.in_f, mc.out_f, h, acc, bias, st, c,
                    ride ? d.z_save + (int64_t)rb * d.save_stride + c.row * d.save_ld + 4 * c.kq : nullptr, zs);
        ML_STAMP(2);                                       // GEMM (+ staging)
        __syncthreads();                                   // slab hand-off: the next slab is complete, this one is free
        ML_STAMP(5);                                       // hand-off barrier
        if (mc.kind == 0) {
#pragma unroll
            for (int m = 0; m < 8; ++m) a[m] = acc[m];
        } else if (mc.kind == 1) {
            // z = W1 LN(a) + b1 (saved); h = act(z)
            // the saves (block input a, pre-activation z) go out HERE, behind the GEMM whose slab staging has just completed: vmcnt
            // retires in order, so a store issued in front of staging loads makes the wait for those loads a wait for the store's
            // round trip to HBM (the block input written before the GEMM cost the forward ~20 %)
            if (d.z_save && c.rowok && mc.out_f != 128) {
                float* zp = d.z_save + (int64_t)rb * d.save_stride + c.row * d.save_ld + 4 * c.kq;
                float* as = d.a_save + (int64_t)rb * d.save_stride + c.row * d.save_ld + 4 * c.kq;
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    if (m < mf) {
                        *reinterpret_cast<ml_f32x4*>(zp + 16 * m) = acc[m];
                        *reinterpret_cast<ml_f32x4*>(as + 16 * m) = a[m];
                    }
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) zs[m] = acc[m];
            auto epi = [&](auto SILU_) {
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[m][r] = decltype(SILU_)::value ? sda_act(SDA_ACT_SILU, acc[m][r]) : sda_act(d.act, acc[m][r]);
            };
            if (silu) epi(std::true_type{});
            else epi(std::false_type{});
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) a[m] += acc[m];
            ++rb;
        }
        ML_STAMP(4);                                       // epilogue
        mc = mn; mn = mm;
    }
    if constexpr (WIN) {
        // fold (score.py:155-164) + eps = (cx0 + cx1 sigma) x + cn s + the likelihood cotangent, as sda_net1d_fwd_fused's epilogue
        if (c.rowok) {
            const MlWinRow wr = ml_win_row(w, c);
            const int k = (w.len - w.nw) / 2, wc = (2 * k + 1) * w.c;
            const float mu = w.coef[0], sg = w.coef[1];
            const bool bare = w.cx0 == 0.f && w.cx1 == 0.f && w.cn == 1.f;
            const float cx = w.cx0 + w.cx1 * sg;
            const float rr = __fdiv_rn(sg, mu);
            const float var = __fadd_rn(__fmul_rn(w.std, w.std), __fmul_rn(w.gamma, __fmul_rn(rr, rr)));
            const int n_oc = (w.c_stop - w.c_start + w.c_step - 1) / w.c_step;
            const float* yb = w.y + (int64_t)wr.b * w.y_sn;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 4 * c.kq + r;
                if (f >= wc) continue;
                const int j = f / w.c, ch = f - j * w.c;
                if (!ml_win_sel(wr, j, k)) continue;
                const int ps = wr.i + j;
                const int64_t o = ((int64_t)wr.b * w.len + ps) * w.c + ch;
                const float xv = w.x[o];
                const float ov = a[0][r];
                const float e = bare ? ov : (xv * cx) + (w.cn * ov);
                w.eps[o] = e;
                const int crel = ch - w.c_start, prel = ps - w.p_start;
                float gv = 0.f;
                if (crel >= 0 && ch < w.c_stop && crel % w.c_step == 0 && prel >= 0 && ps < w.p_stop && prel % w.p_step == 0) {
                    const float xh = (xv - sg * e) / mu;
                    gv = __fdiv_rn(yb[(prel / w.p_step) * n_oc + crel / w.c_step] - xh, var);
                }
                w.ghat[o] = gv;
            }
        }
    } else {
        ml_store_rows(d.out, d.out_ld, d.out_f[d.ngemm - 1], c, a);
    }
    ML_STAMP(6);                                           // output
    ML_DUMP();
}

// ------------------------------------------------------------------------------------------------------------ input VJP
// d.w = the slabs of the TRANSPOSED matrices (backward GEMM of forward GEMM g: out_f[g] -> in_f[g], no bias), same offsets table;
// x = cotangent rows (width out_f[last]), out = input-gradient rows (width in_f[0]); the GEMM list is walked backwards.
template <bool WIN>
__global__ __launch_bounds__(256) void mlp_bwd_kernel(const sda_mlp_desc d, const sda_mlp_win w) {
    extern __shared__ __attribute__((aligned(16))) float ml_lds[];
    MlCtx c;
    ml_ctx(c, d);
    const int gl = d.ngemm - 1;
    MlStage st;
    st.src = ml_rsrc(d.w + d.w_off[gl]); st.toff = 16u * c.tid;
    st.dst = reinterpret_cast<ml_f32x4*>(ml_lds) + c.tid;
    st.npieces = ml_slab_floats(d.out_f[gl], d.in_f[gl]) / ML_PIECE;
    for (int p = 0; p < st.npieces; ++p) { ml_f32x4 t[4]; st.issue(t, p); st.commit(t, p); }
    ml_f32x4 h[8], gacc[8], acc[8], zero[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) zero[m] = ml_f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (WIN) {
        // the cotangent of the window outputs = fold's adjoint of cn ghat: slot j of window (b, i) receives ghat[b][i + j] where fold reads it
        const MlWinRow wr = ml_win_row(w, c);
        const int k = (w.len - w.nw) / 2, wc = (2 * k + 1) * w.c;
#pragma unroll
        for (int m = 0; m < 8; ++m) gacc[m] = ml_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 4 * c.kq + r, fc = f < wc ? f : 0;
            const int j = fc / w.c, ch = fc - j * w.c;
            const bool sel = c.rowok && f < wc && ml_win_sel(wr, j, k);
            const float gv = w.ghat[sel ? ((int64_t)wr.b * w.len + wr.i + j) * w.c + ch : 0];
            gacc[0][r] = sel ? gv * w.cn : 0.f;
        }
    } else {
        ml_load_rows(d.x, d.x_ld, d.out_f[gl], c, gacc);
    }
    __syncthreads();
    const bool silu = d.act == SDA_ACT_SILU;
    int rb = 0;
    for (int g = 0; g < d.ngemm; ++g) rb += d.kind[g] == 2;
    int buf = 0;
    MlMeta mc = ml_meta(d, gl), mn = ml_meta(d, gl - 1);
    for (int g = gl; g >= 0; --g, buf ^= 1) {
        const MlMeta mm = ml_meta(d, g - 2);
        const float* wl = ml_lds + buf * ML_SLAB;
        const bool last = g == 0;
        st.src = ml_rsrc(d.w + (last ? 0 : mn.w_off)); st.toff = 16u * c.tid;
        st.dst = reinterpret_cast<ml_f32x4*>(ml_lds + (buf ^ 1) * ML_SLAB) + c.tid;
        st.npieces = last ? 0 : ml_slab_floats(mn.out_f, mn.in_f) / ML_PIECE;
        if (mc.kind == 2) --rb;
        // what the epilogue reads from the forward: issued before the multiply
        const int cw = mc.in_f, nm = ml_mf(cw);
        const int64_t srow = c.rowok ? c.row : 0;
        ml_f32x4 sv[8];                                    // kind 2: z; kind 1: the block input a
        float mean = 0.f, rs = 0.f;
        if (mc.kind != 0) {
            const float* sp = (mc.kind == 2 ? d.z_save : d.a_save) + (int64_t)rb * d.save_stride + srow * d.save_ld + 4 * c.kq;
#pragma unroll
            for (int m = 0; m < 8; ++m) sv[m] = m < nm ? *reinterpret_cast<const ml_f32x4*>(sp + 16 * m) : ml_f32x4{0.f, 0.f, 0.f, 0.f};
            if (mc.kind == 1) {
                mean = d.mean_save[(int64_t)rb * d.stat_stride + srow];
                rs = d.rstd_save[(int64_t)rb * d.stat_stride + srow];
            }
        }
        // the multiply's input: the cotangent g itself (Linear; a block's second half) or q (its first half)
        if (mc.kind == 1) ml_gemm(wl, mc.out_f, mc.in_f, h, acc, zero, st, c, nullptr, zero);
        else ml_gemm(wl, mc.out_f, mc.in_f, gacc, acc, zero, st, c, nullptr, zero);
        __syncthreads();
        if (mc.kind == 0) {
#pragma unroll
            for (int m = 0; m < 8; ++m) gacc[m] = acc[m];
        } else if (mc.kind == 2) {
            // q = W2^T g, x act'(z)   (features beyond the width: acc = 0 there, so q = 0 x act'(0) = 0)
            auto dact = [&](auto SILU_) {
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        h[m][r] = acc[m][r] * (decltype(SILU_)::value ? sda_dact(SDA_ACT_SILU, sv[m][r]) : sda_dact(d.act, sv[m][r]));
            };
            if (silu) dact(std::true_type{});
            else dact(std::false_type{});
        } else {
            // gh = W1^T q; g += LN^T(gh) = rstd (gh - mean_c(gh) - x_hat mean'_c(gh x_hat))
            const float inv_c = 1.f / (float)cw, inv_v = 1.f / (float)(d.unbiased ? cw - 1 : cw);
            auto lnb = [&](auto FULL_) {
                constexpr bool FULL = decltype(FULL_)::value;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool fok = FULL || 16 * m + 4 * c.kq + r < cw;
                        const float xh = fok ? (sv[m][r] - mean) * rs : 0.f;
                        sv[m][r] = xh;
                        const float gv = fok ? acc[m][r] : 0.f;
                        s1 += gv; s2 += gv * xh;
                    }
                const float av_ = ml_rowsum(s1) * inv_c, bv_ = ml_rowsum(s2) * inv_v;
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool fok = FULL || 16 * m + 4 * c.kq + r < cw;
                        gacc[m][r] += fok ? rs * (acc[m][r] - av_ - sv[m][r] * bv_) : 0.f;
                    }
            };
            if (cw == 128) lnb(std::true_type{});
            else lnb(std::false_type{});
        }
        mc = mn; mn = mm;
    }
    if constexpr (WIN) {
        // the window part of the input gradient, 16 floats per row (the embedding's part is not formed); sda_mc_finish sums the overlaps
        if (c.rowok) *reinterpret_cast<ml_f32x4*>(w.gwin + c.row * 16 + 4 * c.kq) = gacc[0];
    } else {
        ml_store_rows(d.out, d.out_ld, d.in_f[0], c, gacc);
    }
}

static int mlp_check(const sda_mlp_desc* d, bool bwd, bool win) {
    if (!d || d->rows < 1 || d->ngemm < 1 || d->ngemm > SDA_MLP_MAXG) return SDA_E_UNSUPPORTED;
    if ((!win && (!d->x || !d->out)) || !d->w || (!bwd && !d->bias)) return SDA_E_BADARG;
    int nres = 0;
    for (int g = 0; g < d->ngemm; ++g) {
        if (d->in_f[g] < 1 || d->out_f[g] < 1 || d->in_f[g] > 128 || d->out_f[g] > 128 || d->kind[g] < 0 || d->kind[g] > 2) return SDA_E_UNSUPPORTED;
        if (g > 0 && d->in_f[g] != d->out_f[g - 1]) return SDA_E_BADARG;
        if (d->kind[g] == 1) {
            if (g + 1 >= d->ngemm || d->kind[g + 1] != 2 || d->in_f[g] != d->out_f[g] || d->out_f[g + 1] != d->in_f[g]) return SDA_E_BADARG;
            if (d->unbiased && d->in_f[g] < 2) return SDA_E_UNSUPPORTED;
            ++nres;
        }
        if (d->kind[g] == 2 && (g == 0 || d->kind[g - 1] != 1)) return SDA_E_BADARG;
        if ((d->w_off[g] & 3) || (d->b_off[g] & 3)) return SDA_E_BADARG;
    }
    if ((reinterpret_cast<uintptr_t>(d->w) & 15) || (!bwd && (reinterpret_cast<uintptr_t>(d->bias) & 15))) return SDA_E_BADARG;
    if (!bwd && d->b_off[d->ngemm - 1] + 16 * ml_mf(d->out_f[d->ngemm - 1]) > ML_BIAS) return SDA_E_UNSUPPORTED;
    const bool saves = d->a_save && d->z_save && d->mean_save && d->rstd_save;
    if (nres > 0) {
        if (bwd && !saves) return SDA_E_BADARG;
        if (!bwd && (d->a_save || d->z_save || d->mean_save || d->rstd_save) && !saves) return SDA_E_BADARG;
        if (saves && (d->save_ld < 128 || (d->save_ld & 3) || (reinterpret_cast<uintptr_t>(d->a_save) & 15) ||
                      (reinterpret_cast<uintptr_t>(d->z_save) & 15) || (d->save_stride & 3)))
            return SDA_E_BADARG;
    }
    return SDA_OK;
}

static int mlp_win_check(const sda_mlp_desc* d, const sda_mlp_win* w, bool bwd) {
    if (!w || w->nw < 1 || w->c < 1 || w->len < w->nw || ((w->len - w->nw) & 1)) return SDA_E_BADARG;
    const int wc = (w->len - w->nw + 1) * w->c;
    if (wc > 16) return SDA_E_UNSUPPORTED;                 // (the window values of a row live in one D fragment)
    if (d->rows % w->nw || !w->ghat || !w->coef) return SDA_E_BADARG;
    if (d->out_f[d->ngemm - 1] != wc) return SDA_E_BADARG;
    if (!bwd) {
        if (!w->x || !w->eps || !w->y || w->emb_n < 0 || (w->emb_n > 0 && !w->emb) || d->in_f[0] != wc + w->emb_n) return SDA_E_BADARG;
        if (w->p_step < 1 || w->c_step < 1 || w->p_start < 0 || w->c_start < 0 || w->p_stop > w->len || w->c_stop > w->c ||
            w->p_stop <= w->p_start || w->c_stop <= w->c_start)
            return SDA_E_BADARG;
    } else {
        if (!w->gwin || (reinterpret_cast<uintptr_t>(w->gwin) & 15) || d->in_f[0] < wc) return SDA_E_BADARG;
    }
    return SDA_OK;
}

template <bool BWD, bool WIN>
static int mlp_launch(const sda_mlp_desc* d, const sda_mlp_win* w, hipStream_t stream) {
    int rc = mlp_check(d, BWD, WIN);
    if (rc != SDA_OK) return rc;
    if (WIN && (rc = mlp_win_check(d, w, BWD)) != SDA_OK) return rc;
    const int64_t tiles = ((int64_t)d->rows + 63) / 64;
    if (tiles > 0x7fffffffLL) return SDA_E_UNSUPPORTED;
    constexpr int lds = (2 * ML_SLAB + ML_BIAS) * 4;
    static bool raised[SDA_MAX_DEVICES];
    const void* kern = BWD ? reinterpret_cast<const void*>(mlp_bwd_kernel<WIN>) : reinterpret_cast<const void*>(mlp_fwd_kernel<WIN>);
    const int rr = sda_raise_dyn_lds(kern, lds, raised);
    if (rr != SDA_OK) return rr;
    sda_mlp_win wv = {};
    if (WIN) wv = *w;
#ifdef SDA_ML_TRACE
    sda_mlp_desc dd = *d;
    if (!BWD && getenv("SDA_ML_DBG")) {                     // tooling: which save stream costs what (results of a later VJP are wrong)
        const int b = atoi(getenv("SDA_ML_DBG"));
        if (b & 1) dd.a_save = nullptr;
        if (b & 2) dd.z_save = nullptr;
        if (b & 4) { dd.mean_save = nullptr; dd.rstd_save = nullptr; }
    }
    d = &dd;
#endif
    if (BWD) hipLaunchKernelGGL(mlp_bwd_kernel<WIN>, dim3((unsigned)tiles), dim3(256), lds, stream, *d, wv);
    else hipLaunchKernelGGL(mlp_fwd_kernel<WIN>, dim3((unsigned)tiles), dim3(256), lds, stream, *d, wv);
    return sda_launch_status();
}

extern "C" int sda_mlp_fwd(const sda_mlp_desc* d, void* stream) { return mlp_launch<false, false>(d, nullptr, (hipStream_t)stream); }
extern "C" int sda_mlp_bwd(const sda_mlp_desc* d, void* stream) { return mlp_launch<true, false>(d, nullptr, (hipStream_t)stream); }
extern "C" int sda_mlp_fwd_win(const sda_mlp_desc* d, const sda_mlp_win* w, void* stream) { return mlp_launch<false, true>(d, w, (hipStream_t)stream); }
extern "C" int sda_mlp_bwd_win(const sda_mlp_desc* d, const sda_mlp_win* w, void* stream) { return mlp_launch<true, true>(d, w, (hipStream_t)stream); }
// floats of GEMM (in_f -> out_f)'s slab, for the packer
extern "C" int sda_mlp_slab_floats(int in_f, int out_f) {
    if (in_f < 1 || out_f < 1 || in_f > 128 || out_f > 128) return SDA_E_UNSUPPORTED;
    return ml_slab_floats(in_f, out_f);
}
