// Row-keyed standard-normal draws for the Langevin corrector (the `z = torch.randn_like(x)` of sda/score.py:257) when the
// batch is sharded over GPUs: element j of trajectory (global row) r in draw d is a function of (seed, r, d, j) only, so
//   * the union over ranks is the same tensor for every world size (1-GPU and N-GPU jobs sample the same trajectories),
//   * a rank generates exactly its own rows (no global draw, nothing proportional to the world size),
//   * the draw index may live in device memory (a step counter), which makes the launch hipGraph-replayable.
// Counter-based Philox4x32-10 (Salmon et al., SC'11; the generator behind torch's device RNG) + Box-Muller.  HBM-bound:
// one 16-byte store per lane and Philox call; no reads.
#include "sda_common.hpp"

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

struct philox4 { uint32_t v[4]; };

__host__ __device__ __forceinline__ philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                          uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c0, p1 = (uint64_t)PHILOX_M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    philox4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// two uniforms in (0, 1) from the top 24 bits of each word (never 0 or 1) -> two independent N(0, 1)
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.28318530717958647692f * u2, &s, &c);
    z0 = r * c; z1 = r * s;
}

// out: [rows][per_row]; quad q of a row (elements 4q .. 4q+3) comes from counter {q_lo, row, draw_lo, draw_hi ^ q_hi << 16}
__global__ __launch_bounds__(256) void randn_rows_kernel(float* __restrict__ out, int rows, int64_t per_row, uint32_t k0,
                                                         uint32_t k1, int64_t row0, int64_t draw, const int64_t* draw_dev,
                                                         int64_t mul, int64_t add) {
    if (draw_dev) draw = draw_dev[0] * mul + add;
    const int64_t quads = (per_row + 3) >> 2;
    const int64_t total = quads * rows;
    const bool vec = (per_row & 3) == 0 && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / quads, q = i - r * quads;
        const uint64_t grow = (uint64_t)(row0 + r);
        const philox4 x = philox4x32_10((uint32_t)q, (uint32_t)grow, (uint32_t)draw,
                                        (uint32_t)((uint64_t)draw >> 32) ^ ((uint32_t)((uint64_t)q >> 32) << 16) ^
                                            ((uint32_t)(grow >> 32) << 24),
                                        k0, k1);
        float z[4];
        box_muller(x.v[0], x.v[1], z[0], z[1]);
        box_muller(x.v[2], x.v[3], z[2], z[3]);
        float* dst = out + r * per_row + 4 * q;
        if (vec) {
            *reinterpret_cast<float4*>(dst) = make_float4(z[0], z[1], z[2], z[3]);
        } else {
            const int64_t left = per_row - 4 * q;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < left) dst[e] = z[e];
        }
    }
}

extern "C" int sda_randn_rows(float* out, int rows, int64_t per_row, uint64_t seed, int64_t row0, int64_t draw,
                              const int64_t* draw_dev, int64_t draw_mul, int64_t draw_add, void* stream) {
    if (!out || rows <= 0 || per_row <= 0 || row0 < 0) return SDA_E_BADARG;
    const int64_t total = ((per_row + 3) >> 2) * rows;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(randn_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, rows, per_row,
                       (uint32_t)seed, (uint32_t)(seed >> 32), row0, draw, draw_dev, draw_mul, draw_add);
    return sda_launch_status();
}

// the raw Philox words (tests: the counter/key schedule is checked bit-for-bit against a numpy restatement)
__global__ void philox_words_kernel(uint32_t* __restrict__ out, int64_t n, uint32_t k0, uint32_t k1, uint32_t c1, uint32_t c2,
                                    uint32_t c3) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const philox4 x = philox4x32_10((uint32_t)i, c1, c2, c3, k0, k1);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[4 * i + e] = x.v[e];
    }
}

extern "C" int sda_philox_words(uint32_t* out, int64_t n, uint64_t seed, uint32_t c1, uint32_t c2, uint32_t c3, void* stream) {
    if (!out || n <= 0) return SDA_E_BADARG;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(philox_words_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n, (uint32_t)seed,
                       (uint32_t)(seed >> 32), c1, c2, c3);
    return sda_launch_status();
}
