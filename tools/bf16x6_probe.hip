// VERDICT r3 item 7 -- split-bf16 ("bf16x6") emulation of the fp32 Winograd multiply: hardware numerics, MFMA rate and the LDS-fed
// consumer stream, measured before anyone writes the kernel.
//   x = hi + mid + lo (three bf16, 24 mantissa bits);  x y ~ hh + hm + mh + hl + lh + mm  (six exact bf16 x bf16 products, fp32 accumulate)
//   v_mfma_f32_16x16x32_bf16: K = 32 = 16 channels x 2 product terms, three instructions per 16 channels:
//     [Uh Uh] x [Vh Vm]   (hh + hm)      [Um Um] x [Vh Vm]   (mh + mm)      [Uh Ul] x [Vl Vh]   (hl + lh)
// Part A  numerics: C = A B, 96 x 64 x 384, against float64 -- the fp32 MFMA (what conv_wino4 issues) beside the six-product form.
// Part B  rate: register-resident MFMA streams, cycles per instruction (one wave per SIMD).
// Part C  the consumer stream fed from LDS, per "stage equivalent" (16 positions x 96 couts x 32 tiles x 8 channels = what one
//         conv_wino4 K-stage multiplies: 3072 cycles of fp32 MFMAs, 3435 measured in the kernel, profiles/r03_wino4_trace.txt), in
//         two register geometries: G1 = conv_wino4's (a wave owns all 16 positions of 48 couts x 16 tiles: 9 A + 2 B 16-byte reads per
//         9 MFMAs) and G2 = positions split over the waves (4 positions of 96 couts x 32 tiles: 18 A + 4 B reads per 36 MFMAs; needs an
//         LDS exchange for the inverse transform).
//   hipcc --offload-arch=gfx950 -O3 tools/bf16x6_probe.hip -o /tmp/bf16x6_probe && /tmp/bf16x6_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__host__ __device__ inline u16 f2bf(float x) {            // round to nearest even
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
__host__ __device__ inline float bf2f(u16 h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__host__ __device__ inline void split3(float x, u16& h, u16& m, u16& l) {
    h = f2bf(x); const float r1 = x - bf2f(h);
    m = f2bf(r1); const float r2 = r1 - bf2f(m);
    l = f2bf(r2);
}

// ---------------------------------------------------------------------------------------------- Part A
// one wave per 16 x 16 output tile; A [M][K], B [K][N] row major fp32
__global__ __launch_bounds__(64) void gemm_f32(const float* A, const float* B, float* C, int M, int N, int K) {
    const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(m0 + i) * K + k + kq], B[(k + kq) * N + n0 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(m0 + 4 * kq + r) * N + n0 + i] = acc[r];
}
__global__ __launch_bounds__(64) void gemm_bf16x6(const float* A, const float* B, float* C, int M, int N, int K, int terms) {
    const int lane = threadIdx.x, i = lane & 15, kg = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 16) {
        // lane (i, kg): k index 8 kg + e <-> term t = kg >> 1, channel c = 8 (kg & 1) + e
        u16 ah[8], am[8], al[8], bh[8], bm[8], bl[8];
        for (int e = 0; e < 8; ++e) {
            const int c = k + 8 * (kg & 1) + e;
            split3(A[(m0 + i) * K + c], ah[e], am[e], al[e]);
            split3(B[c * N + n0 + i], bh[e], bm[e], bl[e]);
        }
        const int t = kg >> 1;
        bf16x8 a1, b1, a2, a3, b3;
        for (int e = 0; e < 8; ++e) {
            a1[e] = __builtin_bit_cast(__bf16, ah[e]);
            b1[e] = __builtin_bit_cast(__bf16, t ? bm[e] : bh[e]);
            a2[e] = __builtin_bit_cast(__bf16, am[e]);
            a3[e] = __builtin_bit_cast(__bf16, t ? al[e] : ah[e]);
            b3[e] = __builtin_bit_cast(__bf16, t ? bh[e] : bl[e]);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);                 // hh + hm
        if (terms >= 6) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc, 0, 0, 0);             // mh + mm
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b3, acc, 0, 0, 0);             // hl + lh
        }
    }
    for (int r = 0; r < 4; ++r) C[(m0 + 4 * kg + r) * N + n0 + i] = acc[r];
}

// ---------------------------------------------------------------------------------------------- Part B / C
template <int MODE>                    // 0 fp32 registers, 1 bf16 registers, 2 fp32 LDS-fed (conv_wino4's step), 3 bf16 G1, 4 bf16 G2
__global__ __launch_bounds__(256) void stream(float* out, long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 36864; i += 256) sm[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    constexpr int NACC = MODE == 4 ? 48 : 48;
    f32x4 acc[NACC];
    for (int m = 0; m < NACC; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {
        const float a = lane * 1e-3f, b = 1.f + wave;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 48; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
    } else if (MODE == 1) {
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(lane * 1e-3f + e); b[e] = (__bf16)(1.f + wave); }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 48; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m], 0, 0, 0);
    } else if (MODE == 2) {
        // conv_wino4's consumer step: per position pair 3 x ds_read_b128 (A: both positions, both K quads) + 2 x ds_read_b64 (B), 12 MFMAs;
        // 8 pairs per stage, 192 accumulators -> here 48 accumulators reused over the 16 positions (the stream's shape, not its values)
        const float* ua = sm + (3 * (wave >> 1) * 64 + lane) * 4;
        const float* va = sm + 24576 + (lane >> 4) * 128 + (16 * (wave & 1) + (lane & 15)) * 2;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                f32x4 av[3];
                float bv[2][2];
#pragma unroll
                for (int m = 0; m < 3; ++m) av[m] = *reinterpret_cast<const f32x4*>(ua + s * 1536 + m * 256);
#pragma unroll
                for (int k4 = 0; k4 < 2; ++k4) { bv[k4][0] = va[s * 512 + 64 * k4]; bv[k4][1] = va[s * 512 + 64 * k4 + 1]; }
#pragma unroll
                for (int k4 = 0; k4 < 2; ++k4)
#pragma unroll
                    for (int m = 0; m < 3; ++m)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int p = (2 * s + h) % 16;
                            acc[p * 3 + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][2 * h + k4], bv[k4][h], acc[p * 3 + m], 0, 0, 0);
                        }
            }
    } else if (MODE == 3) {
        // G1, bf16x6: per position and 16-channel chunk 3 cout fragments x 3 A reads + 2 B reads (16 bytes each), 9 MFMAs.
        // A 16-channel chunk is two conv_wino4 stages: per stage equivalent (8 channels) half of it.
        const float* ua = sm + lane * 4;
        const float* va = sm + 24576 + lane * 4;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                bf16x8 a[3][3], b[2];
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int j = 0; j < 3; ++j) a[m][j] = *reinterpret_cast<const bf16x8*>(ua + ((p * 9 + m * 3 + j) & 95) * 256);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(va + ((p * 2 + j) & 31) * 256);
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    acc[p * 3 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][0], b[0], acc[p * 3 + m], 0, 0, 0);
                    acc[p * 3 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][1], b[0], acc[p * 3 + m], 0, 0, 0);
                    acc[p * 3 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][2], b[1], acc[p * 3 + m], 0, 0, 0);
                }
            }
    } else {
        // G2, bf16x6: a wave owns 4 positions of 96 couts x 32 tiles (6 x 2 fragments, 48 accumulators): per position and 16-channel
        // chunk 18 A + 4 B reads, 36 MFMAs
        const float* ua = sm + lane * 4;
        const float* va = sm + 24576 + lane * 4;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                bf16x8 b[2][2];
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[n][j] = *reinterpret_cast<const bf16x8*>(va + ((p * 4 + n * 2 + j) & 31) * 256);
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    bf16x8 a[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) a[j] = *reinterpret_cast<const bf16x8*>(ua + ((p * 18 + m * 3 + j) & 95) * 256);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const int ai = (p * 6 + m) * 2 + n;
                        acc[ai] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[n][0], acc[ai], 0, 0, 0);
                        acc[ai] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[n][0], acc[ai], 0, 0, 0);
                        acc[ai] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[n][1], acc[ai], 0, 0, 0);
                    }
                }
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int m = 0; m < NACC; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE>
static double run_stream(int iters, int grid) {
    float* out; long long* clk;
    hipMalloc(&out, grid * 256 * sizeof(float)); hipMalloc(&clk, grid * sizeof(long long));
    hipFuncSetAttribute(reinterpret_cast<const void*>(stream<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 36864 * 4);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(stream<MODE>, dim3(grid), dim3(256), 36864 * 4, 0, out, clk, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), clk, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    hipFree(out); hipFree(clk);
    return s / grid / iters;
}

int main() {
    // ---- Part A
    const int M = 96, N = 64, K = 384;
    std::vector<float> A(M * K), B(K * N), C(M * N);
    std::vector<double> R(M * N, 0.0);
    srand(1);
    auto rnd = []() { double u = 0; for (int i = 0; i < 12; ++i) u += rand() / (double)RAND_MAX; return (float)(u - 6.0); };
    for (auto& v : A) v = rnd() * 0.3f;
    for (auto& v : B) v = rnd();
    for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[k * N + j]; R[i * N + j] = s; }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    auto err = [&](const char* name) {
        hipDeviceSynchronize();
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double e = 0, sc = 0;
        for (int i = 0; i < M * N; ++i) { e = fmax(e, fabs(C[i] - R[i])); sc = fmax(sc, fabs(R[i])); }
        printf("A  %-46s max |err| / max |ref| = %.3e\n", name, e / sc);
    };
    hipLaunchKernelGGL(gemm_f32, dim3(N / 16, M / 16), dim3(64), 0, 0, dA, dB, dC, M, N, K); err("fp32 MFMA 16x16x4 (what conv_wino4 issues)");
    hipLaunchKernelGGL(gemm_bf16x6, dim3(N / 16, M / 16), dim3(64), 0, 0, dA, dB, dC, M, N, K, 6); err("bf16 16x16x32, six products (hh hm mh mm hl lh)");
    hipLaunchKernelGGL(gemm_bf16x6, dim3(N / 16, M / 16), dim3(64), 0, 0, dA, dB, dC, M, N, K, 2); err("bf16 16x16x32, two products (hh hm) -- for scale");
    // ---- Part B / C: one workgroup per CU
    const int grid = 256, it = 200;
    const double f32r = run_stream<0>(it, grid) / 48, bfr = run_stream<1>(it, grid) / 48;
    printf("B  register-resident stream, cycles per MFMA (one wave per SIMD): fp32 16x16x4 %.1f, bf16 16x16x32 %.1f\n", f32r, bfr);
    printf("B  per 16 couts x 16 tiles x 32 channels: fp32 8 MFMAs = %.0f cycles, bf16x6 6 MFMAs = %.0f cycles (%.2fx)\n", 8 * f32r, 6 * bfr,
           8 * f32r / (6 * bfr));
    const double c2 = run_stream<2>(it, grid), c3 = run_stream<3>(it, grid) / 2, c4 = run_stream<4>(it, grid) / 2;
    printf("C  LDS-fed consumer stream, cycles per stage equivalent (16 pos x 96 couts x 32 tiles x 8 ch; nothing else on the CU):\n");
    printf("C    fp32, conv_wino4's step (3 x b128 + 4 x b32 reads per 12 MFMAs)         %.0f   (ideal 3072; 3435 inside the kernel)\n", c2);
    printf("C    bf16x6 G1: all 16 positions x 48 couts x 16 tiles per wave (11 reads / 9 MFMAs)   %.0f   (%.2fx the fp32 stream)\n", c3, c2 / c3);
    printf("C    bf16x6 G2: 4 positions x 96 couts x 32 tiles per wave (22 reads / 36 MFMAs)       %.0f   (%.2fx the fp32 stream)\n", c4, c2 / c4);
    return 0;
}
