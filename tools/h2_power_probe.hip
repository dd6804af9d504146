// Microbenchmark: what the f16 matrix pipe sustains under conv_h2's instruction mix -- v_mfma_f32_32x32x16_f16 from one wave per SIMD
// (+ an idle or VALU-busy partner), with / without the kernel's LDS operand traffic (10 ds_read_b128 per 18 MFMAs), on zeros,
// small integers and full-mantissa random halves.  Reports per variant: time, shader cycles (s_memtime) per MFMA, sustained clock
// (s_memtime / s_memrealtime), TFLOP/s.  The question it answers: is conv_h2 bound by issue or by the power-managed clock?
//   hipcc --offload-arch=gfx950 -O3 tools/h2_power_probe.hip -o /tmp/h2_power_probe && /tmp/h2_power_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// DATA: 0 zeros, 1 small integers, 2 random full-mantissa halves.  LDS: 0 operands stay in registers, 1 re-read from LDS per tap.
// PARTNER: 0 four waves per CU, 1 four more waves that spin on VALU work (the producers' split arithmetic)
template <int LDSR, int PARTNER>
__global__ __launch_bounds__(512) void probe(const h8* __restrict__ src, float* out, long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 64 KiB of operand data in LDS
    for (int i = tid; i < 4096; i += blockDim.x) reinterpret_cast<h8*>(smem)[i] = src[i];
    __syncthreads();
    if (wave >= 4) {
        if (!PARTNER) return;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = (float)src[lane][i];
        for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const _Float16 h = (_Float16)v[i];
                v[i] = (v[i] - (float)h) * 1.0009765625f + 0.37f;
            }
        }
        float s = 0;
        for (int i = 0; i < 8; ++i) s += v[i];
        if (s == 1.2345f) out[tid] = s;
        return;
    }
    f16v acc[6];
    for (int m = 0; m < 6; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const unsigned char* p = smem + lane * 16;
    h8 A[2][6], B[2][4];
    for (int j = 0; j < 6; ++j) A[0][j] = *reinterpret_cast<const h8*>(p + j * 1024);
    for (int j = 0; j < 4; ++j) B[0][j] = *reinterpret_cast<const h8*>(p + (6 + j) * 1024);
    long long t0 = __builtin_readcyclecounter();
    long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tap = 0; tap < 2; ++tap) {
            const int s = tap & 1;
            __builtin_amdgcn_sched_barrier(0);
            if (LDSR) {
                const unsigned char* q = p + ((it * 2 + tap + 1) & 5) * 10240;
                for (int j = 0; j < 6; ++j) A[s ^ 1][j] = *reinterpret_cast<const h8*>(q + j * 1024);
                for (int j = 0; j < 4; ++j) B[s ^ 1][j] = *reinterpret_cast<const h8*>(q + (6 + j) * 1024);
            } else {
                for (int j = 0; j < 6; ++j) A[s ^ 1][j] = A[s][j];
                for (int j = 0; j < 4; ++j) B[s ^ 1][j] = B[s][j];
            }
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc[m * 2 + f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][2 * m], B[s][2 * f + 1], acc[m * 2 + f], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc[m * 2 + f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][2 * m + 1], B[s][2 * f], acc[m * 2 + f], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc[m * 2 + f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s][2 * m], B[s][2 * f], acc[m * 2 + f], 0, 0, 0);
            if (LDSR) {
#pragma unroll
                for (int k = 0; k < 10; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    float s = 0;
    for (int m = 0; m < 6; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

static uint16_t f2h(float f) { union { _Float16 h; uint16_t u; } c; c.h = (_Float16)f; return c.u; }

template <int LDSR, int PARTNER>
static void run(const char* name, const h8* dsrc, float* dout, long long* dclk, int blocks, int iters) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<LDSR, PARTNER>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<LDSR, PARTNER>), dim3(blocks), dim3(512), 160 * 1024, 0, dsrc, dout, dclk, iters);
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe<LDSR, PARTNER>), dim3(blocks), dim3(512), 160 * 1024, 0, dsrc, dout, dclk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    long long h[2]; hipMemcpy(h, dclk, sizeof(h), hipMemcpyDeviceToHost);
    const double mf = 36.0 * iters;                               // MFMAs per wave
    const double fl = mf * 4 * blocks * 32768.0;                  // 32x32x16x2 flops each
    printf("%-44s %8.3f ms  %6.1f cycles/MFMA  clock %5.3f GHz  %7.1f TFLOP/s (%.3f of 2.5 PF)\n", name, ms, h[0] / mf, (double)h[0] / h[1] * 0.1, fl / ms / 1e9,
           fl / ms / 1e9 / 2500);
}

int main() {
    const int blocks = 256, iters = 4000;
    h8* dsrc; float* dout; long long* dclk;
    hipMalloc(&dsrc, 65536); hipMalloc(&dout, blocks * 512 * 4); hipMalloc(&dclk, blocks * 16);
    uint16_t* h = (uint16_t*)malloc(65536);
    for (int data = 0; data < 3; ++data) {
        srand(1);
        for (int i = 0; i < 32768; ++i) {
            float v = data == 0 ? 0.f : data == 1 ? (float)(rand() % 5 - 2) : ((rand() / (float)RAND_MAX) * 2 - 1) * ((i >> 3) & 8 ? 1.0f : 2048.f);
            h[i] = f2h(v);
        }
        hipMemcpy(dsrc, h, 65536, hipMemcpyHostToDevice);
        const char* dn = data == 0 ? "zeros" : data == 1 ? "small ints" : "random halves";
        char name[128];
        snprintf(name, sizeof name, "%s, registers only", dn);              run<0, 0>(name, dsrc, dout, dclk, blocks, iters);
        snprintf(name, sizeof name, "%s, + LDS operand reads", dn);         run<1, 0>(name, dsrc, dout, dclk, blocks, iters);
        snprintf(name, sizeof name, "%s, + LDS reads + VALU partner", dn);  run<1, 1>(name, dsrc, dout, dclk, blocks, iters);
    }
    return 0;
}
