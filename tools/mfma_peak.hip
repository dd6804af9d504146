// Microbenchmark: fp32 MFMA issue rate and the shader clock it sustains, with and without LDS operand traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDS>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters) {
    __shared__ float sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = (float)(i & 7) * 0.01f;
    __syncthreads();
    f32x16 acc[6];
    for (int m = 0; m < 6; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    long long t0 = __builtin_readcyclecounter();
    long long w0 = wall_clock64();
    const float* p = sm + (threadIdx.x & 63);
    if (LDS == 2) {
        float av[3], bv[2], an[3], bn[2];
        for (int j = 0; j < 3; ++j) av[j] = p[j * 64];
        for (int j = 0; j < 2; ++j) bv[j] = p[(3 + j) * 64 + 4096];
        for (int it = 0; it < iters; ++it) {
            for (int j = 0; j < 3; ++j) an[j] = p[(((it + 1) * 5 + j) & 63) * 64];
            for (int j = 0; j < 2; ++j) bn[j] = p[(((it + 1) * 5 + 3 + j) & 63) * 64 + 4096];
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[m * 2 + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[q], acc[m * 2 + q], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            for (int j = 0; j < 3; ++j) av[j] = an[j];
            for (int j = 0; j < 2; ++j) bv[j] = bn[j];
        }
    } else
    for (int it = 0; it < iters; ++it) {
        float av[3], bv[2];
        if (LDS) {
            for (int j = 0; j < 3; ++j) av[j] = p[((it * 5 + j) & 63) * 64];
            for (int j = 0; j < 2; ++j) bv[j] = p[((it * 5 + 3 + j) & 63) * 64 + 4096];
        } else {
            for (int j = 0; j < 3; ++j) av[j] = a + j;
            for (int j = 0; j < 2; ++j) bv[j] = b + j;
        }
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[m * 2 + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[q], acc[m * 2 + q], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    float s = 0;
    for (int m = 0; m < 6; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main() {
    const int blocks = 256, iters = 200000;
    float* out; long long* clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int lds = 0; lds < 3; ++lds) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (lds == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
            else if (lds) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
            else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            double flops = 2.0 * 32 * 32 * 2 * 6.0 * iters * 4 * blocks;
            printf("lds=%d  %.2f ms  %.1f TFLOP/s  shader cycles %lld  wall ticks %lld  -> clock %.3f GHz (wall 100 MHz)\n", lds, ms,
                   flops / ms / 1e9, h[0], h[1], (double)h[0] / (double)h[1] * 0.1);
        }
    }
    return 0;
}
