// sda_clock_probe -- the shader clock the matrix cores sustain on THIS box, measured by the device itself.
//
// Measurement support for bench.py (prompt section 4: roofline.frac is quoted against the 2.4 GHz peak, and the boxes of the pool
// sustain 2.17-2.29 GHz under the fp32 MFMA stream -- DESIGN.md 5.3b; a slow box must not read as a regression).  No counterpart in
// the reference (it has no measurement code: SURVEY.md section 6).
//
// One workgroup per CU-slot streams independent v_mfma_f32_16x16x4_f32 (the instruction conv_wino4 issues) from registers for `iters`
// rounds; wave 0 of every workgroup brackets the stream with s_memtime (shader clock) and s_memrealtime (constant 100 MHz) and
// stores both differences.  clock = d_cycles / d_ticks x 100 MHz.  Power state is the caller's business: run it right after the
// warm-up steps of the workload it normalises.
#include "sda_common.hpp"

typedef float probe_f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long* __restrict__ out, float* __restrict__ sink, int iters) {
    probe_f4 acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = probe_f4{0.f, 0.f, 0.f, 0.f};
    const float a = 1e-3f * (float)(threadIdx.x & 63), b = 1.0f + 1e-3f * (float)(threadIdx.x >> 6);
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + (float)m, b, acc[m], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    if (s == 12345.678f) sink[0] = s;                     // keeps the stream alive; never true
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = c1 - c0;
        out[2 * blockIdx.x + 1] = w1 - w0;
    }
}

extern "C" int sda_clock_probe(unsigned long long* out, int blocks, float* sink, int iters, void* stream) {
    if (!out || !sink || blocks <= 0 || iters <= 0) return SDA_E_BADARG;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, sink, iters);
    return (int)hipGetLastError();
}
