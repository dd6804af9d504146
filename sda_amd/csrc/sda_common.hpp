// Shared device helpers for libsda_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sda_hip.h"

#define SDA_WAVE 64

// ---- timing-ablation switches ($SDA_CONV_DEBUG bits: skip the loaders / the multiply / the epilogue stores ...).  They make
// results WRONG, so the product library does not contain them: SDA_DBG() is a compile-time `false` and the environment
// variable is never read unless the library is a tooling build (SDA_EXTRA_HIPCC_FLAGS=-DSDA_ABLATE python -m sda_amd.build
// --force; tools/conv_bench.py and tools/wino4_check.py say when they need one).  tests/test_abi_and_host.py checks that the
// shipped .so does not carry the variable's name.
#if defined(SDA_ABLATE) || defined(SDA_W4_ABLATE) || defined(SDA_W4_VARIANTS)
#include <stdlib.h>
#define SDA_ABLATE_BUILD 1
#define SDA_DBG(g, bits) ((g).debug & (bits))
static inline int sda_debug_env() { const char* e = getenv("SDA_CONV_" "DEBUG"); return e ? atoi(e) : 0; }
#else
#define SDA_DBG(g, bits) (false)
static inline int sda_debug_env() { return 0; }
#endif

static inline int sda_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SDA_OK : (int)e;
}

// ---- per-device launch state.  A process may drive several GPUs (one after the other or from several threads), so what a
// launcher caches -- "this kernel's dynamic-LDS limit is raised", "this many CUs" -- is indexed by the current device.
#define SDA_MAX_DEVICES 64
static inline int sda_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SDA_MAX_DEVICES) return -1;
    return dev;
}
// CU count of the current device (0 on error)
static inline int sda_cu_count() {
    static int cus[SDA_MAX_DEVICES];
    const int dev = sda_current_device();
    if (dev < 0) return 0;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus[dev] = prop.multiProcessorCount;
    }
    return cus[dev];
}
// raise a kernel's dynamic-LDS limit once per device; `done` is the caller's per-kernel (function-local static) flag array
static inline int sda_raise_dyn_lds(const void* kern, int lds, bool (&done)[SDA_MAX_DEVICES]) {
    const int dev = sda_current_device();
    if (dev < 0) return SDA_E_BADARG;
    if (!done[dev]) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        done[dev] = true;
    }
    return SDA_OK;
}

// ---- activations (sda/utils.py:19-25) and their derivatives w.r.t. the pre-activation ----
// sigmoid: on the device exp and the reciprocal are the hardware v_exp_f32 / v_rcp_f32 (1 ulp each; measured
// end-to-end error of SiLU ~1e-7 relative) -- the accurate libm sequences cost ~30 VALU per element, which made the
// conv loaders VALU-bound (and __frcp_rn is a correctly rounded division: a 10-instruction v_div_scale / fmas / fixup
// sequence, not v_rcp_f32).  The host build (emulator) keeps libm.
__host__ __device__ __forceinline__ float sda_sigmoid(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
#else
    return 1.0f / (1.0f + expf(-v));
#endif
}

__host__ __device__ __forceinline__ float sda_act(int a, float v) {
    switch (a) {
        case SDA_ACT_SILU: return v * sda_sigmoid(v);
        case SDA_ACT_RELU: return v > 0.f ? v : 0.f;
        case SDA_ACT_ELU:  return v > 0.f ? v : expm1f(v);
        case SDA_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        case SDA_ACT_SELU: return 1.0507009873554804934193349852946f *
                                  (v > 0.f ? v : 1.6732632423543772848170429916717f * expm1f(v));
        default: return v;
    }
}

__host__ __device__ __forceinline__ float sda_dact(int a, float z) {
    switch (a) {
        case SDA_ACT_SILU: { float s = sda_sigmoid(z); return s * (1.0f + z * (1.0f - s)); }
        case SDA_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case SDA_ACT_ELU:  return z > 0.f ? 1.f : expf(z);
        case SDA_ACT_GELU: return 0.5f * (1.0f + erff(z * 0.70710678118654752440f)) +
                                  z * 0.39894228040143267794f * expf(-0.5f * z * z);
        case SDA_ACT_SELU: return 1.0507009873554804934193349852946f *
                                  (z > 0.f ? 1.f : 1.6732632423543772848170429916717f * expf(z));
        default: return 1.f;
    }
}

// sum_j w[j] v[j] as EIGHT interleaved partial sums (j mod 8), each accumulated in index order, combined as a balanced tree: the
// dependent-add chain is n / 8 long instead of n (a 256-term sequential chain is ~7 000 cycles on one wave: tools/step1d_trace.py).
// THE dot product of the time-embedding layers (sda_time_embed and sda_step1d_prologue share it: identical results).
__host__ __device__ __forceinline__ float sda_dot8(const float* w, const float* v, int n) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j + 8 <= n; j += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] += w[j + i] * v[j + i];
    }
    for (int i = 0; j < n; ++j, ++i) a[i] += w[j] * v[j];
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

// 64-wide wavefront sum (CDNA wave = 64 lanes)
__device__ __forceinline__ float sda_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, SDA_WAVE);
    return v;
}
