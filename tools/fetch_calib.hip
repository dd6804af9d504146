// FETCH_SIZE / WRITE_SIZE calibration on known byte counts, in the access shapes conv_wino4 uses.
//
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/fetch -o p -- /tmp/fetch_calib
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/write -o p -- /tmp/fetch_calib
//   python tools/fetch_calib_post.py out        -> bytes the counters report per kernel vs the bytes below
//
// Every kernel touches each byte of a 1.5 GiB buffer region at most once (nothing to hit in L2 / the 256 MiB Infinity Cache),
// launched 3 times (rotating regions so that launch k does not find launch k-1's lines in the Infinity Cache).
//   stream16      : 16 B per lane, wave-contiguous (global_load_dwordx4)      -- the guide's reference shape (FETCH = 1/2)
//   stream4       : 4 B per lane, wave-contiguous (global_load_dword, 256 B per wave instruction)
//   halo_iso      : conv_wino4's halo read: a workgroup-sized unit reads the 10 x 18-float halo of a 16 x 8-pixel block of a
//                   W-wide plane, lane l -> halo position l + 64 i (i = 0..2), row start at float 16 bx - 1 (byte 64 bx - 4);
//                   ISOLATED blocks (every 4th column, every 2nd block row, distinct planes): no segment is shared between
//                   two blocks.  Useful bytes 720 per (block, plane); 64-B segments touched 3 x 10 = 1920 B; 128-B lines
//                   touched 2 x 10 = 2560 B.
//   halo_dense    : the same read over ALL blocks of each plane in raster order by consecutive workgroups (neighbours share
//                   segments): what an ideal L2 turns into ~1.41 x the plane (18 x 10 / (16 x 8)), i.e. fetched once = 1.0 x.
//   store8        : conv_wino4's output write: 8-byte stores, lane -> (2 x 2 block of a 16 x 8 tile), 64-B row segments
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void stream16(const float4* __restrict__ src, size_t n16, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1234.5678f) *sink = acc;
}

__global__ void stream4(const float* __restrict__ src, size_t n4, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 1234.5678f) *sink = acc;
}

// one WAVE per (block, plane) unit, unit u -> (plane, by, bx) in raster order.  The units are dealt out the way conv_wino4 deals
// out blocks: XCD (blockIdx & 7) owns a contiguous eighth of the list, and the waves of that XCD's workgroups walk it
// interleaved, so at any moment an XCD works on consecutive units.
template <int DENSE>
__global__ void halo_read(const float* __restrict__ src, int W, int H, int planes, int bx_step, int by_step, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int bxn = W / 16 / bx_step, byn = H / 8 / by_step;
    const long units = (long)planes * bxn * byn;
    const long ubeg = units * xcd / 8, uend = units * (xcd + 1) / 8;
    float acc = 0.f;
    for (long u = ubeg + slot * wpb + wave; u < uend; u += (long)per_xcd * wpb) {
        const int bx = (int)(u % bxn) * bx_step, by = (int)((u / bxn) % byn) * by_step;
        const long plane = u / ((long)bxn * byn);
        const float* p = src + plane * (long)W * H;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pos = lane + 64 * i;
            if (pos < 180) {
                const int hy = pos / 18, hx = pos - 18 * hy;
                int y = 8 * by - 1 + hy, x = 16 * bx - 1 + hx;
                y = y < 0 ? y + H : (y >= H ? y - H : y);
                x = x < 0 ? x + W : (x >= W ? x - W : x);
                acc += p[(long)y * W + x];
            }
        }
    }
    if (acc == 1234.5678f) *sink = acc;
}

// conv_wino4's store shape: a wave writes, per cout plane, the 2 x 2 outputs of 16 tiles (8-byte stores, two rows)
__global__ void store8(float* __restrict__ dst, int W, int H, int planes) {
    const int lane = threadIdx.x & 63;
    const int wave_g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int waves = (gridDim.x * blockDim.x) >> 6;
    const int bxn = W / 16, byn = H / 8;
    // unit = (plane group of 4, block, half): lanes (kq = lane >> 4 -> plane, li = lane & 15 -> tile of the half)
    const long units = (long)(planes / 4) * bxn * byn * 2;
    for (long u = wave_g; u < units; u += waves) {
        const int half = (int)(u & 1);
        const long b = u >> 1;
        const int bx = (int)(b % bxn), by = (int)((b / bxn) % byn);
        const long pg = b / ((long)bxn * byn);
        const int t = 16 * half + (lane & 15);
        const int oy = 8 * by + 2 * (t >> 3), ox = 16 * bx + 2 * (t & 7);
        float* p = dst + (pg * 4 + (lane >> 4)) * (long)W * H + (long)oy * W + ox;
        *reinterpret_cast<float2*>(p) = make_float2(1.f, 2.f);
        *reinterpret_cast<float2*>(p + W) = make_float2(3.f, 4.f);
    }
}

int main() {
    const size_t region = (size_t)1536 << 20;              // 1.5 GiB per launch
    const int reps = 3;
    char* buf = nullptr;
    float* sink = nullptr;
    CHECK(hipMalloc(&buf, region * reps));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 0, region * reps));
    CHECK(hipDeviceSynchronize());
    const int W = 256, H = 256;
    const int planes = (int)(region / ((size_t)W * H * 4));   // 6144 planes of 256 KiB
    printf("region_bytes %zu planes %d W %d H %d\n", region, planes, W, H);
    printf("expect stream16 %zu\nexpect stream4 %zu\n", region, region);
    // isolated: every 4th block column, every 2nd block row
    const long iso_units = (long)planes * (W / 16 / 4) * (H / 8 / 2);
    printf("expect halo_iso useful %ld seg64 %ld line128 %ld\n", iso_units * 720, iso_units * 1920, iso_units * 2560);
    printf("expect halo_dense plane_bytes %zu requested %ld\n", region, (long)planes * (W / 16) * (H / 8) * 720);
    printf("expect store8 %zu\n", region);
    for (int r = 0; r < reps; ++r) {
        char* base = buf + (size_t)r * region;
        hipLaunchKernelGGL(stream16, dim3(2048), dim3(256), 0, 0, (const float4*)base, region / 16, sink);
        hipLaunchKernelGGL(stream4, dim3(2048), dim3(256), 0, 0, (const float*)base, region / 4, sink);
    }
    for (int r = 0; r < reps; ++r) {
        char* base = buf + (size_t)((r + 1) % reps) * region;
        hipLaunchKernelGGL(halo_read<0>, dim3(2048), dim3(256), 0, 0, (const float*)base, W, H, planes, 4, 2, sink);
    }
    for (int r = 0; r < reps; ++r) {
        char* base = buf + (size_t)((r + 2) % reps) * region;
        hipLaunchKernelGGL(halo_read<1>, dim3(2048), dim3(256), 0, 0, (const float*)base, W, H, planes, 1, 1, sink);
    }
    for (int r = 0; r < reps; ++r) {
        char* base = buf + (size_t)r * region;
        hipLaunchKernelGGL(store8, dim3(2048), dim3(256), 0, 0, (float*)base, W, H, planes);
    }
    CHECK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
