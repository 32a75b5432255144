// Microbenchmark: how fast can gfx950 stream a 512 MiB fp32 field through "compare -> sign word"?
// Variants differ in load width and batching.  Build: hipcc --offload-arch=gfx950 -O3 stream_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int I, int N, typename F> __device__ inline void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int J> __device__ inline void write_lane(int& w, int v) { asm("v_writelane_b32 %0, %1, %2" : "+v"(w) : "s"(v), "n"(J)); }

// V0: plain float4 read + reduction (read ceiling)
__global__ void __launch_bounds__(256) k_read4(const float4* __restrict__ g, size_t n4, float t, u32* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    u32 acc = 0;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 a = g[i], b = g[i + stride], c = g[i + 2 * stride], d = g[i + 3 * stride];
        acc += (a.x > t) + (a.y > t) + (a.z > t) + (a.w > t) + (b.x > t) + (b.y > t) + (b.z > t) + (b.w > t) +
               (c.x > t) + (c.y > t) + (c.z > t) + (c.w > t) + (d.x > t) + (d.y > t) + (d.z > t) + (d.w > t);
    }
    for (; i < n4; i += stride) { float4 a = g[i]; acc += (a.x > t) + (a.y > t) + (a.z > t) + (a.w > t); }
    if (acc == 0xffffffffu) out[0] = acc;
}

// V1: dword per lane, B loads batched, ballot -> writelane, coalesced word store. wave handles 64 units/iter
template <int B>
__global__ void __launch_bounds__(256) k_dword(const float* __restrict__ g, size_t nunits, float t, u64* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const size_t wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t u0 = wave * 64; u0 < nunits; u0 += nwaves * 64) {
        const float* p = g + u0 * 64 + lane;
        int wlo = 0, whi = 0;
        static_for<0, 64 / B>([&](auto gc) {
            constexpr int gi = decltype(gc)::value;
            float v[B];
#pragma unroll
            for (int k = 0; k < B; ++k) v[k] = p[(size_t)(gi * B + k) * 64];
            static_for<0, B>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const u64 m = __ballot(v[k] > t);
                write_lane<gi * B + k>(wlo, (int)(u32)m);
                write_lane<gi * B + k>(whi, (int)(u32)(m >> 32));
            });
        });
        bits[u0 + lane] = ((u64)(u32)whi << 32) | (u32)wlo;
    }
}

// V2: dwordx4 per lane, B loads batched (each = 256 voxels = 4 interleaved words)
template <int B>
__global__ void __launch_bounds__(256) k_dwordx4(const float4* __restrict__ g, size_t nunits, float t, u64* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const size_t wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t u0 = wave * 64; u0 < nunits; u0 += nwaves * 64) {  // 64 words = 16 x4-loads
        const float4* p = g + u0 * 16 + lane;
        int wlo = 0, whi = 0;
        static_for<0, 16 / B>([&](auto gc) {
            constexpr int gi = decltype(gc)::value;
            float4 v[B];
#pragma unroll
            for (int k = 0; k < B; ++k) v[k] = p[(size_t)(gi * B + k) * 64];
            static_for<0, B>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int w = (gi * B + k) * 4;
                u64 m;
                m = __ballot(v[k].x > t); write_lane<w + 0>(wlo, (int)(u32)m); write_lane<w + 0>(whi, (int)(u32)(m >> 32));
                m = __ballot(v[k].y > t); write_lane<w + 1>(wlo, (int)(u32)m); write_lane<w + 1>(whi, (int)(u32)(m >> 32));
                m = __ballot(v[k].z > t); write_lane<w + 2>(wlo, (int)(u32)m); write_lane<w + 2>(whi, (int)(u32)(m >> 32));
                m = __ballot(v[k].w > t); write_lane<w + 3>(wlo, (int)(u32)m); write_lane<w + 3>(whi, (int)(u32)(m >> 32));
            });
        });
        bits[u0 + lane] = ((u64)(u32)whi << 32) | (u32)wlo;
    }
}

// V3: dwordx4 with nontemporal loads
template <int B>
__global__ void __launch_bounds__(256) k_dwordx4_nt(const float4* __restrict__ g, size_t nunits, float t, u64* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const size_t wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t u0 = wave * 64; u0 < nunits; u0 += nwaves * 64) {
        const float4* p = g + u0 * 16 + lane;
        int wlo = 0, whi = 0;
        static_for<0, 16 / B>([&](auto gc) {
            constexpr int gi = decltype(gc)::value;
            float4 v[B];
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const float* q = (const float*)(p + (size_t)(gi * B + k) * 64);
                v[k].x = __builtin_nontemporal_load(q); v[k].y = __builtin_nontemporal_load(q + 1);
                v[k].z = __builtin_nontemporal_load(q + 2); v[k].w = __builtin_nontemporal_load(q + 3);
            }
            static_for<0, B>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int w = (gi * B + k) * 4;
                u64 m;
                m = __ballot(v[k].x > t); write_lane<w + 0>(wlo, (int)(u32)m); write_lane<w + 0>(whi, (int)(u32)(m >> 32));
                m = __ballot(v[k].y > t); write_lane<w + 1>(wlo, (int)(u32)m); write_lane<w + 1>(whi, (int)(u32)(m >> 32));
                m = __ballot(v[k].z > t); write_lane<w + 2>(wlo, (int)(u32)m); write_lane<w + 2>(whi, (int)(u32)(m >> 32));
                m = __ballot(v[k].w > t); write_lane<w + 3>(wlo, (int)(u32)m); write_lane<w + 3>(whi, (int)(u32)(m >> 32));
            });
        });
        bits[u0 + lane] = ((u64)(u32)whi << 32) | (u32)wlo;
    }
}

// V4: block-contiguous layout: each block owns one contiguous span
__global__ void __launch_bounds__(256) k_read4_span(const float4* __restrict__ g, size_t n4, float t, u32* out) {
    const size_t per_block = n4 / gridDim.x;
    const float4* p = g + (size_t)blockIdx.x * per_block;
    u32 acc = 0;
    for (size_t i = threadIdx.x; i + 768 < per_block; i += 1024) {
        float4 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
        acc += (a.x > t) + (a.y > t) + (a.z > t) + (a.w > t) + (b.x > t) + (b.y > t) + (b.z > t) + (b.w > t) +
               (c.x > t) + (c.y > t) + (c.z > t) + (c.w > t) + (d.x > t) + (d.y > t) + (d.z > t) + (d.w > t);
    }
    if (acc == 0xffffffffu) out[0] = acc;
}

__global__ void __launch_bounds__(256) k_copy4(const float4* __restrict__ g, float4* __restrict__ o, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n4; i += stride) o[i] = g[i];
}

template <typename F> float time_it(F&& launch, int reps = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / reps;
}

int main() {
    const size_t n = 512ull * 512 * 512;
    float* g; u64* bits; u32* out;
    CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&bits, n / 8 + 4096)); CK(hipMalloc(&out, 64));
    std::vector<float> h(n);
    u32 s = 12345; for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(int)(s >> 8) / 8388608.f - 1.f; }
    CK(hipMemcpy(g, h.data(), n * 4, hipMemcpyHostToDevice));
    const size_t nunits = n / 64;
    const double gb = n * 4 / 1e9;
    {
        float* g2; CK(hipMalloc(&g2, n * 4));
        float ms = time_it([&] { CK(hipMemcpyAsync(g2, g, n * 4, hipMemcpyDeviceToDevice, 0)); });
        printf("hipMemcpy D2D 512MiB      %8.1f us  %7.1f GB/s (read+write %7.1f)\n", ms * 1e3, gb / ms * 1e3, 2 * gb / ms * 1e3);
        for (int grid : {2048, 8192, 65536}) {
            ms = time_it([&] { hipLaunchKernelGGL(k_copy4, dim3(grid), dim3(256), 0, 0, (const float4*)g, (float4*)g2, n / 4); });
            printf("grid %5d copy4          %8.1f us  read+write %7.1f GB/s\n", grid, ms * 1e3, 2 * gb / ms * 1e3);
        }
        CK(hipFree(g2));
    }
    for (int bpc : {8, 32, 64, 256}) {
        int grid = 256 * bpc;
        float ms;
        ms = time_it([&] { hipLaunchKernelGGL(k_read4, dim3(grid), dim3(256), 0, 0, (const float4*)g, n / 4, 0.f, out); });
        printf("grid %5d  read4        %8.1f us  %7.1f GB/s\n", grid, ms * 1e3, gb / ms * 1e3);
        ms = time_it([&] { hipLaunchKernelGGL(k_read4_span, dim3(grid), dim3(256), 0, 0, (const float4*)g, n / 4, 0.f, out); });
        printf("grid %5d  read4 span   %8.1f us  %7.1f GB/s\n", grid, ms * 1e3, gb / ms * 1e3);
        ms = time_it([&] { hipLaunchKernelGGL(k_dwordx4_nt<4>, dim3(grid), dim3(256), 0, 0, (const float4*)g, nunits, 0.f, bits); });
        printf("grid %5d  dwordx4nt B=4%8.1f us  %7.1f GB/s\n", grid, ms * 1e3, gb / ms * 1e3);
        ms = time_it([&] { hipLaunchKernelGGL(k_dword<8>, dim3(grid), dim3(256), 0, 0, g, nunits, 0.f, bits); });
        printf("grid %5d  dword  B=8   %8.1f us  %7.1f GB/s\n", grid, ms * 1e3, gb / ms * 1e3);
        ms = time_it([&] { hipLaunchKernelGGL(k_dword<16>, dim3(grid), dim3(256), 0, 0, g, nunits, 0.f, bits); });
        printf("grid %5d  dword  B=16  %8.1f us  %7.1f GB/s\n", grid, ms * 1e3, gb / ms * 1e3);
        ms = time_it([&] { hipLaunchKernelGGL(k_dword<32>, dim3(grid), dim3(256), 0, 0, g, nunits, 0.f, bits); });
        printf("grid %5d  dword  B=32  %8.1f us  %7.1f GB/s\n", grid, ms * 1e3, gb / ms * 1e3);
        ms = time_it([&] { hipLaunchKernelGGL(k_dwordx4<2>, dim3(grid), dim3(256), 0, 0, (const float4*)g, nunits, 0.f, bits); });
        printf("grid %5d  dwordx4 B=2  %8.1f us  %7.1f GB/s\n", grid, ms * 1e3, gb / ms * 1e3);
        ms = time_it([&] { hipLaunchKernelGGL(k_dwordx4<4>, dim3(grid), dim3(256), 0, 0, (const float4*)g, nunits, 0.f, bits); });
        printf("grid %5d  dwordx4 B=4  %8.1f us  %7.1f GB/s\n", grid, ms * 1e3, gb / ms * 1e3);
        ms = time_it([&] { hipLaunchKernelGGL(k_dwordx4<8>, dim3(grid), dim3(256), 0, 0, (const float4*)g, nunits, 0.f, bits); });
        printf("grid %5d  dwordx4 B=8  %8.1f us  %7.1f GB/s\n", grid, ms * 1e3, gb / ms * 1e3);
    }
    return 0;
}
