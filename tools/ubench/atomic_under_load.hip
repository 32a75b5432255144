// Latency of a returning atomic (scalar-memory and vector-memory, 8 bytes, every wave its own 128-byte line) and of a plain
// sc1 load while the rest of the chip streams a 2 GiB buffer (HBM saturated), against the idle chip.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/atomic_under_load.hip -o tools/ubench/atomic_under_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k(const f4* big, size_t n4, int loaders_per_block, char* words, u64* ticks, u64* rt, float* sink, int mode, int reps) {
    const u32 wib = threadIdx.x / 64;
    if ((int)wib < loaders_per_block) {   // streaming waves
        f4 acc = {0, 0, 0, 0};
        const size_t stride = (size_t)gridDim.x * loaders_per_block * 64;
        for (size_t i = ((size_t)blockIdx.x * loaders_per_block + wib) * 64 + (threadIdx.x & 63); i < n4; i += stride) {
            const f4 v = __builtin_nontemporal_load(big + i);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (acc.x == 1234.5f) sink[0] = acc.y + acc.z + acc.w;
        return;
    }
    if (wib != 3) return;
    char* p = words + (size_t)blockIdx.x * 128;
    u64 tsum = 0, acc = 0;
    const u64 r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < reps; ++i) {
        for (int g = 0; g < 4; ++g) __builtin_amdgcn_s_sleep(100);
        const u64 t0 = __builtin_amdgcn_s_memtime();
        if (mode == 0) {
            u64 r = 1;
            asm volatile("s_atomic_add_x2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(r) : "s"(p) : "memory");
            acc += r;
        } else if (mode == 1) {
            u64 r = 0;
            if ((threadIdx.x & 63) == 0) r = __hip_atomic_fetch_add((u64*)p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += __builtin_amdgcn_readfirstlane((u32)r);
        } else {
            u64 r = 0;
            if ((threadIdx.x & 63) == 0) r = __hip_atomic_load((u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += __builtin_amdgcn_readfirstlane((u32)r);
        }
        tsum += __builtin_amdgcn_s_memtime() - t0;
    }
    const u64 r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) { ticks[blockIdx.x] = tsum + (acc == 0x123456789ull); rt[blockIdx.x] = r1 - r0; }
}

int main() {
    const int blocks = 1024, reps = 16;
    const size_t bytes = 2ull << 30, n4 = bytes / 16;
    f4* big; char* words; u64 *ticks, *rt; float* sink;
    (void)hipMalloc(&big, bytes); (void)hipMalloc(&words, blocks * 128); (void)hipMalloc(&ticks, blocks * 8); (void)hipMalloc(&rt, blocks * 8); (void)hipMalloc(&sink, 4);
    (void)hipMemset(big, 0, bytes);
    const char* names[3] = {"scalar atomic add x2", "vector atomic add x2", "vector sc1 load     "};
    for (int load = 0; load <= 3; load += 3)
        for (int mode = 0; mode < 3; ++mode) {
            (void)hipMemset(words, 0, blocks * 128);
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            k<<<blocks, 256>>>(big, n4, load, words, ticks, rt, sink, mode, reps);
            (void)hipEventRecord(e1);
            if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); return 1; }
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<u64> t(blocks), r(blocks);
            (void)hipMemcpy(t.data(), ticks, blocks * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(r.data(), rt, blocks * 8, hipMemcpyDeviceToHost);
            double s = 0, rr = 0; for (int i = 0; i < blocks; ++i) { s += (double)t[i]; rr += (double)r[i]; }
            printf("%s, %s: %.0f s_memtime ticks per op; kernel %.3f ms%s\n", names[mode], load ? "3 streaming waves per block beside it" : "idle chip                           ",
                   s / blocks / reps, ms, load ? "" : "");
            if (load) printf("    (streamed %.1f GB/s)\n", bytes / (ms * 1e-3) / 1e9);
        }
    // tick calibration: s_memtime vs the 100 MHz realtime over a sleep
    return 0;
}
