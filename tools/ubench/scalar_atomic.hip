// Does gfx950 execute scalar-memory atomics (s_atomic_add ... glc: one returning atomic per WAVE, counted by lgkmcnt, not
// by vmcnt)?  Every wave adds its wave number + 1 to one counter `reps` times and keeps what came back: the returned values
// must be strictly increasing per wave, and the final counter the exact total.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/scalar_atomic.hip -o tools/ubench/scalar_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;

__global__ void k(u32* counter, u32* bad, u64* ticks, int reps) {
    const u32 wave = __builtin_amdgcn_readfirstlane((u32)(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64));
    u32 prev = 0, fails = 0;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; ++i) {
        u32 r = wave + 1;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(r) : "s"(counter) : "memory");
        if (i > 0 && r <= prev) ++fails;
        prev = r;
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        if (fails) atomicAdd(bad, fails);
        ticks[wave] = t1 - t0;
    }
}

int main() {
    u32 *counter, *bad; u64* ticks;
    const int blocks = 1024, threads = 256, reps = 64, waves = blocks * threads / 64;
    (void)hipMalloc(&counter, 4); (void)hipMalloc(&bad, 4); (void)hipMalloc(&ticks, waves * 8);
    (void)hipMemset(counter, 0, 4); (void)hipMemset(bad, 0, 4);
    k<<<blocks, threads>>>(counter, bad, ticks, reps);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    u32 hc = 0, hb = 0;
    (void)hipMemcpy(&hc, counter, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    u64 expect = 0;
    for (int w = 0; w < waves; ++w) expect += (u64)(w + 1) * reps;
    std::vector<u64> t(waves);
    (void)hipMemcpy(t.data(), ticks, waves * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : t) sum += (double)v;
    printf("counter %u, expected %llu (mod 2^32: %u); non-increasing returns: %u; %.0f ticks per returning scalar atomic (one address, %d waves)\n",
           hc, (unsigned long long)expect, (u32)expect, hb, sum / waves / reps, waves);
    return (hc == (u32)expect && hb == 0) ? 0 : 2;
}
