// Latency of returning scalar-memory atomics by width and by how the addresses of the waves are laid out: every wave adds 1
// to ITS OWN word `reps` times (wait after each) -- words packed (8 or 4 bytes apart: 16 or 32 waves per 128-byte line) or
// one per line.  1024 blocks x 1 leader wave (as the plane claims of k_fused DYN) or all 4096 waves.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/scalar_atomic64.hip -o tools/ubench/scalar_atomic64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;

template <int WIDTH>
__global__ void k(char* base, int stride, int leaders_only, u64* ticks, int reps, int gap) {
    const u32 wave = __builtin_amdgcn_readfirstlane((u32)(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64));
    if (leaders_only && (threadIdx.x / 64) != 0) return;
    const u32 slot = leaders_only ? blockIdx.x : wave;
    char* p = base + (size_t)slot * stride;
    u64 acc = 0, tsum = 0;
    for (int i = 0; i < reps; ++i) {
        const u64 t0 = __builtin_amdgcn_s_memtime();
        if (WIDTH == 8) {
            u64 r = 1;
            asm volatile("s_atomic_add_x2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(r) : "s"(p) : "memory");
            acc += r;
        } else {
            u32 r = 1;
            asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(r) : "s"(p) : "memory");
            acc += r;
        }
        tsum += __builtin_amdgcn_s_memtime() - t0;
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(10);
    }
    if ((threadIdx.x & 63) == 0) ticks[wave] = tsum + (acc == 0x123456789ull ? 1 : 0);
}

int main() {
    const int blocks = 1024, threads = 256, reps = 32, waves = blocks * 4;
    char* base; u64* ticks;
    (void)hipMalloc(&base, (size_t)waves * 4096 + 4096); (void)hipMalloc(&ticks, waves * 8);
    for (int leaders = 1; leaders >= 0; --leaders)
        for (int width : {4, 8})
            for (int stride : {8, 128, 256, 4096})
                for (int gap : {0, 8}) {
                    (void)hipMemset(base, 0, (size_t)waves * 4096);
                    (void)hipMemset(ticks, 0, waves * 8);
                    if (width == 8) k<8><<<blocks, threads>>>(base, stride, leaders, ticks, reps, gap);
                    else k<4><<<blocks, threads>>>(base, stride, leaders, ticks, reps, gap);
                    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
                    std::vector<u64> t(waves);
                    (void)hipMemcpy(t.data(), ticks, waves * 8, hipMemcpyDeviceToHost);
                    double sum = 0; int n = 0;
                    for (auto v : t) if (v) { sum += (double)v; ++n; }
                    printf("%s waves, %d-byte atomic, words %4d B apart, gap %d: %.0f ticks (100 MHz x?) per returning atomic (%d waves)\n",
                           leaders ? "1024 leader" : "all 4096", width, stride, gap, sum / n / reps, n);
                }
    return 0;
}
