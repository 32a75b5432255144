// The plane hand-out protocol of the streaming kernel (primitive3d_amd/csrc/range_sched.h) ALONE: persistent blocks of four
// waves, "processing" a plane = the wave counts itself into counts[(col * nplanes + x) * 4 + wave] and sleeps for a
// pseudo-random, column-dependent time (skewed work: some columns cost 4x the others, so a lot is stolen).  At the end every
// (column, plane, wave) must have been counted exactly once.
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-atomic-optimizer-strategy=None tools/ubench/range_sched_test.hip -o tools/ubench/range_sched_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "range_sched.h"
typedef unsigned int u32;
typedef unsigned long long u64;

__global__ void __launch_bounds__(256) k(u64* table, RsGeom g, u32* counts, u32* steals, int skew) {
    __shared__ u32 s_rs[kRsLdsWords];
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool leader = wave == 0;
    const u32 me = blockIdx.x;
    RsBlock s;
    rs_start(s, leader, g, table, me, s_rs);
    bool have = true;
    while (have) {
        if (s.claimed > s.xb) {
            for (u32 x = s.xb;; ++x) {
                const bool own = leader ? rs_leader_own_next(s, x, s_rs) : rs_sibling_own_next(s, x, s_rs);
                if (rs_wants_claim(s, leader)) rs_issue_claim(s, table + (size_t)me * kRsStride);   // (k_fused: inside its plane work, in front of the plane prefetch)
                if (lane == 0) atomicAdd(&counts[((size_t)s.col * g.nplanes + x) * 4 + wave], 1u);
                u32 h = (s.col * 2654435761u) ^ (x * 40503u) ^ (wave * 977u);
                h ^= h >> 13;
                const u32 cost = 2 + (h & 3) + ((s.col % 5 == 0) ? skew : 0);
                for (u32 i = 0; i < cost; ++i) __builtin_amdgcn_s_sleep(20);
                if (!own) break;
            }
        }
        have = rs_switch(s, leader, table, g.nb, me, s_rs);
        if (have && leader && lane == 0) atomicAdd(steals, 1u);
    }
}

int main() {
    int fails = 0;
    const int cfgs[][4] = {{1024, 43, 512, 8}, {1024, 352, 256, 6}, {1024, 172, 1024, 0}, {512, 7, 9000, 12}, {1000, 999, 64, 4}, {1024, 1, 30000, 3}};
    for (auto& c : cfgs) {
        const u32 nb = c[0], ncol = c[1], npl = c[2];
        const RsGeom g = rs_make_geom(nb, ncol, npl);
        u64* table; u32 *counts, *steals;
        const size_t n = (size_t)ncol * npl * 4;
        (void)hipMalloc(&table, (size_t)nb * 8 * kRsStride); (void)hipMalloc(&counts, n * 4); (void)hipMalloc(&steals, 4);
        (void)hipMemset(table, 0, (size_t)nb * 8 * kRsStride); (void)hipMemset(counts, 0, n * 4); (void)hipMemset(steals, 0, 4);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        k<<<nb, 256>>>(table, g, counts, steals, c[3]);
        (void)hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<u32> h(n); u32 hs = 0;
        (void)hipMemcpy(h.data(), counts, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&hs, steals, 4, hipMemcpyDeviceToHost);
        size_t bad = 0, zero = 0, multi = 0;
        for (auto v : h) { if (v != 1) ++bad; if (v == 0) ++zero; if (v > 1) ++multi; }
        printf("blocks %u columns %u planes %u skew %d: %zu wave-planes, %zu wrong (%zu never, %zu more than once), %u steals, %.3f ms\n",
               nb, ncol, npl, c[3], n, bad, zero, multi, hs, ms);
        if (bad) ++fails;
        (void)hipFree(table); (void)hipFree(counts); (void)hipFree(steals);
    }
    return fails ? 2 : 0;
}
