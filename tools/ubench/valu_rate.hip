// How many cycles does one SIMD of gfx950 need per wave64 vector instruction, by instruction kind and by the number of
// waves resident on the SIMD?  (Decides whether a kernel's SQ_INSTS_VALU x 4 cycles or x 2 cycles is its issue floor.)
// Each wave runs a long unrolled stream of independent instructions of one kind over 8 registers; blocks of 256 x W
// threads put W waves on every SIMD of a CU, one block per CU.  Reports cycles per instruction per SIMD from s_memtime.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned int u32;
typedef unsigned long long u64;

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void __launch_bounds__(1024) k(u32* out, u64* cyc, int iters) {
    u32 a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 ^ 0x55, a5 = a0 + 9, a6 = a0 * 11, a7 = ~a0;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
    __syncthreads();
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {   // v_bitop3_b32 (any 3-input boolean function)
            REP64(asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n"
                               "v_bitop3_b32 %2, %2, %3, %4 bitop3:0xe8\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0x96\n"
                               "v_bitop3_b32 %4, %4, %5, %6 bitop3:0xe8\n v_bitop3_b32 %5, %5, %6, %7 bitop3:0x96\n"
                               "v_bitop3_b32 %6, %6, %7, %0 bitop3:0xe8\n v_bitop3_b32 %7, %7, %0, %1 bitop3:0x96\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 1) {   // v_and_b32 / v_add_u32 (VOP2)
            REP64(asm volatile("v_and_b32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_add_u32 %3, %3, %4\n"
                               "v_and_b32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_and_b32 %6, %6, %7\n v_add_u32 %7, %7, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 2) {   // v_bcnt_u32_b32 (popcount + add, VOP3)
            REP64(asm volatile("v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %1, %2, %1\n v_bcnt_u32_b32 %2, %3, %2\n"
                               "v_bcnt_u32_b32 %3, %4, %3\n v_bcnt_u32_b32 %4, %5, %4\n v_bcnt_u32_b32 %5, %6, %5\n"
                               "v_bcnt_u32_b32 %6, %7, %6\n v_bcnt_u32_b32 %7, %0, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 3) {   // v_fma_f32
            REP64(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n"
                               "v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n"
                               "v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1\n"
                               : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));)
        } else if (KIND == 4) {   // v_alignbit_b32 / v_bfe_u32 / v_lshl_add_u32 (VOP3 integer)
            REP64(asm volatile("v_alignbit_b32 %0, %0, %1, %2\n v_bfe_u32 %1, %1, 3, 5\n v_lshl_add_u32 %2, %2, 2, %3\n"
                               "v_alignbit_b32 %3, %3, %4, %5\n v_bfe_u32 %4, %4, 3, 5\n v_lshl_add_u32 %5, %5, 2, %6\n"
                               "v_cndmask_b32 %6, %6, %7, vcc\n v_mov_b32 %7, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");)
        } else if (KIND == 5) {   // DPP mov (row_shr) + add: the wave scans
            REP64(asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_add_u32_dpp %1, %2, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                               "v_add_u32_dpp %2, %3, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n"
                               "v_add_u32_dpp %3, %4, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                               "v_add_u32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_add_u32_dpp %5, %6, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                               "v_add_u32_dpp %6, %7, %6 row_shr:4 row_mask:0xf bank_mask:0xf\n"
                               "v_add_u32_dpp %7, %0, %7 row_shr:8 row_mask:0xf bank_mask:0xf\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 6) {   // v_cmp (writes an SGPR pair) + v_cndmask
            REP64(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_gt_u32 vcc, %2, %3\n"
                               "v_cndmask_b32 %3, %3, %4, vcc\n v_cmp_gt_u32 vcc, %4, %5\n v_cndmask_b32 %5, %5, %6, vcc\n"
                               "v_cmp_gt_u32 vcc, %6, %7\n v_cndmask_b32 %7, %7, %0, vcc\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");)
        } else if (KIND == 7) {   // scalar ALU beside nothing: s_add / s_and
            REP64(asm volatile("s_add_u32 s20, s20, s21\n s_and_b32 s21, s21, s22\n s_add_u32 s22, s22, s23\n s_and_b32 s23, s23, s20\n"
                               "s_add_u32 s20, s20, s21\n s_and_b32 s21, s21, s22\n s_add_u32 s22, s22, s23\n s_and_b32 s23, s23, s20\n"
                               : : : "s20", "s21", "s22", "s23", "scc");)
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (u32)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(const char* name) {
    u32* out;
    u64* cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 256 * 16 * 8);
    const int iters = 64;                       // 64 x 64 x 8 = 32768 instructions per wave
    printf("%-34s", name);
    for (int wps : {1, 2, 3, 4}) {               // waves per SIMD
        k<KIND><<<256, 256 * wps>>>(out, cyc, iters);   // warm-up
        k<KIND><<<256, 256 * wps>>>(out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<u64> h(256 * 4 * wps);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];   // s_memtime ticks (100 MHz on this part? reported raw) per wave
        // all wps waves of a SIMD run concurrently: the SIMD executed wps * 32768 instructions in `med` ticks
        printf("  %dw: %7.3f", wps, med / (32768.0 * wps));
    }
    printf("   [s_memtime ticks per wave-instruction per SIMD]\n");
    hipFree(out);
    hipFree(cyc);
}

__global__ void k_clock(u64* o) {
    const u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    u32 a = threadIdx.x;
    for (int i = 0; i < 200000; ++i) asm volatile("v_add_u32 %0, %0, 1" : "+v"(a));
    const u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { o[0] = t1 - t0; o[1] = r1 - r0; o[2] = a; }
}

int main() {
    u64* o;
    hipMalloc(&o, 64);
    k_clock<<<1, 64>>>(o);
    u64 h[3];
    hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
    printf("s_memtime ticks per 100 MHz realtime tick: %.2f (x 100 MHz = shader clock in MHz); 200000 dependent v_add_u32 in %llu ticks = %.2f ticks each\n",
           (double)h[0] / (double)h[1], (unsigned long long)h[0], (double)h[0] / 200000.0);
    run<0>("v_bitop3_b32");
    run<1>("v_and_b32 / v_add_u32");
    run<2>("v_bcnt_u32_b32");
    run<3>("v_fma_f32");
    run<4>("alignbit/bfe/lshl_add/cndmask/mov");
    run<5>("v_add_u32_dpp row_shr");
    run<6>("v_cmp + v_cndmask");
    run<7>("s_add_u32 / s_and_b32 (scalar)");
    return 0;
}
