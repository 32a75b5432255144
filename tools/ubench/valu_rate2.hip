// Per-instruction issue cost on one SIMD with 4 resident waves (the regime of the face and streaming kernels): which
// of the integer / select / shift instructions run at the 2-cycle rate and which at the 4-cycle one?
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate2.hip -o tools/ubench/valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned int u32;
typedef unsigned long long u64;
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KERNEL(NAME, BODY)                                                                                           \
    __global__ void __launch_bounds__(1024) NAME(u32* out, u64* cyc, int iters) {                                    \
        u32 a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 ^ 0x55, a5 = a0 + 9,      \
            a6 = a0 * 11, a7 = ~a0;                                                                                  \
        __syncthreads();                                                                                             \
        const u64 t0 = __builtin_amdgcn_s_memtime();                                                                 \
        for (int i = 0; i < iters; ++i) {                                                                            \
            REP64(asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),        \
                               "+v"(a7) : : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)        \
        }                                                                                                            \
        const u64 t1 = __builtin_amdgcn_s_memtime();                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                          \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;              \
    }

#define EIGHT(op) op(0, 1, 2) op(1, 2, 3) op(2, 3, 4) op(3, 4, 5) op(4, 5, 6) op(5, 6, 7) op(6, 7, 0) op(7, 0, 1)
#define S(x) #x
#define OP3(name) "v_" name " %%0, %%0, %%1, %%2\n"

KERNEL(k_and, "v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %4\n v_and_b32 %4, %4, %5\n v_and_b32 %5, %5, %6\n v_and_b32 %6, %6, %7\n v_and_b32 %7, %7, %0\n")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7\n")
KERNEL(k_lshrv, "v_lshrrev_b32 %0, %1, %0\n v_lshrrev_b32 %1, %2, %1\n v_lshrrev_b32 %2, %3, %2\n v_lshrrev_b32 %3, %4, %3\n v_lshrrev_b32 %4, %5, %4\n v_lshrrev_b32 %5, %6, %5\n v_lshrrev_b32 %6, %7, %6\n v_lshrrev_b32 %7, %0, %7\n")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 5\n v_bfe_u32 %1, %1, 3, 5\n v_bfe_u32 %2, %2, 3, 5\n v_bfe_u32 %3, %3, 3, 5\n v_bfe_u32 %4, %4, 3, 5\n v_bfe_u32 %5, %5, 3, 5\n v_bfe_u32 %6, %6, 3, 5\n v_bfe_u32 %7, %7, 3, 5\n")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2\n v_alignbit_b32 %1, %1, %2, %3\n v_alignbit_b32 %2, %2, %3, %4\n v_alignbit_b32 %3, %3, %4, %5\n v_alignbit_b32 %4, %4, %5, %6\n v_alignbit_b32 %5, %5, %6, %7\n v_alignbit_b32 %6, %6, %7, %0\n v_alignbit_b32 %7, %7, %0, %1\n")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1\n v_lshl_add_u32 %1, %1, 2, %2\n v_lshl_add_u32 %2, %2, 2, %3\n v_lshl_add_u32 %3, %3, 2, %4\n v_lshl_add_u32 %4, %4, 2, %5\n v_lshl_add_u32 %5, %5, 2, %6\n v_lshl_add_u32 %6, %6, 2, %7\n v_lshl_add_u32 %7, %7, 2, %0\n")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %4\n v_add3_u32 %3, %3, %4, %5\n v_add3_u32 %4, %4, %5, %6\n v_add3_u32 %5, %5, %6, %7\n v_add3_u32 %6, %6, %7, %0\n v_add3_u32 %7, %7, %0, %1\n")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %1, %1, %2, %3\n v_or3_b32 %2, %2, %3, %4\n v_or3_b32 %3, %3, %4, %5\n v_or3_b32 %4, %4, %5, %6\n v_or3_b32 %5, %5, %6, %7\n v_or3_b32 %6, %6, %7, %0\n v_or3_b32 %7, %7, %0, %1\n")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %1, %1, %2, %3\n v_and_or_b32 %2, %2, %3, %4\n v_and_or_b32 %3, %3, %4, %5\n v_and_or_b32 %4, %4, %5, %6\n v_and_or_b32 %5, %5, %6, %7\n v_and_or_b32 %6, %6, %7, %0\n v_and_or_b32 %7, %7, %0, %1\n")
KERNEL(k_cndmask_vcc, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n")
KERNEL(k_cndmask_sgpr, "v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %1, %1, %2, s[22:23]\n v_cndmask_b32 %2, %2, %3, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[22:23]\n v_cndmask_b32 %4, %4, %5, s[20:21]\n v_cndmask_b32 %5, %5, %6, s[22:23]\n v_cndmask_b32 %6, %6, %7, s[20:21]\n v_cndmask_b32 %7, %7, %0, s[22:23]\n")
KERNEL(k_cmp_vcc, "v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %1, %2\n v_cmp_gt_u32 vcc, %2, %3\n v_cmp_gt_u32 vcc, %3, %4\n v_cmp_gt_u32 vcc, %4, %5\n v_cmp_gt_u32 vcc, %5, %6\n v_cmp_gt_u32 vcc, %6, %7\n v_cmp_gt_u32 vcc, %7, %0\n")
KERNEL(k_cmp_sgpr, "v_cmp_gt_u32 s[20:21], %0, %1\n v_cmp_gt_u32 s[22:23], %1, %2\n v_cmp_gt_u32 s[24:25], %2, %3\n v_cmp_gt_u32 s[26:27], %3, %4\n v_cmp_gt_u32 s[20:21], %4, %5\n v_cmp_gt_u32 s[22:23], %5, %6\n v_cmp_gt_u32 s[24:25], %6, %7\n v_cmp_gt_u32 s[26:27], %7, %0\n")
KERNEL(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %1, %2, %1\n v_bcnt_u32_b32 %2, %3, %2\n v_bcnt_u32_b32 %3, %4, %3\n v_bcnt_u32_b32 %4, %5, %4\n v_bcnt_u32_b32 %5, %6, %5\n v_bcnt_u32_b32 %6, %7, %6\n v_bcnt_u32_b32 %7, %0, %7\n")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %4\n v_mul_u32_u24 %4, %4, %5\n v_mul_u32_u24 %5, %5, %6\n v_mul_u32_u24 %6, %6, %7\n v_mul_u32_u24 %7, %7, %0\n")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %4\n v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %5, %5, %6\n v_mul_lo_u32 %6, %6, %7\n v_mul_lo_u32 %7, %7, %0\n")
KERNEL(k_sub_sdwa, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_sub_u32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_sub_u32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_sub_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_sub_u32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_sub_u32_sdwa %5, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_sub_u32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_sub_u32_sdwa %7, %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n v_readlane_b32 s24, %4, 3\n v_readlane_b32 s25, %5, 3\n v_readlane_b32 s26, %6, 3\n v_readlane_b32 s27, %7, 3\n")
KERNEL(k_writelane, "v_writelane_b32 %0, s20, 3\n v_writelane_b32 %1, s21, 3\n v_writelane_b32 %2, s22, 3\n v_writelane_b32 %3, s23, 3\n v_writelane_b32 %4, s24, 3\n v_writelane_b32 %5, s25, 3\n v_writelane_b32 %6, s26, 3\n v_writelane_b32 %7, s27, 3\n")

template <typename K>
void run(const char* name, K kern, int per_rep = 8) {
    u32* out; u64* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipMalloc(&cyc, 256 * 16 * 8);
    const int iters = 32;
    printf("%-22s", name);
    for (int wps : {1, 2, 4}) {
        kern<<<256, 256 * wps>>>(out, cyc, iters);
        kern<<<256, 256 * wps>>>(out, cyc, iters);
        (void)hipDeviceSynchronize();
        std::vector<u64> h(256 * 4 * wps);
        (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("  %dw: %6.2f", wps, (double)h[h.size() / 2] / (32.0 * 64 * per_rep * wps));
    }
    printf("\n");
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    printf("s_memtime ticks per wave64 instruction per SIMD, by waves resident on the SIMD\n");
    run("v_and_b32", k_and); run("v_lshrrev_b32 imm", k_lshr); run("v_lshrrev_b32 var", k_lshrv); run("v_bfe_u32", k_bfe);
    run("v_alignbit_b32", k_alignbit); run("v_lshl_add_u32", k_lshl_add); run("v_add3_u32", k_add3); run("v_or3_b32", k_or3);
    run("v_and_or_b32", k_and_or); run("v_cndmask vcc", k_cndmask_vcc); run("v_cndmask sgpr", k_cndmask_sgpr);
    run("v_cmp -> vcc", k_cmp_vcc); run("v_cmp -> sgpr", k_cmp_sgpr); run("v_mov_b32", k_mov); run("v_bcnt_u32_b32", k_bcnt);
    run("v_mul_u32_u24", k_mul24); run("v_mul_lo_u32", k_mullo); run("v_sub_u32_sdwa", k_sub_sdwa);
    run("v_readlane_b32", k_readlane); run("v_writelane_b32", k_writelane);
    return 0;
}
