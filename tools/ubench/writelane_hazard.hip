// Does v_writelane_b32 see an SGPR written by the IMMEDIATELY preceding VALU (v_cmp)?  gfx950 probe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;

template <int MODE>
__global__ void k(const float* in, u32* out) {
    const int lane = threadIdx.x;
    float v0 = in[lane], v1 = in[64 + lane], v2 = in[128 + lane], v3 = in[192 + lane];
    int w = 0;
    // 4 compares, each result written into lanes 0..3 of w (low half only)
    if (MODE == 0) {  // adjacent
        asm volatile(
            "v_cmp_gt_f32 s[20:21], %1, 0\n v_writelane_b32 %0, s20, 0\n"
            "v_cmp_gt_f32 s[22:23], %2, 0\n v_writelane_b32 %0, s22, 1\n"
            "v_cmp_gt_f32 s[20:21], %3, 0\n v_writelane_b32 %0, s20, 2\n"
            "v_cmp_gt_f32 s[22:23], %4, 0\n v_writelane_b32 %0, s22, 3\n"
            : "+v"(w) : "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "s20", "s21", "s22", "s23", "vcc");
    } else if (MODE == 1) {  // vcc adjacent
        asm volatile(
            "v_cmp_gt_f32 vcc, %1, 0\n v_writelane_b32 %0, vcc_lo, 0\n"
            "v_cmp_gt_f32 vcc, %2, 0\n v_writelane_b32 %0, vcc_lo, 1\n"
            "v_cmp_gt_f32 vcc, %3, 0\n v_writelane_b32 %0, vcc_lo, 2\n"
            "v_cmp_gt_f32 vcc, %4, 0\n v_writelane_b32 %0, vcc_lo, 3\n"
            : "+v"(w) : "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "vcc");
    } else if (MODE == 2) {  // through s_mov
        asm volatile(
            "v_cmp_gt_f32 s[20:21], %1, 0\n s_mov_b32 s24, s20\n v_writelane_b32 %0, s24, 0\n"
            "v_cmp_gt_f32 s[22:23], %2, 0\n s_mov_b32 s24, s22\n v_writelane_b32 %0, s24, 1\n"
            "v_cmp_gt_f32 s[20:21], %3, 0\n s_mov_b32 s24, s20\n v_writelane_b32 %0, s24, 2\n"
            "v_cmp_gt_f32 s[22:23], %4, 0\n s_mov_b32 s24, s22\n v_writelane_b32 %0, s24, 3\n"
            : "+v"(w) : "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "s20", "s21", "s22", "s23", "s24", "vcc");
    } else if (MODE == 3) {  // s_nop 3 between
        asm volatile(
            "v_cmp_gt_f32 s[20:21], %1, 0\n s_nop 3\n v_writelane_b32 %0, s20, 0\n"
            "v_cmp_gt_f32 s[22:23], %2, 0\n s_nop 3\n v_writelane_b32 %0, s22, 1\n"
            "v_cmp_gt_f32 s[20:21], %3, 0\n s_nop 3\n v_writelane_b32 %0, s20, 2\n"
            "v_cmp_gt_f32 s[22:23], %4, 0\n s_nop 3\n v_writelane_b32 %0, s22, 3\n"
            : "+v"(w) : "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "s20", "s21", "s22", "s23", "vcc");
    } else if (MODE == 4) {  // s_nop 0
        asm volatile(
            "v_cmp_gt_f32 s[20:21], %1, 0\n s_nop 0\n v_writelane_b32 %0, s20, 0\n"
            "v_cmp_gt_f32 s[22:23], %2, 0\n s_nop 0\n v_writelane_b32 %0, s22, 1\n"
            "v_cmp_gt_f32 s[20:21], %3, 0\n s_nop 0\n v_writelane_b32 %0, s20, 2\n"
            "v_cmp_gt_f32 s[22:23], %4, 0\n s_nop 0\n v_writelane_b32 %0, s22, 3\n"
            : "+v"(w) : "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "s20", "s21", "s22", "s23", "vcc");
    } else if (MODE == 5) {  // grouped: 4 cmps then 4 writelanes
        asm volatile(
            "v_cmp_gt_f32 s[20:21], %1, 0\n v_cmp_gt_f32 s[22:23], %2, 0\n v_cmp_gt_f32 s[24:25], %3, 0\n v_cmp_gt_f32 s[26:27], %4, 0\n"
            "v_writelane_b32 %0, s20, 0\n v_writelane_b32 %0, s22, 1\n v_writelane_b32 %0, s24, 2\n v_writelane_b32 %0, s26, 3\n"
            : "+v"(w) : "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc");
    }
    out[lane] = (u32)w;
}

int main() {
    std::vector<float> h(256);
    u32 pat[4] = {0x0000ffffu, 0xffff0000u, 0x12345678u, 0x0f0f0f0fu};
    for (int j = 0; j < 4; ++j) for (int l = 0; l < 64; ++l) h[j * 64 + l] = (l < 32 && ((pat[j] >> l) & 1)) ? 1.f : -1.f;
    float* d; u32* o; hipMalloc(&d, 1024); hipMalloc(&o, 256);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    u32 r[64];
    const char* names[] = {"sgpr adjacent", "vcc adjacent", "via s_mov", "s_nop 3", "s_nop 0", "grouped x4"};
    for (int m = 0; m < 6; ++m) {
        int bad = 0;
        for (int rep = 0; rep < 50; ++rep) {
            switch (m) { case 0: k<0><<<1, 64>>>(d, o); break; case 1: k<1><<<1, 64>>>(d, o); break; case 2: k<2><<<1, 64>>>(d, o); break;
                         case 3: k<3><<<1, 64>>>(d, o); break; case 4: k<4><<<1, 64>>>(d, o); break; case 5: k<5><<<1, 64>>>(d, o); break; }
            hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
            for (int j = 0; j < 4; ++j) if (r[j] != pat[j]) ++bad;
        }
        printf("%-14s: %s (bad=%d) got %08x %08x %08x %08x\n", names[m], bad ? "WRONG" : "ok", bad, r[0], r[1], r[2], r[3]);
    }
    return 0;
}
