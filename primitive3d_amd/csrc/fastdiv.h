// fastdiv.h -- division of a 32-bit number by a run-time constant as multiply-high + shifts.  The magic number is made
// on the host; on the device a 32-bit division without it costs ~25 vector instructions, and the face kernel does three
// per block.  Scheme: libdivide's branch-free form -- exact for every 32-bit n and every d >= 1
// (tests/test_oracle_cpu.py::test_fastdiv_is_exact compiles this header with the host compiler and checks it).
#ifndef P3D_FASTDIV_H_
#define P3D_FASTDIV_H_
#include <stdint.h>

#if defined(__HIPCC__)
#define P3D_HD __host__ __device__
#else
#define P3D_HD
#endif

struct FastDiv {
    uint32_t d, m, sh;   // d == 1: identity
};

inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f{d, 0u, 0u};
    if (d <= 1u) return f;
    const uint32_t fl = 31u - (uint32_t)__builtin_clz(d);
    if ((d & (d - 1u)) == 0u) {   // power of two: ((n - 0) >> 1) >> (log2 - 1)
        f.sh = fl - 1u;
        return f;
    }
    const uint64_t num = 1ull << (32 + fl);
    uint64_t m = num / d;
    const uint64_t rem = num % d;
    m += m;
    if (2 * rem >= d) m += 1;
    f.m = (uint32_t)(m + 1);
    f.sh = fl;
    return f;
}

P3D_HD inline uint32_t fd_div(uint32_t n, const FastDiv& f) {
    if (f.d == 1u) return n;   // (uniform)
    const uint32_t q = (uint32_t)(((uint64_t)f.m * n) >> 32);
    return (((n - q) >> 1) + q) >> f.sh;
}
#endif
