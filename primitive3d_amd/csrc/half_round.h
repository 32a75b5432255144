// half_round.h -- the largest fp16 value <= t, as a bit pattern (NaN for a NaN threshold).
// For an fp16 sample v, `float(v) > t` is exactly `v > half_round_down(t)`: no fp16 value lies between the two
// thresholds -- so an fp16 grid is classified with 16-bit compares on the values as they come out of memory (the
// reference up-casts and compares in fp32, marching_cubes.py:87, marching_cubes.cu:25).
// Edge cases: t above the fp16 range -> 65504 (only +inf is inside); t below it -> -inf (everything but -inf and NaN is
// inside); t in (-2^-24, 0) -> the negative subnormal next to zero (both zeros are inside).
// tests/test_oracle_cpu.py::test_half_round_down_is_exact checks it against every fp16 value.
#ifndef P3D_HALF_ROUND_H_
#define P3D_HALF_ROUND_H_
#include <stdint.h>

inline uint32_t half_round_down(float t) {
    if (t != t) return 0x7e00u;
    const _Float16 h = (_Float16)t;   // round to nearest
    unsigned short b = __builtin_bit_cast(unsigned short, h);
    if ((float)h > t) {   // rounded up: one step towards -inf
        if ((b & 0x7fffu) == 0) b = 0x8001u;   // (+-0 -> the smallest negative subnormal)
        else if (b & 0x8000u) b = (unsigned short)(b + 1);
        else b = (unsigned short)(b - 1);
    }
    return b;
}
#endif
