// range_sched.h -- dynamic hand-out of x planes to the PERSISTENT blocks of the streaming kernel (k_fused, DYN).
//
// Why: with a fixed x-slab per block the halo plane of every slab is read twice (1/12 of the field at 12 planes per
// slab, 1/3 in the short slabs that keep the end of the launch balanced: profiles/r04/l2_summary_before.txt).  Here the
// launch has exactly as many blocks as the chip holds at once; a block owns a RANGE of planes of one tile column and
// marches through it plane by plane (state stays in registers / LDS: no halo re-read inside a range).  The ranges are
// equal at first, so x-halo = blocks per column / planes.  A block whose range is exhausted STEALS the upper half of the
// longest range left anywhere (one extra halo plane per steal), so the end of the launch balances itself without short
// slabs.
//
// One 64-bit word per block in device memory, library-owned, zero before the launch:
//     bits 38..63 end (SIGNED) | bit 37 valid | bits 25..36 tile column | bits 0..24 next   (planes relative to the launch's first)
//   owner : claims plane `next` with ONE 64-bit atomic add of 1 and owns it iff the returned next < the returned end
//           (claims run two planes ahead of the plane being processed: the answer is there before it is needed);
//   thief : reads every block's word (one batch of loads), picks a range with many unclaimed planes and lowers its `end`
//           with ONE 64-bit atomic SUBTRACTION of take << 38.  Whatever the word was at that instant comes back with it, and
//           the thief owns exactly [max(next, end - take), end) of that value: planes below `next` had been claimed by the
//           owner, planes from the new end on can no longer be granted to it.  No compare-and-swap: under load an atomic's
//           round trip takes as long as the owner needs for a plane, so a swap that insists on an unchanged `next` loses
//           nearly every time (measured: the blocks of a 512^3 launch spent 43 us each in failed swaps).  Several thieves
//           may subtract from one word; `end` is signed and the top field, so it may go below `next` or below zero without
//           harm (nothing more is granted there) and without touching the other fields.
//   A word of 0 belongs to a block that has not started: thieves skip it (its owner will come).
// Every (column, plane) is owned exactly once: tools/ubench/range_sched_test.hip runs the protocol alone with skewed work
// and counts.
#ifndef P3D_RANGE_SCHED_H_
#define P3D_RANGE_SCHED_H_
#include <stdint.h>

#if defined(__HIPCC__)
#define P3D_RS_HD __host__ __device__
#else
#define P3D_RS_HD
#endif

constexpr int kRsBits = 25;                               // planes of a launch < 2^25
constexpr unsigned long long kRsMask = (1ull << kRsBits) - 1ull;
constexpr int kRsColBits = 12;                            // tile columns < 2^12
constexpr int kRsEndShift = kRsBits + kRsColBits + 1;      // 38
constexpr int kRsMaxBlocks = 1024;                        // words of a table (a thief reads all of them: 16 per lane)
#ifndef P3D_RS_STRIDE
#define P3D_RS_STRIDE 16
#endif
constexpr int kRsStride = P3D_RS_STRIDE;                  // u64 words between the words of two blocks (16: a 128-byte line each)
constexpr int kRsTableWords = kRsMaxBlocks * kRsStride;   // u64 words of a table
#ifndef P3D_RS_ATTEMPTS
#define P3D_RS_ATTEMPTS 2
#endif
#ifndef P3D_RS_MINSTEAL
#define P3D_RS_MINSTEAL 3
#endif
#ifndef P3D_RS_INHAND
#define P3D_RS_INHAND 3
#endif
constexpr unsigned kRsMinSteal = P3D_RS_MINSTEAL;         // unclaimed planes a range must have for a thief to come
constexpr unsigned kRsInHand = P3D_RS_INHAND;             // planes an owner has claimed beyond `next` - in the split they count as its share
constexpr unsigned kRsAhead = 2;                          // planes a fresh range pre-claims for its owner

struct RsGeom {
    uint32_t nb;        // blocks of the launch (>= ncol)
    uint32_t ncol;      // tile columns (y tile x z tile x item)
    uint32_t nplanes;   // planes per column
    uint32_t base, extra;   // nb / ncol, nb % ncol: the first `extra` columns are cut into base + 1 ranges, the rest into base
};

P3D_RS_HD inline RsGeom rs_make_geom(uint32_t nb, uint32_t ncol, uint32_t nplanes) {
    RsGeom g;
    g.nb = nb;
    g.ncol = ncol;
    g.nplanes = nplanes;
    g.base = nb / ncol;
    g.extra = nb % ncol;
    return g;
}

// Initial range of block b.  Ranges are numbered column by column; XCD r (blocks b = r mod 8 under round-robin placement)
// takes the r-th eighth of that order, i.e. a contiguous group of columns: the halo row a tile shares with its y neighbour
// is then read by two blocks of ONE L2 at about the same time.
P3D_RS_HD inline void rs_initial(const RsGeom& g, uint32_t b, uint32_t& col, uint32_t& s, uint32_t& e) {
#ifndef P3D_RS_LINEAR
#define P3D_RS_LINEAR 0
#endif
    const uint32_t i = (g.nb % 8u == 0u && !P3D_RS_LINEAR) ? (b & 7u) * (g.nb >> 3) + (b >> 3) : b;
    const uint32_t hi = g.extra * (g.base + 1u);
    uint32_t k, cnt;
    if (i < hi) {
        col = i / (g.base + 1u);
        k = i - col * (g.base + 1u);
        cnt = g.base + 1u;
    } else {
        const uint32_t t = i - hi;
        const uint32_t q = t / g.base;
        col = g.extra + q;
        k = t - q * g.base;
        cnt = g.base;
    }
    s = (uint32_t)((uint64_t)k * g.nplanes / cnt);
    e = (uint32_t)((uint64_t)(k + 1u) * g.nplanes / cnt);
}

P3D_RS_HD inline unsigned long long rs_pack(uint32_t col, uint32_t next, uint32_t end) {
    return ((unsigned long long)end << kRsEndShift) | (1ull << (kRsBits + kRsColBits)) | ((unsigned long long)col << kRsBits) | next;
}
P3D_RS_HD inline bool rs_valid(unsigned long long w) { return ((w >> (kRsBits + kRsColBits)) & 1ull) != 0ull; }
P3D_RS_HD inline int32_t rs_next(unsigned long long w) { return (int32_t)(w & kRsMask); }
P3D_RS_HD inline int32_t rs_end(unsigned long long w) { return (int32_t)((long long)w >> kRsEndShift); }   // (signed)
P3D_RS_HD inline uint32_t rs_col(unsigned long long w) { return (uint32_t)((w >> kRsBits) & ((1u << kRsColBits) - 1u)); }
// the claim whose atomic add returned `old` was granted
P3D_RS_HD inline bool rs_granted(unsigned long long old) { return rs_next(old) < rs_end(old); }
// unclaimed planes of a word (0 for a block that has not started or whose range is used up)
P3D_RS_HD inline uint32_t rs_unclaimed(unsigned long long w) {
    const int32_t r = rs_end(w) - rs_next(w);
    return (rs_valid(w) && r > 0) ? (uint32_t)r : 0u;
}

#if defined(__HIPCC__)
__device__ inline unsigned long long rs_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by ALL 64 lanes of one wave (wave-uniform arguments and results).  On success the wave's block owns planes
// [xb, xend) of column `col`; its own word `table[me]` holds the range with the first min(kRsAhead, length) planes already
// claimed (`claimed` = xb + that many), open to other thieves.
//   probe: the words of kRsProbe OTHER blocks, one per lane, in ONE load (sc1: the words are written by atomics of all XCDs)
//          -- block me + o for a fixed set of offsets o spread over the whole table (1, 2, 3, 5, 8, 13, ... and their
//          multiples in the second attempt).  Not every block's word: at the end of a launch a thousand blocks run dry
//          within microseconds of each other, and a thousand scans of a thousand lines each (tried first) slowed the
//          blocks still working -- their claims are atomics on those very lines -- by a third;
//   pick : the probed range with the most unclaimed planes (at least kRsMinSteal);
//   take : ONE atomic subtraction of half of what was probed from the victim's `end`; the returned word says exactly which
//          planes that bought (see the top of this file) -- possibly fewer than hoped for, possibly none (then: probe the
//          second set).
constexpr int kRsProbe = 32;
__device__ inline bool rs_steal(unsigned long long* table, uint32_t nb, uint32_t me, uint32_t& col, uint32_t& xb,
                                uint32_t& claimed, uint32_t& xend) {
    const uint32_t lane = threadIdx.x & 63u;
    for (int attempt = 0; attempt < P3D_RS_ATTEMPTS; ++attempt) {
        asm volatile("" ::: "memory");   // (the table is re-read in every attempt)
        // offsets: lane l < 16 -> Fibonacci-like steps up to ~nb/2 forwards, lanes 16..31 the same backwards; the second
        // attempt scales them by 7 (mod nb)
        uint32_t ln = lane;
        asm volatile("" : "+v"(ln));   // (opaque per attempt: nothing below is hoisted out of the caller's loops)
        const uint32_t fib[16] = {1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987, 1597};
        // (reduced with a mask, not a division by nb: the reciprocal of a run-time divisor is a vector register that would be
        //  computed once per kernel and held -- spilled -- across the caller's hot loop)
        const uint32_t pmask = (1u << (31 - __builtin_clz(nb))) - 1u;   // largest 2^n - 1 below nb  (nb >= 8)
        uint32_t o = (fib[ln & 15u] * (attempt ? 7u : 1u)) & pmask;
        o = o == 0u ? 1u : o;
        uint32_t k = (ln & 16u) ? me + nb - o : me + o;    // o < nb: one conditional subtraction wraps it
        k = k >= nb ? k - nb : k;
        unsigned long long w = 0;
        if (ln < (uint32_t)kRsProbe && k != me) w = rs_load(table + (size_t)k * kRsStride);
        const uint32_t rem = rs_unclaimed(w);
        uint32_t key = ((rem > 0xfffffu ? 0xfffffu : rem) << 12) | k;   // (ties: the higher block index)
        key = rem ? key : 0u;
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)key, sh, 64);
            key = other > key ? other : key;
        }
        key = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);   // (uniform by value; now also for the compiler)
        const uint32_t vrem = key >> 12, victim = key & 0xfffu;
        if (vrem < kRsMinSteal) continue;   // nothing worth splitting among the probed
        // (the owner keeps the ~3 planes it has already claimed: split what both have, not only what is unclaimed)
        const uint32_t half = (vrem + kRsInHand) / 2u;
        const uint32_t take = half < vrem ? half : vrem;   // >= 1
        unsigned long long old = 0;
        if (lane == 0)
            old = __hip_atomic_fetch_sub(table + (size_t)victim * kRsStride, (unsigned long long)take << kRsEndShift,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)old);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(old >> 32));
        const unsigned long long wv = ((unsigned long long)hi << 32) | lo;
        const int32_t en = rs_end(wv), nx = rs_next(wv);
        const int32_t from = nx > en - (int32_t)take ? nx : en - (int32_t)take;
        if (from >= en) continue;   // the owner (or another thief) got there first: nothing left of what was probed
        col = rs_col(wv);
        xb = (uint32_t)from;
        xend = (uint32_t)en;
        claimed = xb + ((uint32_t)(en - from) < kRsAhead ? (uint32_t)(en - from) : kRsAhead);
        if (lane == 0)   // (an atomic exchange, like every other access to the word: one ordering domain)
            (void)__hip_atomic_exchange(table + (size_t)me * kRsStride, rs_pack(col, claimed, xend), __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    return false;
}

// ---- the block-level protocol: wave 0 of a block (the LEADER) talks to the table; its three siblings follow it through
// two LDS words per generation (a generation = one range of the block).  Siblings may lag behind the leader but never
// run ahead of its decisions; the four waves meet at a barrier only when the block changes its range.
// The leader's claims are made by the caller, ONE per plane while the range is open (`rs_wants_claim` -> `rs_issue_claim`,
// somewhere inside the plane's work): a 64-bit atomic add of 1 on the block's word by lane 0, a VECTOR-memory atomic whose
// returned value stays in flight in `pending` until the top of the next plane reads it (the compiler places the wait
// there: by then the plane loads issued behind it have been waited for anyway).  Not a scalar-memory atomic: its result
// would share the lgkmcnt counter with the LDS traffic of the plane's work, and the first LDS wait behind it would wait for
// the atomic's round trip to memory -- measured on k_fused: 7.0 instead of 4.8 us per plane.
struct RsBlock {          // (all members wave-uniform)
    uint32_t col, xb;     // the block's current range: tile column, first plane
    uint32_t claimed;     // planes [xb, claimed) were owned when the range was taken (claimed == xb: an empty range)
    uint32_t known;       // leader: planes below `known` are owned ...
    bool final;           // ... and no plane at or beyond it will be once this is set
    uint32_t gen;         // ranges this block has had
    unsigned long long pending;  // leader, lane 0: what the claim made during the previous plane returns (a vector register
                                 // pair still in flight when the plane ends)
#if P3D_RS_FAKE
    uint32_t fake_end;           // dev-only timing builds: the initial range's end; no claims, no steals
#endif
};
#ifndef P3D_RS_NOSTEAL
#define P3D_RS_NOSTEAL 0
#endif
#ifndef P3D_RS_FAKE   // dev-only timing ablations (WRONG results as soon as anything would be stolen): 1 = no claim atomics, no steals; 2 = also no sibling pacing
#define P3D_RS_FAKE 0
#endif
constexpr int kRsLdsWords = 2 + 2 * 4;   // u32 words of LDS the protocol needs: front[2], range[2][4]

__device__ inline void rs_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ inline uint32_t rs_front_word(uint32_t known, bool final) { return known | (final ? 0x80000000u : 0u); }

// all waves of the block, once, at the start of the kernel (contains a block barrier)
__device__ inline void rs_start(RsBlock& s, bool leader, const RsGeom& g, unsigned long long* table, uint32_t me,
                                volatile uint32_t* lds) {
    uint32_t col, b, e;
    rs_initial(g, me, col, b, e);
    s.gen = 0;
    s.col = col;
    s.xb = b;
    s.claimed = b + kRsAhead < e ? b + kRsAhead : e;
    s.known = s.claimed;
    s.final = s.claimed >= e;
    s.pending = 0;
#if P3D_RS_FAKE
    s.fake_end = e;
#endif
    if (leader) {
        // (an atomic exchange, like every later access to the word: one ordering domain)
        if ((threadIdx.x & 63u) == 0u)
            (void)__hip_atomic_exchange(table + (size_t)me * kRsStride, rs_pack(col, s.claimed, e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds[0] = rs_front_word(s.known, s.final);
    }
    rs_lds_barrier();
}

// top of plane x (owned): will plane x + 1 be processed by this block?  Leader: books the claim made during plane x - 1.
__device__ inline bool rs_leader_own_next(RsBlock& s, uint32_t x, volatile uint32_t* lds) {
    if (!s.final && x + 1u >= s.known) {
#if P3D_RS_FAKE == 3   // (the claim is made and read, the decision is still the fixed range's)
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s.pending);
        if (s.known < s.fake_end + (lo_ == 0x7fffffffu ? 1u : 0u)) ++s.known;
#elif P3D_RS_FAKE
        if (s.known < s.fake_end && (P3D_RS_FAKE != 5 || s.gen == 0)) ++s.known;
#else
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s.pending);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(s.pending >> 32));
        if (rs_granted(((unsigned long long)hi << 32) | lo)) ++s.known;
#endif
        else s.final = true;
        lds[s.gen & 1u] = rs_front_word(s.known, s.final);
    }
    return x + 1u < s.known;
}
// the leader makes a claim (for plane `known`) during every plane it processes while its range is open
__device__ inline bool rs_wants_claim(const RsBlock& s, bool leader) { return leader && !s.final && (!P3D_RS_FAKE || P3D_RS_FAKE >= 3); }
__device__ inline void rs_issue_claim(RsBlock& s, unsigned long long* myword) {
    unsigned long long r = 0;
    if ((threadIdx.x & 63u) == 0u) r = __hip_atomic_fetch_add(myword, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s.pending = r;
}

__device__ inline bool rs_sibling_own_next(const RsBlock& s, uint32_t x, volatile uint32_t* lds) {
#if P3D_RS_FAKE == 2
    return x + 1u < s.fake_end;
#endif
    for (;;) {
        const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds[s.gen & 1u]);
        if (x + 1u < (f & 0x7fffffffu)) return true;
        if (f >> 31) return false;
        __builtin_amdgcn_s_sleep(2);   // the leader has not reached plane x yet
    }
}

// all waves, when the block's range is done: the leader steals, the others wait for it at the barrier.  false: no work
// is left anywhere that is worth splitting -- the block ends.
__device__ inline bool rs_switch(RsBlock& s, bool leader, unsigned long long* table, uint32_t nb, uint32_t me,
                                 volatile uint32_t* lds) {
    const uint32_t g1 = (s.gen + 1u) & 1u;
    volatile uint32_t* rec = lds + 2 + 4 * g1;
    if (leader) {
        uint32_t col = 0, xb = 0, claimed = 0, xend = 0;
        const bool ok = ((P3D_RS_FAKE && P3D_RS_FAKE != 5) || P3D_RS_NOSTEAL) ? false : rs_steal(table, nb, me, col, xb, claimed, xend);
        s.known = claimed;
        s.final = claimed >= xend;
        rec[0] = ok ? 1u : 0u;
        rec[1] = col;
        rec[2] = xb;
        rec[3] = claimed;
        lds[g1] = rs_front_word(s.known, s.final);
    }
    rs_lds_barrier();
    ++s.gen;
    // (LDS reads land in vector registers; the values are wave-uniform and feed scalar code)
    s.col = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[1]);
    s.xb = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[2]);
    s.claimed = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[3]);
    return __builtin_amdgcn_readfirstlane((int)rec[0]) != 0;
}
#endif  // __HIPCC__
#endif  // P3D_RANGE_SCHED_H_
