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
//     bit 63 valid | bits 50..62 tile column | bits 25..49 end | bits 0..24 next      (planes relative to the launch's first)
//   owner : claims plane `next` with ONE 64-bit atomic add of 1 and owns it iff the returned next < the returned end
//           (claims run two planes ahead of the plane being processed, so the answer is never waited for);
//   thief : reads every block's word (one coalesced pass: 8 KiB for 1024 blocks), picks the largest end - next and
//           lowers its `end` with ONE 64-bit compare-and-swap of the whole word -- which fails if the owner claimed a
//           plane or another thief came first in between -- then publishes [new end, old end) as its own range.
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
constexpr int kRsColBits = 13;                            // tile columns < 2^13
constexpr int kRsMaxBlocks = 1024;                        // words of a table (a thief reads all of them: 16 per lane)
constexpr unsigned kRsMinSteal = 4;                       // unclaimed planes a range must have to be split (thief takes half)
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
    const uint32_t i = (g.nb % 8u == 0u) ? (b & 7u) * (g.nb >> 3) + (b >> 3) : b;
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
    return (1ull << 63) | ((unsigned long long)col << (2 * kRsBits)) | ((unsigned long long)end << kRsBits) | next;
}
P3D_RS_HD inline uint32_t rs_next(unsigned long long w) { return (uint32_t)(w & kRsMask); }
P3D_RS_HD inline uint32_t rs_end(unsigned long long w) { return (uint32_t)((w >> kRsBits) & kRsMask); }
P3D_RS_HD inline uint32_t rs_col(unsigned long long w) { return (uint32_t)((w >> (2 * kRsBits)) & ((1u << kRsColBits) - 1u)); }
// the claim whose atomic add returned `old` was granted
P3D_RS_HD inline bool rs_granted(unsigned long long old) { return rs_next(old) < rs_end(old); }

#if defined(__HIPCC__)
__device__ inline unsigned long long rs_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by ALL 64 lanes of one wave (wave-uniform arguments and results).  On success the wave's block owns planes
// [xb, xend) of column `col`; its own word `table[me]` holds the range with the first min(kRsAhead, length) planes already
// claimed (`claimed` = xb + that many), open to other thieves.
//   scan : every block's word in ONE batch of loads (sc1: the words are written by atomics of all XCDs; up to 32 per lane),
//          R = the largest number of unclaimed planes;
//   pick : among the ranges with at least max(kRsMinSteal, R / 2) unclaimed planes the one that follows this block most
//          closely in block order -- blocks that run dry at the same moment (most do, at the end of a launch) then go for
//          different victims instead of all for the one longest range;
//   take : compare-and-swap loop on the victim's word -- a failed swap returns the word as it is now (the owner claimed a
//          plane, another thief took a part), the split is redone on that and tried again at once: the window of a try is
//          one atomic round trip, the owner changes its word once per plane.
__device__ inline bool rs_steal(unsigned long long* table, uint32_t nb, uint32_t me, uint32_t& col, uint32_t& xb,
                                uint32_t& claimed, uint32_t& xend) {
    const uint32_t lane = threadIdx.x & 63u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (int)(nb * 8u), 0x00020000);
    constexpr int kGroups = kRsMaxBlocks / 64;
#pragma nounroll
    for (int attempt = 0; attempt < 6; ++attempt) {
        asm volatile("" ::: "memory");   // (the table is re-read in every attempt)
        uint32_t rem[kGroups];
        uint32_t best = 0;
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        u32x2 v[kGroups];
        // (unconditional: words beyond the table come back as 0 = "not started" from the buffer's range check, without a
        //  memory access -- a branch per group would put a wait between the loads)
#pragma unroll
        for (int gI = 0; gI < kGroups; ++gI)
            v[gI] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)((lane + 64u * (uint32_t)gI) * 8u), 0, 16 /* sc1 */);
#pragma unroll
        for (int gI = 0; gI < kGroups; ++gI) {
            const unsigned long long w = ((unsigned long long)v[gI].y << 32) | v[gI].x;
            const uint32_t nx = rs_next(w), en = rs_end(w);
            rem[gI] = ((w >> 63) && en > nx && lane + 64u * (uint32_t)gI != me) ? en - nx : 0u;
            best = rem[gI] > best ? rem[gI] : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)best, o, 64);
            best = other > best ? other : best;
        }
        if (best < kRsMinSteal) return false;   // nothing worth splitting anywhere
        const uint32_t thr = best / 2u > kRsMinSteal ? best / 2u : kRsMinSteal;
        uint32_t near = 0;                      // nb - cyclic distance from this block: larger = closer behind it
#pragma unroll
        for (int gI = 0; gI < kGroups; ++gI) {
            const uint32_t k = lane + 64u * (uint32_t)gI;
            const uint32_t dist = k > me ? k - me : k + nb - me;   // 1 .. nb (for the words that exist)
            const uint32_t score = rem[gI] >= thr ? ((nb - dist + 1u) << 12) | k : 0u;
            near = score > near ? score : near;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)near, o, 64);
            near = other > near ? other : near;
        }
        const uint32_t victim = near & 0xfffu;
        unsigned long long cur = 0;
        if (lane == 0) cur = rs_load(table + victim);
#pragma nounroll
        for (int tries = 0; tries < 8; ++tries) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cur);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(cur >> 32));
            const unsigned long long w = ((unsigned long long)hi << 32) | lo;
            const uint32_t nx = rs_next(w), en = rs_end(w);
            if (!(w >> 63) || en <= nx || en - nx < kRsMinSteal) break;   // shrunk meanwhile: scan again
            const uint32_t take = (en - nx) / 2u;                         // >= 2
            const uint32_t mid = en - take;
            int ok = 0;
            if (lane == 0) {
                unsigned long long expect = w;
                ok = __hip_atomic_compare_exchange_strong(table + victim, &expect, rs_pack(rs_col(w), nx, mid),
                                                          __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
                cur = expect;
            }
            ok = __builtin_amdgcn_readfirstlane(ok);
            if (!ok) continue;
            col = rs_col(w);
            xb = mid;
            xend = en;
            claimed = mid + (take < kRsAhead ? take : kRsAhead);
            if (lane == 0)   // (an atomic exchange, like every other access to the word: one ordering domain)
                (void)__hip_atomic_exchange(table + me, rs_pack(col, claimed, en), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
    }
    return false;
}

// ---- the block-level protocol: wave 0 of a block (the LEADER) talks to the table; its three siblings follow it through
// two LDS words per generation (a generation = one range of the block).  Siblings may lag behind the leader but never
// run ahead of its decisions; the four waves meet at a barrier only when the block changes its range.
struct RsBlock {          // (all members wave-uniform)
    uint32_t col, xb;     // the block's current range: tile column, first plane
    uint32_t claimed;     // planes [xb, claimed) were owned when the range was taken (claimed == xb: an empty range)
    uint32_t known;       // leader: planes below `known` are owned ...
    bool final;           // ... and no plane at or beyond it will be once this is set
    uint32_t gen;         // ranges this block has had
    unsigned long long pending;   // leader, lane 0: what the claim in flight (for plane `known`) returned
};
constexpr int kRsLdsWords = 2 + 2 * 4;   // u32 words of LDS the protocol needs: front[2], range[2][4]

__device__ inline void rs_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ inline void rs_issue_claim(RsBlock& s, unsigned long long* myword) {
    unsigned long long r = 0;
    if ((threadIdx.x & 63u) == 0u) r = __hip_atomic_fetch_add(myword, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s.pending = r;
}
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
    if (leader) {
        // (an atomic exchange, like every later access to the word: one ordering domain)
        if ((threadIdx.x & 63u) == 0u)
            (void)__hip_atomic_exchange(table + me, rs_pack(col, s.claimed, e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds[0] = rs_front_word(s.known, s.final);
        if (!s.final) rs_issue_claim(s, table + me);
    }
    rs_lds_barrier();
}

// top of plane x (owned): will plane x + 1 be processed by this block?  Leader: consumes the claim issued one plane ago.
__device__ inline bool rs_leader_own_next(RsBlock& s, uint32_t x, unsigned long long* myword, volatile uint32_t* lds) {
    if (!s.final && x + 1u >= s.known) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s.pending);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(s.pending >> 32));
        if (rs_granted(((unsigned long long)hi << 32) | lo)) {
            ++s.known;
            rs_issue_claim(s, myword);
        } else {
            s.final = true;
        }
        lds[s.gen & 1u] = rs_front_word(s.known, s.final);
    }
    return x + 1u < s.known;
}
__device__ inline bool rs_sibling_own_next(const RsBlock& s, uint32_t x, volatile uint32_t* lds) {
    for (;;) {
        const uint32_t f = lds[s.gen & 1u];
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
        const bool ok = rs_steal(table, nb, me, col, xb, claimed, xend);
        s.known = claimed;
        s.final = claimed >= xend;
        rec[0] = ok ? 1u : 0u;
        rec[1] = col;
        rec[2] = xb;
        rec[3] = claimed;
        lds[g1] = rs_front_word(s.known, s.final);
        if (ok && !s.final) rs_issue_claim(s, table + me);
    }
    rs_lds_barrier();
    ++s.gen;
    s.col = rec[1];
    s.xb = rec[2];
    s.claimed = rec[3];
    return rec[0] != 0u;
}
#endif  // __HIPCC__
#endif  // P3D_RANGE_SCHED_H_
