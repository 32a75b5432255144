// p3d_mt.hip -- MI355X (gfx950) marching tetrahedra + its C ABI (include/p3d_mt.h).
//
// What this replaces (paths into lzhnb/Primitive3D): prim3d/utility/marching_tetrahedras.py:89-235, a chain of ~25
// PyTorch ops (orientation test by batched determinant, boolean-mask compactions, gathers, a row-wise torch.unique over
// the edges of all active tetrahedra, table lookups).  Here: five kernels around ONE radix sort of 64-bit edge keys.
//
//   k_mt_occ        one occupancy bit per point (sdf > 0, :151): the later gathers hit a table that stays in L2
//   k_mt_classify   per tet: orientation fix in place (:147-148), occupancy case (:151-154, :194-196); the wave's ballot
//                   of active tets and its popcount are all that is kept about activity
//   (scan)          over the wave counts (nt / 64 words): slot of a wave's first active tet
//   k_mt_edges      per active tet: its CROSSING edges (endpoints differ in occupancy: the only ones that become
//                   vertices, :163-168) as keys  lo << hb | hi  (:157-159, hb = bits of a point index) into a hash set
//   (select)        the distinct keys out of the hash table
//   (radix sort)    of the distinct keys only (V of them, not 6 per active tet): ascending = their order among
//                   torch.unique's lexicographically sorted rows (:160), so the rank of a key IS its vertex id (:165-170)
//   k_mt_vertices   one thread per sorted key: interpolation with the reference's float32 operation order (:178-190);
//                   the key's rank (= vertex id) is left at its slot of the hash table
//   k_mt_faces      triangles by the 16-case table, one-triangle tets first (:205-224), tet index per face (:226-234);
//                   8 lanes per tet: six look up the ids of its edges (one probe each), the corners pick by shuffle
//
// HBM-bound integer work; no MFMA.  Scans, the selection and the sort are rocPRIM device primitives.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cstring>
#include <chrono>
#include <mutex>
#include <unordered_map>
#include <rocprim/rocprim.hpp>

#include "../../include/p3d_mt.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int kBlock = 256;

// marching_tetrahedras.py:8-29 (6 slots per case, -1 = none) and :31-34
__device__ const signed char k_tri_table[16][6] = {
    {-1, -1, -1, -1, -1, -1}, {1, 0, 2, -1, -1, -1}, {4, 0, 3, -1, -1, -1}, {1, 4, 2, 1, 3, 4},
    {3, 1, 5, -1, -1, -1},    {2, 3, 0, 2, 5, 3},    {1, 4, 0, 1, 5, 4},    {4, 2, 5, -1, -1, -1},
    {4, 5, 2, -1, -1, -1},    {4, 1, 0, 4, 5, 1},    {3, 2, 0, 3, 5, 2},    {1, 3, 5, -1, -1, -1},
    {4, 1, 2, 4, 3, 1},       {3, 0, 4, -1, -1, -1}, {2, 0, 1, -1, -1, -1}, {-1, -1, -1, -1, -1, -1}};
__device__ const unsigned char k_num_tri[16] = {0, 1, 1, 2, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 0};
// :35-45 corner pairs of the six edges
__device__ const unsigned char k_edge_a[6] = {0, 0, 0, 1, 1, 2};
__device__ const unsigned char k_edge_b[6] = {1, 2, 3, 2, 3, 3};

struct MtWs {  // byte offsets into the workspace
    size_t hdr, occ, cas, wmask, wcnt, wbase, vlist, tcount, tscan, table, rank, uniq, sorted, temp, temp_bytes, total;
};
enum { M_NVALID = 0, M_V = 1, M_N1 = 2, M_N2 = 3, M_BADIDX = 4 /* some tet index is out of range */ };
constexpr u64 kEmpty = ~0ull;   // (no key: lo < hi < 2^32 and hb <= 32)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// hash set of the crossing edges: a power of two >= 6 slots per active tet (a tet has at most 4 crossing edges, and an
// edge is shared by several tets: the load stays well below 2/3)
inline size_t table_slots(int64_t nvalid) {
    size_t s = 1024;
    while (s < (size_t)nvalid * 6) s <<= 1;
    return s;
}

struct NotEmpty {
    __device__ bool operator()(const u64& k) const { return k != kEmpty; }
};

// merge sort only for small inputs (rocPRIM's default limit is 1 Mi keys)
constexpr size_t kMergeSortLimit = 262144;
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config,
                                              kMergeSortLimit>;

size_t temp_bytes_for(int64_t nt) {
    // the largest temporary any of the primitives asks for at the worst-case sizes
    size_t a = 0, a2 = 0, b = 0, c = 0, d = 0;
    const size_t n4 = (size_t)nt * 4;
    (void)rocprim::radix_sort_keys<SortConfig>(nullptr, a, (u64*)nullptr, (u64*)nullptr, n4, 0u, 64u, (hipStream_t)0);
    (void)rocprim::radix_sort_keys<SortConfig>(nullptr, a2, (u64*)nullptr, (u64*)nullptr, std::min(n4, kMergeSortLimit), 0u,
                                               64u, (hipStream_t)0);
    a = std::max(a, a2);
    (void)rocprim::inclusive_scan(nullptr, b, (u32*)nullptr, (u32*)nullptr, (size_t)(nt + 63) / 64, rocprim::plus<u32>(),
                                  (hipStream_t)0);
    (void)rocprim::select(nullptr, c, (u64*)nullptr, (u64*)nullptr, (u64*)nullptr, table_slots(nt), NotEmpty(),
                          (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, d, (u64*)nullptr, (u64*)nullptr, 0ull, (size_t)nt, rocprim::plus<u64>(),
                                  (hipStream_t)0);
    return std::max(std::max(a, b), std::max(c, d));
}

MtWs make_ws(int64_t nv, int64_t nt) {
    MtWs w;
    const size_t n = (size_t)std::max<int64_t>(nt, 1), nw = (n + 63) / 64;
    const size_t occ_words = ((size_t)std::max<int64_t>(nv, 1) + kBlock - 1) / kBlock * (kBlock / 64);
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t r = o;
        o = align_up(o + bytes, 256);
        return r;
    };
    w.hdr = take(256);
    w.occ = take(occ_words * 8);
    w.cas = take(n);
    w.wmask = take(nw * 8);
    w.wcnt = take(nw * 4);
    w.wbase = take(nw * 4);
    w.vlist = take(n * 4);
    w.tcount = take(n * 8);
    w.tscan = take(n * 8);
    w.table = take(table_slots((int64_t)n) * 8);
    w.rank = take(table_slots((int64_t)n) * 4);
    w.uniq = take(n * 4 * 8);
    w.sorted = take(n * 4 * 8);
    w.temp_bytes = temp_bytes_for((int64_t)n);
    w.temp = take(w.temp_bytes);
    w.total = o;
    return w;
}

// ---------------------------------------------------------------------------------------------
struct P3 {
    float x, y, z;   // (one 12-byte load per point)
};

__device__ inline u32 occ_bit(const u32* __restrict__ occ, u64 i) { return (occ[i >> 5] >> (i & 31u)) & 1u; }

__global__ void __launch_bounds__(kBlock) k_mt_occ(const float* __restrict__ sdf, int64_t nv, u64* __restrict__ occ,
                                                    u64* __restrict__ hdr) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < 32) hdr[i] = 0ull;   // the 256-byte header (ordinary kernels instead of the runtime's fill path, which starts late)
    const u64 m = __ballot(i < nv && sdf[i < nv ? i : 0] > 0.f);   // :151
    if ((threadIdx.x & 63) == 0) occ[i >> 6] = m;
}

// ALIGNED: the tet array starts on a 16-byte boundary (two 16-byte loads per tet instead of four 8-byte ones)
template <bool ALIGNED>
__global__ void __launch_bounds__(kBlock) k_mt_classify(const float* __restrict__ vertices, int64_t* __restrict__ tets,
                                                        int64_t nt, int64_t nv, const u32* __restrict__ occ,
                                                        unsigned char* __restrict__ cas, u64* __restrict__ wmask,
                                                        u32* __restrict__ wcnt, u64* __restrict__ hdr) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    bool active = false, bad = false;
    if (t < nt) {
        int64_t i0, i1, i2, i3;
        if (ALIGNED) {
            const longlong2 lo = ((const longlong2*)tets)[2 * t], hi = ((const longlong2*)tets)[2 * t + 1];
            i0 = lo.x, i1 = lo.y, i2 = hi.x, i3 = hi.y;
        } else {
            i0 = tets[4 * t], i1 = tets[4 * t + 1], i2 = tets[4 * t + 2], i3 = tets[4 * t + 3];
        }
        // an index outside [0, nv) is never dereferenced (the reference's indexing raises): the tet is skipped and the
        // call fails with P3D_MT_EINDEX
        bad = (u64)i0 >= (u64)nv || (u64)i1 >= (u64)nv || (u64)i2 >= (u64)nv || (u64)i3 >= (u64)nv;
        if (bad) i0 = i1 = i2 = i3 = 0;   // (nv >= 1 whenever a tet exists and is valid; see the host check)
        // orientation: sign of det [1 p0; 1 p1; 1 p2; 1 p3] = det [p1-p0; p2-p0; p3-p0]  (:50-65; float64 from the
        // float32 coordinates -- the reference's float32 LU gives the same sign on non-degenerate cells)
        const P3* __restrict__ pts = (const P3*)vertices;
        const P3 q0 = pts[i0], q1 = pts[i1], q2 = pts[i2], q3 = pts[i3];
        const double ax = (double)q1.x - (double)q0.x, ay = (double)q1.y - (double)q0.y, az = (double)q1.z - (double)q0.z;
        const double bx = (double)q2.x - (double)q0.x, by = (double)q2.y - (double)q0.y, bz = (double)q2.z - (double)q0.z;
        const double cx = (double)q3.x - (double)q0.x, cy = (double)q3.y - (double)q0.y, cz = (double)q3.z - (double)q0.z;
        const double det = ax * (by * cz - bz * cy) - ay * (bx * cz - bz * cx) + az * (bx * cy - by * cx);
        if (det < 0.0) {  // :148  tets[flip, :2] = tets[flip][:, [1, 0]]
            const int64_t tmp = i0;
            i0 = i1;
            i1 = tmp;
            if (ALIGNED) {
                ((longlong2*)tets)[2 * t] = longlong2{i0, i1};
            } else {
                tets[4 * t] = i0;
                tets[4 * t + 1] = i1;
            }
        }
        const u32 c = occ_bit(occ, (u64)i0) | occ_bit(occ, (u64)i1) << 1 | occ_bit(occ, (u64)i2) << 2 |
                      occ_bit(occ, (u64)i3) << 3;   // :194-195
        cas[t] = (unsigned char)c;
        active = c != 0u && c != 15u;                                                       // :153-154
    }
    if (t < nt && bad) {   // inactive
        cas[t] = 0;
        active = false;
    }
    const u64 m = __ballot(active);
    const u64 anybad = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && t < nt) {
        wmask[t >> 6] = m;
        wcnt[t >> 6] = (u32)__popcll(m);
        if (anybad) hdr[M_BADIDX] = 1ull;   // (every writer stores the same value)
    }
}

__global__ void __launch_bounds__(kBlock) k_mt_fill_empty(ulonglong2* __restrict__ table, size_t n2) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (size_t)gridDim.x * kBlock)
        table[i] = ulonglong2{kEmpty, kEmpty};
}

__device__ inline u64 mix64(u64 k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    return k;
}

__device__ inline u64 edge_key(int64_t a, int64_t b, unsigned hb) {   // sorted pair (:67-83)
    return a < b ? ((u64)a << hb | (u64)b) : ((u64)b << hb | (u64)a);
}

__global__ void __launch_bounds__(kBlock) k_mt_edges(const int64_t* __restrict__ tets, int64_t nt,
                                                     const unsigned char* __restrict__ cas, const u64* __restrict__ wmask,
                                                     const u32* __restrict__ wbase, u32* __restrict__ vlist,
                                                     u64* __restrict__ table, u64 table_mask, u64* __restrict__ tcount,
                                                     unsigned hb) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= nt) return;
    const u64 m = wmask[t >> 6];
    const u32 lane = (u32)(t & 63);
    if (!((m >> lane) & 1ull)) return;
    // rank among the active tets, in tet order (wbase = inclusive scan of the wave counts)
    const u32 s = wbase[t >> 6] - (u32)__popcll(m) + (u32)__popcll(m & ((1ull << lane) - 1ull));
    vlist[s] = (u32)t;
    const int64_t idx[4] = {tets[4 * t], tets[4 * t + 1], tets[4 * t + 2], tets[4 * t + 3]};
    const u32 c = cas[t];
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        if (((c >> k_edge_a[e]) ^ (c >> k_edge_b[e])) & 1u) {   // endpoints differ in occupancy (:163)
            const u64 key = edge_key(idx[k_edge_a[e]], idx[k_edge_b[e]], hb);
            u64 h = mix64(key) & table_mask;
            for (;;) {
                const u64 prev = atomicCAS((unsigned long long*)&table[h], (unsigned long long)kEmpty, (unsigned long long)key);
                if (prev == kEmpty || prev == key) break;
                h = (h + 1) & table_mask;
            }
        }
    }
    tcount[s] = k_num_tri[c] == 1 ? 1ull : (1ull << 32);   // low word: one-triangle tets, high word: two-triangle tets
}

struct Sizes {   // what phase 1 found; kept in the last 64 bytes of the workspace's 256-byte header for phase 2
    int64_t nv, nt, nvalid, nv_out, n1, n2;
};
constexpr size_t kSizesOffset = 192;

// Sizes reach the host through a slot of pinned, host-coherent memory that the kernel which learns them writes
// (payload, then the call's sequence number with a system-scope release) and the host polls -- instead of a device-to-host
// copy plus a stream synchronisation (each costs the call ~15 us more).  Any failure falls back to copy + synchronise.
constexpr int kMtSlots = 64, kMtSlotWords = 8;

__device__ inline void mt_report(u64* mb, u64 seq, u64 a, u64 b, u64 c) {
    if (!mb) return;
    mb[1] = a;
    mb[2] = b;
    mb[3] = c;
    __threadfence_system();
    __hip_atomic_store(&mb[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_mt_report_nvalid(const u32* __restrict__ last_inclusive, const u64* __restrict__ hdr, u64* mb, u64 seq) {
    mt_report(mb, seq, (u64)*last_inclusive, hdr[M_BADIDX], 0ull);
}

__global__ void k_mt_totals(const u64* __restrict__ nuniq, const u64* __restrict__ tcount, const u64* __restrict__ tscan,
                            int64_t nv, int64_t nt, int64_t nvalid, u64* __restrict__ hdr, u64* mb, u64 seq) {
    const u64 v = nvalid > 0 ? *nuniq : 0ull;
    const u64 both = nvalid > 0 ? tscan[nvalid - 1] + tcount[nvalid - 1] : 0ull;
    hdr[M_V] = v;
    hdr[M_N1] = both & 0xffffffffull;
    hdr[M_N2] = both >> 32;
    Sizes* sz = (Sizes*)((char*)hdr + kSizesOffset);
    *sz = Sizes{nv, nt, nvalid, (int64_t)v, (int64_t)(both & 0xffffffffull), (int64_t)(both >> 32)};
    mt_report(mb, seq, v, both & 0xffffffffull, both >> 32);
}

// slot of a key that IS in the hash set
__device__ inline u64 find_slot(const u64* __restrict__ table, u64 table_mask, u64 key) {
    // (bounded: a caller that hands p3d_mt_emit other tets than p3d_mt_prepare corrected gets a wrong id, not a hang)
    u64 h = mix64(key) & table_mask;
    for (u64 probes = 0; table[h] != key && probes <= table_mask; ++probes) h = (h + 1) & table_mask;
    return h;
}

// one thread per sorted key: its rank IS the vertex id (:165-171) -- left at the key's hash slot, where the faces find
// it -- and the interpolated position
__global__ void __launch_bounds__(kBlock) k_mt_vertices(const u64* __restrict__ sorted, int64_t nkeys, unsigned hb,
                                                        const float* __restrict__ vertices, const float* __restrict__ sdf,
                                                        const u64* __restrict__ table, u64 table_mask,
                                                        u32* __restrict__ rank, float* __restrict__ out,
                                                        int64_t* __restrict__ pairs) {
    const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (v >= nkeys) return;
    const u64 k = sorted[v];
    rank[find_slot(table, table_mask, k)] = (u32)v;
    const int64_t a = (int64_t)(k >> hb), b = (int64_t)(k & ((1ull << hb) - 1ull));
    // :178-190, operation for operation in float32: [s_a, -s_b], their sum, the flipped pair divided by it, then
    // p_a * w0 + p_b * w1 (build flag -ffp-contract=off keeps the products and the sum separately rounded)
    const float sa = sdf[a], nsb = sdf[b] * -1.0f;
    const float den = __fadd_rn(sa, nsb);
    const float w0 = __fdiv_rn(nsb, den), w1 = __fdiv_rn(sa, den);
#pragma unroll
    for (int c = 0; c < 3; ++c)
        out[v * 3 + c] = __fadd_rn(__fmul_rn(vertices[a * 3 + c], w0), __fmul_rn(vertices[b * 3 + c], w1));
    if (pairs) {
        pairs[v * 2] = a;
        pairs[v * 2 + 1] = b;
    }
}

// 8 lanes per active tet: lane e < 6 looks up edge e's vertex id, lane k < 3n writes corner k of the tet's triangles
__global__ void __launch_bounds__(kBlock) k_mt_faces(const int64_t* __restrict__ tets, const u32* __restrict__ vlist,
                                                     const unsigned char* __restrict__ cas, const u64* __restrict__ tscan,
                                                     const u64* __restrict__ table, u64 table_mask,
                                                     const u32* __restrict__ rank, unsigned hb, int64_t nvalid,
                                                     const u64* __restrict__ hdr, int64_t* __restrict__ faces,
                                                     int64_t* __restrict__ tet_idx) {
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t s = g >> 3;
    const int e = (int)(g & 7);
    const bool live = s < nvalid;
    u32 t = 0, c = 0;
    u32 id = 0;
    if (live) {
        t = vlist[s];
        c = cas[t];
        if (e < 6 && (((c >> k_edge_a[e]) ^ (c >> k_edge_b[e])) & 1u)) {
            const u64 key = edge_key(tets[4 * (int64_t)t + k_edge_a[e]], tets[4 * (int64_t)t + k_edge_b[e]], hb);
            id = rank[find_slot(table, table_mask, key)];
        }
    }
    const int n = live ? k_num_tri[c] : 0;
    const int src = (e < 3 * n) ? k_tri_table[c][e] : 0;
    const u32 corner = (u32)__shfl((int)id, ((int)threadIdx.x & 56) + src, 64);   // (all lanes take part)
    if (e < 3 * n) {
        const u64 sc = tscan[s];
        // one-triangle tets first, in tet order, then the two-triangle tets (:205-224)
        const int64_t f0 = n == 1 ? (int64_t)(sc & 0xffffffffull) : (int64_t)hdr[M_N1] + 2 * (int64_t)(sc >> 32);
        faces[f0 * 3 + e] = (int64_t)corner;
        if (tet_idx && e < n) tet_idx[f0 + e] = (int64_t)t;
    }
}

thread_local char g_err[512] = "";
int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}
#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return fail(P3D_MT_EHIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

int check_sizes(int64_t nv, int64_t nt) {
    if (nv < 0 || nt < 0) return fail(P3D_MT_EINVAL, "negative size%s");
    if (nv >= (1ll << 32)) return fail(P3D_MT_ERANGE, "more than 2^32 - 1 vertices%s");
    if (nt * 6 >= (1ll << 32)) return fail(P3D_MT_ERANGE, "more than 2^32 / 6 tetrahedra%s");
    return P3D_MT_OK;
}

inline u32 blocks_for(int64_t n) { return (u32)std::max<int64_t>(1, (n + kBlock - 1) / kBlock); }

// Host-side copy of what p3d_mt_prepare found, keyed by workspace: saves p3d_mt_emit a device read-back and a
// synchronisation when it follows on the same workspace (the usual case); the device copy in the header stays the
// authority for any other caller.
struct SizesCache {
    std::mutex mu;
    std::unordered_map<const void*, Sizes> map;
    void put(const void* ws, const Sizes& sz) {
        std::lock_guard<std::mutex> g(mu);
        if (map.size() >= 64) map.clear();
        map[ws] = sz;
    }
    bool take(const void* ws, Sizes* sz) {
        std::lock_guard<std::mutex> g(mu);
        auto it = map.find(ws);
        if (it == map.end()) return false;
        *sz = it->second;
        map.erase(it);
        return true;
    }
};
SizesCache g_sizes;

struct MtMailbox {
    u64 *host = nullptr, *dev = nullptr;
    bool failed = false;
    u64 next_seq = 1;
};
std::mutex g_mtmb_mu;
MtMailbox g_mtmb[64];

// a slot for one report: its device pointer (null = unavailable), its host view and the sequence number to wait for
u64* mt_mailbox_take(volatile u64** host_slot, u64* seq) {
    static const bool disabled = getenv("P3D_NO_MAILBOX") != nullptr && atoi(getenv("P3D_NO_MAILBOX")) != 0;
    int dev = 0;
    if (disabled || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(g_mtmb_mu);
    MtMailbox& m = g_mtmb[dev];
    if (m.failed) return nullptr;
    if (!m.host) {
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, (size_t)kMtSlots * kMtSlotWords * 8,
                          hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess ||
            hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
            (void)hipGetLastError();
            m.failed = true;
            return nullptr;
        }
        memset(h, 0, (size_t)kMtSlots * kMtSlotWords * 8);
        m.host = (u64*)h;
        m.dev = (u64*)d;
    }
    *seq = m.next_seq++;
    const size_t off = (size_t)(*seq % kMtSlots) * kMtSlotWords;
    *host_slot = m.host + off;
    return m.dev + off;
}

// polls the slot; false = recycled by a much newer call or timed out (the caller then copies + synchronises)
bool mt_mailbox_wait(volatile u64* slot, u64 seq, u64 out[3]) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spin = 0;; ++spin) {
        const u64 sv = __atomic_load_n(&slot[0], __ATOMIC_ACQUIRE);
        if (sv == seq) break;
        if (sv > seq) return false;
        if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return false;
        __builtin_ia32_pause();
    }
    for (int i = 0; i < 3; ++i) out[i] = __atomic_load_n(&slot[1 + i], __ATOMIC_ACQUIRE);
    return __atomic_load_n(&slot[0], __ATOMIC_ACQUIRE) == seq;   // (seqlock: not recycled while the payload was read)
}

inline unsigned index_bits(int64_t nv) {   // bits of a point index (>= 1)
    unsigned hb = 1;
    while (hb < 32 && (1ull << hb) < (u64)nv) ++hb;
    return hb;
}

}  // namespace

extern "C" {

int p3d_mt_abi_version(void) { return P3D_MT_ABI_VERSION; }
const char* p3d_mt_last_error(void) { return g_err; }

int p3d_mt_workspace_bytes(int64_t num_vertices, int64_t num_tets, size_t* bytes) {
    if (!bytes) return fail(P3D_MT_EINVAL, "bytes is null%s");
    if (int rc = check_sizes(num_vertices, num_tets)) return rc;
    *bytes = make_ws(num_vertices, num_tets).total;
    return P3D_MT_OK;
}

int p3d_mt_prepare(const float* vertices, int64_t num_vertices, int64_t* tets, int64_t num_tets, const float* sdf,
                   void* ws_, int64_t* out_vertices, int64_t* out_faces, void* stream) {
    if (!ws_ || !out_vertices || !out_faces) return fail(P3D_MT_EINVAL, "null pointer%s");
    if (num_tets > 0 && (!vertices || !tets || !sdf)) return fail(P3D_MT_EINVAL, "null input%s");
    if (int rc = check_sizes(num_vertices, num_tets)) return rc;
    if (num_tets > 0 && num_vertices < 1) return fail(P3D_MT_EINDEX, "tets without vertices%s");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)ws_;
    const MtWs w = make_ws(num_vertices, num_tets);
    u64* hdr = (u64*)(ws + w.hdr);
    u64* occ = (u64*)(ws + w.occ);
    unsigned char* cas = (unsigned char*)(ws + w.cas);
    u64* wmask = (u64*)(ws + w.wmask);
    u32 *wcnt = (u32*)(ws + w.wcnt), *wbase = (u32*)(ws + w.wbase), *vlist = (u32*)(ws + w.vlist);
    u64 *tcount = (u64*)(ws + w.tcount), *tscan = (u64*)(ws + w.tscan);
    u64 *table = (u64*)(ws + w.table), *uniq = (u64*)(ws + w.uniq);
    void* temp = ws + w.temp;
    size_t tb = w.temp_bytes;
    const unsigned hb = index_bits(num_vertices);
    Sizes sz{num_vertices, num_tets, 0, 0, 0, 0};
    if (num_tets <= 0) HIP_TRY(hipMemsetAsync(hdr, 0, 256, st));
    if (num_tets > 0) {
        const int64_t nw = (num_tets + 63) / 64;
        hipLaunchKernelGGL(k_mt_occ, dim3(blocks_for(num_vertices)), dim3(kBlock), 0, st, sdf, num_vertices, occ, hdr);
        if (((uintptr_t)tets & 15u) == 0)
            hipLaunchKernelGGL((k_mt_classify<true>), dim3(blocks_for(num_tets)), dim3(kBlock), 0, st, vertices, tets, num_tets,
                               num_vertices, (const u32*)occ, cas, wmask, wcnt, hdr);
        else
            hipLaunchKernelGGL((k_mt_classify<false>), dim3(blocks_for(num_tets)), dim3(kBlock), 0, st, vertices, tets,
                               num_tets, num_vertices, (const u32*)occ, cas, wmask, wcnt, hdr);
        // inclusive: its last element is the number of active tets (a wave's first slot = its element - its count)
        HIP_TRY(rocprim::inclusive_scan(temp, tb, wcnt, wbase, (size_t)nw, rocprim::plus<u32>(), st));
        // (the reference synchronises here too: tets[valid_tets], :157)
        volatile u64* slot = nullptr;
        u64 seq = 0, got[3];
        u64* mb = mt_mailbox_take(&slot, &seq);
        if (mb) hipLaunchKernelGGL(k_mt_report_nvalid, dim3(1), dim3(1), 0, st, wbase + (nw - 1), (const u64*)hdr, mb, seq);
        u64 bad_index = 0;
        if (mb && mt_mailbox_wait(slot, seq, got)) {
            sz.nvalid = (int64_t)got[0];
            bad_index = got[1];
        } else {
            u32 nvalid = 0;
            HIP_TRY(hipMemcpyAsync(&nvalid, wbase + (nw - 1), sizeof(u32), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&bad_index, hdr + M_BADIDX, sizeof(u64), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            sz.nvalid = (int64_t)nvalid;
        }
        if (bad_index) return fail(P3D_MT_EINDEX, "a tet index is outside [0, num_vertices)%s");
    }
    if (sz.nvalid > 0) {
        const size_t slots = table_slots(sz.nvalid);
        hipLaunchKernelGGL(k_mt_fill_empty, dim3((u32)std::min<size_t>(slots / 2 / kBlock, 4096)), dim3(kBlock), 0, st,
                           (ulonglong2*)table, slots / 2);
        hipLaunchKernelGGL(k_mt_edges, dim3(blocks_for(num_tets)), dim3(kBlock), 0, st, tets, num_tets, cas, wmask, wbase,
                           vlist, table, (u64)(slots - 1), tcount, hb);
        tb = w.temp_bytes;
        HIP_TRY(rocprim::select(temp, tb, table, uniq, hdr + M_V, slots, NotEmpty(), st));
        tb = w.temp_bytes;
        HIP_TRY(rocprim::exclusive_scan(temp, tb, tcount, tscan, 0ull, (size_t)sz.nvalid, rocprim::plus<u64>(), st));
    }
    // totals, and the sizes for p3d_mt_emit in the workspace header (written by the device: no upload, no wait)
    volatile u64* slot2 = nullptr;
    u64 seq2 = 0, got2[3];
    u64* mb2 = sz.nvalid > 0 ? mt_mailbox_take(&slot2, &seq2) : nullptr;
    hipLaunchKernelGGL(k_mt_totals, dim3(1), dim3(1), 0, st, hdr + M_V, tcount, tscan, num_vertices, num_tets, sz.nvalid, hdr,
                       mb2, seq2);
    if (sz.nvalid > 0) {
        if (mb2 && mt_mailbox_wait(slot2, seq2, got2)) {
            sz.nv_out = (int64_t)got2[0];
            sz.n1 = (int64_t)got2[1];
            sz.n2 = (int64_t)got2[2];
        } else {
            u64 h[4] = {0, 0, 0, 0};
            HIP_TRY(hipMemcpyAsync(h, hdr, sizeof(h), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            sz.nv_out = (int64_t)h[M_V];
            sz.n1 = (int64_t)h[M_N1];
            sz.n2 = (int64_t)h[M_N2];
        }
    }
    HIP_TRY(hipGetLastError());
    g_sizes.put(ws_, sz);
    *out_vertices = sz.nv_out;
    *out_faces = sz.n1 + 2 * sz.n2;
    return P3D_MT_OK;
}

int p3d_mt_emit(const float* vertices, const int64_t* tets, const float* sdf, void* ws_, float* out_vertices,
                int64_t* out_edge_pairs, int64_t* out_faces, int64_t* out_tet_idx, void* stream) {
    if (!ws_) return fail(P3D_MT_EINVAL, "null pointer%s");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)ws_;
    Sizes sz;   // what phase 1 found: this process's copy, else the one in the workspace header
    if (!g_sizes.take(ws_, &sz)) {
        HIP_TRY(hipMemcpyAsync(&sz, ws + kSizesOffset, sizeof(sz), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    const MtWs w = make_ws(sz.nv, sz.nt);
    if (sz.nvalid <= 0) return P3D_MT_OK;
    if ((sz.nv_out > 0 && (!out_vertices || !vertices || !sdf)) || (sz.n1 + sz.n2 > 0 && (!out_faces || !tets)))
        return fail(P3D_MT_EINVAL, "null output%s");
    const unsigned hb = index_bits(sz.nv);
    u64 *uniq = (u64*)(ws + w.uniq), *sorted = (u64*)(ws + w.sorted);
    const u64* table = (const u64*)(ws + w.table);
    u32* rank = (u32*)(ws + w.rank);
    const u64 table_mask = (u64)(table_slots(sz.nvalid) - 1);
    if (sz.nv_out > 0) {
        size_t tb = w.temp_bytes;
        HIP_TRY(rocprim::radix_sort_keys<SortConfig>(ws + w.temp, tb, uniq, sorted, (size_t)sz.nv_out, 0u, 2u * hb, st));
        hipLaunchKernelGGL(k_mt_vertices, dim3(blocks_for(sz.nv_out)), dim3(kBlock), 0, st, (const u64*)sorted, sz.nv_out, hb,
                           vertices, sdf, table, table_mask, rank, out_vertices, out_edge_pairs);
    }
    if (sz.n1 + sz.n2 > 0)
        hipLaunchKernelGGL(k_mt_faces, dim3(blocks_for(sz.nvalid * 8)), dim3(kBlock), 0, st, tets, (const u32*)(ws + w.vlist),
                           (const unsigned char*)(ws + w.cas), (const u64*)(ws + w.tscan), table, table_mask,
                           (const u32*)rank, hb, sz.nvalid, (const u64*)(ws + w.hdr), out_faces, out_tet_idx);
    HIP_TRY(hipGetLastError());
    return P3D_MT_OK;
}

}  // extern "C"
