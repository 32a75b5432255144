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
//   k_mt_edges      per active tet: six sorted edge keys  lo << hb | hi  (:157-159, hb = bits of a point index) + where
//                   they came from
//   (radix sort)    keys ascending = torch.unique's lexicographic row order (:160); 2 hb bits, onesweep above 64 Ki keys
//   k_mt_flags      first key of every run whose endpoints differ in occupancy = one output vertex (:163-168)
//   (scan)          vertex id = rank of that run
//   k_mt_map        vertex id (or -1) of every (tet, edge) slot (:169)
//   k_mt_vertices   interpolation with the reference's float32 operation order (:178-190)
//   k_mt_faces      triangles by the 16-case table, one-triangle tets first (:205-224), tet index per face (:226-234)
//
// HBM-bound integer work; no MFMA.  Scans and the sort are rocPRIM device primitives.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cstring>
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
    size_t hdr, occ, cas, wmask, wcnt, wbase, vlist, keys_a, keys_b, vals_a, vals_b, cs, map, tcount, tscan, temp, temp_bytes,
        total;
};
enum { M_NVALID = 0, M_V = 1, M_N1 = 2, M_N2 = 3 };

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// merge sort only for small inputs: the 2 hb <= 64 key bits are 6-8 onesweep passes, cheaper than ~10 merge passes
// from 64 Ki keys on (measured: tools/bench_next.py)
constexpr size_t kMergeSortLimit = 65536;
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config,
                                              kMergeSortLimit>;

size_t temp_bytes_for(int64_t nt) {
    // the largest temporary any of the primitives asks for at the worst-case sizes
    size_t a = 0, a2 = 0, b = 0, c = 0, d = 0;
    const size_t n6 = (size_t)nt * 6;
    (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, a, (u64*)nullptr, (u64*)nullptr, (u32*)nullptr, (u32*)nullptr, n6,
                                                0u, 64u, (hipStream_t)0);
    (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, a2, (u64*)nullptr, (u64*)nullptr, (u32*)nullptr, (u32*)nullptr,
                                                std::min(n6, kMergeSortLimit), 0u, 64u, (hipStream_t)0);
    a = std::max(a, a2);
    (void)rocprim::exclusive_scan(nullptr, b, (u32*)nullptr, (u32*)nullptr, 0u, (size_t)(nt + 63) / 64, rocprim::plus<u32>(),
                                  (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, c, (u32*)nullptr, (u32*)nullptr, n6, rocprim::plus<u32>(), (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, d, (u64*)nullptr, (u64*)nullptr, 0ull, (size_t)nt, rocprim::plus<u64>(),
                                  (hipStream_t)0);
    return std::max(std::max(a, b), std::max(c, d));
}

MtWs make_ws(int64_t nv, int64_t nt) {
    MtWs w;
    const size_t n = (size_t)std::max<int64_t>(nt, 1), n6 = n * 6, nw = (n + 63) / 64;
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
    w.keys_a = take(n6 * 8);
    w.keys_b = take(n6 * 8);
    w.vals_a = take(n6 * 4);
    w.vals_b = take(n6 * 4);
    w.cs = take(n6 * 4);
    w.map = take(n6 * 4);
    w.tcount = take(n * 8);
    w.tscan = take(n * 8);
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

__global__ void __launch_bounds__(kBlock) k_mt_occ(const float* __restrict__ sdf, int64_t nv, u64* __restrict__ occ) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const u64 m = __ballot(i < nv && sdf[i < nv ? i : 0] > 0.f);   // :151
    if ((threadIdx.x & 63) == 0) occ[i >> 6] = m;
}

// ALIGNED: the tet array starts on a 16-byte boundary (two 16-byte loads per tet instead of four 8-byte ones)
template <bool ALIGNED>
__global__ void __launch_bounds__(kBlock) k_mt_classify(const float* __restrict__ vertices, int64_t* __restrict__ tets,
                                                        int64_t nt, const u32* __restrict__ occ,
                                                        unsigned char* __restrict__ cas, u64* __restrict__ wmask,
                                                        u32* __restrict__ wcnt) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    bool active = false;
    if (t < nt) {
        int64_t i0, i1, i2, i3;
        if (ALIGNED) {
            const longlong2 lo = ((const longlong2*)tets)[2 * t], hi = ((const longlong2*)tets)[2 * t + 1];
            i0 = lo.x, i1 = lo.y, i2 = hi.x, i3 = hi.y;
        } else {
            i0 = tets[4 * t], i1 = tets[4 * t + 1], i2 = tets[4 * t + 2], i3 = tets[4 * t + 3];
        }
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
    const u64 m = __ballot(active);
    if ((threadIdx.x & 63) == 0 && t < nt) {
        wmask[t >> 6] = m;
        wcnt[t >> 6] = (u32)__popcll(m);
    }
}

__global__ void k_mt_nvalid(const u32* __restrict__ wcnt, const u32* __restrict__ wbase, int64_t nw, u64* __restrict__ hdr) {
    hdr[M_NVALID] = nw > 0 ? (u64)wbase[nw - 1] + wcnt[nw - 1] : 0ull;
}

__global__ void __launch_bounds__(kBlock) k_mt_edges(const int64_t* __restrict__ tets, int64_t nt,
                                                     const unsigned char* __restrict__ cas, const u64* __restrict__ wmask,
                                                     const u32* __restrict__ wbase, u32* __restrict__ vlist,
                                                     u64* __restrict__ keys, u32* __restrict__ vals, u64* __restrict__ tcount,
                                                     unsigned hb) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= nt) return;
    const u64 m = wmask[t >> 6];
    const u32 lane = (u32)(t & 63);
    if (!((m >> lane) & 1ull)) return;
    const u32 s = wbase[t >> 6] + (u32)__popcll(m & ((1ull << lane) - 1ull));   // rank among the active tets (tet order)
    vlist[s] = (u32)t;
    const int64_t idx[4] = {tets[4 * t], tets[4 * t + 1], tets[4 * t + 2], tets[4 * t + 3]};
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const u64 a = (u64)idx[k_edge_a[e]], b = (u64)idx[k_edge_b[e]];
        keys[(size_t)s * 6 + e] = a < b ? (a << hb | b) : (b << hb | a);   // sorted pair (:67-83)
        vals[(size_t)s * 6 + e] = s * 6u + (u32)e;
    }
    tcount[s] = k_num_tri[cas[t]] == 1 ? 1ull : (1ull << 32);   // low word: one-triangle tets, high word: two-triangle tets
}

__device__ inline bool key_crosses(const u32* __restrict__ occ, u64 k, unsigned hb) {
    return occ_bit(occ, k >> hb) != occ_bit(occ, k & ((1ull << hb) - 1ull));   // :163
}

__global__ void __launch_bounds__(kBlock) k_mt_flags(const u64* __restrict__ keys, int64_t n6, const u32* __restrict__ occ,
                                                     unsigned hb, u32* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n6) return;
    const u64 k = keys[i];
    const bool head = i == 0 || keys[i - 1] != k;
    flag[i] = (head && key_crosses(occ, k, hb)) ? 1u : 0u;
}

__global__ void __launch_bounds__(kBlock) k_mt_map(const u64* __restrict__ keys, const u32* __restrict__ vals, int64_t n6,
                                                   const u32* __restrict__ occ, const u32* __restrict__ cs, unsigned hb,
                                                   int32_t* __restrict__ map) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n6) return;
    // every element of a run sees the run's own count (only its first element was flagged): :164-169
    map[vals[i]] = key_crosses(occ, keys[i], hb) ? (int32_t)(cs[i] - 1u) : -1;
}

struct Sizes {   // what phase 1 found; kept in the last 64 bytes of the workspace's 256-byte header for phase 2
    int64_t nv, nt, nvalid, nv_out, n1, n2;
};
constexpr size_t kSizesOffset = 192;

__global__ void k_mt_totals(const u32* __restrict__ cs, int64_t n6, const u64* __restrict__ tcount,
                            const u64* __restrict__ tscan, int64_t nv, int64_t nt, int64_t nvalid, u64* __restrict__ hdr) {
    const u64 v = n6 > 0 ? (u64)cs[n6 - 1] : 0ull;
    const u64 both = nvalid > 0 ? tscan[nvalid - 1] + tcount[nvalid - 1] : 0ull;
    hdr[M_V] = v;
    hdr[M_N1] = both & 0xffffffffull;
    hdr[M_N2] = both >> 32;
    Sizes* sz = (Sizes*)((char*)hdr + kSizesOffset);
    *sz = Sizes{nv, nt, nvalid, (int64_t)v, (int64_t)(both & 0xffffffffull), (int64_t)(both >> 32)};
}

__global__ void __launch_bounds__(kBlock) k_mt_vertices(const u64* __restrict__ keys, const u32* __restrict__ flag,
                                                        const u32* __restrict__ cs, int64_t n6, unsigned hb,
                                                        const float* __restrict__ vertices, const float* __restrict__ sdf,
                                                        float* __restrict__ out, int64_t* __restrict__ pairs) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n6 || !flag[i]) return;
    const u64 k = keys[i];
    const int64_t a = (int64_t)(k >> hb), b = (int64_t)(k & ((1ull << hb) - 1ull));
    const size_t v = (size_t)cs[i] - 1;
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

__global__ void __launch_bounds__(kBlock) k_mt_faces(const u32* __restrict__ vlist, const unsigned char* __restrict__ cas,
                                                     const u64* __restrict__ tscan, const int32_t* __restrict__ map,
                                                     int64_t nvalid, const u64* __restrict__ hdr,
                                                     int64_t* __restrict__ faces, int64_t* __restrict__ tet_idx) {
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= nvalid) return;
    const u32 t = vlist[s];
    const int c = cas[t];
    const int n = k_num_tri[c];
    const u64 sc = tscan[s];
    // one-triangle tets first, in tet order, then the two-triangle tets (:205-224)
    const int64_t f0 = n == 1 ? (int64_t)(sc & 0xffffffffull) : (int64_t)hdr[M_N1] + 2 * (int64_t)(sc >> 32);
    for (int k = 0; k < n; ++k) {
#pragma unroll
        for (int j = 0; j < 3; ++j) faces[(f0 + k) * 3 + j] = (int64_t)map[(size_t)s * 6 + k_tri_table[c][3 * k + j]];
        if (tet_idx) tet_idx[f0 + k] = (int64_t)t;
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
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)ws_;
    const MtWs w = make_ws(num_vertices, num_tets);
    u64* hdr = (u64*)(ws + w.hdr);
    u64* occ = (u64*)(ws + w.occ);
    unsigned char* cas = (unsigned char*)(ws + w.cas);
    u64* wmask = (u64*)(ws + w.wmask);
    u32 *wcnt = (u32*)(ws + w.wcnt), *wbase = (u32*)(ws + w.wbase), *vlist = (u32*)(ws + w.vlist);
    u64 *keys_a = (u64*)(ws + w.keys_a), *keys_b = (u64*)(ws + w.keys_b);
    u32 *vals_a = (u32*)(ws + w.vals_a), *vals_b = (u32*)(ws + w.vals_b), *cs = (u32*)(ws + w.cs);
    int32_t* map = (int32_t*)(ws + w.map);
    u64 *tcount = (u64*)(ws + w.tcount), *tscan = (u64*)(ws + w.tscan);
    void* temp = ws + w.temp;
    size_t tb = w.temp_bytes;
    const unsigned hb = index_bits(num_vertices);
    Sizes sz{num_vertices, num_tets, 0, 0, 0, 0};
    HIP_TRY(hipMemsetAsync(hdr, 0, 256, st));
    if (num_tets > 0) {
        const int64_t nw = (num_tets + 63) / 64;
        hipLaunchKernelGGL(k_mt_occ, dim3(blocks_for(num_vertices)), dim3(kBlock), 0, st, sdf, num_vertices, occ);
        if (((uintptr_t)tets & 15u) == 0)
            hipLaunchKernelGGL((k_mt_classify<true>), dim3(blocks_for(num_tets)), dim3(kBlock), 0, st, vertices, tets, num_tets,
                               (const u32*)occ, cas, wmask, wcnt);
        else
            hipLaunchKernelGGL((k_mt_classify<false>), dim3(blocks_for(num_tets)), dim3(kBlock), 0, st, vertices, tets,
                               num_tets, (const u32*)occ, cas, wmask, wcnt);
        HIP_TRY(rocprim::exclusive_scan(temp, tb, wcnt, wbase, 0u, (size_t)nw, rocprim::plus<u32>(), st));
        hipLaunchKernelGGL(k_mt_nvalid, dim3(1), dim3(1), 0, st, wcnt, wbase, nw, hdr);
        u64 nvalid = 0;
        HIP_TRY(hipMemcpyAsync(&nvalid, hdr + M_NVALID, sizeof(u64), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));   // (the reference synchronises here too: tets[valid_tets], :157)
        sz.nvalid = (int64_t)nvalid;
    }
    const int64_t n6 = sz.nvalid * 6;
    if (sz.nvalid > 0) {
        hipLaunchKernelGGL(k_mt_edges, dim3(blocks_for(num_tets)), dim3(kBlock), 0, st, tets, num_tets, cas, wmask, wbase,
                           vlist, keys_a, vals_a, tcount, hb);
        tb = w.temp_bytes;
        HIP_TRY(rocprim::radix_sort_pairs<SortConfig>(temp, tb, keys_a, keys_b, vals_a, vals_b, (size_t)n6, 0u, 2u * hb, st));
        hipLaunchKernelGGL(k_mt_flags, dim3(blocks_for(n6)), dim3(kBlock), 0, st, keys_b, n6, (const u32*)occ, hb, vals_a);
        tb = w.temp_bytes;
        HIP_TRY(rocprim::inclusive_scan(temp, tb, vals_a, cs, (size_t)n6, rocprim::plus<u32>(), st));
        hipLaunchKernelGGL(k_mt_map, dim3(blocks_for(n6)), dim3(kBlock), 0, st, keys_b, vals_b, n6, (const u32*)occ, cs, hb,
                           map);
        tb = w.temp_bytes;
        HIP_TRY(rocprim::exclusive_scan(temp, tb, tcount, tscan, 0ull, (size_t)sz.nvalid, rocprim::plus<u64>(), st));
    }
    // totals, and the sizes for p3d_mt_emit in the workspace header (written by the device: no upload, no wait)
    hipLaunchKernelGGL(k_mt_totals, dim3(1), dim3(1), 0, st, cs, n6, tcount, tscan, num_vertices, num_tets, sz.nvalid, hdr);
    if (sz.nvalid > 0) {
        u64 h[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(h, hdr, sizeof(h), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        sz.nv_out = (int64_t)h[M_V];
        sz.n1 = (int64_t)h[M_N1];
        sz.n2 = (int64_t)h[M_N2];
    }
    HIP_TRY(hipGetLastError());
    g_sizes.put(ws_, sz);
    *out_vertices = sz.nv_out;
    *out_faces = sz.n1 + 2 * sz.n2;
    return P3D_MT_OK;
}

int p3d_mt_emit(const float* vertices, const int64_t* tets, const float* sdf, void* ws_, float* out_vertices,
                int64_t* out_edge_pairs, int64_t* out_faces, int64_t* out_tet_idx, void* stream) {
    (void)tets;
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
    if ((sz.nv_out > 0 && (!out_vertices || !vertices || !sdf)) || (sz.n1 + sz.n2 > 0 && !out_faces))
        return fail(P3D_MT_EINVAL, "null output%s");
    const int64_t n6 = sz.nvalid * 6;
    const unsigned hb = index_bits(sz.nv);
    if (sz.nv_out > 0)
        hipLaunchKernelGGL(k_mt_vertices, dim3(blocks_for(n6)), dim3(kBlock), 0, st, (const u64*)(ws + w.keys_b),
                           (const u32*)(ws + w.vals_a), (const u32*)(ws + w.cs), n6, hb, vertices, sdf, out_vertices,
                           out_edge_pairs);
    if (sz.n1 + sz.n2 > 0)
        hipLaunchKernelGGL(k_mt_faces, dim3(blocks_for(sz.nvalid)), dim3(kBlock), 0, st, (const u32*)(ws + w.vlist),
                           (const unsigned char*)(ws + w.cas), (const u64*)(ws + w.tscan), (const int32_t*)(ws + w.map),
                           sz.nvalid, (const u64*)(ws + w.hdr), out_faces, out_tet_idx);
    HIP_TRY(hipGetLastError());
    return P3D_MT_OK;
}

}  // extern "C"
