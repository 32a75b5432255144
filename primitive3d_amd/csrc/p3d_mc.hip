// p3d_mc.hip -- MI355X (gfx950 / CDNA4) marching-cubes hot path + its C ABI (include/p3d_mc.h).
//
// What this replaces (paths into lzhnb/Primitive3D):
//   count_vertices_faces_kernel / gen_vertices_kernel / gen_faces_kernel and their host driver,
//   src/prim3d/Utility/marching_cubes.cu:4-305.
//
// Design: the scalar field is read ONCE, coalesced along the contiguous z axis, with
// one lane per voxel of a 64-voxel "unit"; the wave64 compare result (v_cmp -> SGPR pair) IS the
// unit's sign word, so classification costs one VALU instruction per 64 voxels.  Everything after
// that (edge crossings, cell masks, counts, prefix sums, vertex ids) is bit arithmetic on those
// words (1 bit/voxel, L2/Infinity-Cache resident) instead of the reference's 12 B/voxel
// vertex_grids and per-cell global atomics.
//
//   unit u = (x*ry + y)*ncz + c   <->  voxels (x, y, 64c .. 64c+63),  ncz = ceil(rz/64)
//   bits[u]  : u64, bit k = field(x,y,64c+k) > thresh           (strict >, marching_cubes.cu:25)
//   rec[u]   : {base, offY | offZ<<16}: vertex ids of the unit's edges are
//                 axis0 (x edges): base        + rank among Cx bits below
//                 axis1 (y edges): base + offY + rank among Cy bits below
//                 axis2 (z edges): base + offZ + rank among Cz bits below
//              with Cx = bits[u]^bits[u+P], Cy = bits[u]^bits[u+ncz], Cz = bits[u]^(bits[u]>>1|next<<63)
//
// No CUDA-compat shims, no dual paths: gfx950 only (wave64 hard-coded).
#include <hip/hip_fp16.h>
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <type_traits>
#include <vector>

#include "../../include/p3d_mc.h"
#include "fastdiv.h"
#include "half_round.h"
#include "tri_table_packed.inc"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

#include "tri_count_bitsliced.inc"

// The case table (marching_cubes.h:21-277, nibble-packed in tri_table_packed.inc) as k_faces reads it: indexed by the
// INTERLEAVED corner mask (bit 2k = column k at z, bit 2k+1 = column k at
// z+1 -- what one shift per column yields; the reference's mask, marching_cubes.cu:49-57, has bits 0-3 at z and 4-7 at
// z+1: a fixed permutation of the index), with the row's triangle count in the top nibble (a row uses 15 nibbles).
struct TriRows {
    u64 r[256];
};
constexpr TriRows make_interleaved_rows() {
    constexpr u64 packed[256] = {P3D_TRI_TABLE_PACKED};
    constexpr unsigned char count[256] = {P3D_TRI_COUNT};
    TriRows t{};
    for (int i = 0; i < 256; ++i) {
        int m = 0;
        for (int k = 0; k < 4; ++k) m |= (((i >> (2 * k)) & 1) << k) | (((i >> (2 * k + 1)) & 1) << (k + 4));
        t.r[i] = (packed[m] & 0x0fffffffffffffffull) | ((u64)count[m] << 60);
    }
    return t;
}
__device__ const TriRows g_tri_rows = make_interleaved_rows();

struct Dims {
    int64_t rx, ry, rz;   // rx = ALL planes of the call (a batch of B grids is a stack of B * xper planes)
    int64_t P;  // units per x plane = ry * ncz
    int64_t U;  // total units
    int ncz;    // 64-voxel chunks per z row
    int ztail;  // rz % 64
    int64_t xper;   // planes per item: plane x belongs to item x / xper and has an upper neighbour unless it is the
    int nitems;     // item's last plane (a single grid: xper = rx, nitems = 1)
    int stack;      // 1: the batched entry (per-item cursor blocks in the workspace, item offsets written), even for B = 1
};

struct Ws {  // byte offsets into the workspace
    size_t hdr, bits, rec, cnt, bsum_v, bbase_v, chunk_sum, chunk_pre, wave_off, tile_tris, cur, total;
    int64_t nchunks;  // face chunks (tile column x face_chunk_planes planes), all items
    int64_t cpi;      // chunks per item
    int xw;
    int64_t nb_v, nb_f, tpp;  // unit blocks, face blocks, face tiles per plane
};

constexpr int kBlock = 256;
constexpr int kPreMinChunks = 1024;   // more face chunks than this: a one-block scan (k_chunk_prefix / k_stack_finish) leaves their
                                      // exclusive prefix between the counting and the face launch
constexpr int kHdrBytes = 8192;   // [0,256) totals/flags; [256,4352) the 32 cursors of a multi-part extraction (a whole-grid
                                  // call uses a block of the library's ring, see cursor_block_for); [4352,4608) prefixes
// header slots (u64)
enum { H_V = 0, H_T = 1, H_FLAGS = 2, H_RECFORM = 3 /* 1: rec[].x still region * 2^26 + slot; 2: rec[].x is a row of the
                                                         region layout, rows at or beyond V are remapped (H_TAIL_*) */,
       H_CURSORS = 32 /* u64 index */, H_PREFIX = 32 + 32 * 16,
       H_LAYOUT = 600 /* [41] first row of every region and of every spill area, and the end: copied there by the streaming kernel */,
       // the 40 intervals of the layout (32 regions + 8 spill areas), as the counting launch's riding block 0 leaves them:
       H_LAY_OCC = 650 /* [40] rows interval i really holds */,
       H_TAIL_TS = 700 /* [40] first row at or beyond V of interval i */, H_TAIL_TP = 750 /* [40] such rows in intervals < i */,
       H_TAIL_HS = 800 /* [40] first free row below V behind interval i's rows */, H_TAIL_HP = 850 /* [41] free rows behind intervals < i */,
       H_TAIL_TE = 900 /* [40] end of interval i's rows at or beyond V (= ts_i when it has none) */ };
constexpr int kLayIv = 40;   // intervals of a region layout: the 32 regions and the 8 spill areas (kRegions + kSpillAreas)

__host__ __device__ inline Dims make_dims(int64_t rx, int64_t ry, int64_t rz) {
    Dims d;
    d.rx = rx;
    d.ry = ry;
    d.rz = rz;
    d.ncz = (int)((rz + 63) / 64);
    d.ztail = (int)(rz % 64);
    d.P = ry * d.ncz;
    d.U = rx * d.P;
    d.xper = rx;
    d.nitems = 1;
    d.stack = 0;
    return d;
}
// B grids of [rx, ry, rz], back to back in memory
__host__ __device__ inline Dims make_dims_stack(int64_t nitems, int64_t rx, int64_t ry, int64_t rz) {
    Dims d = make_dims(nitems * rx, ry, rz);
    d.xper = rx;
    d.nitems = (int)nitems;
    d.stack = 1;
    return d;
}

// face chunk = one tile column over this many planes: 8, doubled while there are more than 4096 chunks and a chunk does
// not yet span the whole item (k_faces adds the chunk totals before a tile's chunk, or reads their prefix); small grids
// take 4-plane chunks: with 8 planes a 256^3 grid has 128 chunk blocks for 256 CUs and each walks 8 planes in sequence
// (11.4 -> 8.2 us there).  A stack of very many small items can still have more than 4096 chunks (one per item and tile
// column at least): every consumer of the chunk arrays loops over their real number.
inline int face_chunk_planes(int64_t rx, int64_t tpp, int64_t nitems = 1) {
    int xw = 8;
    if (((rx - 1 + 7) / 8) * tpp * nitems < 512) xw = 4;
    while (xw < rx - 1 && ((rx - 1 + xw - 1) / xw) * tpp * nitems > 4096) xw *= 2;
    return xw;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

Ws make_ws(const Dims& d) {
    Ws w;
    w.nb_v = (d.U + kBlock - 1) / kBlock;
    w.tpp = (d.P + kBlock - 1) / kBlock;
    w.nb_f = (d.rx > 1 ? d.rx - 1 : 0) * w.tpp;
    size_t o = 0;
    w.hdr = o;
    o += kHdrBytes;
    w.bits = o;
    o = align_up(o + (size_t)(d.U + 2) * 8, 256);
    w.rec = o;
    o = align_up(o + (size_t)(d.U + 2) * 8, 256);
    w.cnt = o;
    o = align_up(o + (size_t)d.U * 4, 256);
    // scan arrays: padded to 1024 runs of a multiple of 4 entries (k_scan_blocks reads/writes whole runs)
    auto scan_pad = [](int64_t nb) { return (size_t)(((nb + 1023) / 1024 + 3) / 4 * 4) * 1024 + 16; };
    w.bsum_v = o;
    o = align_up(o + scan_pad(w.nb_v) * 4, 256);
    w.bbase_v = o;
    o = align_up(o + scan_pad(w.nb_v) * 4, 256);
    w.xw = face_chunk_planes(d.xper, w.tpp, d.nitems);
    w.cpi = d.xper > 1 ? ((d.xper - 1 + w.xw - 1) / w.xw) * w.tpp : 0;
    w.nchunks = w.cpi * d.nitems;
    w.chunk_sum = o;   // triangles per chunk
    o = align_up(o + (size_t)(w.nchunks + 1) * 4, 256);
    w.chunk_pre = o;   // exclusive prefix of chunk_sum (by k_chunk_prefix / k_stack_finish when there are many chunks)
    o = align_up(o + (size_t)(w.nchunks + 1) * 4, 256);
    w.wave_off = o;    // first face of every (tile, wave), relative to its chunk
    o = align_up(o + (size_t)(w.nb_f + 1) * 4 * 4, 256);
    w.tile_tris = o;   // triangles of every face tile (an empty tile's block returns at once)
    o = align_up(o + (size_t)(w.nb_f + 1) * 4, 256);
    w.cur = o;         // a stack of items keeps one block of 32 vertex cursors per item here
    if (d.stack) o = align_up(o + (size_t)d.nitems * 32 * 16 * 8, 256);
    w.total = o;
    return w;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ inline u64 zmask(const Dims& d, int c) {  // voxels of chunk c that exist (z < rz)
    return (c == d.ncz - 1 && d.ztail) ? ((1ull << d.ztail) - 1ull) : ~0ull;
}
__device__ inline u64 zedge(const Dims& d, int c) {  // voxels of chunk c whose z+1 neighbour exists
    if (c != d.ncz - 1) return ~0ull;
    const int last = d.ztail ? d.ztail : 64;  // voxels in the last chunk
    return last >= 2 ? ((1ull << (last - 1)) - 1ull) : 0ull;  // last-1 <= 63
}
__device__ inline int popc64(u64 v) { return __popcll(v); }
__device__ inline u64 below(int k) { return (1ull << k) - 1ull; }  // k in [0,63]

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N)
template <int I, int N, typename F>
__device__ inline void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// (lo,hi)[lane J] = the two halves of the wave-uniform 64-bit mask m (a v_cmp result).
//  * The lane select must be an inline constant: on gfx9 a VOP3 may read only one SGPR, and the value is one.
//  * HAZARD (measured on gfx950, tools/ubench/writelane_hazard.hip): v_writelane_b32 that reads an SGPR or VCC
//    written by the IMMEDIATELY preceding VALU instruction (the v_cmp) receives the register's OLD contents.
//    One wait state is enough; the compiler cannot see into an asm statement, so the s_nop is in the string.
template <int J>
__device__ inline void write_lane2(int& lo, int& hi, u64 m) {
    asm("s_nop 0\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
        : "+v"(lo), "+v"(hi)
        : "s"((int)(u32)m), "s"((int)(u32)(m >> 32)), "n"(J)
        : "vcc");
}
__device__ inline float load_f32(const float* p) { return *p; }
__device__ inline float load_f32(const __half* p) { return __half2float(*p); }

__device__ inline u32 block_reduce_sum(u32 v, u32* s_tmp /* >= 4 */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) s_tmp[wave] = v;
    __syncthreads();
    return s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
}

// exclusive scan over the 256 threads of a block; *total receives the block sum
__device__ inline u32 block_excl_scan(u32 v, u32* s_tmp /* >= 4 */, u32* total) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32 inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        u32 t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    u32 wbase = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) wbase += s_tmp[w];
    *total = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
    return wbase + inc - v;
}

// ---------------------------------------------------------------------------------------------
// K1: classify.  HBM-bound stream over the field: one lane per voxel, the wave-wide compare result
// is the unit's sign word.  64 words are collected across lanes (v_writelane) and stored coalesced.
// Replaces the `density > thresh` tests of marching_cubes.cu:25,31,38,45,50-57,103,116,129,169-176.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock) k_classify(const T* __restrict__ grid, float thresh, Dims d,
                                                     u64* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
    const int64_t nrows = d.rx * d.ry;
    const int64_t nvox = nrows * d.rz;
    for (int64_t u0 = wave * 64; u0 < d.U; u0 += nwaves * 64) {
        int64_t row = u0 / d.ncz;
        int c = (int)(u0 - row * d.ncz);
        int wlo = 0, whi = 0;
        static_for<0, 64>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int z = c * 64 + lane;
            int64_t idx = row * d.rz + z;
            idx = idx < nvox ? idx : nvox - 1;  // clamp: the load is unconditional, the predicate is not
            const float v = load_f32(grid + idx);
            const bool inside = (row < nrows) && (z < d.rz) && (v > thresh);
            const u64 m = __ballot(inside);
            write_lane2<j>(wlo, whi, m);
            if (++c == d.ncz) {
                c = 0;
                ++row;
            }
        });
        if (u0 + lane < d.U) bits[u0 + lane] = ((u64)(u32)whi << 32) | (u32)wlo;
    }
}

// ---------------------------------------------------------------------------------------------
// K2: per-unit edge-crossing counts (lane = unit) + block sums.  All inputs are sign words.
// Replaces the three atomicAdd(counters, 1) of marching_cubes.cu:29-45.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_unit_counts(const u64* __restrict__ bits, Dims d, int halo_last,
                                                        u32* __restrict__ cnt, u32* __restrict__ bsum) {
    __shared__ u32 s_tmp[4];
    const int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    u32 n = 0;
    if (u < d.U) {
        const int64_t x = u / d.P;
        const int64_t p = u - x * d.P;
        const int64_t y = p / d.ncz;
        const int c = (int)(p - y * d.ncz);
        const u64 W = bits[u];
        const u64 zm = zmask(d, c);
        u32 nX = 0, nY = 0, nZ = 0;
        if (x + 1 < d.rx) nX = popc64((W ^ bits[u + d.P]) & zm);
        if (!(halo_last && x == d.rx - 1)) {
            if (y + 1 < d.ry) nY = popc64((W ^ bits[u + d.ncz]) & zm);
            const u64 nb = (c + 1 < d.ncz) ? (bits[u + 1] & 1ull) : 0ull;
            nZ = popc64((W ^ ((W >> 1) | (nb << 63))) & zedge(d, c));
        }
        cnt[u] = nX | (nY << 8) | (nZ << 16);
        n = nX + nY + nZ;
    }
    const u32 tot = block_reduce_sum(n, s_tmp);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// Result mailbox: a 64-byte slot of pinned, host-coherent memory per call.  The kernel that learns a total stores it
// there (payload first, then the call's sequence number with system-scope release), so the host can size the
// output tensors while the remaining kernels are still running -- no copy engine, no stream synchronisation.
//   slot[0] = seq (vertices valid)   slot[1] = V   slot[2] = flags     slot[3] = seq (faces valid)   slot[4] = F
__device__ inline void mb_publish_v(u64* slot, u64 seq, u64 nv, u64 flags) {
    if (!slot) return;
    __hip_atomic_store(slot + 1, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(slot + 2, flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(slot + 0, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (the 32 region totals of the streaming kernel travel in the same slot, words 8..39: written before the sequence word)
__device__ inline void mb_put_region(u64* slot, int r, u64 n) {
    if (slot) __hip_atomic_store(slot + 8 + r, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline void mb_publish_f(u64* slot, u64 seq, u64 nf) {
    if (!slot) return;
    __hip_atomic_store(slot + 4, nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(slot + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// exclusive scan of block sums by ONE block of 1024 threads.  Thread t owns the contiguous run
// [t*per, (t+1)*per), per a multiple of 4: 16-byte loads with no dependence between them, one block-wide scan of
// the run totals, 16-byte stores.  The arrays are padded to 1024*per entries (make_ws).
__global__ void __launch_bounds__(1024) k_scan_blocks(const u32* __restrict__ bsum, u32* __restrict__ bbase,
                                                      int64_t nb, u64* __restrict__ total_out, u64* mb, u64 seq,
                                                      int faces) {
    __shared__ u64 s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t per = ((nb + 1023) / 1024 + 3) / 4 * 4;
    const int64_t i0 = (int64_t)tid * per;
    const uint4* src = (const uint4*)(bsum + i0);
    u64 sum = 0;
    for (int64_t q = 0; q < per / 4; ++q) {
        const uint4 v = src[q];
        const int64_t i = i0 + q * 4;
        sum += (i < nb ? v.x : 0u) + (u64)(i + 1 < nb ? v.y : 0u) + (i + 2 < nb ? v.z : 0u) + (i + 3 < nb ? v.w : 0u);
    }
    u64 inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u64 t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    u64 wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        if (w < wave) wbase += s_w[w];
        total += s_w[w];
    }
    if (tid == 0) {  // the total first: the host is waiting for it
        *total_out = total;
        if (faces) mb_publish_f(mb, seq, total);
        else mb_publish_v(mb, seq, total, 0ull);
    }
    u64 run = wbase + inc - sum;
    uint4* dst = (uint4*)(bbase + i0);
    for (int64_t q = 0; q < per / 4; ++q) {
        const uint4 v = src[q];
        const int64_t i = i0 + q * 4;
        uint4 o;
        o.x = (u32)(run > 0xffffffffull ? 0xffffffffull : run);
        run += (i < nb ? v.x : 0u);
        o.y = (u32)(run > 0xffffffffull ? 0xffffffffull : run);
        run += (i + 1 < nb ? v.y : 0u);
        o.z = (u32)(run > 0xffffffffull ? 0xffffffffull : run);
        run += (i + 2 < nb ? v.z : 0u);
        o.w = (u32)(run > 0xffffffffull ? 0xffffffffull : run);
        run += (i + 3 < nb ? v.w : 0u);
        dst[q] = o;
    }
}

// per-unit vertex-id records from counts + scanned block bases
__global__ void __launch_bounds__(kBlock) k_unit_records(const u32* __restrict__ cnt, const u32* __restrict__ bbase,
                                                         Dims d, int halo_last, uint2* __restrict__ rec) {
    __shared__ u32 s_tmp[4];
    const int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const u32 cw = (u < d.U) ? cnt[u] : 0;
    const u32 nX = cw & 0xff, nY = (cw >> 8) & 0xff, nZ = (cw >> 16) & 0xff;
    u32 tot;
    const u32 ex = block_excl_scan(nX + nY + nZ, s_tmp, &tot);
    if (u < d.U) {
        const bool is_halo = halo_last && (u >= (d.rx - 1) * d.P);
        if (!is_halo) rec[u] = make_uint2(bbase[blockIdx.x] + ex, nX | ((nX + nY) << 16));
    }
}

// ---------------------------------------------------------------------------------------------
// K3 (exact mode): vertex emission by gather -- one wave per unit with crossings, lane = voxel.
// Arithmetic follows gen_vertices_kernel (marching_cubes.cu:100-135) and the epilogue (:290-298)
// operation for operation: IEEE divide, float(coord)+dt, then a separately rounded multiply and add.
// ---------------------------------------------------------------------------------------------
struct Xform {
    float sx, sy, sz, ox, oy, oz;
};

__device__ inline void put_vertex(float* __restrict__ verts, int64_t* __restrict__ keys, int64_t cap, u32 vid,
                                  float px, float py, float pz, const Xform& t, int64_t key) {
    if ((int64_t)vid < cap) {
        float* o = verts + (int64_t)vid * 3;
        o[0] = __fadd_rn(__fmul_rn(px, t.sx), t.ox);
        o[1] = __fadd_rn(__fmul_rn(py, t.sy), t.oy);
        o[2] = __fadd_rn(__fmul_rn(pz, t.sz), t.oz);
        if (keys) keys[vid] = key;
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock) k_emit_vertices(const T* __restrict__ grid, float thresh, Dims d,
                                                          const u64* __restrict__ bits, const u32* __restrict__ cnt,
                                                          const uint2* __restrict__ rec, Xform t, int64_t x_origin,
                                                          float* __restrict__ verts, int64_t cap,
                                                          int64_t* __restrict__ keys) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
    const u64 lt = below(lane);
    for (int64_t u = wave; u < d.U; u += nwaves) {
        const u32 cw = cnt[u];
        if (cw == 0) continue;  // wave-uniform
        const int64_t x = u / d.P;
        const int64_t p = u - x * d.P;
        const int64_t y = p / d.ncz;
        const int c = (int)(p - y * d.ncz);
        const u64 W = bits[u];
        const u64 zm = zmask(d, c);
        const u64 Cx = (cw & 0xff) ? ((W ^ bits[u + d.P]) & zm) : 0ull;
        const u64 Cy = ((cw >> 8) & 0xff) ? ((W ^ bits[u + d.ncz]) & zm) : 0ull;
        u64 Cz = 0;
        if ((cw >> 16) & 0xff) {
            const u64 nb = (c + 1 < d.ncz) ? (bits[u + 1] & 1ull) : 0ull;
            Cz = (W ^ ((W >> 1) | (nb << 63))) & zedge(d, c);
        }
        const uint2 r = rec[u];
        const int z = c * 64 + lane;
        const int64_t lin = (x * d.ry + y) * d.rz + z;
        const bool zin = z < d.rz;
        const float d0 = zin ? load_f32(grid + lin) : 0.f;
        const float fx = (float)(x + x_origin), fy = (float)y, fz = (float)z;
        if ((Cx >> lane) & 1ull) {
            const float d1 = load_f32(grid + lin + d.ry * d.rz);
            const float dt = (thresh - d0) / (d1 - d0);
            put_vertex(verts, keys, cap, r.x + popc64(Cx & lt), fx + dt, fy, fz, t, lin * 3 + 0);
        }
        if ((Cy >> lane) & 1ull) {
            const float d1 = load_f32(grid + lin + d.rz);
            const float dt = (thresh - d0) / (d1 - d0);
            put_vertex(verts, keys, cap, r.x + (r.y & 0xffff) + popc64(Cy & lt), fx, fy + dt, fz, t, lin * 3 + 1);
        }
        if ((Cz >> lane) & 1ull) {
            const float d1 = load_f32(grid + lin + 1);
            const float dt = (thresh - d0) / (d1 - d0);
            put_vertex(verts, keys, cap, r.x + (r.y >> 16) + popc64(Cz & lt), fx, fy, fz + dt, t, lin * 3 + 2);
        }
    }
}

#include "fused_stream.inc"

// ---------------------------------------------------------------------------------------------
// Faces.  Two kernels, both over TILES of 256 consecutive units of one x plane (4 waves x 64 units):
//   k_face_count_walk : triangles per (tile, wave), by a boolean network over the sign words -> face offsets
//   k_faces           : the triangles themselves, lane = active cell, then lane = triangle
// Face order: chunk-major (chunk = one tile column over xw planes), then plane, tile, wave, cell.  The reference's
// order is whatever its atomicAdd produced (marching_cubes.cu:200), so any order is in spec.
// Corner bit weights and edge numbering follow marching_cubes.cu:49-57 and :178-192.
// ---------------------------------------------------------------------------------------------
struct FaceArgs {
    int xlate;  // rec[].x may be region * 2^26 + slot (straight from the streaming kernel) and is made dense on the fly:
                // 0 never; 1 yes, region bases from the call's cursors (the header is not finished yet);
                // 2 what hdr[H_RECFORM] says (a later p3d_mc_emit / part 6): region bases from hdr[H_PREFIX], or layout form;
                // 3 layout form (RegionLayout): rec[].x is a row; rows at or beyond V are remapped with the header's tail tables
    int halo_last;
    int64_t vid_base, halo_vid_base;
    const int64_t* rank_counts;  // optional: all-gathered V of all ranks on the device; the id bases are derived from it
    int rank, rank_stride;       // (rank r's count is rank_counts[r * rank_stride])
    int64_t tpp;           // tiles per plane
    int xw;                // planes per chunk (face_chunk_planes)
    int cpi;               // chunks per item (a single grid: all chunks)
    const u32* chunk_sum;  // [nchunks] triangles per chunk
    const u32* chunk_pre;  // optional [nchunks] exclusive prefix of chunk_sum (a stack of items has up to 4096 chunks:
                           // one small scan launch instead of a 16 KiB read per face tile); null: tiles add them up
    const u32* wave_off;   // [nb_f * 4] first face of (tile, wave) relative to its chunk
    const u32* tile_tris;  // [nb_f] triangles of the tile
    const u64* cursors;    // xlate: the call's 32 vertex-region cursors
    u64* mb;               // result mailbox slot (or null) and the call's sequence number (compaction block 0 reports)
    u64 seq;
    FastDiv div_tpp = {1, 0, 0}, div_xper = {1, 0, 0}, div_ncz = {1, 0, 0};   // (filled in by launch_faces)
    int xw_shift = 0;                                                           // log2(xw)
    const float* lay_verts = nullptr;   // layout form: the vertex buffer -- a moved row's first word holds the row it went to
    int sparse = 0;   // the caller expects few triangles per tile (launch_faces, from the face capacity): a block first asks
                      // for its tile's triangle count alone -- one scalar load -- and an empty tile leaves before any of the
                      // prologue's vector loads is issued
};

__device__ inline void wave_lds_sync() {  // orders one wave's LDS traffic across its lanes (no block barrier)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Compaction of the streaming kernel's 32 vertex regions, riding in the launches of the two face kernels: their
// first `nblocks` blocks copy while the other blocks count / emit faces (a bandwidth-bound job next to two
// ALU-bound ones, one stream, no events).  Block j serves region j % 32, slice j / 32.  16-byte vectors aligned on the
// destination; the source is read with 4-byte-aligned 16-byte loads.  Block 0 also finishes the header: V, F, the
// overflow flag, the region prefixes, the record form flag, and publishes V and F to the host.
struct CompactArgs {
    const float* scratch;  // null: nothing to copy (counting call), the header is still finished
    float* verts;
    int64_t capv;
    u32 store_rows, region_rows;
    int nblocks;           // multiple of kRegions (0: no compaction blocks in this launch)
    int part0, nparts;     // this launch copies slices part0 .. part0 + nblocks/32 - 1 of every region, out of nparts
    int finish;            // block 0 also finishes the header and reports V and F
    const u32* chunk_sum;  // block 0 adds the chunk sums up to F
    int nchunks;
    const u64* cursors;    // the call's cursor block (a stack of items: one block per item, back to back)
    u32 id_limit;          // vertices a region may number (2^26: ids are region * 2^26 + slot); beyond it ids alias
    int nitems;            // items of a stack (0 / 1: a single grid)
    int64_t* item_offsets; // (unused by the riding blocks; kept so that both call flavours build one argument struct)
    int layout = 0;        // region-layout mode (RegionLayout).  In the COUNTING launch the riding blocks are layout_move_block -- block 0
                           // makes V, the flags and the tail tables, every block moves one region's rows at or beyond V into the
                           // free rows below V --, in the FACE launch the one extra block is layout_report_block: F and the report
                           // to the host (V, flags, region totals)
};
typedef float F4U __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte access at 4-byte alignment
typedef float F4A __attribute__((ext_vector_type(4)));
// Region-layout mode (RegionLayout, fused_stream.inc).  40 intervals: the 32 regions and the 8 spill areas.  Interval i holds
// rows [first_i, first_i + occ_i): occ of a region = its fill mark if a wave-plane of it turned to the spill area, else
// min(total, its rows); occ of a spill area = min(its cursor, its rows).  V = the sum of the 32 region totals (= the sum
// of all occ when the spill area did not overflow).  Rows at or beyond V ("tail") move into the free rows below V ("holes"),
// k-th tail row -> k-th hole, both in ascending order -- one interval of each kind per layout interval:
//     tail of i      [ts_i, ts_i + tl_i) = [max(first_i, V), max(first_i + occ_i, V))        tp_i = tail rows of intervals < i
//     hole behind i  [hs_i, hs_i + hl_i) = [min(first_i + occ_i, V), min(first_(i+1), V))    hp_i = hole rows behind intervals < i
// Every riding block works the tables out for itself (lane i < 40 = interval i); block 0 writes them into the header for
// k_faces (and for tests/ws_keys.py).
struct TailTables {
    u32 V, ts, tl, tp, hs, hl, hp;   // lane i < 40: interval i's entries
};
__device__ inline TailTables tail_tables(const u64* __restrict__ cursors, const u64* __restrict__ hdr_layout, u32* over_out,
                                         u32* occ_out) {
    const int lane = threadIdx.x & 63;
    const bool mine = lane < kLayIv, region = lane < kRegions;
    const u32 first = mine ? (u32)hdr_layout[lane] : 0u, next = mine ? (u32)hdr_layout[lane + 1] : 0u;
    const u32 rows = next - first;
    const u64 cur = region ? cursors[lane * kCursorStride]
                           : (mine ? cursors[(size_t)(4 * (lane - kRegions)) * kCursorStride + kSpillWord] : 0ull);
    const u64 fill = region ? cursors[lane * kCursorStride + kFillWord] : 0ull;
    const u32 occ = fill ? (u32)(fill - 1ull) : (u32)min(cur, (u64)rows);
    if (occ_out) *occ_out = occ;
    if (over_out) *over_out = (__ballot(mine && !region && cur > (u64)rows) != 0ull) ? 1u : 0u;   // a spill area overflowed
    u64 v64 = region ? cur : 0ull;   // V counts every vertex, stored or not (the caller sizes its re-emission from it)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v64 += __shfl_xor(v64, o, 64);
    TailTables t;
    t.V = (u32)min(v64, (u64)0xffffffffull);
    const u32 end = first + occ;
    t.ts = max(first, t.V);
    t.tl = mine ? max(end, t.V) - t.ts : 0u;
    t.hs = min(end, t.V);
    t.hl = mine ? min(next, t.V) - t.hs : 0u;
    t.tp = wave_prefix_sum(t.tl) - t.tl;
    t.hp = wave_prefix_sum(t.hl) - t.hl;
    if (!mine) {   // (never selected: the searches stay below 40)
        t.ts = t.tp = t.hp = 0xffffffffu;
        t.hs = 0u;
    }
    return t;
}
// the LAST lane i < 40 with tab_i <= key (tab non-decreasing over the lanes 0..39, ~0 beyond): ds_bpermute, ALL lanes active
__device__ inline u32 last_not_above(u32 tab, u32 key) {
    u32 i = 0;
#pragma unroll
    for (int h = 32; h >= 1; h >>= 1) {
        const u32 v = (u32)__builtin_amdgcn_ds_bpermute((int)(min(i + (u32)h, 63u) << 2), (int)tab);
        i = (i + (u32)h < (u32)kLayIv && v <= key) ? i + (u32)h : i;
    }
    return i;
}
// k-th tail row's place below V
__device__ inline u32 hole_of(u32 k, const TailTables& t) {
    const u32 i = last_not_above(t.hp, k);   // (empty holes are stepped over by taking the LAST)
    const u32 hsi = (u32)__builtin_amdgcn_ds_bpermute((int)(i << 2), (int)t.hs);
    const u32 hpi = (u32)__builtin_amdgcn_ds_bpermute((int)(i << 2), (int)t.hp);
    return hsi + (k - hpi);
}
constexpr u32 kTailMark = 0x80000000u;   // a staged base whose unit's ids may be moved rows: translated id by id (k_faces)
// the face launch's one extra block in layout mode: F, and the report to the host (V and the flags were left in the header by
// the counting launch's block 0; the 32 region totals travel with them: the adapter lays the next call out from them)
__device__ inline void layout_report_block(const CompactArgs& c, u64* __restrict__ hdr, u64* mb, u64 seq, u64* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    if (blockIdx.x != 0) return;
    u64 part_sum = 0;
    for (int i = tid; i < c.nchunks; i += kBlock) part_sum += c.chunk_sum[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part_sum += __shfl_down(part_sum, o, 64);
    if (lane == 0) smem[tid >> 6] = part_sum;
    __syncthreads();
    if (tid < kRegions) mb_put_region(mb, tid, c.cursors[tid * kCursorStride]);
    if (tid == 0) {
        const u64 nf = smem[0] + smem[1] + smem[2] + smem[3];
        hdr[H_T] = nf;
        mb_publish_f(mb, seq, nf);
        mb_publish_v(mb, seq, hdr[H_V], hdr[H_FLAGS]);   // (a release store: the region totals of this wave's lanes go first)
    }
}
// the counting launch's riding blocks in layout mode: block 0 writes V, the flags and the tail tables; all of them together
// move the rows at or beyond V into the holes (thread <-> k-th tail row; few rows: the blocks are done long before the count)
__device__ inline void layout_move_block(const CompactArgs& c, u64* __restrict__ hdr) {
    const int tid = threadIdx.x, lane = tid & 63;
    u32 over = 0, occ = 0;
    const TailTables t = tail_tables(c.cursors, hdr + H_LAYOUT, &over, &occ);   // (every wave of every block: ~100 loads)
    if (blockIdx.x == 0 && tid < 64) {
        if (lane < kLayIv) {
            hdr[H_TAIL_TS + lane] = t.ts;
            hdr[H_TAIL_TP + lane] = t.tp;
            hdr[H_TAIL_HS + lane] = t.hs;
            hdr[H_TAIL_HP + lane] = t.hp;
            hdr[H_TAIL_TE + lane] = t.ts + t.tl;
            hdr[H_LAY_OCC + lane] = occ;
        }
        const u64 cur = lane < kRegions ? c.cursors[lane * kCursorStride] : 0ull;
        // (what a later p3d_mc_emit reads: the region totals and their prefix, as the scratch mode leaves them)
        if (lane < kRegions && c.cursors != hdr + H_CURSORS) hdr[H_CURSORS + lane * kCursorStride] = cur;
        u64 inc = cur;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u64 tt = __shfl_up(inc, o, 64);
            if (lane >= o) inc += tt;
        }
        if (lane < kRegions) hdr[H_PREFIX + lane] = inc - cur;
        if (lane == kLayIv - 1) hdr[H_TAIL_HP + kLayIv] = t.hp + t.hl;
        if (lane == kRegions - 1) {
            hdr[H_V] = inc;
            hdr[H_FLAGS] = (over ? 4ull : 0ull) | (inc > 0x7fffffffull ? 2ull : 0ull);
            hdr[H_RECFORM] = 2ull;
        }
    }
    // the k-th tail row (ascending) goes to the k-th free row below V
    const u32 ntail = (u32)__builtin_amdgcn_readlane((int)(t.tp + t.tl), kLayIv - 1);
    float* const v = c.verts;
    const u32 stride = (u32)c.nblocks * kBlock;
    for (u32 k0 = (u32)blockIdx.x * kBlock; k0 < ntail; k0 += stride) {   // (block-uniform trip count)
        const u32 k = min(k0 + (u32)tid, ntail - 1u);
        const u32 j = last_not_above(t.tp, k);   // (intervals without tail rows are stepped over by taking the LAST)
        const u32 tsj = (u32)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)t.ts);
        const u32 tpj = (u32)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)t.tp);
        const u32 src = tsj + (k - tpj);
        const u32 dst = hole_of(k, t);
        if (k0 + (u32)tid < ntail && (int64_t)src < c.capv && (int64_t)dst < c.capv) {
            const float x = v[(size_t)src * 3], y = v[(size_t)src * 3 + 1], z = v[(size_t)src * 3 + 2];
            v[(size_t)dst * 3] = x;
            v[(size_t)dst * 3 + 1] = y;
            v[(size_t)dst * 3 + 2] = z;
            v[(size_t)src * 3] = __uint_as_float(dst);   // where the row went: k_faces translates the ids of moved rows through it
        }
    }
}

// A stack of items (a batch of grids) has one cursor block and 32 scratch regions per item: the launch's compaction
// blocks are split evenly over the items (c.nblocks = nitems * slices * 32) and an item's vertices land behind those of
// the items before it; the totals and per-item offsets of a stack are written by k_stack_finish, not here.
constexpr int kCompactSmemWords = 2 * kRegions + 1 + 4;   // u64 words of LDS a compaction block needs (from its caller:
                                                          // the face kernel lends a corner of its id columns)
__device__ inline void compact_block(const CompactArgs& c, u64* __restrict__ hdr, u64* mb, u64 seq, u64* smem) {
    u64* const s_cur = smem;
    u64* const s_pre = smem + kRegions;
    u64& s_item_base = smem[2 * kRegions];
    u64* const s_red = smem + 2 * kRegions + 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int nitems = c.nitems > 0 ? c.nitems : 1;
    const int per_item = c.nblocks / nitems;                       // compaction blocks per item (multiple of 32)
    const int item = per_item > 0 ? (int)blockIdx.x / per_item : 0;
    const int jb = per_item > 0 ? (int)blockIdx.x - item * per_item : 0;
    const u64* const cursors = c.cursors + (size_t)item * kCursorBlockWords;
    const bool finisher = blockIdx.x == 0 && c.finish;
    if (tid < 64) {
        // vertices of the items before this one (a single grid: nothing to add)
        u64 before = 0;
        for (int i = lane; i < item * kRegions; i += 64) before += c.cursors[(size_t)i * kCursorStride];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) before += __shfl_down(before, o, 64);
        before = readlane64(before, 0);
        const u64 cur = lane < kRegions ? cursors[lane * kCursorStride] : 0ull;
        u64 inc = cur;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u64 tt = __shfl_up(inc, o, 64);
            if (lane >= o) inc += tt;
        }
        if (lane < kRegions) {
            s_cur[lane] = cur;
            s_pre[lane] = inc - cur;
        }
        if (lane == 0) s_item_base = before;
        if (finisher) {
            const u64 over = __ballot(cur > (u64)(c.scratch ? c.store_rows : c.region_rows));
            const u64 wrap = __ballot(cur > (u64)c.id_limit);   // a region outgrew its id space: ids are ambiguous
            const u64 flags = (over ? 1ull : 0ull) | (wrap ? 2ull : 0ull);
            if (lane < kRegions) hdr[H_PREFIX + lane] = inc - cur;
            // (the region counts stay with the workspace: a caller whose OUTPUT buffers turned out too small -- not the scratch --
            //  can have the faces and the compaction run again into larger ones, p3d_mc_slab.part = 6, without streaming the
            //  field a second time; the call's cursor block goes back to the stream's ring)
            if (lane < kRegions && c.nitems <= 1 && c.cursors != hdr + H_CURSORS) hdr[H_CURSORS + lane * kCursorStride] = cur;
            if (lane < kRegions && c.nitems <= 1) mb_put_region(mb, lane, cur);   // (the adapter lays the next call's regions out from them)
            if (lane == kRegions - 1) {
                hdr[H_V] = inc;
                hdr[H_FLAGS] = flags;
                hdr[H_RECFORM] = 1ull;
                mb_publish_v(mb, seq, inc, flags);
            }
        }
    }
    __syncthreads();
    if (finisher) {
        u64 part_sum = 0;
        for (int i = tid; i < c.nchunks; i += kBlock) part_sum += c.chunk_sum[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part_sum += __shfl_down(part_sum, o, 64);
        if (lane == 0) s_red[tid >> 6] = part_sum;
        __syncthreads();
        if (tid == 0) {
            const u64 nf = s_red[0] + s_red[1] + s_red[2] + s_red[3];
            hdr[H_T] = nf;
            mb_publish_f(mb, seq, nf);
        }
    }
    if (!c.scratch || c.capv <= 0) return;
    const int r = jb % kRegions, part = c.part0 + jb / kRegions, nparts = c.nparts;
    const int64_t rows = (int64_t)min(s_cur[r], (u64)c.store_rows);
    const int64_t dst0 = (int64_t)(s_item_base + s_pre[r]) * 3;
    const int64_t n = min(rows * 3, c.capv * 3 - dst0);  // floats to move (<= 0: nothing fits)
    if (n <= 0) return;
    const float* __restrict__ src = c.scratch + ((size_t)item * kRegions + r) * c.store_rows * 3;
    float* __restrict__ dst = c.verts + dst0;
    const int64_t head = min(n, (int64_t)((4 - (dst0 & 3)) & 3));  // floats before the first 16-byte boundary of dst
    const int64_t nvec = (n - head) >> 2;
    const int64_t tail0 = head + nvec * 4;
    if (part == 0 && tid < 8) {
        if (tid < head) dst[tid] = src[tid];
        if (tid >= 4 && tail0 + (tid - 4) < n) dst[tail0 + (tid - 4)] = src[tail0 + (tid - 4)];
    }
    const F4U* __restrict__ s4 = (const F4U*)(src + head);
    F4A* __restrict__ d4 = (F4A*)(dst + head);
    const int64_t stride = (int64_t)nparts * kBlock;
    int64_t i = (int64_t)part * kBlock + tid;
    for (; i + 3 * stride < nvec; i += 4 * stride) {  // four loads in flight per lane
        const F4U a0 = __builtin_nontemporal_load(s4 + i), a1 = __builtin_nontemporal_load(s4 + i + stride);
        const F4U a2 = __builtin_nontemporal_load(s4 + i + 2 * stride), a3 = __builtin_nontemporal_load(s4 + i + 3 * stride);
        // streaming stores: the copy is never read again by this call, and keeping its 63 MB out of the caches lets
        // the NEXT call's streaming kernel start on a clean L2 / Infinity Cache (measured: -7 us on k_fused in a
        // back-to-back call stream).  The same policy on the 12-byte face stores costs far more than it saves.
        __builtin_nontemporal_store(a0, d4 + i);
        __builtin_nontemporal_store(a1, d4 + i + stride);
        __builtin_nontemporal_store(a2, d4 + i + 2 * stride);
        __builtin_nontemporal_store(a3, d4 + i + 3 * stride);
    }
    for (; i < nvec; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(s4 + i), d4 + i);
}

// Exclusive prefix of the chunk totals by ONE block of 1024 threads (a thread takes a run of consecutive chunks): what a
// face tile reads instead of adding the totals in front of its chunk up when there are more than kPreMinChunks of them.
__device__ inline void block_chunk_prefix(const u32* __restrict__ chunk_sum, int nchunks, u32* __restrict__ chunk_pre) {
    __shared__ u32 s_wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (nchunks + 1023) / 1024;
    const int i0 = tid * per;
    u32 sum = 0;
    for (int k = 0; k < per; ++k)
        if (i0 + k < nchunks) sum += chunk_sum[i0 + k];
    const u32 inc = wave_prefix_sum(sum);
    if (lane == 63) s_wsum[wv] = inc;
    __syncthreads();
    u32 base = 0;
    for (int w = 0; w < wv; ++w) base += s_wsum[w];
    u32 run = base + inc - sum;
    for (int k = 0; k < per; ++k)
        if (i0 + k < nchunks) {
            chunk_pre[i0 + k] = run;
            run += chunk_sum[i0 + k];
        }
    __syncthreads();
}
__global__ void __launch_bounds__(1024) k_chunk_prefix(const u32* __restrict__ chunk_sum, int nchunks, u32* __restrict__ chunk_pre) {
    block_chunk_prefix(chunk_sum, nchunks, chunk_pre);
}

// Totals, flags and per-item offsets of a stack of items (one block of 1024 threads, launched after the face kernel of a
// batched call): 32 lanes per item add up its 32 cursors and its chunk totals, thread 0 then walks the items.
// item_offsets: [nitems + 1] vertex offsets, then [nitems + 1] face offsets.
__global__ void __launch_bounds__(1024) k_stack_finish(const u64* __restrict__ cursors, const u32* __restrict__ chunk_sum,
                                                       int nchunks, int nitems, u32 rows_limit, u32 id_limit,
                                                       int64_t* __restrict__ item_offsets, u64* __restrict__ hdr, u64* mb,
                                                       u64 seq, u32* __restrict__ chunk_pre) {
    __shared__ u64 s_nv[32], s_nf[32];
    __shared__ u32 s_fl[32];
    if (chunk_pre) block_chunk_prefix(chunk_sum, nchunks, chunk_pre);
    const int tid = threadIdx.x, sub = tid & 31, grp = tid >> 5;   // 32 groups of 32 lanes
    const int cpi = nchunks / nitems;
    u64 run_v = 0, run_f = 0;
    u32 run_fl = 0;
    for (int i0 = 0; i0 < nitems; i0 += 32) {
        const int i = i0 + grp;
        u64 nv = 0, nf = 0;
        u32 fl = 0;
        if (i < nitems) {
            const u64 cur = cursors[((size_t)i * kRegions + sub) * kCursorStride];
            nv = cur;
            fl = (cur > (u64)rows_limit ? 1u : 0u) | (cur > (u64)id_limit ? 2u : 0u);
            for (int k = sub; k < cpi; k += 32) nf += chunk_sum[(size_t)i * cpi + k];
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {   // (xor butterflies stay inside each half wave)
            nv += __shfl_xor(nv, o, 64);
            nf += __shfl_xor(nf, o, 64);
            fl |= (u32)__shfl_xor((int)fl, o, 64);
        }
        if (sub == 0) {
            s_nv[grp] = nv;
            s_nf[grp] = nf;
            s_fl[grp] = fl;
        }
        __syncthreads();
        if (tid == 0) {
            for (int k = 0; k < 32 && i0 + k < nitems; ++k) {
                item_offsets[i0 + k] = (int64_t)run_v;
                item_offsets[nitems + 1 + i0 + k] = (int64_t)run_f;
                run_v += s_nv[k];
                run_f += s_nf[k];
                run_fl |= s_fl[k];
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        item_offsets[nitems] = (int64_t)run_v;
        item_offsets[2 * nitems + 1] = (int64_t)run_f;
        hdr[H_V] = run_v;
        hdr[H_T] = run_f;
        hdr[H_FLAGS] = run_fl;
        hdr[H_RECFORM] = 1ull;
        mb_publish_v(mb, seq, run_v, run_fl);
        mb_publish_f(mb, seq, run_f);
    }
}

// Triangle counts.  A block owns a CHUNK = one tile column over `xw` consecutive planes and walks it along x: every
// sign word is loaded once per chunk (plane x+1 of one step is plane x of the next; all loads of a sub-batch of PB
// planes are issued together).  Lane = unit: the 8 corner signs of the unit's 64 cells are 8 words (the four column
// words and the same shifted by one voxel), and the triangle count of all 64 cells comes out of a boolean network
// over those words (tri_count_bitsliced.inc: ~200 v_bitop3 per unit) -- no table lookups, no cell list, no
// divergence.  Output: wave_off[(x * tpp + tile) * 4 + w] = triangles of the chunk before wave w of that tile,
// chunk_sum[chunk] = chunk total.  No scan follows: k_faces adds the chunk totals up itself.
// Replaces the atomicAdd(counters + 1, ...) of marching_cubes.cu:60-65.
#ifndef P3D_COUNT_PB
#define P3D_COUNT_PB 8   // planes per sub-batch of k_face_count_walk
#endif
#ifndef P3D_COUNT_STAMP   // dev-only: s_memtime stamps of every counting wave's phases -> a debug buffer (tools/dev/count_stamps.py)
#define P3D_COUNT_STAMP 0
#endif
#if P3D_COUNT_STAMP
__device__ u64* g_count_stamps = nullptr;   // [chunks * 4 waves][8]
#define CSTAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); cstamp[k] = __builtin_amdgcn_s_memtime(); } while (0)
#define CSTAMP_NOWAIT(k) do { cstamp[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CSTAMP(k) do { } while (0)
#define CSTAMP_NOWAIT(k) do { } while (0)
#endif
template <int PB>
__global__ void __launch_bounds__(kBlock) k_face_count_walk(const u64* __restrict__ bits, Dims d, int64_t tpp, int xw,
                                                            int cpi, u32* __restrict__ chunk_sum, u32* __restrict__ wave_off,
                                                            u32* __restrict__ tile_tris, CompactArgs cp, u64* __restrict__ hdr) {
    if ((int)blockIdx.x < cp.nblocks) {  // the launch's first blocks move vertices (uniform per block)
        __shared__ u64 s_cb[kCompactSmemWords];
        if (cp.layout) layout_move_block(cp, hdr);
        else compact_block(cp, hdr, nullptr, 0, s_cb);
        return;
    }
    __shared__ u32 s_part[PB][4];
#if P3D_COUNT_STAMP
    u64 cstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    CSTAMP_NOWAIT(0);
    const u64 crt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t chunk = (int64_t)blockIdx.x - cp.nblocks;
    // chunks are numbered item by item (a single grid is one item): cpi chunks each, none straddles two items
    // (32-bit arithmetic; the item split only for a stack)
    const u32 item = d.stack ? (u32)chunk / (u32)cpi : 0u;
    const u32 cl = (u32)chunk - item * (u32)cpi;
    const int64_t xc = cl / (u32)tpp, tile = cl - (u32)xc * (u32)tpp;
    const int64_t x_begin = item * d.xper + xc * xw;
    const int64_t x_end = min(x_begin + xw, item * d.xper + d.xper - 1);  // cell layers [x_begin, x_end)
    const int64_t p = tile * kBlock + tid;
    const int64_t y = (u32)p / (u32)d.ncz;
    const int c = (int)(p - y * d.ncz);
    const bool valid = (p < d.P) && (y + 1 < d.ry);
    const bool more = c + 1 < d.ncz;
    const u64 cells = valid ? zedge(d, c) : 0ull;
    // the wave's last unit and whether its row goes on behind it (wave-uniform)
    const int64_t p63 = tile * kBlock + (int64_t)__builtin_amdgcn_readfirstlane(wave) * 64 + 63;
    const bool need63 = __builtin_amdgcn_readlane((int)(valid && more), 63) != 0;
    u32 running = 0;  // first wave: triangles of the chunk before the current sub-batch
    for (int64_t xs = x_begin; xs < x_end; xs += PB) {
        u64 Wp[PB + 1], Wq[PB + 1];  // columns (x', y) and (x', y+1) of this unit for x' = xs .. xs+PB
        u32 first = 0, nbs = 0;
        // first bit of the next chunk of both columns, all planes in one word: the next lane has them (same row).  The
        // wave's last lane needs them of the unit behind the wave's 64: lane 2i (+1) loads the low dword of that unit's
        // column in plane xs + i (its row-above column) -- ONE vector load in front of the columns' own, travelling with
        // them, and a ballot over its bit 0 is the word (as loads of the last lane behind the shuffle below they were a
        // second memory round trip per wave)
        u32 nxv;
        {
            const bool mine = need63 && lane < 2 * (PB + 1);
            const int64_t un = mine ? min(xs + (lane >> 1), d.rx - 1) * d.P + p63 + 1 + ((lane & 1) ? d.ncz : 0) : 0;
            nxv = ((const u32*)bits)[2 * un];   // (always a valid address: no branch)
            nxv = mine ? nxv & 1u : 0u;
        }
        static_for<0, PB + 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int64_t u = min(xs + i, d.rx - 1) * d.P + p;
            Wp[i] = valid ? bits[u] : 0ull;
            Wq[i] = valid ? bits[u + d.ncz] : 0ull;
        });
        CSTAMP_NOWAIT(1);   // all loads of the sub-batch issued (the compiler is free to move this stamp: read 1 + 2 together)
        CSTAMP(2);          // ... and returned
        static_for<0, PB + 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            first |= ((u32)(Wp[i] & 1ull) | ((u32)(Wq[i] & 1ull) << 1)) << (2 * i);
        });
        nbs = (u32)__shfl_down((int)first, 1, 64);
        {
            const u32 n63 = (u32)__ballot(nxv != 0u);   // bit 2i / 2i+1 = lane 2i / 2i+1: the layout of `first`
            if (lane == 63) nbs = n63;
        }
        if (!more) nbs = 0;
        // A wave all of whose units are entirely outside (every word 0) or entirely inside (every word all ones)
        // over the whole sub-batch has no cell with a sign change: it skips the networks (most waves of a sparse
        // field -- an object's SDF in a box).  One OR / AND over the words already loaded; conservative on tail chunks.
        u64 acc_or = 0, acc_and = ~0ull;
        static_for<0, PB + 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            acc_or |= Wp[i] | Wq[i];
            acc_and &= Wp[i] & Wq[i];
        });
        constexpr u32 kAllNext = (1u << (2 * (PB + 1))) - 1u;
        const bool flat = !valid || (acc_or == 0ull && nbs == 0u) || (acc_and == ~0ull && (!more || nbs == kAllNext));
        const bool wave_active = __ballot(!flat) != 0ull;
        static_for<0, PB>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            u32 n = 0;
            if (xs + i < x_end && wave_active) {  // uniform
                const u64 W0 = Wp[i], W1 = Wp[i + 1], W2 = Wq[i + 1], W3 = Wq[i];
                const u64 S0 = (W0 >> 1) | ((u64)((nbs >> (2 * i)) & 1u) << 63);
                const u64 S3 = (W3 >> 1) | ((u64)((nbs >> (2 * i + 1)) & 1u) << 63);
                const u64 S1 = (W1 >> 1) | ((u64)((nbs >> (2 * i + 2)) & 1u) << 63);
                const u64 S2 = (W2 >> 1) | ((u64)((nbs >> (2 * i + 3)) & 1u) << 63);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32 o0, o1, o2;
                    tri_count_bitsliced((u32)(W0 >> (32 * h)), (u32)(W1 >> (32 * h)), (u32)(W2 >> (32 * h)),
                                        (u32)(W3 >> (32 * h)), (u32)(S0 >> (32 * h)), (u32)(S1 >> (32 * h)),
                                        (u32)(S2 >> (32 * h)), (u32)(S3 >> (32 * h)), o0, o1, o2);
                    const u32 m = (u32)(cells >> (32 * h));
                    n += (u32)__popc(o0 & m) + 2u * (u32)__popc(o1 & m) + 4u * (u32)__popc(o2 & m);
                }
            }
            n = (u32)__builtin_amdgcn_readlane((int)wave_prefix_sum(n), 63);
            if (lane == 0) s_part[i][wave] = n;
        });
        CSTAMP(3);          // networks + per-plane wave sums done
        __syncthreads();
        CSTAMP_NOWAIT(4);   // barrier passed
        // offsets of the sub-batch in one scan by the first wave: lane l = (plane i = l / 4, wave w = l % 4), so the
        // exclusive prefix over the lanes is "triangles of the chunk before wave w of plane i" (every lane keeps `running`)
        if (tid < 64) {
            const int i = lane >> 2, w = lane & 3;
            const bool live = i < PB && xs + i < x_end;
            const u32 val = live ? s_part[i < PB ? i : 0][w] : 0u;
            const u32 inc = wave_prefix_sum(val);
            if (live) {
                wave_off[((xs + i) * tpp + tile) * 4 + w] = running + inc - val;
                // (lane i*4+3 holds the inclusive prefix at the end of plane i; the plane's total = that minus the
                //  exclusive prefix at its first lane)
                const u32 at_end = (u32)__builtin_amdgcn_ds_bpermute((lane | 3) << 2, (int)inc);
                const u32 at_begin = (u32)__builtin_amdgcn_ds_bpermute((lane & ~3) << 2, (int)(inc - val));
                if (w == 0) tile_tris[(xs + i) * tpp + tile] = at_end - at_begin;
            }
            running += (u32)__builtin_amdgcn_readlane((int)inc, 63);
        }
        __syncthreads();
    }
    // (With many chunks -- a stack of items, a very large grid -- a face tile reads the exclusive prefix of these totals
    //  instead of adding them up: k_chunk_prefix / k_stack_finish make it, one small block between the two launches.  The
    //  last counting block used to do that inside this launch; its hand-off -- an agent-scope store, a drained wait and a
    //  returning atomic per block -- cost every block about a microsecond: 110 -> 104 us on the 32 x 256^3 stack.)
    if (tid == 0) chunk_sum[chunk] = running;
#if P3D_COUNT_STAMP
    CSTAMP_NOWAIT(5);
    if (lane == 0 && g_count_stamps) {
        u64* o = g_count_stamps + ((size_t)chunk * 4 + wave) * 8;
        for (int k = 0; k < 6; ++k) o[k] = cstamp[k];
        o[6] = crt0;
        o[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// F = sum of the chunk totals -> header + host mailbox (the counting call and the slab path; the one-pass call lets
// block 0 of the k_faces launch do it)
__global__ void __launch_bounds__(kBlock) k_face_total(const u32* __restrict__ chunk_sum, int nchunks,
                                                       u64* __restrict__ hdr, u64* mb, u64 seq, int also_v) {
    __shared__ u64 s_red[4];
    const int tid = threadIdx.x;
    u64 part_sum = 0;
    for (int i = tid; i < nchunks; i += kBlock) part_sum += chunk_sum[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part_sum += __shfl_down(part_sum, o, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = part_sum;
    __syncthreads();
    if (tid == 0) {
        const u64 nf = s_red[0] + s_red[1] + s_red[2] + s_red[3];
        hdr[H_T] = nf;
        mb_publish_f(mb, seq, nf);
        if (also_v) mb_publish_v(mb, seq, hdr[H_V], hdr[H_FLAGS]);  // (k_early_header left them in the header)
    }
}

// V, the overflow flag and the region prefixes as soon as the streaming kernel is done (one wave).  The multi-GPU
// path wants them before the face count: the all-gather of V and the export of the first plane's records can then
// travel while the counting and compaction kernels run.  (The finishing block of the k_faces launch writes the
// same values again and reports to the host.)
// With `out` the same launch also exports one plane's vertex-id records in dense form (p3d_mc_slab.export_first_plane_to:
// what p3d_mc_export_plane_records does as a launch of its own; a one-wave kernel behind another one-wave kernel costs the
// stream 4-5 us each): every block works the region prefixes out of the cursors for itself, block 0 writes the header.
__global__ void __launch_bounds__(kBlock) k_early_header(u64* __restrict__ hdr, const u64* __restrict__ cursors, u32 rows_per_region,
                                                         u32 id_limit, const uint2* __restrict__ rec, int64_t first, int64_t n,
                                                         uint2* __restrict__ out) {
    __shared__ u32 s_pre[kRegions];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        const u64 cur = lane < kRegions ? cursors[lane * kCursorStride] : 0ull;
        u64 inc = cur;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u64 tt = __shfl_up(inc, o, 64);
            if (lane >= o) inc += tt;
        }
        if (lane < kRegions) s_pre[lane] = (u32)(inc - cur);
        if (blockIdx.x == 0) {
            const u64 over = __ballot(cur > (u64)rows_per_region);
            const u64 wrap = __ballot(cur > (u64)id_limit);
            if (lane < kRegions) hdr[H_PREFIX + lane] = inc - cur;
            // (the cursors come from the stream's ring of pre-cleared blocks: the later parts of the extraction find the region
            //  counts in the workspace header)
            if (lane < kRegions && cursors != hdr + H_CURSORS) hdr[H_CURSORS + lane * kCursorStride] = cur;
            if (lane == kRegions - 1) {
                hdr[H_V] = inc;
                hdr[H_FLAGS] = (over ? 1ull : 0ull) | (wrap ? 2ull : 0ull);
                hdr[H_RECFORM] = 1ull;
            }
        }
    }
    if (!out) return;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint2 r = rec[first + i];   // (fresh from the streaming kernel: region form)
    const u32 reg = r.x >> 26;
    if (reg < (u32)kRegions) r.x = (r.x & 0x3ffffffu) + s_pre[reg];
    out[i] = r;
}

__global__ void __launch_bounds__(kBlock) k_zero_words(u64* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) p[i] = 0ull;
}

// Dense copy of one plane's vertex-id records (the multi-GPU path ships the first plane to the previous rank, whose
// halo plane it is).  After a one-pass call rec[] still holds region-form ids: translated here with the region
// prefixes of the header.
__global__ void __launch_bounds__(kBlock) k_export_plane_records(const uint2* __restrict__ rec, int64_t first,
                                                                 int64_t n, const u64* __restrict__ hdr,
                                                                 uint2* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint2 r = rec[first + i];
    if (hdr[H_RECFORM] != 0ull) {
        const u32 reg = r.x >> 26;
        if (reg < (u32)kRegions) r.x = (r.x & 0x3ffffffu) + (u32)hdr[H_PREFIX + reg];
    }
    out[i] = r;
}

// ---------------------------------------------------------------------------------------------
// k_faces -- faces from sign words + vertex-id records (second version, round 3; the first is in the history up to
// commit 6013781 and profiles/r03/ablation.txt compares them).  One block = one tile; after ONE block barrier (the
// staging) its four waves never meet again: wave w owns the cells of units 64w .. 64w+63 and knows where its faces start
// (wave_off from k_face_count_walk + the chunk totals before its chunk).
//   phase A (lane = staged unit, block): sign words of planes x and x+1 -> LDS as a DWORD stream; the vertex-id records
//            -> per-HALF-unit form {first id of the unit, byte offsets of the first x / y / z edge id of the low half, the
//            same for the high half}: the high half's offsets include the popcounts of the low halves of the unit's
//            crossing words (computed here, once per staged unit, from words loaded in the same round trip).  Region-form
//            ids (region * 2^26 + slot, straight from the streaming kernel) are made dense on the fly from the 32 region
//            cursors, so no pass over the records precedes the faces.
//   phase B (lane = unit, wave)  : dense list of the wave's active cells (2 .. 16 rounds over z slices if it does not fit)
//   phase C (lane = cell, wave)  : everything per cell is 32-bit.  Per column the cell reads the two dwords starting at
//            the half unit that holds z (index 2t + z / 32): bit z and bit z+1 are always inside that window, also at
//            z = 31 and z = 63 (the next dword is the next half / the next chunk of the same row) -- no 64-bit shifts, no
//            side table for the first bit of the next chunk, no divergent branch.  An id is
//            popcount(crossing32 & below(z % 32)) + base + offset (and, bcnt, byte-add); the four edges at z+1 are the
//            ids at z plus "edge at z crosses" (bits of the corner mask), or the next unit's first ids when z = 63.  Every
//            listed cell is active, so the ids are computed unconditionally and all LDS reads of a batch are issued
//            before the first wait.  The 12 ids go to the lane's column of the wave's LDS slice and come back by table
//            index (a register array cannot be indexed per lane); the k-th triangles of all cells are written together,
//            the lanes that have one writing a dense run (one 12-byte streaming store per lane).  The case table row
//            (global memory, 2 KiB, hot in the vector L1: keeps the block's LDS at six blocks per CU) carries its
//            triangle count in the top nibble.
// No block barrier and no dependent global load sits between a face store and the next batch.
// NHALO: units staged beyond the tile's 256 for the y+1 columns -- 32 (rows of at most 32 chunks, rz <= 2048: 26.5 KiB of
// LDS, six tiles per CU) or 256 (any row length: fewer tiles per CU).
// Where the time goes (512^3 Perlin, profiles/r03/ablation.txt): prologue 20 us, cell lists 4, per-cell header + round
// control + stores 27, id arithmetic 11, id read-back by table index 10.
// ---------------------------------------------------------------------------------------------
__device__ inline u32 add_byte0(u32 a, u32 packed) { return a + (packed & 0xffu); }
__device__ inline u32 add_byte1(u32 a, u32 packed) { return a + ((packed >> 8) & 0xffu); }
__device__ inline u32 add_byte2(u32 a, u32 packed) { return a + ((packed >> 16) & 0xffu); }
__device__ inline u32 rank32(u32 crossing, u32 lowm, u32 base) {   // base + crossings below z in this half
    return (u32)__builtin_popcount(crossing & lowm) + base;
}

#ifndef P3D_FACES_ABL   // dev-only timing ablation (wrong results): 1 = return after the staging barrier
#define P3D_FACES_ABL 0
#endif
#ifndef P3D_FACES_STAMP   // dev-only: s_memtime stamps of every face wave's phases -> a debug buffer (tools/dev/faces_stamps.py)
#define P3D_FACES_STAMP 0
#endif
#if P3D_FACES_STAMP
__device__ u64* g_face_stamps = nullptr;   // [blocks * 4 waves][16]
#define FSTAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stamp[k] = __builtin_amdgcn_s_memtime(); } while (0)
#define FSTAMP_NOWAIT(k) do { stamp[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FSTAMP(k) do { } while (0)
#define FSTAMP_NOWAIT(k) do { } while (0)
#endif
template <int N>
__device__ inline u32 row_nibble(u32 lo, u32 hi) {   // nibble N of the 64-bit table row (lo, hi)
    if constexpr (N < 8) return (lo >> (4 * N)) & 15u;
    else return (hi >> (4 * (N - 8))) & 15u;
}

template <int NHALO, bool LAYOUT>   // LAYOUT: rec[].x is a row of the region layout (FaceArgs::xlate == 3) -- its own instantiation,
                                    // so that the tables it carries do not cost the other forms a wave per SIMD
__global__ void __launch_bounds__(kBlock) k_faces(const u64* __restrict__ bits, const uint2* __restrict__ rec, Dims d,
                                                   FaceArgs a, CompactArgs cp, u64* __restrict__ hdr,
                                                   int32_t* __restrict__ faces, int64_t cap_faces) {
    constexpr int NS = (kBlock + NHALO + 1) > 2 * (kBlock + 1) ? (kBlock + NHALO + 1) : (NHALO >= kBlock ? 2 * (kBlock + 1) : kBlock + NHALO + 1);
    // LDS: 22.2 KiB with NHALO = 32 -> SEVEN tiles per CU (23 405 B is the limit; the sixth came with the case table moving
    // to global memory, the seventh with the three savings noted below)
    __shared__ u64 s_w[2][NS + 1];                       // sign words of planes x and x+1 (read as a dword stream; one pad)
    __shared__ u32 s_e0[3 * NS + 3];                     // plane x, per staged unit: {first id, offsets of the low half, of the high half}
    __shared__ u32 s_e1[2 * NS + 2];                     // plane x+1: {first id, y / z offsets of both halves in four bytes} (its x
                                                         // edges belong to the next cell layer: 2 dwords instead of 3)
    __shared__ __attribute__((aligned(8))) u32 s_ids[4][12][64];   // per wave: vertex ids of the batch's cells' 12 edges; row 0 doubles as
                                                         // the batch's unit markers (read before the ids are written), and the
                                                         // launch's compaction blocks borrow a corner as their scratch
    __shared__ u32 s_tmp[4];
    __shared__ u32 s_strad[4];                           // layout form: a wave staged a unit whose ids are translated one by one
    if ((int)blockIdx.x < cp.nblocks) {  // the launch's first blocks move the vertices (uniform per block)
        if (cp.layout) layout_report_block(cp, hdr, a.mb, a.seq, (u64*)&s_ids[0][0][0]);
        else compact_block(cp, hdr, a.mb, a.seq, (u64*)&s_ids[0][0][0]);
        return;
    }
#if P3D_FACES_STAMP
    u64 stamp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    FSTAMP_NOWAIT(0);
    const u64 rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // The kernel arguments the prologue's address arithmetic needs are touched here, together: the compiler otherwise
    // fetches them piecemeal, basic block by basic block, and every piece is a scalar-memory wait of its own (eight in a
    // row in front of the first vector load).
    {
        auto pin32 = [](u32 v) { asm volatile("" ::"s"(v)); };
        auto pin64 = [](u64 v) { asm volatile("" ::"s"(v)); };
        pin64((u64)d.P); pin64((u64)d.U); pin32((u32)d.ncz); pin64((u64)d.ry); pin64((u64)d.xper); pin32((u32)d.stack);
        pin64((u64)a.tpp); pin32((u32)a.cpi); pin32((u32)a.xw_shift); pin32((u32)a.xlate); pin32((u32)a.halo_last);
        pin32(a.div_tpp.m); pin32(a.div_xper.m); pin32(a.div_ncz.m);
        pin64((u64)a.chunk_sum); pin64((u64)a.chunk_pre); pin64((u64)a.wave_off); pin64((u64)a.tile_tris);
        pin64((u64)a.cursors); pin64((u64)a.rank_counts); pin64((u64)bits); pin64((u64)rec);
    }
    // (32-bit index arithmetic by multiply-high: three divisions by run-time constants)
    const u32 b = (u32)blockIdx.x - (u32)cp.nblocks;
    // sparse fields (an object's SDF in a box: nearly every tile is empty): the tile's count by a scalar load, in front of
    // everything -- an empty tile's block lives for one scalar round trip instead of issuing its 17 vector loads per lane and
    // waiting for the first (sphere in 512^3: 21.6 -> 15.2 us).  On a dense field it costs a round trip per block in front
    // of the prologue (512^3 noise: 67.6 -> 68.1 us over three runs each), so only launches that expect few triangles per
    // tile take it.
    if (a.sparse && a.tile_tris[b] == 0u) return;
    const bool XLATE = !LAYOUT && (a.xlate == 1 || (a.xlate == 2 && hdr[H_RECFORM] == 1ull));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 x32 = fd_div(b, a.div_tpp);
    const int64_t x = x32;
    const int64_t tile = b - x32 * (u32)a.tpp;
    const u32 item = d.stack ? fd_div(x32, a.div_xper) : 0u;
    const u32 xl = x32 - item * (u32)d.xper;
    if (xl == (u32)d.xper - 1u) return;   // the last plane of an item has no cell layer above it (stack of items only)
    // ---- prologue: every global load of the block is issued before the first wait ----
    // One "misc" dword per lane gathers the small per-block inputs in ONE vector load (uniform addresses would become
    // scalar loads, and every s_waitcnt lgkmcnt(0) behind one of those is a serial memory round trip -- the compiler had
    // strung five of them in front of the staging loads: 2.9 us of a wave's 9.6, tools/dev/faces_stamps.py):
    //   lanes 0..31  the 32 vertex-region cursors of the item (their exclusive prefix makes region-form ids dense)
    //   lane 32      the tile's triangle count          lanes 33..36  the first face of the tile's four waves in its chunk
    //   lane 37      the chunk's exclusive prefix (when the counting launch left one)
    const u32 mychunk32 = item * (u32)a.cpi + (xl >> a.xw_shift) * (u32)a.tpp + (u32)tile;
    // (multi-GPU slabs: the id bases come from the all-gathered vertex counts -- lane r loads rank r's count, ONE vector
    //  load that travels with the prologue's batch below; a scalar loop over the ranks was rank + 1 serial round trips in
    //  front of every tile's loads: the higher a rank, the slower its face launch)
    u32 b0 = (u32)a.vid_base, bhalo = (u32)a.halo_vid_base;
    const u32* rcp = (const u32*)a.tile_tris;   // (idle lanes: any valid address)
    if (a.rank_counts && (int)(threadIdx.x & 63) <= a.rank) rcp = (const u32*)(a.rank_counts + (size_t)(threadIdx.x & 63) * a.rank_stride);
    // (every lane has a valid address and the loads are unconditional: one basic block, so that the compiler lets all the
    //  prologue's loads leave before it waits for the first)
    const u32* mp = a.tile_tris + b;
    {
        const u32* const curp = a.xlate == 2 ? (const u32*)(hdr + H_PREFIX + (lane & 31))
                                             : (const u32*)(a.cursors + (size_t)item * kCursorBlockWords + (lane & 31) * kCursorStride);
        mp = (lane < kRegions && XLATE) ? curp : mp;
        mp = (lane > 32 && lane < 37) ? a.wave_off + (size_t)b * 4 + (lane - 33) : mp;
        mp = (lane == 37 && a.chunk_pre) ? a.chunk_pre + mychunk32 : mp;
        mp = (lane == 38 && LAYOUT) ? (const u32*)(hdr + H_V) : mp;   // (layout form: V, left there by the counting launch)
        mp = (lane == 39 && LAYOUT) ? (const u32*)(hdr + H_LAYOUT + kLayIv) : mp;   // (... and the end of the layout's rows)
    }
    const int64_t p = tile * kBlock + tid;
    const int64_t y = fd_div((u32)p, a.div_ncz);   // (p < 2^31: check_dims)
    const int c = (int)(p - y * d.ncz);
    const bool valid = (p < d.P) && (y + 1 < d.ry);
    const bool more = c + 1 < d.ncz;

    // phase A: staging.  All global loads of the prologue are issued before the first wait: the tile's triangle count,
    // the cursors, per staged unit the sign words of both planes, of both planes one row up (for the y crossings of the
    // low half) and both records, this thread's share of the chunk totals.
    const bool one_range = d.ncz <= NHALO;
    const int hoff = one_range ? d.ncz : kBlock + 1;
    const int nstage = one_range ? kBlock + d.ncz + 1 : 2 * (kBlock + 1);
    const u32 p0 = (u32)tile * kBlock;
    const bool xhalo = a.halo_last && (x + 1 == d.rx - 1);
    constexpr int NST = (NS + kBlock - 1) / kBlock;
    const int wave0 = wave * 64;
    // plane bases are block-uniform (scalar registers); a lane adds a 32-bit unit offset inside the plane (check_dims: a
    // plane has fewer than 2^29 units) -- no 64-bit vector arithmetic on the 17 addresses of the prologue
    const u64* const bw0 = bits + x * d.P;
    const u64* const bw1 = bw0 + d.P;          // (plane x + 1 exists: the last cell layer is x = rx - 2)
    const uint2* const br0 = rec + x * d.P;
    const uint2* const br1 = br0 + d.P;
    u32 so[NST], sy[NST];        // offsets of the staged unit and of the unit one row up (or the unit itself when there is none)
    bool st_in[NST], st_up[NST];
#pragma unroll
    for (int q = 0; q < NST; ++q) {
        // (entries that do not exist -- behind the staged range, behind the plane, a row above the plane's last -- read a
        //  valid address and are masked afterwards: conditional loads made the compiler wait between them; all idle lanes
        //  read the tile's first unit: one line)
        const u32 i = (u32)tid + (u32)q * kBlock;
        const u32 pi = (one_range || i <= (u32)kBlock) ? p0 + i : p0 + (u32)d.ncz + (i - (u32)kBlock - 1u);
        st_in[q] = i < (u32)nstage && pi < (u32)d.P;
        st_up[q] = st_in[q] && pi + (u32)d.ncz < (u32)d.P;   // the row above exists (same plane)
        so[q] = st_in[q] ? pi : p0;
        sy[q] = st_up[q] ? so[q] + (u32)d.ncz : so[q];
    }
    // this thread's share of the chunk totals in front of its chunk (at most kPreMinChunks = 1024 chunks without a prefix:
    // four independent loads; more only under the P3D_NO_CHUNK_PRE test switch, added up behind the staging)
    const u32* cp4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cp4[k] = a.chunk_sum + min((u32)tid + (u32)k * kBlock, mychunk32);
    // ---- the loads: one batch ----
    // (relaxed workgroup-scope ATOMIC loads: the same plain instructions, but the compiler may not sink them below the
    //  empty-tile exit that follows -- ordinary loads it moved behind that branch, i.e. behind the wait for the first one)
    auto ld32 = [](const void* p) -> u32 { return __hip_atomic_load((const u32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto ld64 = [](const void* p) -> u64 { return __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    const u32 misc = ld32(mp);
    const u32 rcv = ld32(rcp);   // (the low dword of a rank's count: the totals stay below 2^31, SlabExtractor checks)
    u32 part[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) part[k] = ld32(cp4[k]);
    u64 lw[4 * NST];
    uint2 lr[2 * NST];
#pragma unroll
    for (int q = 0; q < NST; ++q) {
        // (32-bit BYTE offsets -- a plane has fewer than 2^29 units -- so that the loads take the scalar base + 32-bit
        //  vector offset form instead of a 64-bit vector address each)
        const u32 bo = so[q] * 8u, by = sy[q] * 8u;
        lw[4 * q + 0] = ld64((const char*)bw0 + bo);
        lw[4 * q + 1] = ld64((const char*)bw1 + bo);
        lw[4 * q + 2] = ld64((const char*)bw0 + by);
        lw[4 * q + 3] = ld64((const char*)bw1 + by);
        const u64 r0 = ld64((const char*)br0 + bo), r1 = ld64((const char*)br1 + bo);
        lr[2 * q + 0] = make_uint2((u32)r0, (u32)(r0 >> 32));
        lr[2 * q + 1] = make_uint2((u32)r1, (u32)(r1 >> 32));
    }
    FSTAMP_NOWAIT(1);   // all prologue loads issued
    // a tile without triangles leaves here, in front of the staging (sparse fields -- an object's SDF in a box -- are mostly
    // such tiles)
    const u32 my_tris = (u32)__builtin_amdgcn_readlane((int)misc, 32);
    if (my_tris == 0u) return;   // (block-uniform)
    FSTAMP(2);          // ... and returned
    u64 st_w0[NST], st_w1[NST], st_y0[NST], st_y1[NST];
    uint2 st_r0[NST], st_r1[NST];
#pragma unroll
    for (int q = 0; q < NST; ++q) {
        st_w0[q] = st_in[q] ? lw[4 * q + 0] : 0ull;
        st_w1[q] = st_in[q] ? lw[4 * q + 1] : 0ull;
        st_y0[q] = st_up[q] ? lw[4 * q + 2] : 0ull;
        st_y1[q] = st_up[q] ? lw[4 * q + 3] : 0ull;
        st_r0[q] = st_in[q] ? lr[2 * q + 0] : make_uint2(0u, 0u);
        st_r1[q] = st_in[q] ? lr[2 * q + 1] : make_uint2(0u, 0u);
    }
    if (a.rank_counts) {   // (wave-uniform; every wave computes the two bases for itself)
        const u32 inc = wave_prefix_sum(lane <= a.rank ? rcv : 0u);
        bhalo = (u32)__builtin_amdgcn_readlane((int)inc, 63);
        b0 = bhalo - (u32)__builtin_amdgcn_readlane((int)rcv, a.rank & 63);
    }
    u32 cs = 0;
    if (!a.chunk_pre) {
#pragma unroll
        for (int k = 0; k < 4; ++k) cs += ((u32)tid + (u32)k * kBlock < mychunk32) ? part[k] : 0u;
        if (mychunk32 > 4u * kBlock)
            for (u32 i = (u32)tid + 4u * kBlock; i < mychunk32; i += kBlock) cs += a.chunk_sum[i];
    }
    u32 pref = 0;
    if (XLATE) {
        if (a.xlate == 1) {
            const u32 cnt = lane < kRegions ? misc : 0u;
            pref = wave_prefix_sum(cnt) - cnt;
        } else if (lane < kRegions) {
            pref = misc;
        }
    }
    auto dense = [&](u32 v) -> u32 {
        return XLATE ? (v & 0x3ffffffu) + (u32)__builtin_amdgcn_ds_bpermute((int)(((v >> 26) & (kRegions - 1)) << 2), (int)pref) : v;
    };
    // Layout form: a staged unit's first id is a ROW, final unless the unit's run of at most 192 ids can reach V: such a unit
    // is MARKED (bit 31 of its base) and its ids are translated one by one in the cell batches -- a row at or beyond V was
    // moved by the counting launch's riding blocks, which left the row it went to in the old row's first word.  (Records of
    // units without vertices hold whatever the memory held: only a row inside the layout can be a moved one.)
    const u32 layV = LAYOUT ? (u32)__builtin_amdgcn_readlane((int)misc, 38) : 0u;
    const u32 layEnd = LAYOUT ? (u32)__builtin_amdgcn_readlane((int)misc, 39) : 0u;
    u32 strad = 0;
    if (a.chunk_pre) cs = lane == 37 && wave == 0 ? misc : 0u;
    u32* const E0 = s_e0;
    u32* const E1 = s_e1;
#pragma unroll
    for (int q = 0; q < NST; ++q) {   // (wave-uniform trip count: the translation shuffles across the wave's lanes)
        const int i = tid + q * kBlock;
        if (wave0 + q * kBlock >= nstage) continue;
        u32 base0 = dense(st_r0[q].x) + b0;
        u32 base1 = xhalo ? st_r1[q].x + bhalo : dense(st_r1[q].x) + b0;
        if (LAYOUT) {
            const bool m0 = st_in[q] && st_r0[q].x + 191u >= layV && st_r0[q].x < layEnd;
            const bool m1 = st_in[q] && st_r1[q].x + 191u >= layV && st_r1[q].x < layEnd && !xhalo;
            base0 = m0 ? (base0 | kTailMark) : base0;
            base1 = m1 ? (base1 | kTailMark) : base1;
            strad |= (m0 || m1) ? 1u : 0u;
        }
        // offsets of the unit's first x / y / z edge id (bytes 0 / 1 / 2): low half = the record's, high half = plus
        // the crossings of the low half (x crossings of plane x+1 belong to the next cell layer: not needed)
        const u64 w0 = st_w0[q], w1 = st_w1[q];
        const u32 oY0 = st_r0[q].y & 0xffffu, oZ0 = st_r0[q].y >> 16, oY1 = st_r1[q].y & 0xffffu, oZ1 = st_r1[q].y >> 16;
        const u32 pX = (u32)__builtin_popcount((u32)(w0 ^ w1));
        const u32 pY0 = (u32)__builtin_popcount((u32)(w0 ^ st_y0[q])), pY1 = (u32)__builtin_popcount((u32)(w1 ^ st_y1[q]));
        const u32 pZ0 = (u32)__builtin_popcount((u32)(w0 ^ (w0 >> 1))), pZ1 = (u32)__builtin_popcount((u32)(w1 ^ (w1 >> 1)));
        if (i < nstage) {
            s_w[0][i] = w0;
            s_w[1][i] = w1;
            E0[3 * i] = base0;
            E0[3 * i + 1] = (oY0 << 8) | (oZ0 << 16);
            E0[3 * i + 2] = pX | ((oY0 + pY0) << 8) | ((oZ0 + pZ0) << 16);
            E1[2 * i] = base1;
            E1[2 * i + 1] = oY1 | (oZ1 << 8) | ((oY1 + pY1) << 16) | ((oZ1 + pZ1) << 24);   // (oY <= 96, oZ <= 160)
        }
    }
    if (tid == 0) {   // the pad dword behind the last staged unit (read by a window at z >= 32 of that unit, never used)
        s_w[0][nstage] = 0ull;
        s_w[1][nstage] = 0ull;
    }
    {
        cs = (u32)__builtin_amdgcn_readlane((int)wave_prefix_sum(cs), 63);
        if (lane == 0) s_tmp[wave] = cs;
        const u32 wave_strad = __ballot(strad != 0u) != 0ull ? 1u : 0u;
        if (LAYOUT && lane == 0) s_strad[wave] = wave_strad;
    }
    FSTAMP(3);          // staged (LDS writes done)
    __syncthreads();  // the only block barrier: staging done (no store is in flight yet)
    FSTAMP_NOWAIT(4);   // barrier passed
#if P3D_FACES_ABL == 1
    return;
#endif
    // layout form: does any cell of this tile read a marked unit?  (block-uniform; the tables again -- few tiles get here)
    const bool per_id = LAYOUT && __builtin_amdgcn_readfirstlane((int)(s_strad[0] | s_strad[1] | s_strad[2] | s_strad[3])) != 0;
    // first face of this wave and the capacity, relative to it (a wave-tile emits at most 64 * 64 * 5 faces)
    const u32 wrun0 = (u32)__builtin_amdgcn_readfirstlane((int)(s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3])) +
                      (u32)__builtin_amdgcn_readlane((int)misc, 33 + wave);
    const u32 cap32 = (u32)min(cap_faces, (int64_t)0x7fffffff);
    const u32 cap_rel = cap32 > wrun0 ? cap32 - wrun0 : 0u;
    int32_t* const wfaces = faces + (size_t)wrun0 * 3;   // (wave-uniform base: the stores take a 32-bit offset)
    u32 rel = 0;                                          // faces this wave has written so far (uniform)

    // per unit (lane = unit of this wave): active cells.  A cell is inactive when its eight corners agree, i.e. when bits z
    // and z+1 of all four columns are 0 (of their OR) or all 1 (of their AND): two words and ONE shifted copy of each
    // instead of four columns with a shifted copy each.  The unit's own columns are still in the registers of the prologue
    // (first staging pass: lane = unit); bit z+1 at z = 63 is bit 0 of the next chunk's columns, read from the staged words.
    u64 act_all = 0;
    {
        const u64 Xo = st_w0[0] | st_w1[0] | st_y0[0] | st_y1[0], Na = st_w0[0] & st_w1[0] & st_y0[0] & st_y1[0];
        const u32* const L0 = (const u32*)s_w[0];
        const u32* const L1 = (const u32*)s_w[1];
        const u32 n0 = L0[2 * (tid + 1)], n1 = L1[2 * (tid + 1)], n3 = L0[2 * (tid + 1 + hoff)], n2 = L1[2 * (tid + 1 + hoff)];
        const u64 xn = (valid && more) ? (u64)((n0 | n1 | n2 | n3) & 1u) : 0ull;
        const u64 nn = (valid && more) ? (u64)((n0 & n1 & n2 & n3) & 1u) : 0ull;
        const u64 Sx = (Xo >> 1) | (xn << 63), Sn = (Na >> 1) | (nn << 63);
        if (valid) act_all = (Xo | Sx) & ~(Na & Sn) & zedge(d, c);
    }
    // phase B (lane = unit): the wave's active cells are numbered unit by unit, ascending z (no list is built: a list
    // needs one loop trip per active cell of the wave's FULLEST unit -- tens of trips where the surface runs along z --
    // and cost 1-2 us of a wave's 9.6, tools/dev/faces_stamps.py)
    const u32 pc = (u32)popc64(act_all);
    const u32 inc0 = wave_prefix_sum(pc);
    const u32 na = (u32)__builtin_amdgcn_readlane((int)inc0, 63);
    const u32 off = inc0 - pc;                      // this unit's cells are slots [off, off + pc)
    const u32 act_lo = (u32)act_all, act_hi = (u32)(act_all >> 32);
    const bool owner = pc > 0u;
    u32* const mark = &s_ids[wave][0][0];
    const u32* const W0 = (const u32*)s_w[0];
    const u32* const W1 = (const u32*)s_w[1];
    u32* const ids = &s_ids[wave][0][lane];   // this lane's column: edge e at ids[e * 64]
#if P3D_FACES_STAMP
    FSTAMP(5);
    stamp[10] = na;
#endif
    // the cell batches, in two versions of the same text: the second translates the ids of marked units one by one
    // (layout form, the few tiles that stage such a unit).  Two versions, not a branch inside the loop: a conditional
    // region with LDS look-ups in the batch makes the compiler drain the batch's LDS reads at its borders -- the
    // reads this loop issues together -- and cost every tile 40 % (k_faces 58 -> 82 us with four tiles on that path)
    auto batches = [&](auto per_id_tag) {
        constexpr bool PER_ID = decltype(per_id_tag)::value;
        for (u32 i0 = 0; i0 < na; i0 += 64) {
            // phase C (lane = cell slot i0 + lane).  Its unit: the units that start inside this batch drop a marker on
            // their first slot, a prefix max spreads it (slots in front of the first marker belong to the last unit that
            // starts before the batch); its z: the (slot - off)-th set bit of that unit's active word.
            mark[lane] = 0u;
            if (owner && off >= i0 && off < i0 + 64u) mark[off - i0] = (u32)lane + 1u;
            wave_lds_sync();
            const u32 mk = mark[lane];
            const u64 before = __ballot(owner && off < i0);
            const u32 prev = before ? 63u - (u32)__builtin_clzll(before) : 0u;
            const u32 um = wave_prefix_max(mk);
            const u32 tl = (um ? um : prev + 1u) - 1u;
            const bool on = i0 + (u32)lane < na;
            const int tsel = (int)(tl << 2);
            const u32 u_lo = (u32)__builtin_amdgcn_ds_bpermute(tsel, (int)act_lo);
            const u32 u_hi = (u32)__builtin_amdgcn_ds_bpermute(tsel, (int)act_hi);
            const u32 u_off = (u32)__builtin_amdgcn_ds_bpermute(tsel, (int)off);
            // (idle lanes of the last batch shadow the first active cell of the batch's last unit: a valid active cell)
            const u32 rnk = on ? i0 + (u32)lane - u_off : 0u;
            const u32 cz = select_bit(((u64)u_hi << 32) | u_lo, rnk);
            const u32 cell = (((u32)wave0 + tl) << 6) | cz;
            // Every listed cell has a sign change, i.e. at least one triangle.
            const u32 t = cell >> 6, zz = cell & 31u, D = cell >> 5, Dh = D + 2u * (u32)hoff;
            const bool z63 = (cell & 63u) == 63u;
            // the dword windows of the four columns: a = the half unit holding z, b = the dword behind it
            const u32 a0 = W0[D], c0 = W0[D + 1], a1 = W1[D], c1 = W1[D + 1];
            const u32 a3 = W0[Dh], c3 = W0[Dh + 1], a2 = W1[Dh], c2 = W1[Dh + 1];
            // per-half id records: {base, offsets of this half}; the next unit's {base, low-half offsets} for z = 63
            const u32 hsel = 1u + ((cell >> 5) & 1u);
            const u32 th = t + (u32)hoff;
            const u32 hsh = (cell >> 1) & 16u;   // plane x+1: the half's two offset bytes sit at bit 0 or bit 16
            const u32 e3 = __umul24(t, 3u), eh3 = __umul24(th, 3u);   // (v_mul_u32_u24: full rate; a 32-bit multiply is not)
            const u32 B0 = E0[e3], O0 = E0[e3 + hsel], B1 = E1[2 * t], O1 = E1[2 * t + 1] >> hsh;
            const u32 B3 = E0[eh3], O3 = E0[eh3 + hsel], B2 = E1[2 * th], O2 = E1[2 * th + 1] >> hsh;
            const u32 N0 = E0[e3 + 3], NO0 = E0[e3 + 4], N1 = E1[2 * t + 2], NO1 = E1[2 * t + 3], N3 = E0[eh3 + 3];
            // (touched here so that the five reads leave with the others: left to itself the compiler sinks them into a
            //  divergent "some lane has z = 63" branch -- true in most batches -- with a wait of its own inside)
            asm volatile("" ::"v"(N0), "v"(NO0), "v"(N1), "v"(NO1), "v"(N3));
            // corner mask (interleaved): bits z and z+1 of every column
            const u32 t0 = __builtin_amdgcn_alignbit(c0, a0, zz), t1 = __builtin_amdgcn_alignbit(c1, a1, zz);
            const u32 t2 = __builtin_amdgcn_alignbit(c2, a2, zz), t3 = __builtin_amdgcn_alignbit(c3, a3, zz);
            const u32 mask = (t0 & 3u) | ((t1 & 3u) << 2) | ((t2 & 3u) << 4) | ((t3 & 3u) << 6);
            const u64 row = g_tri_rows.r[mask];   // (2 KiB, hot in the vector L1)
            const u32 row_lo = (u32)row, row_hi = (u32)(row >> 32);
            const u32 nt = on ? row_hi >> 28 : 0u;
#if P3D_FACES_STAMP
            if (i0 == 0) { asm volatile("" :: "v"(nt)); FSTAMP(6); }   // first batch: header (windows, records, table row) in
#endif
            // crossing words of this half (axis 0: columns (x,y) and (x,y+1); axis 1: (x,y) and (x+1,y); axis 2: all four)
            const u32 lowm = (1u << zz) - 1u;
            const u32 Cx0 = a0 ^ a1, Cx3 = a3 ^ a2, Cy0 = a0 ^ a3, Cy1 = a1 ^ a2;
            const u32 Cz0 = a0 ^ __builtin_amdgcn_alignbit(c0, a0, 1), Cz1 = a1 ^ __builtin_amdgcn_alignbit(c1, a1, 1);
            const u32 Cz2 = a2 ^ __builtin_amdgcn_alignbit(c2, a2, 1), Cz3 = a3 ^ __builtin_amdgcn_alignbit(c3, a3, 1);
            // edges at z (Bourke numbering, marching_cubes.cu:178-192)
            const u32 id0 = rank32(Cx0, lowm, add_byte0(B0, O0));
            const u32 id3 = rank32(Cy0, lowm, add_byte1(B0, O0));
            const u32 id8 = rank32(Cz0, lowm, add_byte2(B0, O0));
            const u32 id1 = rank32(Cy1, lowm, add_byte0(B1, O1));
            const u32 id9 = rank32(Cz1, lowm, add_byte1(B1, O1));
            const u32 id10 = rank32(Cz2, lowm, add_byte1(B2, O2));
            const u32 id2 = rank32(Cx3, lowm, add_byte0(B3, O3));
            const u32 id11 = rank32(Cz3, lowm, add_byte2(B3, O3));
            // edges at z+1: one more if the edge at z crosses (mask bit 2k = column k at z); at z = 63 they are the
            // first ids of the next chunk of the row
            const u32 q = mask ^ (mask >> 2), r = mask ^ (mask >> 6);
            // (both sides are computed and pinned, then selected: left alone the compiler turns every one of these into a
            //  divergent branch pair)
            u32 a4 = id0 + (q & 1u), a7 = id3 + (r & 1u), a5 = id1 + ((q >> 2) & 1u), a6 = id2 + ((q >> 4) & 1u);
            u32 n7 = add_byte1(N0, NO0), n5 = add_byte0(N1, NO1);
            asm volatile("" : "+v"(a4), "+v"(a7), "+v"(a5), "+v"(a6), "+v"(n7), "+v"(n5));
            const u32 id4 = z63 ? N0 : a4;
            const u32 id7 = z63 ? n7 : a7;
            const u32 id5 = z63 ? n5 : a5;
            const u32 id6 = z63 ? N3 : a6;
            u32 idv[12] = {id0, id1, id2, id3, id4, id5, id6, id7, id8, id9, id10, id11};
            if constexpr (PER_ID) {   // (ids of marked units carry kTailMark and are rows of the layout)
#pragma unroll
                for (int e = 0; e < 12; ++e) {
                    const u32 row = idv[e] & ~kTailMark;
                    if ((idv[e] & kTailMark) && row >= layV)
                        idv[e] = __float_as_uint(__builtin_nontemporal_load(a.lay_verts + (size_t)min(row, layEnd - 1u) * 3));
                    else idv[e] = row;
                }
            }
#pragma unroll
            for (int e = 0; e < 12; ++e) ids[e * 64] = idv[e];
#if P3D_FACES_STAMP
            if (i0 == 0) FSTAMP(7);   // first batch: ids computed and written
#endif
            // the batch's triangles, k-th triangle of every cell together: the lanes that have one write a DENSE run.
            // The three ids come back out of the lane's LDS column by table index.
            // The ids of the first two triangles are read back before the first store (most cells have two; a cell with one
            // reads three ids it does not use: row nibble 15 is clamped to a valid column).
            typedef int i3u __attribute__((ext_vector_type(3), aligned(4)));
            i3u tri0, tri1;
            tri0.x = (int32_t)ids[row_nibble<0>(row_lo, row_hi) * 64];
            tri0.y = (int32_t)ids[row_nibble<1>(row_lo, row_hi) * 64];
            tri0.z = (int32_t)ids[row_nibble<2>(row_lo, row_hi) * 64];
            tri1.x = (int32_t)ids[min(row_nibble<3>(row_lo, row_hi), 11u) * 64];
            tri1.y = (int32_t)ids[min(row_nibble<4>(row_lo, row_hi), 11u) * 64];
            tri1.z = (int32_t)ids[min(row_nibble<5>(row_lo, row_hi), 11u) * 64];
            {   // k = 0: every listed cell has one
                const u64 have = __ballot(on);
                const u32 f = __builtin_amdgcn_mbcnt_hi((u32)(have >> 32), __builtin_amdgcn_mbcnt_lo((u32)have, rel));
                if (on && f < cap_rel) __builtin_nontemporal_store(tri0, (i3u*)((char*)wfaces + __umul24(f, 12u)));
                rel += (u32)popc64(have);
            }
            bool go = true;
            static_for<1, 5>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (!go) return;
                const u64 have = __ballot((u32)k < nt);
                if (!have) {   // wave-uniform
                    go = false;
                    return;
                }
                const u32 f = __builtin_amdgcn_mbcnt_hi((u32)(have >> 32), __builtin_amdgcn_mbcnt_lo((u32)have, rel));
                if ((u32)k < nt && f < cap_rel) {
                    i3u tv;
                    if constexpr (k == 1) {
                        tv = tri1;
                    } else {
                        tv.x = (int32_t)ids[row_nibble<3 * k>(row_lo, row_hi) * 64];
                        tv.y = (int32_t)ids[row_nibble<3 * k + 1>(row_lo, row_hi) * 64];
                        tv.z = (int32_t)ids[row_nibble<3 * k + 2>(row_lo, row_hi) * 64];
                    }
                    __builtin_nontemporal_store(tv, (i3u*)((char*)wfaces + __umul24(f, 12u)));
                }
                rel += (u32)popc64(have);
            });
#if P3D_FACES_STAMP
            if (i0 == 0) FSTAMP_NOWAIT(8);   // first batch: stores issued
#endif
        }
    };
    if (per_id) batches(std::true_type{});
    else batches(std::false_type{});
#if P3D_FACES_STAMP
    FSTAMP_NOWAIT(9);
    stamp[11] = rel;
    if (lane == 0 && g_face_stamps) {
        u64* o = g_face_stamps + ((size_t)b * 4 + wave) * 16;
        for (int k = 0; k < 12; ++k) o[k] = stamp[k];
        o[12] = __builtin_amdgcn_s_memrealtime();
        u32 xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[13] = xcc;
        o[14] = rt0;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) return fail(P3D_EHIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

int check_dims(int64_t rx, int64_t ry, int64_t rz) {
    if (rx < 1 || ry < 1 || rz < 1) return fail(P3D_EINVAL, "grid dims must be >= 1%s");
    if (ry * rz >= (1ll << 29) || rz >= (1ll << 24)) return fail(P3D_ERANGE, "plane too large%s");
    const Dims d = make_dims(rx, ry, rz);
    // int32 vertex ids / 32-bit block bases: refuse grids whose worst case cannot be indexed safely
    if (d.U >= (1ll << 31) || rx * ry >= (1ll << 40)) return fail(P3D_ERANGE, "grid too large%s");
    return P3D_OK;
}

int grid_for(int64_t work_items, int per_block, int64_t cap) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ---- optional stage timing (bench/roofline): hipEvents recorded on the caller's stream ----------
enum { ST_CLASSIFY = 0, ST_UNIT_COUNTS, ST_SCAN_V, ST_UNIT_RECORDS, ST_FACES_COUNT, ST_SCAN_F, ST_EMIT_VERTS,
       ST_EMIT_FACES, ST_FUSED, ST_FINALIZE, ST_FUSED_INTERIOR, ST_N };
const char* const k_stage_names[ST_N] = {"k_classify",    "k_unit_counts", "k_scan_blocks(v)", "k_unit_records",
                                         "k_face_count_walk", "k_face_total", "k_emit_vertices", "k_faces",
                                         "k_fused", "(unused)", "k_fused(interior part)"};
// (measurement hooks: one global set of events, NOT thread-safe -- meant for a single benchmarking thread)
int g_prof_mode = 0;  // 0 off, 1 dominant kernel only (k_classify), 2 every stage
hipEvent_t g_ev[ST_N][2];
bool g_ev_made = false;
bool g_ev_used[ST_N];

struct StageTimer {
    int stage;
    hipStream_t st;
    bool on;
    StageTimer(int stage_, hipStream_t st_) : stage(stage_), st(st_) {
        on = g_prof_mode == 2 || (g_prof_mode == 1 && (stage == ST_CLASSIFY || stage == ST_FUSED));
        if (on) (void)hipEventRecord(g_ev[stage][0], st);
    }
    ~StageTimer() {
        if (on) {
            (void)hipEventRecord(g_ev[stage][1], st);
            g_ev_used[stage] = true;
        }
    }
};

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// What the library reads from the environment -- ONCE, at the first call (no getenv in the per-call host path).
//   SUPPORTED KNOBS (documented in include/p3d_mc.h and INTEGRATION.md; every build reads them):
//     P3D_FUSED_BLOCKS, P3D_FUSED_XT, P3D_COMPACT_BLOCKS, P3D_COMPACT_EARLY, P3D_FACES_SPARSE, P3D_NO_MAILBOX
//   DEVELOPER SWEEPS AND TEST HOOKS: compiled in only with -DP3D_DEV_HOOKS=1 (primitive3d_amd/_build.py builds that variant
//   as dev/libp3dmc.so for the tests that need a hook); the default library does not contain their names and uses the
//   built-in value -- a stray P3D_TEST_ID_LIMIT in a deployment's environment changes nothing.
#ifndef P3D_DEV_HOOKS
#define P3D_DEV_HOOKS 0
#endif
#if P3D_DEV_HOOKS
#define P3D_DEV_KNOB(name, dflt) env_int(name, dflt)
#else
#define P3D_DEV_KNOB(name, dflt) (dflt)
#endif
struct Tuning {
    int fused_blocks, fused_xt, compact_blocks, compact_early, faces_sparse;   // supported; -1 = "use the built-in rule"
    // dev hooks
    int fused_xt_tail, fused_tail_div, split_rows, small16, test_id_limit, no_chunk_pre, test_index_limit, stack_nparts,
        stack_early, fused_nbig, fused_nmid, fused_xt_mid, parts_ring, fail_after_lease;
};
Tuning read_tuning() {
    Tuning t{};
    t.fused_blocks = env_int("P3D_FUSED_BLOCKS", 2048);     // streaming launch: about this many blocks (x-slabs of <= 16 planes)
    t.fused_xt = env_int("P3D_FUSED_XT", -1);               // streaming launch: planes per block
    t.compact_blocks = env_int("P3D_COMPACT_BLOCKS", 256);  // blocks that copy the vertex regions to their dense place
    t.compact_early = env_int("P3D_COMPACT_EARLY", 3);      // ... of which this many slices ride with the counting launch
    t.faces_sparse = env_int("P3D_FACES_SPARSE", -1);       // empty-tile pre-check of the face launch: -1 by rule, 0 never, 1 always
    t.fused_xt_tail = P3D_DEV_KNOB("P3D_FUSED_XT_TAIL", -1);
    t.fused_tail_div = P3D_DEV_KNOB("P3D_FUSED_TAIL_DIV", 4);
    t.split_rows = P3D_DEV_KNOB("P3D_FUSED_SPLIT_ROWS", 1);
    t.small16 = P3D_DEV_KNOB("P3D_FUSED_SMALL16", 1);
    t.test_id_limit = P3D_DEV_KNOB("P3D_TEST_ID_LIMIT", 1 << 26);           // pretend a region's id space is smaller
    t.no_chunk_pre = P3D_DEV_KNOB("P3D_NO_CHUNK_PRE", 0);                   // every face tile adds the chunk totals up itself
    t.test_index_limit = P3D_DEV_KNOB("P3D_TEST_INDEX_LIMIT", 0x7fffffff);  // pretend the int32 limit is smaller
    t.stack_nparts = P3D_DEV_KNOB("P3D_STACK_NPARTS", -1);
    t.stack_early = P3D_DEV_KNOB("P3D_STACK_EARLY", -1);
    t.fused_nbig = P3D_DEV_KNOB("P3D_FUSED_NBIG", -1);
    t.fused_nmid = P3D_DEV_KNOB("P3D_FUSED_NMID", -1);
    t.fused_xt_mid = P3D_DEV_KNOB("P3D_FUSED_XT_MID", -1);
    t.parts_ring = P3D_DEV_KNOB("P3D_PARTS_RING", 1);
    t.fail_after_lease = P3D_DEV_KNOB("P3D_TEST_FAIL_AFTER_LEASE", 0);      // fail this many whole-grid calls between taking
                                                                            // their cursor block and their first launch
    // (knobs that are divided by or used as counts: a zero or negative value from the environment means "the smallest legal")
    t.fused_blocks = std::max(1, t.fused_blocks);
    t.fused_tail_div = std::max(1, t.fused_tail_div);
    t.compact_blocks = std::max(kRegions, t.compact_blocks);
    t.compact_early = std::max(0, t.compact_early);
    t.test_id_limit = std::max(1, t.test_id_limit);
    t.test_index_limit = std::max(1, t.test_index_limit);
    return t;
}
Tuning g_tuning;
std::once_flag g_tuning_once;
#if P3D_DEV_HOOKS
std::atomic<int> g_fail_after_lease{0};
#endif
// what has been launched so far (p3d_mc_debug_counters): fixed-slab / dynamic streaming launches, streaming passes, count+emit calls,
// emissions without a streaming pass of their own (part 6)
std::atomic<int64_t> g_counters[5];
std::atomic<int64_t> g_ring_bytes{0};   // device memory the cursor rings hold right now (p3d_mc_debug_counters, out[6])
void apply_tuning(const Tuning& t) {
    g_tuning = t;
#if P3D_DEV_HOOKS
    g_fail_after_lease.store(t.fail_after_lease);
#endif
}
const Tuning& tuning() {
    std::call_once(g_tuning_once, [] { apply_tuning(read_tuning()); });
    return g_tuning;
}

// ---- cursor blocks ---------------------------------------------------------------------------------
// The streaming kernel's 32 output cursors must be zero when a call starts.
// Instead of a fill kernel per call the library owns, per (device, stream), a ring of kCursorRing blocks: a starting call
// takes the block its predecessor's streaming kernel cleared for it (`next_slot`), and its own kernel clears the block the
// NEXT starting call will take -- the first one behind its own that no extraction is holding (stream order makes the
// clearing safe: that block's last user ran earlier on the same stream).  A slab streamed in two parts HOLDS its block from
// part 1 to part 2 / 3: starting calls step over held blocks, however many of them run in between (until round 5 a held
// block was simply overrun after 13 other starts, and the continuing part refused -- ADVICE r05).
constexpr int kCursorRing = 16;
constexpr int kRingSlotWords = kCursorBlockWords;   // the 32 cursors of one call
struct CursorRing {
    int dev = -1;
    hipStream_t stream = nullptr;
    u64* base = nullptr;
    int next_slot = 0;      // the block the next starting call takes: cleared (by the memset at creation, or by the kernel of
                            // the call that started before it)
    unsigned held = 0;      // bit s: block s is held by an extraction between its part 1 and its part 2 / 3
    std::mutex mu;   // held by a call from the moment it takes its block until its LAST kernel is enqueued
};
// (device, stream) -> ring.  shared_ptr: a call's lease keeps its ring alive while p3d_mc_release_stream / p3d_mc_shutdown
// on another thread takes it out of the map.
std::mutex g_ring_mu;
std::map<std::pair<int, hipStream_t>, std::shared_ptr<CursorRing>> g_rings;

// The cursor block an extraction in several calls took with its first part (part 1): the parts that continue the streaming
// (2, 3) use the same block; kept in the workspace's entry of the protocol table.
struct HeldBlock {
    u64* cursors = nullptr;               // null: the extraction keeps its cursors in its workspace header
    std::shared_ptr<CursorRing> ring;
    u64* ring_base = nullptr;             // the ring's allocation when the block was taken (a released ring is gone)
    int slot = -1;                        // its index in the ring (bit of CursorRing::held)
};
// gives a held block back to its ring (the continuing part has been enqueued; or the extraction was abandoned: a new start
// on its workspace, its entry leaving the protocol table)
void release_held(HeldBlock* h) {
    if (h->cursors && h->ring) {
        std::lock_guard<std::mutex> g(h->ring->mu);
        if (h->ring->base == h->ring_base && h->slot >= 0) h->ring->held &= ~(1u << h->slot);
    }
    *h = HeldBlock();
}

// A call's hold on its stream's ring.  Two host threads may share one stream (through ctypes the GIL is released): the
// block a call uses is cleared by the streaming kernel of the call BEFORE it in ring order, and is read by all three of
// its kernels -- so the calls of one stream must enqueue their launches as whole calls, in ring order.  The lease keeps
// the ring locked until the caller has enqueued its last kernel (a few tens of microseconds of host time; the GPU
// serialises the stream anyway).
struct RingLease {
    std::shared_ptr<CursorRing> ring;
    std::unique_lock<std::mutex> lock;
    int take = 0, clears = 0;
    // The ring moves on only when the call's streaming kernel IS enqueued: that kernel is what dirties the call's block and
    // clears the next one.  A call that fails between taking its block and that launch (a HIP error, the dev-build hook
    // P3D_TEST_FAIL_AFTER_LEASE) leaves the ring where it was: the next call takes the same, still clean block, and the
    // block after it is still waiting for a clearing kernel that will now be that call's.
    void commit(bool hold = false) {
        if (ring) {
            ring->next_slot = clears;
            if (hold) ring->held |= 1u << take;
        }
    }
};

// the block a starting call takes (and in *zero_next the one its kernel clears for the starting call after it); the caller
// commits the lease once its streaming kernel is enqueued.  *block = nullptr (and P3D_OK) when every other block of the
// ring is held: the caller then keeps its cursors in its workspace header.
int cursor_block_for(hipStream_t st, RingLease* lease, u64** block, u64** zero_next) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::shared_ptr<CursorRing> r;
    {
        std::lock_guard<std::mutex> g(g_ring_mu);
        auto it = g_rings.find({dev, st});
        if (it != g_rings.end()) {
            r = it->second;
        } else {
            std::shared_ptr<CursorRing> n(new CursorRing);
            n->dev = dev;
            n->stream = st;
            const size_t bytes = (size_t)kCursorRing * kRingSlotWords * sizeof(u64);
            HIP_TRY(hipMalloc((void**)&n->base, bytes));
            if (hipMemsetAsync(n->base, 0, bytes, st) != hipSuccess) {
                (void)hipFree(n->base);
                return fail(P3D_EHIP, "hipMemsetAsync(cursor ring)%s");
            }
            g_ring_bytes.fetch_add((int64_t)bytes, std::memory_order_relaxed);
            n->next_slot = 0;
            g_rings[{dev, st}] = n;
            r = n;
        }
    }
    lease->ring = r;
    lease->lock = std::unique_lock<std::mutex>(r->mu);
    if (!r->base) return fail(P3D_EINVAL, "the stream's state was released while a call on it was starting%s");
    lease->take = r->next_slot;
    int zn = -1;
    for (int k = 1; k < kCursorRing; ++k) {
        const int sl = (lease->take + k) % kCursorRing;
        if (!((r->held >> sl) & 1u)) {
            zn = sl;
            break;
        }
    }
    if (zn < 0) {   // fifteen extractions hold a block each: this one keeps its cursors in its workspace header
        lease->lock.unlock();
        lease->ring.reset();
        *block = nullptr;
        *zero_next = nullptr;
        return P3D_OK;
    }
    lease->clears = zn;
    *block = r->base + (size_t)lease->take * kRingSlotWords;
    *zero_next = r->base + (size_t)zn * kRingSlotWords;
    return P3D_OK;
}

// frees one ring: waits for whoever is enqueueing on it, then for the stream's work (its kernels read and clear the blocks)
int free_ring(const std::shared_ptr<CursorRing>& r, bool stream_alive) {
    std::lock_guard<std::mutex> g(r->mu);
    if (!r->base) return P3D_OK;
    if (stream_alive) (void)hipStreamSynchronize(r->stream);
    const hipError_t e = hipFree(r->base);   // (hipFree itself waits for the device: safe even after the stream is gone)
    r->base = nullptr;
    if (e == hipSuccess) g_ring_bytes.fetch_sub((int64_t)kCursorRing * kRingSlotWords * (int64_t)sizeof(u64), std::memory_order_relaxed);
    if (e != hipSuccess) return fail(P3D_EHIP, "hipFree(cursor ring): %s", hipGetErrorString(e));
    return P3D_OK;
}

// ---- result mailbox (host side) ------------------------------------------------------------------
// One ring of 64-byte slots in pinned host-coherent memory per device; every call that produces totals takes the
// next sequence number, its kernels publish into slot seq % kMbSlots (mb_publish_*), and p3d_mc_read_counts polls that
// slot.  The workspace pointer keys the pending call.  Anything unexpected (no pinned memory, slot recycled by 64
// newer calls, time-out) falls back to the copy + synchronise path, which reads the same totals from the header.
constexpr int kMbSlots = 64, kMbSlotWords = 64;   // (a slot: 5 words of totals + 32 region totals at words 8..39)
struct Mailbox {
    u64* host = nullptr;  // also valid as device pointer (hipHostMallocMapped, unified addressing)
    u64* dev = nullptr;
    u64 next_seq = 1;
    bool failed = false;
};
struct PendingCall {
    const void* ws = nullptr;
    int dev = -1;
    u64 seq = 0;
};
std::mutex g_mb_mu;
Mailbox g_mb[64];
PendingCall g_pending[kMbSlots];
int g_pending_next = 0;

// returns the device pointer of the slot and the sequence number; nullptr when the mailbox is unavailable
u64* mailbox_open(const void* ws, u64* seq_out) {
    static const bool disabled = env_int("P3D_NO_MAILBOX", 0) != 0;
    *seq_out = 0;
    if (disabled) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(g_mb_mu);
    Mailbox& m = g_mb[dev];
    if (m.failed) return nullptr;
    if (!m.host) {
        void* h = nullptr;
        void* dptr = nullptr;
        if (hipHostMalloc(&h, (size_t)kMbSlots * kMbSlotWords * 8,
                          hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess ||
            hipHostGetDevicePointer(&dptr, h, 0) != hipSuccess) {
            (void)hipGetLastError();
            m.failed = true;
            return nullptr;
        }
        memset(h, 0, (size_t)kMbSlots * kMbSlotWords * 8);
        m.host = (u64*)h;
        m.dev = (u64*)dptr;
    }
    const u64 seq = m.next_seq++;
    for (auto& pc : g_pending)  // a workspace has one pending call at most
        if (pc.ws == ws && pc.dev == dev) pc.ws = nullptr;
    PendingCall& pc = g_pending[g_pending_next];
    g_pending_next = (g_pending_next + 1) % kMbSlots;
    pc.ws = ws;
    pc.dev = dev;
    pc.seq = seq;
    *seq_out = seq;
    return m.dev + (seq % kMbSlots) * kMbSlotWords;
}

// polls the slot of the pending call on `ws`; false = not pending / recycled / timed out (use the copy path)
bool mailbox_wait(const void* ws, u64* nv, u64* nf, u64* flags, int64_t* regions = nullptr) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    u64 seq = 0;
    const volatile u64* slot = nullptr;
    {
        std::lock_guard<std::mutex> g(g_mb_mu);
        for (auto& pc : g_pending)
            if (pc.ws == ws && pc.dev == dev) {
                seq = pc.seq;
                pc.ws = nullptr;
                slot = g_mb[dev].host + (seq % kMbSlots) * kMbSlotWords;
                break;
            }
    }
    if (!slot) return false;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spin = 0;; ++spin) {
        const u64 sv = __atomic_load_n(&slot[0], __ATOMIC_ACQUIRE), sf = __atomic_load_n(&slot[3], __ATOMIC_ACQUIRE);
        if (sv == seq && sf == seq) break;
        if (sv > seq || sf > seq) return false;  // slot recycled by a newer call
        if ((spin & 1023) == 1023 &&
            std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return false;
        __builtin_ia32_pause();
    }
    // seqlock: a call 64 sequence numbers newer (another thread or stream) may have recycled the slot between the
    // check above and these loads -- re-check the two sequence words after the payload
    const u64 v1 = __atomic_load_n(&slot[1], __ATOMIC_ACQUIRE), v2 = __atomic_load_n(&slot[2], __ATOMIC_ACQUIRE);
    const u64 v4 = __atomic_load_n(&slot[4], __ATOMIC_ACQUIRE);
    int64_t reg[kRegions];
    for (int r = 0; r < kRegions; ++r) reg[r] = (int64_t)__atomic_load_n(&slot[8 + r], __ATOMIC_ACQUIRE);
    if (__atomic_load_n(&slot[0], __ATOMIC_ACQUIRE) != seq || __atomic_load_n(&slot[3], __ATOMIC_ACQUIRE) != seq)
        return false;
    if (regions) memcpy(regions, reg, sizeof(reg));
    *nv = v1;
    *flags = v2;
    *nf = v4;
    return true;
}

// the counting launch: sub-batches of 8 planes, or of 4 for the 4-plane chunks of small grids
void launch_count_walk(dim3 grid, hipStream_t st, const u64* bits, const Dims& d, const Ws& w, u32* csum, u32* woff,
                       u32* tile_tris, const CompactArgs& cp, u64* hdr) {
    if (w.xw == 4)
        hipLaunchKernelGGL(k_face_count_walk<4>, grid, dim3(kBlock), 0, st, bits, d, w.tpp, w.xw, (int)w.cpi, csum, woff,
                           tile_tris, cp, hdr);
    else
        hipLaunchKernelGGL(k_face_count_walk<P3D_COUNT_PB>, grid, dim3(kBlock), 0, st, bits, d, w.tpp, w.xw, (int)w.cpi,
                           csum, woff, tile_tris, cp, hdr);
}

// the face launch: with `faces_here` one block per face tile, else the compaction blocks only
void launch_faces(const Dims& d, const Ws& w, const u64* bits, const uint2* rec, const FaceArgs& a_in, const CompactArgs& cp,
                  u64* hdr, int32_t* faces, int64_t capf, bool faces_here, hipStream_t st) {
    const dim3 fgrid((u32)((faces_here ? w.nb_f : 0) + cp.nblocks));
    if (fgrid.x == 0) return;
    FaceArgs a = a_in;
    a.div_tpp = make_fastdiv((u32)w.tpp);
    a.div_xper = make_fastdiv((u32)d.xper);
    a.div_ncz = make_fastdiv((u32)d.ncz);
    a.xw_shift = __builtin_ctz((unsigned)w.xw);
    // (by the capacity the caller sized for its expectation: fewer than 128 triangles per tile on average -- at 512^3 a noise
    //  field has 1270 per tile, a sphere 19)
    const int sparse_knob = tuning().faces_sparse;   // (P3D_FACES_SPARSE: -1 the rule, 0 never, 1 always -- dev A/B, tests)
    a.sparse = sparse_knob >= 0 ? sparse_knob : (faces_here && capf < (int64_t)w.nb_f * 128 ? 1 : 0);
    if (a.xlate == 3) {
        if (d.ncz <= 32)
            hipLaunchKernelGGL((k_faces<32, true>), fgrid, dim3(kBlock), 0, st, bits, rec, d, a, cp, hdr, faces, capf);
        else
            hipLaunchKernelGGL((k_faces<256, true>), fgrid, dim3(kBlock), 0, st, bits, rec, d, a, cp, hdr, faces, capf);
    } else if (d.ncz <= 32)
        hipLaunchKernelGGL((k_faces<32, false>), fgrid, dim3(kBlock), 0, st, bits, rec, d, a, cp, hdr, faces, capf);
    else
        hipLaunchKernelGGL((k_faces<256, false>), fgrid, dim3(kBlock), 0, st, bits, rec, d, a, cp, hdr, faces, capf);
}

template <typename T>
int count_impl(const T* grid, const Dims& d, const Ws& w, float thresh, const p3d_mc_slab* slab, char* ws,
               hipStream_t st) {
    const int halo = slab ? slab->halo_last_plane : 0;
    u64* hdr = (u64*)(ws + w.hdr);
    u64* bits = (u64*)(ws + w.bits);
    uint2* rec = (uint2*)(ws + w.rec);
    u32* cnt = (u32*)(ws + w.cnt);
    u32 *bsv = (u32*)(ws + w.bsum_v), *bbv = (u32*)(ws + w.bbase_v);
    u32 *csum = (u32*)(ws + w.chunk_sum), *woff = (u32*)(ws + w.wave_off);

    HIP_TRY(hipMemsetAsync(hdr + H_FLAGS, 0, 2 * sizeof(u64), st));  // no overflow, dense ids
    g_counters[3].fetch_add(1, std::memory_order_relaxed);
    u64 seq = 0;
    u64* mb = mailbox_open(ws, &seq);
    // classify: 64 units (16 KiB of fp32) per wave iteration; cap the grid and stride the rest
    const int64_t wave_iters = (d.U + 63) / 64;
    {
        StageTimer tm(ST_CLASSIFY, st);
        hipLaunchKernelGGL(k_classify<T>, dim3(grid_for(wave_iters, kBlock / 64, 256 * 16)), dim3(kBlock), 0, st,
                           grid, thresh, d, bits);
    }
    {
        StageTimer tm(ST_UNIT_COUNTS, st);
        hipLaunchKernelGGL(k_unit_counts, dim3((u32)w.nb_v), dim3(kBlock), 0, st, bits, d, halo, cnt, bsv);
    }
    {
        StageTimer tm(ST_SCAN_V, st);
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, bsv, bbv, w.nb_v, hdr + H_V, mb, seq, 0);
    }
    {
        StageTimer tm(ST_UNIT_RECORDS, st);
        hipLaunchKernelGGL(k_unit_records, dim3((u32)w.nb_v), dim3(kBlock), 0, st, cnt, bbv, d, halo, rec);
    }
    if (w.nchunks > 0) {
        StageTimer tm(ST_FACES_COUNT, st);
        const CompactArgs none{nullptr, nullptr, 0, 0, 0, 0, 0, 1, 0, nullptr, 0, nullptr, 1u << 26, 1, nullptr};
        launch_count_walk(dim3((u32)w.nchunks), st, bits, d, w, csum, woff, (u32*)(ws + w.tile_tris), none, hdr);
    }
    {
        StageTimer tm(ST_SCAN_F, st);
        hipLaunchKernelGGL(k_face_total, dim3(1), dim3(kBlock), 0, st, csum, (int)w.nchunks, hdr, mb, seq, 0);
    }
    HIP_TRY(hipGetLastError());
    return P3D_OK;
}

template <typename T>
int emit_impl(const T* grid, const Dims& d, const Ws& w, float thresh, const Xform& t, const p3d_mc_slab* slab,
              char* ws, float* verts, int64_t capv, int32_t* faces, int64_t capf, int64_t* keys, hipStream_t st) {
    const int halo = slab ? slab->halo_last_plane : 0;
    u64* bits = (u64*)(ws + w.bits);
    uint2* rec = (uint2*)(ws + w.rec);
    u32* cnt = (u32*)(ws + w.cnt);
    u64* hdr = (u64*)(ws + w.hdr);
    if (capv > 0) {
        // a one-pass call leaves the records in region form: the gather emitter wants them dense
        hipLaunchKernelGGL(k_fix_records, dim3((u32)w.nb_v), dim3(256), 0, st, rec, d.U, hdr, 1);
        HIP_TRY(hipMemsetAsync(hdr + H_RECFORM, 0, sizeof(u64), st));
        // per-unit crossing counts select the units to visit; recomputed here (12 us at 512^3) because the
        // fused streaming kernel does not materialise them
        hipLaunchKernelGGL(k_unit_counts, dim3((u32)w.nb_v), dim3(kBlock), 0, st, bits, d, halo, cnt,
                           (u32*)(ws + w.bsum_v));
        StageTimer tm(ST_EMIT_VERTS, st);
        hipLaunchKernelGGL(k_emit_vertices<T>, dim3(grid_for(d.U, kBlock / 64, 256 * 32)), dim3(kBlock), 0, st, grid,
                           thresh, d, bits, cnt, rec, t, slab ? slab->x_origin : (int64_t)0, verts, capv, keys);
    }
    if (w.nb_f > 0 && capf > 0) {
        StageTimer tm(ST_EMIT_FACES, st);
        const FaceArgs a{2, halo, slab ? slab->vertex_id_base : 0, slab ? slab->halo_vertex_id_base : 0,
                         slab ? slab->rank_counts : nullptr, slab ? slab->rank : 0,
                         (slab && slab->rank_counts_stride > 0) ? slab->rank_counts_stride : 1, w.tpp, w.xw, (int)w.cpi,
                         (const u32*)(ws + w.chunk_sum), nullptr, (const u32*)(ws + w.wave_off),
                         (const u32*)(ws + w.tile_tris), nullptr, nullptr, 0};
        const CompactArgs none{nullptr, nullptr, 0, 0, 0, 0, 0, 1, 0, nullptr, 0, nullptr, 1u << 26, 1, nullptr};
        launch_faces(d, w, bits, rec, a, none, hdr, faces, capf, true, st);
    }
    HIP_TRY(hipGetLastError());
    return P3D_OK;
}


template <typename T, int NC, int RY>
void launch_fused(const T* grid, const Dims& d, float thresh, int halo, const Xform& t, int64_t x_origin, u64* bits,
                  uint2* rec, u64* cursors, u64* zero_next, float* scratch, u32 region_rows, u32 store_rows, int x_lo,
                  int x_hi,
                  hipEvent_t ev0, hipEvent_t ev1, hipStream_t st, const u64* region_first_row, const RegionLayout* layp,
                  u64* lay_out, int cz_base = 0, int cz_count = -1) {
    RegionLayout lay{};
    if (layp) lay = *layp;
    FusedGeom g;
    g.x_lo = x_lo;
    g.x_hi = x_hi;
    g.nxt_item = 0;
    g.n_mid = 0;
    g.XT_mid = 1;
    g.cz_base = cz_base;   // chunks [cz_base, cz_base + cz_count) of every row (default: the whole row)
    if (cz_count < 0) cz_count = (int)d.ncz - cz_base;
    const bool stack = d.stack != 0;   // a batch of grids: every x-slab lies inside one item, all items in one launch
    const int64_t nplanes = stack ? d.xper : x_hi - x_lo;
    if (nplanes <= 0) return;
    g.nzt = (cz_count + NC - 1) / NC;
    g.nyt = (int)((d.ry + kFusedWPB * RY - 1) / (kFusedWPB * RY));
    // planes per block: enough blocks to fill the chip a few times over, but >= 8 planes (x-halo overhead 1/XT)
    // unless the grid is so small that 8-plane blocks would leave most CUs idle (then latency wins over the halo)
    const int64_t per_slab = (int64_t)g.nzt * g.nyt * (stack ? d.nitems : 1);
    const Tuning& tn = tuning();
    const u32 thresh16 = half_round_down(thresh);
    // ---- fixed x-slabs
    // (measured, tools/dev/xt_sweep*.sh / blocks_sweep.sh: about 2000 blocks and at most 16 planes per block -- 12 planes
    //  at 512^3 (120 us vs 124 with 8), 16 at 1024^3 (976 us vs 1140 with the 43 an uncapped rule gave))
    int64_t want_slabs = std::max<int64_t>(1, (std::max(1, tn.fused_blocks) + per_slab - 1) / per_slab);
    int xt = (int)std::min<int64_t>(16, (nplanes + want_slabs - 1) / want_slabs);
    if (xt < 8) {
        const int64_t blocks_at_8 = per_slab * ((nplanes + 7) / 8);
        if (blocks_at_8 >= 1024) xt = 8;
        else {
            // (about 1400 blocks: a block and a half per slot of the chip -- 256^3: 4 planes per block 25.9 us, 6 planes
            //  27.4, 8 planes 28.9 on the round-3 kernel)
            const int64_t slabs_for_fill = (1400 + per_slab - 1) / per_slab;
            xt = (int)std::max<int64_t>(1, std::min<int64_t>(8, (nplanes + slabs_for_fill - 1) / slabs_for_fill));
        }
    }
    if (tn.fused_xt >= 0) xt = tn.fused_xt;
    if (xt < 1) xt = 1;
    if (xt > nplanes) xt = (int)nplanes;
    g.XT = xt;
    // taper: the last quarter of the long slabs is replaced by slabs of XT/2 planes (two thirds of those planes) and of XT/4
    // planes (the rest) -- profiles/r04/dyn_ranges.txt section 11
    const int xt_tail = tn.fused_xt_tail >= 0 ? tn.fused_xt_tail : (xt >= 4 ? xt / 4 : xt);
    const int64_t nslab_all = (nplanes + xt - 1) / xt;
    int64_t n_big = nslab_all - std::max<int64_t>(1, nslab_all / std::max(1, tn.fused_tail_div));
    if (xt_tail >= xt || nslab_all < 8) n_big = nslab_all;
    if (tn.fused_nbig >= 0) n_big = std::min<int64_t>(tn.fused_nbig, nslab_all);
    g.n_big = (int)n_big;
    g.XT_tail = xt_tail > 0 ? xt_tail : 1;
    int64_t rest = nplanes - n_big * xt;
    g.XT_mid = tn.fused_xt_mid > 0 ? tn.fused_xt_mid : std::max(1, xt / 2);
    // (rule: the first two thirds of what the big slabs leave)
    const int64_t want_mid = tn.fused_nmid >= 0 ? tn.fused_nmid : (g.XT_mid > g.XT_tail && g.XT_mid < xt ? rest * 2 / 3 / g.XT_mid : 0);
    g.n_mid = (int)std::max<int64_t>(0, std::min<int64_t>(want_mid, rest / g.XT_mid));
    rest -= (int64_t)g.n_mid * g.XT_mid;
    g.nxt = (int)(n_big + g.n_mid + (rest > 0 ? (rest + g.XT_tail - 1) / g.XT_tail : 0));
    if (stack) {   // uniform slabs, no taper (many items: the tail of the launch is short anyway)
        g.nxt_item = (int)((nplanes + xt - 1) / xt);
        g.nxt = g.nxt_item;   // (per_slab already counts the items)
        g.n_big = g.nxt;
        g.XT_tail = xt;
        g.n_mid = 0;
    }
    const int64_t nblocks = per_slab * g.nxt;
    g_counters[0].fetch_add(1, std::memory_order_relaxed);
    // (the timing events, if any, ride on the dispatch packet itself: no extra barrier packets around the kernel)
    if (ev0 || ev1)   // (rows split over two launches: the first carries the start event, the second the stop event)
        hipExtLaunchKernelGGL((k_fused<T, NC, RY>), dim3((u32)nblocks), dim3(kFusedBlock), 0, st, ev0, ev1, 0, lay, lay_out, grid, thresh,
                              thresh16, d, g, halo, t, x_origin, bits, rec, cursors, zero_next, scratch, region_rows,
                              store_rows, region_first_row);
    else
        hipLaunchKernelGGL((k_fused<T, NC, RY>), dim3((u32)nblocks), dim3(kFusedBlock), 0, st, lay, lay_out, grid, thresh, thresh16, d, g,
                           halo, t, x_origin, bits, rec, cursors, zero_next, scratch, region_rows, store_rows, region_first_row);
}

template <typename T>
void dispatch_fused(const T* grid, const Dims& d, float thresh, int halo, const Xform& t, int64_t x_origin, u64* bits,
                    uint2* rec, u64* cursors, u64* zero_next, float* scratch, u32 region_rows, u32 store_rows, int x_lo,
                  int x_hi,
                    hipEvent_t ev0, hipEvent_t ev1, hipStream_t st, const u64* region_first_row = nullptr,
                    const RegionLayout* layp = nullptr, u64* lay_out = nullptr) {
    // tile geometries (32 unit words per wave-plane unless noted): long rows (8 chunks x 3 rows per wave), rows of 3-4
    // chunks (rz <= 256: 4 chunks x 6 rows -- the 8-chunk tile would be half empty -- or, for a single small grid,
    // 16-unit tiles of 4 chunks x 3 rows), short rows (2 chunks x 15 rows).
    const int rem = (int)(d.ncz % 8);
    if (d.ncz >= 9 && rem >= 1 && rem <= 2 && tuning().split_rows) {
        // rows of 8k + 1..2 chunks (rz = 513, 517, 600, 1025: grids of 2^n + 1 samples are common): the 8-chunk tiles take
        // the first 8k chunks, a second launch with the 2-chunk tile the rest -- a last 8-chunk tile would be 1/8 or 1/4
        // full (513 x 511 x 517: 169 -> 155 us; with 3..4 chunks left over the split measured no gain)
        const int full = (int)d.ncz - rem;
        launch_fused<T, 8, 3>(grid, d, thresh, halo, t, x_origin, bits, rec, cursors, zero_next, scratch, region_rows,
                                     store_rows, x_lo, x_hi, ev0, nullptr, st, region_first_row, layp, lay_out, 0, full);
        launch_fused<T, 2, 15>(grid, d, thresh, halo, t, x_origin, bits, rec, cursors, zero_next, scratch, region_rows,
                                      store_rows, x_lo, x_hi, nullptr, ev1, st, region_first_row, layp, lay_out, full, rem);
        return;
    }
    if (d.ncz >= 5)
        launch_fused<T, 8, 3>(grid, d, thresh, halo, t, x_origin, bits, rec, cursors, zero_next, scratch, region_rows,
                                    store_rows, x_lo, x_hi, ev0, ev1, st, region_first_row, layp, lay_out);
    else if (d.ncz >= 3 &&
             // (only when 8-plane slabs of the wider-in-y tile still give the chip enough blocks: a single small grid is
             //  better off with more, half-empty tiles than with 2-plane slabs)
             ((d.ry + 23) / 24) * (d.stack ? d.nitems : 1) * ((x_hi - x_lo + 7) / 8) >= 1024)
        // (4 x 6 rows, not the 4 x 7 that would fill the 32 unit slots: a wave-plane of 28 units carries ~75 vertices on the
        //  Perlin stacks -- a full batch of 64 and a nearly empty one --, one of 24 units ~64: 32 x 256^3 fp16 342 -> 328 us,
        //  fp32 556 -> 535 us)
        launch_fused<T, 4, 6>(grid, d, thresh, halo, t, x_origin, bits, rec, cursors, zero_next, scratch, region_rows,
                                    store_rows, x_lo, x_hi, ev0, ev1, st, region_first_row, layp, lay_out);
    else if (d.ncz >= 3 && tuning().small16)
        // a single small grid: 16-unit tiles (4 chunks x 3 rows + halo row) -- twice the waves of the 8-chunk tile, none
        // of them half empty
        launch_fused<T, 4, 3>(grid, d, thresh, halo, t, x_origin, bits, rec, cursors, zero_next, scratch, region_rows,
                                     store_rows, x_lo, x_hi, ev0, ev1, st, region_first_row, layp, lay_out);
    else if (d.ncz >= 3)
        launch_fused<T, 8, 3>(grid, d, thresh, halo, t, x_origin, bits, rec, cursors, zero_next, scratch, region_rows,
                                    store_rows, x_lo, x_hi, ev0, ev1, st, region_first_row, layp, lay_out);
    else
        launch_fused<T, 2, 15>(grid, d, thresh, halo, t, x_origin, bits, rec, cursors, zero_next, scratch, region_rows,
                                      store_rows, x_lo, x_hi, ev0, ev1, st, region_first_row, layp, lay_out);
}

Xform make_xform(const Dims& d, const float lower[3], const float upper[3], const int64_t full_res[3]) {
    const int64_t fr[3] = {full_res ? full_res[0] : d.rx, full_res ? full_res[1] : d.ry, full_res ? full_res[2] : d.rz};
    // marching_cubes.cu:293-297 verbatim, including the upper[2]-lower[1] term of :295 (fp32 arithmetic)
    Xform t;
    t.sx = (upper[0] - lower[0]) / static_cast<float>(fr[0]);
    t.sy = (upper[2] - lower[1]) / static_cast<float>(fr[1]);
    t.sz = (upper[2] - lower[2]) / static_cast<float>(fr[2]);
    t.ox = lower[0];
    t.oy = lower[1];
    t.oz = lower[2];
    return t;
}

// p3d_mc_emit behind a counting pass (p3d_mc_count, or any finished one-pass extraction) of a whole grid: a SECOND streaming
// pass.  Which wave-plane adds to which of the 32 cursors depends on the launch geometry alone, so the region totals of the
// counting pass -- their prefix is in the workspace header -- hold for this pass as well: region r is stored at its final
// rows of the caller's exactly sized buffer (no scratch, no copy), the records are rewritten with this pass's slots, and the
// face launch translates them with this pass's cursors.  Replaces gen_vertices_kernel + gen_faces_kernel + the epilogue,
// marching_cubes.cu:266-298 (the reference, too, reads the field a second time there).
template <typename T>
int emit_stream_impl(const T* grid, const Dims& d, const Ws& w, float thresh, const Xform& t, char* ws, float* verts,
                     int64_t capv, int32_t* faces, int64_t capf, hipStream_t st) {
    u64* hdr = (u64*)(ws + w.hdr);
    u64* bits = (u64*)(ws + w.bits);
    uint2* rec = (uint2*)(ws + w.rec);
    u32 *csum = (u32*)(ws + w.chunk_sum), *woff = (u32*)(ws + w.wave_off);
    u32* cpre = (w.nchunks > kPreMinChunks && !tuning().no_chunk_pre) ? (u32*)(ws + w.chunk_pre) : nullptr;
    u64 *cursors = nullptr, *zero_next = nullptr;
    RingLease lease;   // (released when this function returns: both launches are enqueued by then)
    if (int rc = cursor_block_for(st, &lease, &cursors, &zero_next)) return rc;
    if (!cursors) {   // (every block of the stream's ring is held by an extraction in several calls: the header's cursors,
                      //  which this pass leaves with the values the counting pass left there)
        cursors = hdr + H_CURSORS;
        hipLaunchKernelGGL(k_zero_words, dim3((kCursorBlockWords + kBlock - 1) / kBlock), dim3(kBlock), 0, st, cursors,
                           (int64_t)kCursorBlockWords);
    }
    const bool timed = g_prof_mode != 0;
    if (timed) g_ev_used[ST_EMIT_VERTS] = true;
    dispatch_fused<T>(grid, d, thresh, 0, t, 0, bits, rec, cursors, zero_next, verts, 1u << 26,
                      (u32)std::min<int64_t>(capv, 0xffffffffll), 0, (int)d.rx, timed ? g_ev[ST_EMIT_VERTS][0] : nullptr,
                      timed ? g_ev[ST_EMIT_VERTS][1] : nullptr, st, hdr + H_PREFIX);
    HIP_TRY(hipGetLastError());
    lease.commit();
    if (w.nb_f > 0 && capf > 0) {
        StageTimer tm(ST_EMIT_FACES, st);
        const FaceArgs a{1, 0, 0, 0, nullptr, 0, 1, w.tpp, w.xw, (int)w.cpi, csum, cpre, woff, (const u32*)(ws + w.tile_tris),
                         cursors, nullptr, 0};
        const CompactArgs none{nullptr, nullptr, 0, 0, 0, 0, 0, 1, 0, nullptr, 0, nullptr, 1u << 26, 1, nullptr};
        launch_faces(d, w, bits, rec, a, none, hdr, faces, capf, true, st);
    }
    HIP_TRY(hipGetLastError());
    return P3D_OK;
}

template <typename T>
int fused_impl(const T* grid, const Dims& d, const Ws& w, float thresh, const Xform& t, const p3d_mc_slab* slab,
               char* ws, float* verts, int64_t capv, float* scratch, int64_t scratch_rows, int32_t* faces,
               int64_t capf, hipStream_t st, HeldBlock* held) {
    const int halo = slab ? slab->halo_last_plane : 0;
    u64* hdr = (u64*)(ws + w.hdr);
    u64* bits = (u64*)(ws + w.bits);
    uint2* rec = (uint2*)(ws + w.rec);
    u32 *csum = (u32*)(ws + w.chunk_sum), *woff = (u32*)(ws + w.wave_off);
    // exclusive prefix of the chunk totals, made by k_chunk_prefix between the two launches when there are many chunks
    // (P3D_NO_CHUNK_PRE=1: escape hatch and test reference -- every face tile adds the chunk totals up itself)
    u32* cpre = (w.nchunks > kPreMinChunks && !tuning().no_chunk_pre) ? (u32*)(ws + w.chunk_pre) : nullptr;
    // Vertex ids handed out by the streaming kernel are region * 2^26 + slot (a fixed stride, so they stay
    // unambiguous even when a region outgrows its share of the scratch buffer); readers make them dense on the fly.
    // Storage is store_rows rows per region; without a scratch buffer nothing is stored (pure count + ids).
    const u32 region_rows = 1u << 26;
    // (test hook: P3D_TEST_ID_LIMIT pretends the id space of a region is smaller, to reach the callers' fallback)
    const u32 id_limit = (u32)std::min<int64_t>(region_rows, std::max(1, tuning().test_id_limit));
    const u32 store_rows = scratch ? (u32)std::min<int64_t>(scratch_rows / kRegions, (int64_t)region_rows) : 0u;
    const int part = slab ? slab->part : 0;
    // parts (p3d_mc_slab.part): 0 everything; 1 planes [0, split) only; 2 planes [split, rx) + finalize;
    // 3 planes [split, rx) + early header, no finalize; 4 face count (+ first slices of the vertex copy), totals to
    // the host; 5 faces (+ the rest of the vertex copy); 6 faces + the WHOLE vertex copy (part 4 was given no vertex buffer)
    // a predicted layout of the 32 regions inside the caller's vertex buffer (p3d_mc_slab.region_first_rows: whole-grid calls)
    RegionLayout lay_v{};
    const RegionLayout* lay = nullptr;
    if (slab && slab->region_first_rows) {
        if (part != 0 || halo || !verts || capv <= 0)
            return fail(P3D_EINVAL, "region_first_rows: whole-grid calls (part 0, no halo plane) with a vertex buffer only%s");
        for (int r = 0; r < kLayRows; ++r) lay_v.first[r] = slab->region_first_rows[r];
        bool ok = lay_v.first[0] == 0u && (int64_t)lay_v.first[kLayRows - 1] <= capv && lay_v.first[kLayRows - 1] < 0x7fffff00u;
        for (int r = 0; r + 1 < kLayRows; ++r) ok = ok && lay_v.first[r] <= lay_v.first[r + 1];
        if (!ok) return fail(P3D_EINVAL, "region_first_rows must be 41 ascending rows from 0 to at most cap_vertices%s");
        // (the streaming kernel addresses a row by a 32-bit byte offset from its region's or spill area's first row)
        for (int r = 0; r + 1 < kLayRows; ++r)
            if (lay_v.first[r + 1] - lay_v.first[r] > kLayMaxRows)
                return fail(P3D_EINVAL, "region_first_rows: a region or spill area may hold at most 2^28 rows%s");
        lay_v.on = 1u;
        lay = &lay_v;
    }
    int x_lo = 0, x_hi = (int)d.rx;
    if (part == 1) x_hi = (int)slab->split_plane;
    if (part == 2 || part == 3) x_lo = (int)slab->split_plane;
    // The 32 output cursors: every call that STARTS streaming takes a pre-cleared block from the per-stream ring (no fill
    // kernel).  A call that continues the streaming of an extraction (part 2, or part 3 behind part 1) uses the block part 1
    // took -- remembered in the workspace's entry of the protocol table; other extractions may start on the stream in
    // between (the in-process multi-rank harness does exactly that), each taking the next block: the held block stays
    // untouched until the ring has come round, which is checked.  The parts behind the streaming (4, 5, 6) find the region
    // counts in the workspace header (k_early_header / the finishing block copy them there).
    u64 *cursors = nullptr, *zero_next = nullptr;
    RingLease lease;   // (released when this function returns: every launch of the call is enqueued by then)
    const bool new_block = part == 0 || part == 1 || (part == 3 && slab->split_plane == 0);
    const bool ring_block = part == 0 || (new_block && tuning().parts_ring != 0);   // (P3D_PARTS_RING=0: dev A/B, header cursors)
    const bool continues = (part == 2 || (part == 3 && slab->split_plane != 0)) && held && held->cursors;
    if (ring_block) {
        if (int rc = cursor_block_for(st, &lease, &cursors, &zero_next)) return rc;
#if P3D_DEV_HOOKS
        if (g_fail_after_lease.load() > 0 && g_fail_after_lease.fetch_sub(1) > 0)
            return fail(P3D_EHIP, "injected failure between the cursor lease and the first launch (P3D_TEST_FAIL_AFTER_LEASE)%s");
#endif
    } else if (continues) {
        // (the ring's lock from here to the last launch of this call, like a starting call: calls of one stream enqueue whole)
        lease.ring = held->ring;
        lease.lock = std::unique_lock<std::mutex>(held->ring->mu);
        if (held->ring->base != held->ring_base || !((held->ring->held >> held->slot) & 1u))
            return fail(P3D_EINVAL, "the cursor block part 1 took is gone: the stream's state was released "
                                    "(p3d_mc_release_stream / p3d_mc_shutdown) since%s");
        cursors = held->cursors;
        lease.ring.reset();   // (nothing to commit: the ring does not move for a continuing part)
    }
    if (!cursors) {   // no ring block (P3D_PARTS_RING=0, or every block of the ring is held): the workspace header's
        cursors = hdr + H_CURSORS;
        if (new_block)   // (an ordinary kernel: the runtime's fill path starts late, see fused_stack_impl)
            hipLaunchKernelGGL(k_zero_words, dim3((kCursorBlockWords + kBlock - 1) / kBlock), dim3(kBlock), 0, st, cursors,
                               (int64_t)kCursorBlockWords);
    }
    if (part < 4) {
        if (part != 1) g_counters[2].fetch_add(1, std::memory_order_relaxed);
        if (lay) g_counters[1].fetch_add(1, std::memory_order_relaxed);
        const int stage = part == 1 ? ST_FUSED_INTERIOR : ST_FUSED;
        const bool timed = g_prof_mode != 0;
        if (timed) g_ev_used[stage] = true;
        if (new_block && part != 1) g_ev_used[ST_FUSED_INTERIOR] = false;
        const int64_t xo = slab ? slab->x_origin : 0;
        dispatch_fused<T>(grid, d, thresh, halo, t, xo, bits, rec, cursors, zero_next, lay ? verts : scratch, region_rows, store_rows,
                          x_lo, x_hi, timed ? g_ev[stage][0] : nullptr, timed ? g_ev[stage][1] : nullptr, st, nullptr, lay,
                          lay ? hdr + H_LAYOUT : nullptr);
        // (a launch that was refused never ran: the ring stays where it was)
        HIP_TRY(hipGetLastError());
        const bool holds = held && lease.ring && part == 1;   // the parts that continue the streaming find the block through the table
        lease.commit(holds);
        if (holds) {
            held->cursors = cursors;
            held->ring = lease.ring;
            held->ring_base = lease.ring->base;
            held->slot = lease.take;
        } else if (continues && held) {
            // the block goes back to its ring: stream order protects it until this launch is done (the lock is still ours)
            if (held->ring->base == held->ring_base) held->ring->held &= ~(1u << held->slot);
            *held = HeldBlock();
        }
    }
    if (part == 3) {
        // (p3d_mc_slab.export_first_plane_to: the dense records of plane 0 for the previous rank, in the same launch)
        uint2* const exp = (uint2*)slab->export_first_plane_to;
        hipLaunchKernelGGL(k_early_header, dim3(exp ? (u32)((d.P + kBlock - 1) / kBlock) : 1u), dim3(exp ? kBlock : 64), 0, st, hdr,
                           cursors, scratch ? store_rows : region_rows, id_limit, (const uint2*)rec, (int64_t)0, d.P, exp);
    }
    if (part == 1 || part == 3) {  // a later call finalizes
        HIP_TRY(hipGetLastError());
        return P3D_OK;
    }
    // What follows the streaming kernel, on the caller's stream: k_face_count_walk -> k_faces.  The first blocks of
    // the k_faces launch copy the vertex regions to their dense place and finish the header (V, F, flags, region
    // prefixes); the records stay in region form and k_faces makes them dense on the fly.  Without a face buffer, or
    // with a halo plane (its records arrive later: the faces are then written by p3d_mc_emit), the launch consists of
    // those first blocks only.
    const bool faces_here = w.nb_f > 0 && capf > 0 && (!halo || part >= 5);
    if (lay) {
        // Region-layout mode: the vertices are where they stay (but for the rows at or beyond V); the counting launch's 32
        // riding blocks make V, the flags and the tail tables and move those rows; the face launch's one extra block reports.
        u64 seq_l = 0;
        u64* mb_l = mailbox_open(ws, &seq_l);
        if (w.nchunks > 0 || true) {
            StageTimer tm(ST_FACES_COUNT, st);
            CompactArgs cpe{nullptr, verts, capv, 0, region_rows, kRegions, 0, 1, 0, csum, (int)w.nchunks, cursors, id_limit, 1, nullptr};
            cpe.layout = 1;
            launch_count_walk(dim3((u32)(w.nchunks + cpe.nblocks)), st, bits, d, w, csum, woff, (u32*)(ws + w.tile_tris), cpe, hdr);
            if (cpre) hipLaunchKernelGGL(k_chunk_prefix, dim3(1), dim3(1024), 0, st, csum, (int)w.nchunks, cpre);
        }
        FaceArgs a{3, 0, 0, 0, nullptr, 0, 1, w.tpp, w.xw, (int)w.cpi, csum, cpre, woff, (const u32*)(ws + w.tile_tris),
                   cursors, mb_l, seq_l};
        a.lay_verts = verts;
        CompactArgs cp{nullptr, verts, capv, 0, region_rows, 1, 0, 1, 1, csum, (int)w.nchunks, cursors, id_limit, 1, nullptr};
        cp.layout = 1;
        StageTimer tm(ST_EMIT_FACES, st);
        launch_faces(d, w, bits, rec, a, cp, hdr, faces, capf, faces_here, st);
        HIP_TRY(hipGetLastError());
        return P3D_OK;
    }
    u64 seq = 0;
    // (p3d_mc_slab.defer_totals: part 4 reports nothing, the part 5 behind it reports V and F from its first block)
    const bool defer = slab && slab->defer_totals != 0 && (part == 4 || part == 5);
    u64* mb = (part >= 5 && !(defer && part == 5)) || (defer && part == 4) ? nullptr : mailbox_open(ws, &seq);
    // the copy of the vertex regions is split over the two launches: `early` of `nparts` slices of every region ride
    // with the counting kernel (VALU-bound, HBM idle), the rest with k_faces
    const bool copy = scratch && capv > 0;
    const int nparts = copy ? std::max(1, tuning().compact_blocks / kRegions) : 1;
    const int early = (copy && w.nchunks > 0 && part != 6) ? std::min(nparts - 1, tuning().compact_early) : 0;
    if (w.nchunks > 0 && part < 5) {
        StageTimer tm(ST_FACES_COUNT, st);
        const CompactArgs cpe{copy ? scratch : nullptr, verts, capv, store_rows, region_rows, early * kRegions, 0, nparts, 0,
                              csum, (int)w.nchunks, cursors, id_limit, 1, nullptr};
        launch_count_walk(dim3((u32)(w.nchunks + cpe.nblocks)), st, bits, d, w, csum, woff, (u32*)(ws + w.tile_tris), cpe, hdr);
        if (cpre) hipLaunchKernelGGL(k_chunk_prefix, dim3(1), dim3(1024), 0, st, csum, (int)w.nchunks, cpre);
    }
    if (part == 4) {  // totals to the host now; the faces (and the rest of the vertex copy) follow in part 5
        if (!defer) {
            StageTimer tm(ST_SCAN_F, st);
            hipLaunchKernelGGL(k_face_total, dim3(1), dim3(kBlock), 0, st, csum, (int)w.nchunks, hdr, mb, seq, 1);
        }
        HIP_TRY(hipGetLastError());
        return P3D_OK;
    }
    if (part == 6) g_counters[4].fetch_add(1, std::memory_order_relaxed);
    const FaceArgs a{1, halo, slab ? slab->vertex_id_base : 0, slab ? slab->halo_vertex_id_base : 0,
                     slab ? slab->rank_counts : nullptr, slab ? slab->rank : 0,
                         (slab && slab->rank_counts_stride > 0) ? slab->rank_counts_stride : 1, w.tpp, w.xw, (int)w.cpi,
                     csum, cpre, woff, (const u32*)(ws + w.tile_tris), cursors, mb, seq};
    const CompactArgs cp{copy ? scratch : nullptr, verts, capv, store_rows, region_rows, (nparts - early) * kRegions, early,
                         nparts, (part >= 5 && !(defer && part == 5)) ? 0 : 1, csum, (int)w.nchunks, cursors, id_limit, 1, nullptr};
    StageTimer tm(ST_EMIT_FACES, st);
    launch_faces(d, w, bits, rec, a, cp, hdr, faces, capf, faces_here, st);
    HIP_TRY(hipGetLastError());
    return P3D_OK;
}

// A batch of B grids of one shape, back to back in memory, as ONE stack of B * rx planes: one streaming launch, one
// counting launch, one face launch for the whole batch (a 256^3 grid alone cannot fill the chip: launch ramp and tail
// dominate its three kernels).  Every item fills its own 32 vertex regions, so its vertices are contiguous in the
// output and its face indices are local to it; the item offsets are written to device memory by the face launch.
template <typename T>
int fused_stack_impl(const T* grids, const Dims& d, const Ws& w, float thresh, const Xform& t, char* ws, float* verts,
                     int64_t capv, float* scratch, int64_t scratch_rows, int32_t* faces, int64_t capf,
                     int64_t* item_offsets, hipStream_t st) {
    u64* hdr = (u64*)(ws + w.hdr);
    u64* bits = (u64*)(ws + w.bits);
    uint2* rec = (uint2*)(ws + w.rec);
    u32 *csum = (u32*)(ws + w.chunk_sum), *woff = (u32*)(ws + w.wave_off);
    u64* cursors = (u64*)(ws + w.cur);
    // (P3D_NO_CHUNK_PRE=1: escape hatch and test reference -- every face tile adds the chunk totals up itself)
    u32* cpre = (w.nchunks > kPreMinChunks && !tuning().no_chunk_pre) ? (u32*)(ws + w.chunk_pre) : nullptr;   // (see fused_impl)
    // (an ordinary kernel, not hipMemsetAsync: the runtime's fill path left the GPU idle for 11 us before it ran)
    {
        const int64_t nwords = (int64_t)d.nitems * kCursorBlockWords;
        hipLaunchKernelGGL(k_zero_words, dim3((u32)((nwords + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, cursors, nwords);
    }
    const u32 region_rows = 1u << 26;
    const u32 id_limit = (u32)std::min<int64_t>(region_rows, std::max(1, tuning().test_id_limit));
    const u32 store_rows =
        scratch ? (u32)std::min<int64_t>(scratch_rows / ((int64_t)kRegions * d.nitems), (int64_t)region_rows) : 0u;
    const bool timed = g_prof_mode != 0;
    g_counters[2].fetch_add(1, std::memory_order_relaxed);
    if (timed) g_ev_used[ST_FUSED] = true;
    g_ev_used[ST_FUSED_INTERIOR] = false;
    dispatch_fused<T>(grids, d, thresh, 0, t, 0, bits, rec, cursors, nullptr, scratch, region_rows, store_rows, 0, (int)d.rx,
                      timed ? g_ev[ST_FUSED][0] : nullptr, timed ? g_ev[ST_FUSED][1] : nullptr, st);
    u64 seq = 0;
    u64* mb = mailbox_open(ws, &seq);
    const bool copy = scratch && capv > 0;
    // compaction blocks: every (item, region) is copied in `nparts` slices, `early` of them by blocks riding in the
    // counting launch (VALU-bound, HBM idle), the rest with the faces -- about two thousand blocks for the stack
    int nparts = copy ? std::max(2, std::min(8, 2048 / (kRegions * d.nitems))) : 1;
    if (copy && tuning().stack_nparts > 0) nparts = tuning().stack_nparts;
    int early = (copy && w.nchunks > 0) ? std::max(1, nparts * 3 / 8) : 0;
    if (copy && w.nchunks > 0 && tuning().stack_early >= 0) early = std::min(nparts - 1, tuning().stack_early);
    if (w.nchunks > 0) {
        StageTimer tm(ST_FACES_COUNT, st);
        const CompactArgs cpe{copy ? scratch : nullptr, verts, capv, store_rows, region_rows, early * kRegions * d.nitems, 0,
                              nparts, 0, csum, (int)w.nchunks, cursors, id_limit, d.nitems, item_offsets};
        launch_count_walk(dim3((u32)(w.nchunks + cpe.nblocks)), st, bits, d, w, csum, woff, (u32*)(ws + w.tile_tris), cpe, hdr);
    }
    const FaceArgs a{1, 0, 0, 0, nullptr, 0, 1, w.tpp, w.xw, (int)w.cpi, csum, cpre, woff,
                     (const u32*)(ws + w.tile_tris), cursors, nullptr, 0};
    const CompactArgs cp{copy ? scratch : nullptr, verts, capv, store_rows, region_rows,
                         (nparts - early) * kRegions * d.nitems, early, nparts, 0, csum, (int)w.nchunks, cursors, id_limit,
                         d.nitems, item_offsets};
    // per-item offsets and the totals for the host BEFORE the faces: everything they need (region cursors, chunk sums) is
    // final after the counting launch, and the host then sizes its tensors and queues the next call while the faces are
    // still being written (40 us of idle GPU per batch when this ran last)
    hipLaunchKernelGGL(k_stack_finish, dim3(1), dim3(1024), 0, st, cursors, csum, (int)w.nchunks, d.nitems,
                       scratch ? store_rows : region_rows, id_limit, item_offsets, hdr, mb, seq, cpre);
    {
        StageTimer tm(ST_EMIT_FACES, st);
        launch_faces(d, w, bits, rec, a, cp, hdr, faces, capf, w.nb_f > 0 && capf > 0, st);
    }
    HIP_TRY(hipGetLastError());
    return P3D_OK;
}

// ---- the order of an extraction's calls (p3d_mc_slab.part) ----------------------------------------------------------------
// The reference's boundary is stateless (marching_cubes.h:14-15); an extraction made of several calls is not, and a part
// that runs on a workspace its predecessors never prepared reads whatever the header holds.  The library therefore keeps,
// per (device, workspace pointer), which part it saw last and with which stream / shape / scratch / vertex buffer, and
// refuses every successor the diagram in INTEGRATION.md does not have, synchronously, with P3D_EINVAL and a message.
// Host-side only (a phase word on the device could not fail a call without a synchronisation).  A starting part (0, 1, 3
// with split_plane 0, p3d_mc_count, the batched entry) is always legal and resets the entry, so a workspace address that
// an allocator hands out again never inherits a verdict; the table holds the kMaxExtractions most recently used workspaces.
enum Phase { PH_NONE = 0, PH_INTERIOR /* after part 1 */, PH_STREAMED /* after part 3 */, PH_COUNTED /* after part 4 */,
             PH_DONE /* after part 0, 2, 5, 6 */, PH_COUNT_CALL /* after p3d_mc_count */, PH_STACK /* the batched entry */,
             PH_COUNT_SCAN /* after p3d_mc_count_scan */ };
const char* const k_phase_names[] = {"nothing", "part 1", "part 3", "part 4", "a finished extraction (part 0, 2, 5 or 6)",
                                     "p3d_mc_count", "p3d_mc_extract_fused_batched", "p3d_mc_count_scan"};
struct Extraction {
    int phase = PH_NONE;
    hipStream_t stream = nullptr;
    int64_t rx = 0, ry = 0, rz = 0, split = 0;
    int dtype = 0;
    const float* scratch = nullptr;
    int64_t scratch_rows = 0;
    const float* verts4 = nullptr;   // the vertex buffer part 4 began to fill (null: it copied nothing)
    int64_t capv4 = 0;
    HeldBlock held;                  // part 1's cursor block, for the parts that continue the streaming
    bool layout = false;             // the extraction was a whole-grid call with a predicted region layout (rec[] holds rows)
    uint64_t last_use = 0;
};
constexpr size_t kMaxExtractions = 1024;
std::mutex g_proto_mu;
std::map<std::pair<int, const void*>, Extraction> g_proto;
uint64_t g_proto_clock = 0;

struct ProtoCall {   // what one call presents
    const void* ws;
    int part;        // 0..6, or -1 p3d_mc_count, -2 p3d_mc_emit, -3 the batched entry, -4 p3d_mc_count_scan
    hipStream_t stream;
    int64_t rx, ry, rz, split;
    int dtype;
    const float* scratch;
    int64_t scratch_rows;
    const float* verts;
    int64_t capv;
    bool layout = false;
};

int proto_fail(const char* what, int last_phase) {
    snprintf(g_err, sizeof(g_err), "illegal order of calls on this workspace: %s (last seen on it: %s)", what,
             k_phase_names[last_phase]);
    return P3D_EINVAL;
}

// validates the call against the workspace's entry; *out = the entry as it will be when the call has succeeded
int proto_check(const ProtoCall& c, Extraction* out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    Extraction e;
    {
        std::lock_guard<std::mutex> g(g_proto_mu);
        auto it = g_proto.find({dev, c.ws});
        if (it != g_proto.end()) e = it->second;
    }
    const int last = e.phase;
    const bool start = c.part == 0 || c.part == 1 || (c.part == 3 && c.split == 0) || c.part == -1 || c.part == -3 || c.part == -4;
    if (!start) {
        // every continuation: the same stream and grid shape as the part that started the extraction
        if (last == PH_NONE || last == PH_STACK) {
            if (c.part == -2) return proto_fail("p3d_mc_emit needs a p3d_mc_count or a p3d_mc_extract_fused before it", last);
            return proto_fail("this part continues an extraction, and none is in progress", last);
        }
        if (c.stream != e.stream) return proto_fail("all parts of one extraction must be given the same stream", last);
        if (c.rx != e.rx || c.ry != e.ry || c.rz != e.rz || c.dtype != e.dtype)
            return proto_fail("all parts of one extraction must be given the same grid shape and dtype", last);
    }
    const bool same_scratch = c.scratch == e.scratch && c.scratch_rows == e.scratch_rows;
    switch (c.part) {
        case 0: case 1: case -1: case -3: case -4: break;
        case 2:
            if (last != PH_INTERIOR) return proto_fail("part 2 needs part 1 before it", last);
            if (c.split != e.split) return proto_fail("part 2 must continue at part 1's split_plane", last);
            if (!same_scratch) return proto_fail("part 2 must be given part 1's scratch buffer", last);
            break;
        case 3:
            if (c.split != 0) {
                if (last != PH_INTERIOR) return proto_fail("part 3 with a split_plane needs part 1 before it", last);
                if (c.split != e.split) return proto_fail("part 3 must continue at part 1's split_plane", last);
                if (!same_scratch) return proto_fail("part 3 must be given part 1's scratch buffer", last);
            }
            break;
        case 4:
            if (last != PH_STREAMED) return proto_fail("part 4 needs part 3 before it", last);
            if (!same_scratch) return proto_fail("part 4 must be given part 3's scratch buffer", last);
            break;
        case 5:
            if (last != PH_COUNTED) return proto_fail("part 5 needs part 4 before it", last);
            if (!same_scratch) return proto_fail("part 5 must be given part 4's scratch buffer", last);
            if (e.verts4 ? (c.verts != e.verts4 || c.capv != e.capv4) : c.capv != 0)
                return proto_fail(e.verts4 ? "part 5 must be given the vertex buffer part 4 began to fill"
                                           : "part 4 was given no vertex buffer: part 6 writes the vertices, part 5 cannot", last);
            break;
        case 6:
            if (last != PH_COUNTED && last != PH_DONE) return proto_fail("part 6 needs part 4, or a finished extraction, before it", last);
            if (e.layout) return proto_fail("part 6 copies out of a scratch buffer: the extraction before it had none (region layout)", last);
            if (!same_scratch) return proto_fail("part 6 must be given the scratch buffer the field was streamed into", last);
            break;
        case -2:
            if (last != PH_COUNT_CALL && last != PH_COUNT_SCAN && last != PH_COUNTED && last != PH_DONE)
                return proto_fail("p3d_mc_emit needs finished counts (p3d_mc_count, p3d_mc_count_scan, part 4 or a whole extraction) before it", last);
            break;
        default: return fail(P3D_EINVAL, "bad slab part%s");
    }
    if (start) {
        release_held(&e.held);   // (an extraction abandoned after its part 1: its cursor block goes back to the ring)
        e = Extraction();
        e.stream = c.stream;
        e.rx = c.rx; e.ry = c.ry; e.rz = c.rz; e.dtype = c.dtype; e.split = c.split;
        e.scratch = c.scratch; e.scratch_rows = c.scratch_rows;
        e.layout = c.layout;
    }
    switch (c.part) {
        case 0: case 2: case 5: case 6: e.phase = PH_DONE; break;
        case 1: e.phase = PH_INTERIOR; break;
        case 3: e.phase = PH_STREAMED; break;
        case 4:
            e.phase = PH_COUNTED;
            e.verts4 = (c.scratch && c.capv > 0) ? c.verts : nullptr;
            e.capv4 = e.verts4 ? c.capv : 0;
            break;
        case -1: e.phase = PH_COUNT_CALL; break;
        case -3: e.phase = PH_STACK; break;
        case -4: e.phase = PH_COUNT_SCAN; break;
        default: break;   // (p3d_mc_emit leaves the phase where it is)
    }
    *out = e;
    return P3D_OK;
}

void proto_commit(const void* ws, const Extraction& e_in) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> g(g_proto_mu);
    if (g_proto.size() >= kMaxExtractions && g_proto.find({dev, ws}) == g_proto.end()) {
        // full: the older half goes in one sweep (amortised O(1) per call; entries are stamped with a running clock)
        const uint64_t keep_from = g_proto_clock > kMaxExtractions / 2 ? g_proto_clock - kMaxExtractions / 2 : 0;
        for (auto it = g_proto.begin(); it != g_proto.end();) {
            if (it->second.last_use < keep_from) {
                release_held(&it->second.held);
                it = g_proto.erase(it);
            } else {
                ++it;
            }
        }
    }
    Extraction e = e_in;
    e.last_use = ++g_proto_clock;
    g_proto[{dev, ws}] = e;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int p3d_mc_abi_version(void) { return P3D_MC_ABI_VERSION; }
const char* p3d_last_error(void) { return g_err; }

int p3d_mc_workspace_bytes(int64_t rx, int64_t ry, int64_t rz, size_t* bytes) {
    if (!bytes) return fail(P3D_EINVAL, "bytes is null%s");
    if (int rc = check_dims(rx, ry, rz)) return rc;
    *bytes = make_ws(make_dims(rx, ry, rz)).total;
    return P3D_OK;
}

int p3d_mc_count(const void* grid, int dtype, int64_t rx, int64_t ry, int64_t rz, float thresh,
                 const p3d_mc_slab* slab, void* ws, void* stream) {
    if (!grid || !ws) return fail(P3D_EINVAL, "null pointer%s");
    if (int rc = check_dims(rx, ry, rz)) return rc;
    const Dims d = make_dims(rx, ry, rz);
    const Ws w = make_ws(d);
    hipStream_t st = (hipStream_t)stream;
    if (dtype != P3D_F32 && dtype != P3D_F16) return fail(P3D_EINVAL, "unknown dtype%s");
    Extraction next;
    if (int rc = proto_check(ProtoCall{ws, -1, st, rx, ry, rz, 0, dtype, nullptr, 0, nullptr, 0}, &next)) return rc;
    // the one-pass kernels in count-only form: the streaming kernel without a vertex buffer (sign words, records, the 32
    // region totals) + the face count; the finishing block leaves V, F, the flags and the region prefixes in the header
    p3d_mc_slab whole{};
    if (slab) whole = *slab;
    whole.part = 0;
    const float unit_lo[3] = {0.f, 0.f, 0.f}, unit_hi[3] = {1.f, 1.f, 1.f};
    const Xform t = make_xform(d, unit_lo, unit_hi, nullptr);   // (no vertex is stored: the box does not matter)
    g_counters[3].fetch_add(1, std::memory_order_relaxed);
    const int rc = dtype == P3D_F32
                       ? fused_impl((const float*)grid, d, w, thresh, t, &whole, (char*)ws, nullptr, 0, nullptr, 0, nullptr, 0, st, &next.held)
                       : fused_impl((const __half*)grid, d, w, thresh, t, &whole, (char*)ws, nullptr, 0, nullptr, 0, nullptr, 0, st, &next.held);
    if (rc == P3D_OK) proto_commit(ws, next);
    return rc;
}

int p3d_mc_count_scan(const void* grid, int dtype, int64_t rx, int64_t ry, int64_t rz, float thresh,
                      const p3d_mc_slab* slab, void* ws, void* stream) {
    if (!grid || !ws) return fail(P3D_EINVAL, "null pointer%s");
    if (int rc = check_dims(rx, ry, rz)) return rc;
    const Dims d = make_dims(rx, ry, rz);
    const Ws w = make_ws(d);
    hipStream_t st = (hipStream_t)stream;
    if (dtype != P3D_F32 && dtype != P3D_F16) return fail(P3D_EINVAL, "unknown dtype%s");
    Extraction next;
    if (int rc = proto_check(ProtoCall{ws, -4, st, rx, ry, rz, 0, dtype, nullptr, 0, nullptr, 0}, &next)) return rc;
    const int rc = dtype == P3D_F32 ? count_impl((const float*)grid, d, w, thresh, slab, (char*)ws, st)
                                    : count_impl((const __half*)grid, d, w, thresh, slab, (char*)ws, st);
    if (rc == P3D_OK) proto_commit(ws, next);
    return rc;
}

int p3d_mc_read_counts(const void* ws, int64_t* num_vertices, int64_t* num_faces, int32_t* scratch_overflow,
                       void* stream) {
    return p3d_mc_read_counts_ex(ws, num_vertices, num_faces, scratch_overflow, nullptr, stream);
}

int p3d_mc_read_counts_ex(const void* ws, int64_t* num_vertices, int64_t* num_faces, int32_t* scratch_overflow,
                          int64_t* region_totals, void* stream) {
    if (!ws || !num_vertices || !num_faces) return fail(P3D_EINVAL, "null pointer%s");
    u64 h[3] = {0, 0, 0};
    hipStream_t st = (hipStream_t)stream;
    if (!mailbox_wait(ws, &h[H_V], &h[H_T], &h[H_FLAGS], region_totals)) {
        // (no mailbox: the header -- totals, and the region totals the finishing block copies to the header's cursor lines)
        static_assert(H_CURSORS + kRegions * kCursorStride <= kHdrBytes / 8, "header");
        std::vector<u64> hdr_copy(H_CURSORS + (size_t)kRegions * kCursorStride);
        HIP_TRY(hipMemcpyAsync(hdr_copy.data(), ws, hdr_copy.size() * sizeof(u64), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int i = 0; i < 3; ++i) h[i] = hdr_copy[i];
        if (region_totals)
            for (int r = 0; r < kRegions; ++r) region_totals[r] = (int64_t)hdr_copy[H_CURSORS + (size_t)r * kCursorStride];
    }
    *num_vertices = (int64_t)h[H_V];
    *num_faces = (int64_t)h[H_T];
    if (scratch_overflow) *scratch_overflow = (int32_t)(h[H_FLAGS] & 7ull);
    // (test hook: P3D_TEST_INDEX_LIMIT pretends the int32 limit is smaller, to reach the callers' handling of it)
    const u64 index_limit = (u64)std::max(1, tuning().test_index_limit);
    if (h[H_V] > index_limit || h[H_T] > index_limit)
        return fail(P3D_ERANGE, "vertex/face count exceeds int32 indexing%s");
    return P3D_OK;
}

int p3d_mc_emit(const void* grid, int dtype, int64_t rx, int64_t ry, int64_t rz, float thresh, const float lower[3],
                const float upper[3], const int64_t full_res[3], const p3d_mc_slab* slab, void* ws, float* vertices,
                int64_t cap_vertices, int32_t* faces, int64_t cap_faces, int64_t* vertex_keys, void* stream) {
    if (!grid || !ws || !lower || !upper) return fail(P3D_EINVAL, "null pointer%s");
    if ((cap_vertices > 0 && !vertices) || (cap_faces > 0 && !faces)) return fail(P3D_EINVAL, "null output%s");
    if (slab && slab->rank_counts && (slab->rank < 0 || slab->rank >= 64))
        return fail(P3D_EINVAL, "rank_counts serves ranks 0..63 (one lane of a wave per rank)%s");
    if (int rc = check_dims(rx, ry, rz)) return rc;
    const Dims d = make_dims(rx, ry, rz);
    const Ws w = make_ws(d);
    const Xform t = make_xform(d, lower, upper, full_res);
    hipStream_t st = (hipStream_t)stream;
    if (dtype != P3D_F32 && dtype != P3D_F16) return fail(P3D_EINVAL, "unknown dtype%s");
    Extraction next;
    if (int rc = proto_check(ProtoCall{ws, -2, st, rx, ry, rz, 0, dtype, nullptr, 0, nullptr, 0}, &next)) return rc;
    // A whole grid whose vertices AND faces are (re)written, behind counts made by the one-pass kernels: a second streaming
    // pass into the exactly sized buffers.  Ids handed out before must survive in every other case -- a slab (its first
    // plane's records may already be with the neighbour), vertices or faces alone, the keys of the parity tests, ids
    // renumbered by p3d_mc_count_scan -- and there the gather emitter writes by id.
    if (next.layout && !(cap_vertices > 0 && cap_faces > 0))
        return fail(P3D_EINVAL, "behind a call with a region layout p3d_mc_emit takes both buffers (it streams the field again)%s");
    if (next.layout && (slab || vertex_keys))
        return fail(P3D_EINVAL, "behind a call with a region layout p3d_mc_emit takes neither a slab nor vertex_keys%s");
    if (!slab && !vertex_keys && cap_vertices > 0 && cap_faces > 0 && next.phase != PH_COUNT_SCAN) {
        const int rc = dtype == P3D_F32
                           ? emit_stream_impl((const float*)grid, d, w, thresh, t, (char*)ws, vertices, cap_vertices, faces, cap_faces, st)
                           : emit_stream_impl((const __half*)grid, d, w, thresh, t, (char*)ws, vertices, cap_vertices, faces, cap_faces, st);
        if (rc == P3D_OK && next.layout) {   // (the records are in region form again: a later faces-only emit reads them so)
            next.layout = false;
            proto_commit(ws, next);
        }
        return rc;
    }
    if (dtype == P3D_F32)
        return emit_impl((const float*)grid, d, w, thresh, t, slab, (char*)ws, vertices, cap_vertices, faces,
                         cap_faces, vertex_keys, st);
    return emit_impl((const __half*)grid, d, w, thresh, t, slab, (char*)ws, vertices, cap_vertices, faces,
                     cap_faces, vertex_keys, st);
}

int p3d_mc_extract_fused(const void* grid, int dtype, int64_t rx, int64_t ry, int64_t rz, float thresh,
                         const float lower[3], const float upper[3], const int64_t full_res[3],
                         const p3d_mc_slab* slab, void* ws, float* vertices, int64_t cap_vertices,
                         float* vertex_scratch, int64_t scratch_rows, int32_t* faces, int64_t cap_faces,
                         void* stream) {
    if (!grid || !ws || !lower || !upper) return fail(P3D_EINVAL, "null pointer%s");
    if ((cap_vertices > 0 && !vertices) || (cap_faces > 0 && !faces)) return fail(P3D_EINVAL, "null output%s");
    if (cap_vertices < 0 || cap_faces < 0 || scratch_rows < 0) return fail(P3D_EINVAL, "negative capacity%s");
    const bool has_layout = slab && slab->region_first_rows;
    if (has_layout) {   // (the regions live in `vertices`: a scratch buffer, if given, is not used)
        vertex_scratch = nullptr;
        scratch_rows = 0;
    }
    if (cap_vertices > 0 && !has_layout && (!vertex_scratch || scratch_rows < kRegions))
        return fail(P3D_EINVAL, "vertex output needs a scratch buffer of at least 32 rows (or a region layout)%s");
    if (int rc = check_dims(rx, ry, rz)) return rc;
    if (slab && (slab->part < 0 || slab->part > 6)) return fail(P3D_EINVAL, "bad slab part%s");
    if (slab && slab->rank_counts && (slab->rank < 0 || slab->rank >= 64))
        return fail(P3D_EINVAL, "rank_counts serves ranks 0..63 (one lane of a wave per rank)%s");
    if (slab && (slab->part == 1 || slab->part == 2) && (slab->split_plane < 1 || slab->split_plane >= rx))
        return fail(P3D_EINVAL, "bad split_plane%s");
    if (slab && slab->part == 3 && (slab->split_plane < 0 || slab->split_plane >= rx))
        return fail(P3D_EINVAL, "bad split_plane%s");
    const Dims d = make_dims(rx, ry, rz);
    const Ws w = make_ws(d);
    const Xform t = make_xform(d, lower, upper, full_res);
    hipStream_t st = (hipStream_t)stream;
    if (dtype != P3D_F32 && dtype != P3D_F16) return fail(P3D_EINVAL, "unknown dtype%s");
    Extraction next;
    if (int rc = proto_check(ProtoCall{ws, slab ? slab->part : 0, st, rx, ry, rz, slab ? slab->split_plane : 0, dtype,
                                       vertex_scratch, vertex_scratch ? scratch_rows : 0, vertices, cap_vertices, has_layout}, &next))
        return rc;
    const int rc = dtype == P3D_F32
                       ? fused_impl((const float*)grid, d, w, thresh, t, slab, (char*)ws, vertices, cap_vertices, vertex_scratch,
                                    scratch_rows, faces, cap_faces, st, &next.held)
                       : fused_impl((const __half*)grid, d, w, thresh, t, slab, (char*)ws, vertices, cap_vertices,
                                    vertex_scratch, scratch_rows, faces, cap_faces, st, &next.held);
    if (rc == P3D_OK) proto_commit(ws, next);
    return rc;
}

int p3d_mc_workspace_bytes_batched(int64_t nitems, int64_t rx, int64_t ry, int64_t rz, size_t* bytes) {
    if (!bytes) return fail(P3D_EINVAL, "bytes is null%s");
    if (nitems < 1 || nitems > 65535) return fail(P3D_EINVAL, "nitems must be in [1, 65535]%s");
    if (int rc = check_dims(nitems * rx, ry, rz)) return rc;
    *bytes = make_ws(make_dims_stack(nitems, rx, ry, rz)).total;
    return P3D_OK;
}

int p3d_mc_extract_fused_batched(const void* grids, int dtype, int64_t nitems, int64_t rx, int64_t ry, int64_t rz,
                                 float thresh, const float lower[3], const float upper[3], void* ws, float* vertices,
                                 int64_t cap_vertices, float* vertex_scratch, int64_t scratch_rows, int32_t* faces,
                                 int64_t cap_faces, int64_t* item_offsets, void* stream) {
    if (!grids || !ws || !lower || !upper || !item_offsets) return fail(P3D_EINVAL, "null pointer%s");
    if ((cap_vertices > 0 && !vertices) || (cap_faces > 0 && !faces)) return fail(P3D_EINVAL, "null output%s");
    if (cap_vertices < 0 || cap_faces < 0 || scratch_rows < 0) return fail(P3D_EINVAL, "negative capacity%s");
    if (nitems < 1 || nitems > 65535) return fail(P3D_EINVAL, "nitems must be in [1, 65535]%s");
    if (rx < 1) return fail(P3D_EINVAL, "grid dims must be >= 1%s");
    if (cap_vertices > 0 && (!vertex_scratch || scratch_rows < kRegions * nitems))
        return fail(P3D_EINVAL, "vertex output needs a scratch buffer of at least 32 rows per item%s");
    if (int rc = check_dims(nitems * rx, ry, rz)) return rc;
    const Dims d = make_dims_stack(nitems, rx, ry, rz);
    const Ws w = make_ws(d);
    const Dims d1 = make_dims(rx, ry, rz);   // the box transform is the single grid's (marching_cubes.cu:293-297)
    const Xform t = make_xform(d1, lower, upper, nullptr);
    hipStream_t st = (hipStream_t)stream;
    if (dtype != P3D_F32 && dtype != P3D_F16) return fail(P3D_EINVAL, "unknown dtype%s");
    Extraction next;
    if (int rc = proto_check(ProtoCall{ws, -3, st, nitems * rx, ry, rz, 0, dtype, nullptr, 0, nullptr, 0}, &next)) return rc;
    const int rc = dtype == P3D_F32
                       ? fused_stack_impl((const float*)grids, d, w, thresh, t, (char*)ws, vertices, cap_vertices, vertex_scratch,
                                          scratch_rows, faces, cap_faces, item_offsets, st)
                       : fused_stack_impl((const __half*)grids, d, w, thresh, t, (char*)ws, vertices, cap_vertices,
                                          vertex_scratch, scratch_rows, faces, cap_faces, item_offsets, st);
    if (rc == P3D_OK) proto_commit(ws, next);
    return rc;
}

int p3d_mc_debug_layout(int64_t rx, int64_t ry, int64_t rz, size_t* off_bits, size_t* off_records,
                        int64_t* num_units, int32_t* chunks_per_row) {
    if (!off_bits || !off_records || !num_units || !chunks_per_row) return fail(P3D_EINVAL, "null pointer%s");
    if (int rc = check_dims(rx, ry, rz)) return rc;
    const Dims d = make_dims(rx, ry, rz);
    const Ws w = make_ws(d);
    *off_bits = w.bits;
    *off_records = w.rec;
    *num_units = d.U;
    *chunks_per_row = d.ncz;
    return P3D_OK;
}

#if P3D_COUNT_STAMP
int p3d_mc_debug_count_stamps(void* buf) {   // dev build only: where k_face_count_walk leaves its per-wave phase stamps
    u64* p = (u64*)buf;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_count_stamps), &p, sizeof(p)));
    return P3D_OK;
}
#endif
#if P3D_FACES_STAMP
int p3d_mc_debug_face_stamps(void* buf) {   // dev build only: where k_faces leaves its per-wave phase stamps
    u64* p = (u64*)buf;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_face_stamps), &p, sizeof(p)));
    return P3D_OK;
}
#endif

int p3d_mc_dev_hooks(void) { return P3D_DEV_HOOKS; }

int p3d_mc_reload_tuning(void) {   // re-read the P3D_* knobs (not for use beside running calls)
    (void)tuning();
    apply_tuning(read_tuning());
    return P3D_OK;
}

int p3d_mc_release_stream(void* stream) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::shared_ptr<CursorRing> r;
    {
        std::lock_guard<std::mutex> g(g_ring_mu);
        auto it = g_rings.find({dev, (hipStream_t)stream});
        if (it == g_rings.end()) return P3D_OK;   // nothing was ever kept for it
        r = it->second;
        g_rings.erase(it);
    }
    return free_ring(r, true);
}

int p3d_mc_shutdown(void) {
    int rc = P3D_OK;
    std::map<std::pair<int, hipStream_t>, std::shared_ptr<CursorRing>> rings;
    {
        std::lock_guard<std::mutex> g(g_ring_mu);
        rings.swap(g_rings);
    }
    for (auto& kv : rings)
        if (int e = free_ring(kv.second, false)) rc = e;   // (streams may be gone already: hipFree waits for the device)
    {
        std::lock_guard<std::mutex> g(g_mb_mu);
        for (auto& m : g_mb) {
            if (m.host && hipHostFree(m.host) != hipSuccess) rc = fail(P3D_EHIP, "hipHostFree(mailbox)%s");
            m = Mailbox();
        }
        for (auto& pc : g_pending) pc = PendingCall();
    }
    {
        std::lock_guard<std::mutex> g(g_proto_mu);
        g_proto.clear();
    }
    if (g_ev_made) {
        for (int i = 0; i < ST_N; ++i)
            for (int j = 0; j < 2; ++j) (void)hipEventDestroy(g_ev[i][j]);
        g_ev_made = false;
        g_prof_mode = 0;
    }
    return rc;
}

int p3d_mc_debug_counters(int64_t* out, int n) {
    if (!out || n < 0) return fail(P3D_EINVAL, "null pointer%s");
    const int m = std::min(n, 7);
    for (int i = 0; i < std::min(m, 5); ++i) out[i] = g_counters[i].load(std::memory_order_relaxed);
    if (m > 5) {
        std::lock_guard<std::mutex> g(g_ring_mu);
        out[5] = (int64_t)g_rings.size();
    }
    if (m > 6) out[6] = g_ring_bytes.load(std::memory_order_relaxed);
    return m;
}

int p3d_mc_profile_enable(int mode) {
    if (mode < 0 || mode > 2) return fail(P3D_EINVAL, "profile mode must be 0, 1 or 2%s");
    if (mode && !g_ev_made) {
        for (int i = 0; i < ST_N; ++i)
            for (int j = 0; j < 2; ++j) HIP_TRY(hipEventCreate(&g_ev[i][j]));
        g_ev_made = true;
    }
    for (int i = 0; i < ST_N; ++i) g_ev_used[i] = false;
    g_prof_mode = mode;
    return P3D_OK;
}

int p3d_mc_profile_read(float* stage_ms, int n) {
    if (!stage_ms || n < ST_N) return fail(P3D_EINVAL, "need room for 11 stages%s");
    for (int i = 0; i < ST_N; ++i) {
        stage_ms[i] = -1.f;
        if (g_ev_made && g_ev_used[i]) {
            HIP_TRY(hipEventSynchronize(g_ev[i][1]));
            HIP_TRY(hipEventElapsedTime(&stage_ms[i], g_ev[i][0], g_ev[i][1]));
        }
    }
    return ST_N;
}

const char* p3d_mc_profile_stage_name(int stage) { return (stage >= 0 && stage < ST_N) ? k_stage_names[stage] : ""; }

int p3d_mc_plane_records(void* ws, int64_t rx, int64_t ry, int64_t rz, int64_t plane, void** records,
                         size_t* bytes_per_plane) {
    if (!ws || !records || !bytes_per_plane) return fail(P3D_EINVAL, "null pointer%s");
    if (int rc = check_dims(rx, ry, rz)) return rc;
    if (plane < 0 || plane >= rx) return fail(P3D_EINVAL, "plane out of range%s");
    const Dims d = make_dims(rx, ry, rz);
    const Ws w = make_ws(d);
    *records = (char*)ws + w.rec + (size_t)plane * d.P * sizeof(uint2);
    *bytes_per_plane = (size_t)d.P * sizeof(uint2);
    return P3D_OK;
}

int p3d_mc_export_plane_records(const void* ws, int64_t rx, int64_t ry, int64_t rz, int64_t plane, void* out,
                                void* stream) {
    if (!ws || !out) return fail(P3D_EINVAL, "null pointer%s");
    if (int rc = check_dims(rx, ry, rz)) return rc;
    if (plane < 0 || plane >= rx) return fail(P3D_EINVAL, "plane out of range%s");
    const Dims d = make_dims(rx, ry, rz);
    const Ws w = make_ws(d);
    hipLaunchKernelGGL(k_export_plane_records, dim3((u32)((d.P + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, (const uint2*)((const char*)ws + w.rec), plane * d.P, d.P,
                       (const u64*)((const char*)ws + w.hdr), (uint2*)out);
    HIP_TRY(hipGetLastError());
    return P3D_OK;
}

}  // extern "C"
