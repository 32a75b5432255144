// p3d_rc.hip -- MI355X (gfx950) ray caster: BVH4 build on the host cores, traversal kernel on the GPU, C ABI
// (include/p3d_rc.h).  Replaces the reference's non-OptiX RayCaster: src/prim3d/Utility/ray_cast.cu:340-424,
// src/prim3d/Geometry/bvh.cu:146-346, triangle.h:12-33, bounding_box.h:157-210.
//
// Layout.  One 128-byte node holds the boxes of its FOUR children (SoA: minx[4] miny[4] minz[4] maxx[4] maxy[4] maxz[4])
// and their links, so one node fetch (two 64-byte lines) feeds four slab tests; the reference keeps one box per node and
// fetches four nodes (bvh.cu:176-180).  Triangles are stored in leaf order as three float4 (a|idx, b, c): 48 bytes.
// Traversal: lane = ray; the stack lives in LDS, one column per lane ([depth][lane]: conflict-free); children are visited
// near to far (a 4-element sorting network on the entry distances, as bvh.cu:182) and pruned against the best hit.
// The intersector and the normal are the reference's, operation for operation (triangle.h:16-33, :12-14), built with
// -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/p3d_rc.h"

namespace {

constexpr float kMaxDist = 10.0f;   // bvh.cu:13
constexpr int kLeafTris = 8;        // ray_cast.cu:381
constexpr int kStackMax = 48;       // 3 pushes per level + 1: the balanced build is 12 levels deep at 2^27 triangles
constexpr int kRcBlock = 64;

struct alignas(16) Node {           // 128 bytes
    float lo[3][4];                 // [axis][child]
    float hi[3][4];
    int32_t child[4];               // >= 0: node index; < 0: leaf, -(first * 16 + count) - 1; INT32_MIN: empty
    int32_t pad[4];
};
static_assert(sizeof(Node) == 128, "node = two 64-byte lines");

struct HostTri {
    float a[3], b[3], c[3];
    int32_t idx;
    float cen[3];
};

thread_local char g_err[512] = "";
int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}
#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return fail(P3D_RC_EHIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

// chunks [0, n) over up to 16 host threads (small n: the calling thread alone)
template <class F>
void parallel_chunks(size_t n, F&& fn) {
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nt = n < 65536 ? 1 : std::min<size_t>(16, std::max(1u, hw));
    if (nt <= 1) {
        fn((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (size_t t = 1; t < nt; ++t) th.emplace_back([&, t] { fn(std::min(n, t * per), std::min(n, (t + 1) * per)); });
    fn((size_t)0, std::min(n, per));
    for (auto& x : th) x.join();
}

// ---- host: 4-ary build (bvh.cu:209-300: two rounds of median splits on the axis of largest centroid variance) ----------
// The subtrees of the first two levels (up to 16) are built by their own host threads: the triangle ranges are disjoint,
// nodes come out of one preallocated array through an atomic counter (the reference builds on one thread).
struct Builder {
    std::vector<HostTri>& tris;
    std::unique_ptr<Node[]> nodes;     // preallocated (not zero-filled); `used` of them are taken
    size_t cap = 0;
    std::atomic<int32_t> used{0};
    std::atomic<int> max_depth{0};
    std::atomic<bool> overflow{false};
    int par_levels = 0;                // levels whose children are built concurrently
    explicit Builder(std::vector<HostTri>& t) : tris(t) {
        // inner nodes I(n) <= n / 3 - 1 for n >= 9 triangles (induction over the four children, each of which holds at
        // least two triangles); the balanced median splits produce about 0.15 n
        cap = t.size() / 3 + 16;
        nodes.reset(new Node[cap]);
        const unsigned hw = std::thread::hardware_concurrency();
        par_levels = (t.size() >= 65536 && hw >= 4) ? (hw >= 16 ? 2 : 1) : 0;
    }

    void bounds(size_t b, size_t e, float lo[3], float hi[3]) const {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::numeric_limits<float>::infinity();
            hi[k] = -std::numeric_limits<float>::infinity();
        }
        for (size_t i = b; i < e; ++i)
            for (int k = 0; k < 3; ++k) {
                lo[k] = std::min(lo[k], std::min(tris[i].a[k], std::min(tris[i].b[k], tris[i].c[k])));
                hi[k] = std::max(hi[k], std::max(tris[i].a[k], std::max(tris[i].b[k], tris[i].c[k])));
            }
    }
    size_t split(size_t b, size_t e) {   // median along the axis of largest centroid variance
        const double n = (double)(e - b);
        double mean[3] = {0, 0, 0}, var[3] = {0, 0, 0};
        for (size_t i = b; i < e; ++i)
            for (int k = 0; k < 3; ++k) mean[k] += tris[i].cen[k];
        for (int k = 0; k < 3; ++k) mean[k] /= n;
        for (size_t i = b; i < e; ++i)
            for (int k = 0; k < 3; ++k) {
                const double dlt = tris[i].cen[k] - mean[k];
                var[k] += dlt * dlt;
            }
        const int axis = var[0] >= var[1] ? (var[0] >= var[2] ? 0 : 2) : (var[1] >= var[2] ? 1 : 2);
        const size_t m = b + (e - b) / 2;
        std::nth_element(tris.begin() + b, tris.begin() + m, tris.begin() + e,
                         [axis](const HostTri& p, const HostTri& q) { return p.cen[axis] < q.cen[axis]; });
        return m;
    }
    // returns the link of the subtree over [b, e)
    int32_t build(size_t b, size_t e, int depth) {
        int seen = max_depth.load();
        while (depth > seen && !max_depth.compare_exchange_weak(seen, depth)) {}
        if (e - b <= (size_t)kLeafTris) return -(int32_t)(b * 16 + (e - b)) - 1;
        const int32_t me = used.fetch_add(1);
        if ((size_t)me >= cap) {   // (cannot happen with the bound above; reported, never written past the array)
            overflow = true;
            return INT32_MIN;
        }
        size_t cut[5];
        cut[0] = b;
        cut[4] = e;
        cut[2] = split(b, e);
        if (depth < par_levels) {   // the two halves are independent
            std::thread t([&] { cut[1] = split(cut[0], cut[2]); });
            cut[3] = split(cut[2], cut[4]);
            t.join();
        } else {
            cut[1] = split(cut[0], cut[2]);
            cut[3] = split(cut[2], cut[4]);
        }
        int32_t link[4];
        auto child = [&](int k) {
            float lo[3], hi[3];
            link[k] = INT32_MIN;
            if (cut[k + 1] > cut[k]) {
                bounds(cut[k], cut[k + 1], lo, hi);
                link[k] = build(cut[k], cut[k + 1], depth + 1);
            } else {
                for (int a = 0; a < 3; ++a) {
                    lo[a] = std::numeric_limits<float>::infinity();
                    hi[a] = -std::numeric_limits<float>::infinity();
                }
            }
            Node& nd = nodes[me];   // (child k writes column k only)
            for (int a = 0; a < 3; ++a) {
                nd.lo[a][k] = lo[a];
                nd.hi[a][k] = hi[a];
            }
            nd.child[k] = link[k];
            nd.pad[k] = 0;
        };
        if (depth < par_levels) {
            std::thread th[3];
            for (int k = 0; k < 3; ++k) th[k] = std::thread(child, k);
            child(3);
            for (int k = 0; k < 3; ++k) th[k].join();
        } else {
            for (int k = 0; k < 4; ++k) child(k);
        }
        return me;
    }
};

}  // namespace

struct p3d_rc_caster {
    Node* nodes = nullptr;        // device
    float4* tris = nullptr;       // device, 3 per triangle
    int64_t num_nodes = 0, num_tris = 0;
    int32_t root = 0;             // link of the root (a mesh of <= 8 triangles is a single leaf)
    int32_t max_depth = 0;
    int device = 0;
};

namespace {

struct Hit {
    float t;
    int tri;
};

__device__ inline void tri_intersect(const float4* __restrict__ tr, float ox, float oy, float oz, float dx, float dy, float dz,
                                     float& t_out) {
    // triangle.h:16-33, operation for operation
    const float4 A = tr[0], B = tr[1], C = tr[2];
    const float e1x = B.x - A.x, e1y = B.y - A.y, e1z = B.z - A.z;          // v1v0
    const float e2x = C.x - A.x, e2y = C.y - A.y, e2z = C.z - A.z;          // v2v0
    const float rx = ox - A.x, ry = oy - A.y, rz = oz - A.z;                // rov0
    const float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;   // n = v1v0 x v2v0
    const float qx = ry * dz - rz * dy, qy = rz * dx - rx * dz, qz = rx * dy - ry * dx;               // q = rov0 x rd
    const float d = 1.0f / (dx * nx + dy * ny + dz * nz);
    const float u = d * -(qx * e2x + qy * e2y + qz * e2z);
    const float v = d * (qx * e1x + qy * e1y + qz * e1z);
    float t = d * -(nx * rx + ny * ry + nz * rz);
    if (u < 0.0f || u > 1.0f || v < 0.0f || (u + v) > 1.0f || t < 0.0f) t = 3.402823466e+38f;   // no intersection
    t_out = t;
}

// KS: stack entries per ray (3 * tree depth + 1 fit; a smaller stack leaves room for more waves per CU).
// (Measured and dropped: carrying each entry's box-entry distance to skip boxes popped after a nearer hit was found --
//  the second LDS array and the lost occupancy cost more than the skipped nodes save: 1.24 ms vs 0.95 ms.)
template <int KS>
__global__ void __launch_bounds__(kRcBlock) k_raycast(const Node* __restrict__ nodes, const float4* __restrict__ tris,
                                                      int32_t root, const float* __restrict__ origins,
                                                      const float* __restrict__ directions, int64_t n,
                                                      float* __restrict__ depths, float* __restrict__ normals,
                                                      int32_t* __restrict__ ids) {
    __shared__ int32_t s_stack[KS][kRcBlock];
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kRcBlock + lane;
    if (i >= n) return;
    const float ox = origins[i * 3], oy = origins[i * 3 + 1], oz = origins[i * 3 + 2];
    const float dx = directions[i * 3], dy = directions[i * 3 + 1], dz = directions[i * 3 + 2];
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    float best = kMaxDist;   // bvh.cu:155
    int best_tri = -1;
    int sp = 0;
    s_stack[sp++][lane] = root;
    while (sp > 0) {
        const int32_t link = s_stack[--sp][lane];
        if (link < 0) {   // leaf: triangles [first, first + count)
            const int32_t code = -(link + 1);
            const int first = code >> 4, count = code & 15;
            for (int k = 0; k < count; ++k) {
                float t;
                tri_intersect(tris + (size_t)(first + k) * 3, ox, oy, oz, dx, dy, dz, t);
                if (t < best) {   // strict, bvh.cu:170
                    best = t;
                    best_tri = first + k;
                }
            }
            continue;
        }
        const Node& nd = nodes[link];
        float dist[4];
        int32_t ch[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // slab test (bounding_box.h:157-210), NaN-tolerant form: fminf / fmaxf drop a NaN operand
            const float x0 = (nd.lo[0][k] - ox) * ix, x1 = (nd.hi[0][k] - ox) * ix;
            const float y0 = (nd.lo[1][k] - oy) * iy, y1 = (nd.hi[1][k] - oy) * iy;
            const float z0 = (nd.lo[2][k] - oz) * iz, z1 = (nd.hi[2][k] - oz) * iz;
            const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
            const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
            ch[k] = nd.child[k];
            const bool hit = ch[k] != INT32_MIN && tn <= tf && tf >= 0.0f && tn < best;
            dist[k] = hit ? tn : 3.402823466e+38f;
        }
        // far to near onto the stack so that the nearest child is popped first (sorting network, bvh.cu:46-142)
#define P3D_CSWAP(a_, b_)                      \
    if (dist[a_] > dist[b_]) {                 \
        const float td = dist[a_];             \
        dist[a_] = dist[b_];                   \
        dist[b_] = td;                         \
        const int32_t tc = ch[a_];             \
        ch[a_] = ch[b_];                       \
        ch[b_] = tc;                           \
    }
        P3D_CSWAP(0, 1) P3D_CSWAP(2, 3) P3D_CSWAP(0, 2) P3D_CSWAP(1, 3) P3D_CSWAP(1, 2)
#undef P3D_CSWAP
#pragma unroll
        for (int k = 3; k >= 0; --k)
            if (dist[k] < 3.0e+38f && sp < KS) s_stack[sp++][lane] = ch[k];
    }
    depths[i] = best;
    if (best_tri >= 0) {
        const float4 A = tris[(size_t)best_tri * 3], B = tris[(size_t)best_tri * 3 + 1], C = tris[(size_t)best_tri * 3 + 2];
        // triangle.h:12-14: (b - a) x (c - a), normalised
        const float e1x = B.x - A.x, e1y = B.y - A.y, e1z = B.z - A.z;
        const float e2x = C.x - A.x, e2y = C.y - A.y, e2z = C.z - A.z;
        const float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
        const float len = sqrtf(nx * nx + ny * ny + nz * nz);
        normals[i * 3] = nx / len;
        normals[i * 3 + 1] = ny / len;
        normals[i * 3 + 2] = nz / len;
        ids[i] = __float_as_int(A.w);
    } else {
        normals[i * 3] = normals[i * 3 + 1] = normals[i * 3 + 2] = 0.0f;   // bvh.cu:340-344
        ids[i] = -1;
    }
}

}  // namespace

extern "C" {

int p3d_rc_abi_version(void) { return P3D_RC_ABI_VERSION; }
const char* p3d_rc_last_error(void) { return g_err; }

int p3d_rc_create(const float* vertices, int64_t num_vertices, const int32_t* faces, int64_t num_faces,
                  p3d_rc_caster** out) {
    if (!vertices || !faces || !out) return fail(P3D_RC_EINVAL, "null pointer%s");
    if (num_faces < 1 || num_vertices < 1) return fail(P3D_RC_EINVAL, "need at least one triangle%s");
    if (num_faces >= (1ll << 27)) return fail(P3D_RC_ERANGE, "more than 2^27 triangles%s");
    std::vector<HostTri> tris((size_t)num_faces);
    std::atomic<bool> bad{false};
    parallel_chunks((size_t)num_faces, [&](size_t f0, size_t f1) {
        for (size_t f = f0; f < f1; ++f) {
            HostTri& t = tris[f];
            const int32_t ia = faces[f * 3], ib = faces[f * 3 + 1], ic = faces[f * 3 + 2];
            if (ia < 0 || ib < 0 || ic < 0 || ia >= num_vertices || ib >= num_vertices || ic >= num_vertices) {
                bad = true;
                return;
            }
            for (int k = 0; k < 3; ++k) {
                t.a[k] = vertices[(size_t)ia * 3 + k];
                t.b[k] = vertices[(size_t)ib * 3 + k];
                t.c[k] = vertices[(size_t)ic * 3 + k];
                t.cen[k] = (t.a[k] + t.b[k] + t.c[k]) / 3.0f;   // triangle.h:40-46
                if (!(t.cen[k] == t.cen[k])) t.cen[k] = 0.0f;   // NaN coordinates: keep the split comparator a strict weak order
            }
            t.idx = (int32_t)f;
        }
    });
    if (bad) return fail(P3D_RC_EINVAL, "face index out of range%s");
    Builder bld(tris);
    const int32_t root = bld.build(0, tris.size(), 0);
    if (bld.overflow) return fail(P3D_RC_ERANGE, "node array too small%s");
    const size_t num_nodes = (size_t)bld.used.load();
    if (3 * bld.max_depth.load() + 1 > kStackMax) return fail(P3D_RC_ERANGE, "tree deeper than the traversal stack%s");
    std::unique_ptr<float4[]> packed(new float4[tris.size() * 3]);
    const size_t packed_n = tris.size() * 3;
    parallel_chunks(tris.size(), [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
            float w;
            memcpy(&w, &tris[i].idx, 4);
            packed[i * 3] = make_float4(tris[i].a[0], tris[i].a[1], tris[i].a[2], w);
            packed[i * 3 + 1] = make_float4(tris[i].b[0], tris[i].b[1], tris[i].b[2], 0.f);
            packed[i * 3 + 2] = make_float4(tris[i].c[0], tris[i].c[1], tris[i].c[2], 0.f);
        }
    });
    p3d_rc_caster* c = new p3d_rc_caster();
    c->num_nodes = (int64_t)num_nodes;
    c->num_tris = num_faces;
    c->root = root;
    c->max_depth = bld.max_depth.load();
    if (hipGetDevice(&c->device) != hipSuccess) {
        delete c;
        return fail(P3D_RC_EHIP, "hipGetDevice failed%s");
    }
    const size_t nb = std::max<size_t>(num_nodes, 1) * sizeof(Node);
    if (hipMalloc((void**)&c->nodes, nb) != hipSuccess || hipMalloc((void**)&c->tris, packed_n * sizeof(float4)) != hipSuccess) {
        p3d_rc_destroy(c);
        return fail(P3D_RC_EHIP, "hipMalloc failed%s");
    }
    if ((num_nodes > 0 && hipMemcpy(c->nodes, bld.nodes.get(), num_nodes * sizeof(Node), hipMemcpyHostToDevice) != hipSuccess) ||
        hipMemcpy(c->tris, packed.get(), packed_n * sizeof(float4), hipMemcpyHostToDevice) != hipSuccess) {
        p3d_rc_destroy(c);
        return fail(P3D_RC_EHIP, "hipMemcpy failed%s");
    }
    *out = c;
    return P3D_RC_OK;
}

void p3d_rc_destroy(p3d_rc_caster* c) {
    if (!c) return;
    if (c->nodes) (void)hipFree(c->nodes);
    if (c->tris) (void)hipFree(c->tris);
    delete c;
}

int p3d_rc_invoke(const p3d_rc_caster* c, const float* origins, const float* directions, int64_t num_rays, float* depths,
                  float* normals, int32_t* primitive_ids, void* stream) {
    if (!c) return fail(P3D_RC_EINVAL, "null caster%s");
    if (num_rays < 0) return fail(P3D_RC_EINVAL, "negative ray count%s");
    if (num_rays == 0) return P3D_RC_OK;
    if (!origins || !directions || !depths || !normals || !primitive_ids) return fail(P3D_RC_EINVAL, "null pointer%s");
    const int64_t blocks = (num_rays + kRcBlock - 1) / kRcBlock;
    if (blocks >= (1ll << 31)) return fail(P3D_RC_ERANGE, "too many rays for one launch%s");
    if (3 * c->max_depth + 1 <= 32)
        hipLaunchKernelGGL(k_raycast<32>, dim3((unsigned)blocks), dim3(kRcBlock), 0, (hipStream_t)stream, c->nodes, c->tris,
                           c->root, origins, directions, num_rays, depths, normals, primitive_ids);
    else
        hipLaunchKernelGGL(k_raycast<kStackMax>, dim3((unsigned)blocks), dim3(kRcBlock), 0, (hipStream_t)stream, c->nodes,
                           c->tris, c->root, origins, directions, num_rays, depths, normals, primitive_ids);
    HIP_TRY(hipGetLastError());
    return P3D_RC_OK;
}

int p3d_rc_stats(const p3d_rc_caster* c, int64_t* num_nodes, int64_t* num_triangles, int32_t* max_depth) {
    if (!c) return fail(P3D_RC_EINVAL, "null caster%s");
    if (num_nodes) *num_nodes = c->num_nodes;
    if (num_triangles) *num_triangles = c->num_tris;
    if (max_depth) *max_depth = c->max_depth;
    return P3D_RC_OK;
}

}  // extern "C"
