/*
 * mc_oracle.c -- CPU restatement of the Primitive3D marching-cubes hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product path (primitive3d_amd/) never does.
 *
 * What it restates (all citations are into /root/reference):
 *   - count_vertices_faces_kernel   src/prim3d/Utility/marching_cubes.cu:4-68
 *   - gen_vertices_kernel           src/prim3d/Utility/marching_cubes.cu:70-138
 *   - gen_faces_kernel              src/prim3d/Utility/marching_cubes.cu:140-209
 *   - host driver + scale/offset    src/prim3d/Utility/marching_cubes.cu:212-305
 *   - case table                    src/prim3d/Utility/marching_cubes.h:21-277 (nibble-packed here,
 *                                   see tools/gen_tri_table.py; sha256 of the unpacked bytes is tested)
 *
 * Pinning status: the reference GPU path cannot be built in this image (needs nvcc + CUDA libtorch),
 * and its CPU path is the un-vendored third-party PyMCubes, also absent.  The oracle is therefore
 * pinned against the known answers the reference's own example checks imply and SURVEY.md section 4
 * records (vertex/face counts on sphere200, bunny66, sphere64; closed-manifold invariants; table
 * sha256), and against an independent numpy count (oracle/np_counts.py).  Vertex VALUES and face
 * INDICES have no reference-run vectors: that part of parity is "restatement only".
 *
 * Differences from the reference that do not change results:
 *   - the reference assigns vertex/face slots by atomicAdd arrival order (nondeterministic); the
 *     oracle visits voxels in linear order, which is one admissible outcome.  Comparisons are made
 *     on the canonical form (vertices keyed by edge key = voxel_linear*3+axis, faces as ordered
 *     triples of edge keys).
 *   - 64-bit index arithmetic (the reference's int32 x*(res_y*res_z*3) overflows at N >= 895).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: the reference's mul-then-add epilogue,
 * marching_cubes.cu:298, must not be fused into an FMA).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../primitive3d_amd/csrc/tri_table_packed.inc"

static const uint64_t k_tri_packed[256] = {P3D_TRI_TABLE_PACKED};

/* slot i of case `mask`, -1 when past the end (marching_cubes.h:21-277 semantics) */
static inline int tri_entry(int mask, int i) {
    int v = (int)((k_tri_packed[mask] >> (4 * i)) & 0xF);
    return v == 0xF ? -1 : v;
}

/* unpack to the reference's int8[256][16] layout (for the sha256 test) */
void p3d_oracle_tri_table(int8_t* out) {
    for (int m = 0; m < 256; ++m)
        for (int i = 0; i < 16; ++i) out[m * 16 + i] = (int8_t)tri_entry(m, i);
}

/* marching_cubes.cu:49-57: corner bit weights */
static inline int cell_mask(const float* g, int64_t sy, int64_t sx, int64_t x, int64_t y, int64_t z,
                            float t) {
    const float* p = g + x * sx + y * sy + z;
    int mask = 0;
    if (p[0] > t) mask |= 1;
    if (p[sx] > t) mask |= 2;
    if (p[sx + sy] > t) mask |= 4;
    if (p[sy] > t) mask |= 8;
    if (p[1] > t) mask |= 16;
    if (p[sx + 1] > t) mask |= 32;
    if (p[sx + sy + 1] > t) mask |= 64;
    if (p[sy + 1] > t) mask |= 128;
    return mask;
}

/* marching_cubes.cu:61-65: 3 * #triangles, found by scanning the row */
static inline int tri_index_count(int mask) {
    int n = 0;
    for (; n < 15; n += 3)
        if (tri_entry(mask, n) < 0) break;
    return n;
}

/* count_vertices_faces_kernel, marching_cubes.cu:4-68.  counters[0] = V, counters[1] = 3F. */
int p3d_oracle_count(const float* grid, int64_t rx, int64_t ry, int64_t rz, float thresh,
                     int64_t* num_vertices, int64_t* num_face_indices) {
    if (rx < 1 || ry < 1 || rz < 1) return -1;
    const int64_t sy = rz, sx = ry * rz;
    int64_t nv = 0, nf = 0;
    for (int64_t x = 0; x < rx; ++x)
        for (int64_t y = 0; y < ry; ++y)
            for (int64_t z = 0; z < rz; ++z) {
                const float* p = grid + x * sx + y * sy + z;
                const int inside = p[0] > thresh; /* strict >, :25 */
                if (x < rx - 1 && inside != (p[sx] > thresh)) ++nv;
                if (y < ry - 1 && inside != (p[sy] > thresh)) ++nv;
                if (z < rz - 1 && inside != (p[1] > thresh)) ++nv;
                if (x < rx - 1 && y < ry - 1 && z < rz - 1)
                    nf += tri_index_count(cell_mask(grid, sy, sx, x, y, z, thresh));
            }
    *num_vertices = nv;
    *num_face_indices = nf;
    return 0;
}

static void p3d_oracle_epilogue(int64_t rx, int64_t ry, int64_t rz, const float* lower, const float* upper,
                                float* verts, int64_t cursor);

/*
 * Full extraction.  verts: float[V*3], vkeys: int64[V] (edge key, may be NULL), faces: int32[F*3].
 * V and F must come from p3d_oracle_count.  lower/upper as the pybind boundary hands them over
 * (marching_cubes.h:14-15): already float32.
 */
int p3d_oracle_extract(const float* grid, int64_t rx, int64_t ry, int64_t rz, float thresh,
                       const float* lower, const float* upper, float* verts, int64_t* vkeys,
                       int32_t* faces) {
    if (rx < 1 || ry < 1 || rz < 1) return -1;
    const int64_t sy = rz, sx = ry * rz;
    const int64_t nvox = rx * ry * rz;
    /* vertex_grids, marching_cubes.cu:257-259: [rx,ry,rz,3] int32, 0 = no vertex, else id+1 */
    int32_t* vgrid = (int32_t*)calloc((size_t)nvox * 3, sizeof(int32_t));
    if (!vgrid) return -2;

    /* gen_vertices_kernel, marching_cubes.cu:70-138 */
    int64_t cursor = 0;
    for (int64_t x = 0; x < rx; ++x)
        for (int64_t y = 0; y < ry; ++y)
            for (int64_t z = 0; z < rz; ++z) {
                const int64_t lin = x * sx + y * sy + z;
                const float d0 = grid[lin];
                const int inside = d0 > thresh;
                int32_t* cur = vgrid + lin * 3;
                if (x < rx - 1) {
                    const float d1 = grid[lin + sx];
                    if (inside != (d1 > thresh)) {
                        const float dt = (thresh - d0) / (d1 - d0); /* :105 */
                        cur[0] = (int32_t)(cursor + 1);
                        verts[cursor * 3 + 0] = (float)x + dt;
                        verts[cursor * 3 + 1] = (float)y;
                        verts[cursor * 3 + 2] = (float)z;
                        if (vkeys) vkeys[cursor] = lin * 3 + 0;
                        ++cursor;
                    }
                }
                if (y < ry - 1) {
                    const float d1 = grid[lin + sy];
                    if (inside != (d1 > thresh)) {
                        const float dt = (thresh - d0) / (d1 - d0); /* :118 */
                        cur[1] = (int32_t)(cursor + 1);
                        verts[cursor * 3 + 0] = (float)x;
                        verts[cursor * 3 + 1] = (float)y + dt;
                        verts[cursor * 3 + 2] = (float)z;
                        if (vkeys) vkeys[cursor] = lin * 3 + 1;
                        ++cursor;
                    }
                }
                if (z < rz - 1) {
                    const float d1 = grid[lin + 1];
                    if (inside != (d1 > thresh)) {
                        const float dt = (thresh - d0) / (d1 - d0); /* :131 */
                        cur[2] = (int32_t)(cursor + 1);
                        verts[cursor * 3 + 0] = (float)x;
                        verts[cursor * 3 + 1] = (float)y;
                        verts[cursor * 3 + 2] = (float)z + dt;
                        if (vkeys) vkeys[cursor] = lin * 3 + 2;
                        ++cursor;
                    }
                }
            }

    /* gen_faces_kernel, marching_cubes.cu:140-209 */
    int64_t fcur = 0;
    for (int64_t x = 0; x + 1 < rx; ++x)
        for (int64_t y = 0; y + 1 < ry; ++y)
            for (int64_t z = 0; z + 1 < rz; ++z) {
                const int mask = cell_mask(grid, sy, sx, x, y, z, thresh);
                if (tri_entry(mask, 0) < 0) continue;
                const int64_t lin = x * sx + y * sy + z;
#define VG(dx, dy, dz, a) vgrid[(lin + (dx)*sx + (dy)*sy + (dz)) * 3 + (a)]
                int32_t e[12]; /* :178-192, Bourke edge numbering */
                e[0] = VG(0, 0, 0, 0);
                e[1] = VG(1, 0, 0, 1);
                e[2] = VG(0, 1, 0, 0);
                e[3] = VG(0, 0, 0, 1);
                e[4] = VG(0, 0, 1, 0);
                e[5] = VG(1, 0, 1, 1);
                e[6] = VG(0, 1, 1, 0);
                e[7] = VG(0, 0, 1, 1);
                e[8] = VG(0, 0, 0, 2);
                e[9] = VG(1, 0, 0, 2);
                e[10] = VG(1, 1, 0, 2);
                e[11] = VG(0, 1, 0, 2);
#undef VG
                for (int i = 0; i < 15; ++i) { /* :201-208 */
                    const int j = tri_entry(mask, i);
                    if (j < 0) break;
                    faces[fcur++] = e[j] - 1;
                }
            }
    free(vgrid);
    p3d_oracle_epilogue(rx, ry, rz, lower, upper, verts, cursor);
    return 0;
}

/* epilogue, marching_cubes.cu:290-298. (shared by the serial and the OpenMP driver) */
static void p3d_oracle_epilogue(int64_t rx, int64_t ry, int64_t rz, const float* lower, const float* upper,
                                float* verts, int64_t cursor) {
    /* marching_cubes.cu:290-298.  NOTE the reference quirk at :295: the y scale uses
     * upper[2] - lower[1].  vertices = vertices * scale + offset, two separately rounded ops. */
    const float scale[3] = {(upper[0] - lower[0]) / (float)rx, (upper[2] - lower[1]) / (float)ry,
                            (upper[2] - lower[2]) / (float)rz};
    for (int64_t i = 0; i < cursor; ++i)
        for (int a = 0; a < 3; ++a) {
            volatile float m = verts[i * 3 + a] * scale[a];
            verts[i * 3 + a] = m + lower[a];
        }
}

/*
 * The same extraction on `nthreads` host cores (OpenMP over axis-0 planes; SURVEY.md section 8d "all host cores"
 * CPU baseline).  Every plane's vertex and face-index counts are taken first (the reference's counting kernel,
 * marching_cubes.cu:4-68, per plane), an exclusive prefix over the planes gives each plane its first vertex id and
 * first face slot, and the planes are then filled independently with exactly the loops of p3d_oracle_extract above:
 * the output (values AND order) is identical to the serial oracle's, which tests/test_oracle_cpu.py asserts.
 * Returns the number of threads actually used (>0), negative on error.
 */
#ifdef _OPENMP
#include <omp.h>
#endif
int p3d_oracle_extract_mt(const float* grid, int64_t rx, int64_t ry, int64_t rz, float thresh, const float* lower,
                          const float* upper, float* verts, int64_t* vkeys, int32_t* faces, int nthreads) {
    if (rx < 1 || ry < 1 || rz < 1) return -1;
    const int64_t sy = rz, sx = ry * rz;
    const int64_t nvox = rx * ry * rz;
    /* The dense id grid (the reference's vertex_grids, 12 B per voxel: 1.6 GB at 512^3) is kept between calls and
     * cleared by all threads: a fresh calloc per extraction spent most of a many-core run in page faults (128 threads
     * ran only 2x faster than one), which says nothing about the algorithm on the host's cores.  Not thread-safe across
     * concurrent callers -- the baseline and the tests call it from one thread. */
    static int32_t* vgrid_cache = NULL;
    static size_t vgrid_words = 0;
    if (vgrid_words < (size_t)nvox * 3) {
        free(vgrid_cache);
        vgrid_cache = (int32_t*)malloc((size_t)nvox * 3 * sizeof(int32_t));
        vgrid_words = vgrid_cache ? (size_t)nvox * 3 : 0;
    }
    int32_t* vgrid = vgrid_cache;
    int64_t* vfirst = (int64_t*)calloc((size_t)rx + 1, sizeof(int64_t));
    int64_t* ffirst = (int64_t*)calloc((size_t)rx + 1, sizeof(int64_t));
    if (!vgrid || !vfirst || !ffirst) {
        free(vfirst);
        free(ffirst);
        return -2;
    }
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
    }
#else
    (void)nthreads;
#endif
    /* per-plane counts (count_vertices_faces_kernel restricted to one x); the plane's ids are cleared on the way */
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t x = 0; x < rx; ++x) {
        memset(vgrid + x * sx * 3, 0, (size_t)sx * 3 * sizeof(int32_t));
        int64_t nv = 0, nf = 0;
        for (int64_t y = 0; y < ry; ++y)
            for (int64_t z = 0; z < rz; ++z) {
                const float* p = grid + x * sx + y * sy + z;
                const int inside = p[0] > thresh;
                if (x < rx - 1 && inside != (p[sx] > thresh)) ++nv;
                if (y < ry - 1 && inside != (p[sy] > thresh)) ++nv;
                if (z < rz - 1 && inside != (p[1] > thresh)) ++nv;
                if (x < rx - 1 && y < ry - 1 && z < rz - 1)
                    nf += tri_index_count(cell_mask(grid, sy, sx, x, y, z, thresh));
            }
        vfirst[x + 1] = nv;
        ffirst[x + 1] = nf;
    }
    for (int64_t x = 0; x < rx; ++x) {
        vfirst[x + 1] += vfirst[x];
        ffirst[x + 1] += ffirst[x];
    }
    /* gen_vertices_kernel per plane, marching_cubes.cu:70-138 */
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t x = 0; x < rx; ++x) {
        int64_t cursor = vfirst[x];
        for (int64_t y = 0; y < ry; ++y)
            for (int64_t z = 0; z < rz; ++z) {
                const int64_t lin = x * sx + y * sy + z;
                const float d0 = grid[lin];
                const int inside = d0 > thresh;
                int32_t* cur = vgrid + lin * 3;
                for (int a = 0; a < 3; ++a) {
                    const int64_t lim = a == 0 ? rx : (a == 1 ? ry : rz), pos = a == 0 ? x : (a == 1 ? y : z);
                    const int64_t step = a == 0 ? sx : (a == 1 ? sy : 1);
                    if (pos >= lim - 1) continue;
                    const float d1 = grid[lin + step];
                    if (inside == (d1 > thresh)) continue;
                    const float dt = (thresh - d0) / (d1 - d0); /* :105,:118,:131 */
                    cur[a] = (int32_t)(cursor + 1);
                    verts[cursor * 3 + 0] = (float)x + (a == 0 ? dt : 0.0f);
                    verts[cursor * 3 + 1] = (float)y + (a == 1 ? dt : 0.0f);
                    verts[cursor * 3 + 2] = (float)z + (a == 2 ? dt : 0.0f);
                    if (vkeys) vkeys[cursor] = lin * 3 + a;
                    ++cursor;
                }
            }
    }
    /* gen_faces_kernel per cell layer, marching_cubes.cu:140-209 */
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t x = 0; x < rx - 1; ++x) {
        int64_t fcur = ffirst[x];
        for (int64_t y = 0; y + 1 < ry; ++y)
            for (int64_t z = 0; z + 1 < rz; ++z) {
                const int mask = cell_mask(grid, sy, sx, x, y, z, thresh);
                if (tri_entry(mask, 0) < 0) continue;
                const int64_t lin = x * sx + y * sy + z;
#define VG(dx, dy, dz, a) vgrid[(lin + (dx)*sx + (dy)*sy + (dz)) * 3 + (a)]
                const int32_t e[12] = {VG(0, 0, 0, 0), VG(1, 0, 0, 1), VG(0, 1, 0, 0), VG(0, 0, 0, 1),
                                       VG(0, 0, 1, 0), VG(1, 0, 1, 1), VG(0, 1, 1, 0), VG(0, 0, 1, 1),
                                       VG(0, 0, 0, 2), VG(1, 0, 0, 2), VG(1, 1, 0, 2), VG(0, 1, 0, 2)};
#undef VG
                for (int i = 0; i < 15; ++i) {
                    const int j = tri_entry(mask, i);
                    if (j < 0) break;
                    faces[fcur++] = e[j] - 1;
                }
            }
    }
    const int64_t nv_total = vfirst[rx];
    free(vfirst);
    free(ffirst);
    p3d_oracle_epilogue(rx, ry, rz, lower, upper, verts, nv_total);
    return used;
}
