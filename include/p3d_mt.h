/*
 * p3d_mt.h -- C ABI of the MI355X-native marching tetrahedra (libp3dmt.so).
 *
 * Replaces the tensor-op chain of the reference's prim3d/utility/marching_tetrahedras.py:89-235 (a kaolin-derived,
 * pure-PyTorch function: ~25 small kernels, two boolean-mask compactions, a row-wise torch.unique over all edges of the
 * active tetrahedra) by a handful of hand-written HIP kernels around a hash set of the CROSSING edges and one radix sort
 * of the distinct ones (only those become vertices).  Same results:
 *   - the orientation fix of :147-148 (tets with a negative [1,x,y,z] determinant get corners 0 and 1 swapped, IN
 *     PLACE in the caller's array, as the reference does).  The determinant is taken in float64 from the float32
 *     coordinates; the reference's is a float32 LU (`torch.det`, :50-65).  The two signs can differ only where the
 *     determinant is rounding noise: measured against reference-run vectors that keep their slivers
 *     (tests/golden/tetraslivers_*.npz, ~1600 of 14.9 k tets with |det| < 1e-7): 0 tets at a point jitter of 1e-6, 6 of
 *     14 847 at 3e-8, all with |det| < 1e-25; such a tet's triangles then have the opposite winding, nothing else
 *     changes (tests/tetra_compare.py),
 *   - vertices in the order of torch.unique's sorted rows (:160-171), computed with the reference's float32 operation
 *     order (:178-190),
 *   - faces by the 16-case table (:8-29): all one-triangle tets first, then the two-triangle tets (:205-224), int64
 *     vertex ids, and the tet index of every face (:226-234).
 *
 * Conventions: all pointers are DEVICE pointers owned by the caller (row-major, contiguous); `stream` is a hipStream_t
 * passed as void*; return 0 / negative P3D_MT_E*; p3d_mt_last_error() gives a thread-local message.  p3d_mt_prepare
 * waits for the device twice, where the reference's own ops synchronise (boolean-mask indexing, torch.unique): the
 * number of active tets and the output sizes have to reach the host (through a pinned host slot the kernels write;
 * P3D_NO_MAILBOX=1 or any failure: copy + hipStreamSynchronize).
 */
#ifndef P3D_MT_H_
#define P3D_MT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P3D_MT_ABI_VERSION 2
#define P3D_MT_OK 0
#define P3D_MT_EINVAL (-1)
#define P3D_MT_ERANGE (-2)   /* more than 2^32 - 1 vertices, or more edges than 32-bit slots */
#define P3D_MT_EHIP (-3)
#define P3D_MT_EINDEX (-4)   /* a tet refers to a point outside [0, num_vertices): the reference's indexing raises IndexError */

/* Device scratch for a mesh of num_tets tetrahedra (worst case: every tet active; about 250 bytes per tet: the hash set
 * of edges and its rank table 36 B, the distinct keys and their sorted copy 64 B, rocPRIM's temporaries, per-tet lists). */
int p3d_mt_workspace_bytes(int64_t num_vertices, int64_t num_tets, size_t* bytes);

/* Phase 1: orientation fix (tets is IN/OUT), classification of every tet against sdf > 0, the sorted unique edges
 * of the active tets and their vertex ids.  Returns the sizes of the outputs: out_vertices = crossing edges,
 * out_faces = triangles.  vertices: float32 [num_vertices,3]; sdf: float32 [num_vertices]; tets: int64 [num_tets,4].
 * A tet index outside [0, num_vertices) is never dereferenced: the call returns P3D_MT_EINDEX (tets may already have
 * been partly corrected in place). */
int p3d_mt_prepare(const float* vertices, int64_t num_vertices, int64_t* tets, int64_t num_tets, const float* sdf,
                   void* ws, int64_t* out_vertices, int64_t* out_faces, void* stream);

/* Phase 2 (after p3d_mt_prepare on the same ws): write the interpolated vertices [V,3] f32, the endpoint pair of every
 * vertex [V,2] i64 (nullable; the Python wrapper uses it to rebuild the vertices with autograd when gradients are
 * needed), the faces [F,3] i64 and the tet index of every face [F] i64 (nullable).  `tets` is the array p3d_mt_prepare
 * corrected. */
int p3d_mt_emit(const float* vertices, const int64_t* tets, const float* sdf, void* ws, float* out_vertices,
                int64_t* out_edge_pairs, int64_t* out_faces, int64_t* out_tet_idx, void* stream);

const char* p3d_mt_last_error(void);
int p3d_mt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* P3D_MT_H_ */
