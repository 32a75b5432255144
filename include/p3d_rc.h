/*
 * p3d_rc.h -- C ABI of the MI355X-native ray caster (libp3drc.so): nearest ray / triangle-mesh hit through a 4-wide BVH.
 *
 * Replaces the non-OptiX path of the reference's RayCaster (paths into lzhnb/Primitive3D):
 *   create_raycaster / RayCasterImpl::build_bvh   src/prim3d/Utility/ray_cast.cu:340-384, :437-450
 *   TriangleBvh4::build (CPU, 4-ary, median split on the axis of largest centroid variance, <= 8 triangles per leaf)
 *                                                  src/prim3d/Geometry/bvh.cu:209-300
 *   RayCasterImpl::invoke -> raytrace_kernel -> TriangleBvh4::ray_intersect
 *                                                  ray_cast.cu:387-424, bvh.cu:146-196, :311-346
 *   Triangle::ray_intersect / normal               src/prim3d/Geometry/triangle.h:12-33
 * The OptiX path (ray_cast.cu:61-333, optix_ext/) is NVIDIA RT-core specific and has no counterpart here.
 *
 * Same results as the reference's BVH path: per ray the smallest t < 10 (MAX_DIST, bvh.cu:13,155) over all triangles with
 * the intersector of triangle.h:16-33, the unit normal (b-a) x (c-a) of that triangle and its index in `faces`; a ray
 * without such a hit gets depth 10, normal 0, id -1 (bvh.cu:330-345).  The tree itself is this library's own (child
 * boxes stored in the parent, one 128-byte node per 4 children); among triangles hit at EXACTLY the same t the winner
 * depends on traversal order, in the reference as here.
 *
 * Conventions: return 0 / negative P3D_RC_E*; p3d_rc_last_error() gives a thread-local message; `stream` is a
 * hipStream_t passed as void*.
 */
#ifndef P3D_RC_H_
#define P3D_RC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P3D_RC_ABI_VERSION 1
#define P3D_RC_OK 0
#define P3D_RC_EINVAL (-1)
#define P3D_RC_ERANGE (-2)
#define P3D_RC_EHIP (-3)

typedef struct p3d_rc_caster p3d_rc_caster;   /* opaque: the BVH and the triangles, resident on one GPU */

/* Build (replaces build_bvh): vertices float32 [num_vertices,3] and faces int32 [num_faces,3] are HOST pointers, as in
 * the reference (CHECK_CPU_INPUT, ray_cast.cu:346-347); the tree is built on the host cores and uploaded to the
 * current device.  num_faces >= 1. */
int p3d_rc_create(const float* vertices, int64_t num_vertices, const int32_t* faces, int64_t num_faces,
                  p3d_rc_caster** out);
void p3d_rc_destroy(p3d_rc_caster* caster);

/* Cast (replaces invoke): origins / directions float32 [num_rays,3], depths float32 [num_rays], normals float32
 * [num_rays,3], primitive_ids int32 [num_rays] -- DEVICE pointers on the caster's device; enqueued on `stream`. */
int p3d_rc_invoke(const p3d_rc_caster* caster, const float* origins, const float* directions, int64_t num_rays,
                  float* depths, float* normals, int32_t* primitive_ids, void* stream);

/* Sizes of the tree (diagnostics / tests). */
int p3d_rc_stats(const p3d_rc_caster* caster, int64_t* num_nodes, int64_t* num_triangles, int32_t* max_depth);

const char* p3d_rc_last_error(void);
int p3d_rc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* P3D_RC_H_ */
