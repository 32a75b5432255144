"""Brute-force restatement of the reference's BVH ray caster RESULT (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

The reference walks a BVH (src/prim3d/Geometry/bvh.cu:146-196) but what it returns per ray is fixed by three lines:
  * the intersector, src/prim3d/Geometry/triangle.h:16-33 (restated below in float32, same operation order),
  * the search `t < mint` from mint = MAX_DIST = 10 over every triangle the walk reaches (bvh.cu:13,155,168-173) -- the
    walk only skips boxes the ray enters at or beyond the best t, so the minimum over ALL triangles is the same number,
  * the outputs, bvh.cu:326-345: depth = that minimum (10 on a miss), unit normal (b-a) x (c-a) of the winner
    (triangle.h:12-14), its index in `faces`; a miss gets normal 0 and id -1.
Among triangles hit at exactly the same t the winner depends on the traversal order (unspecified): the tests compare ids
only where the runner-up is farther than a margin.  Parity status: restatement only -- the reference's CUDA build cannot
run here (and nvcc contracts these products into FMAs, so its t can differ from this file's in the last bits).
"""
import numpy as np

MAX_DIST = np.float32(10.0)
FLT_MAX = np.float32(3.402823466e+38)


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def _dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def ray_triangle_t(ro, rd, A, B, C):
    """triangle.h:16-33 for rays [R,3] against triangles [T,3] -> t [R,T] float32."""
    ro, rd = ro[:, None, :], rd[:, None, :]
    v1v0, v2v0 = (B - A)[None], (C - A)[None]
    rov0 = ro - A[None]
    n = _cross(v1v0, v2v0)
    q = _cross(rov0, rd)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        d = np.float32(1.0) / _dot(rd, n)
        u = d * -_dot(q, v2v0)
        v = d * _dot(q, v1v0)
        t = d * -_dot(n, rov0)
        bad = (u < 0) | (u > 1) | (v < 0) | ((u + v) > 1) | (t < 0)
    t = np.where(bad, FLT_MAX, t)
    return t.astype(np.float32)


def raycast_oracle(vertices, faces, origins, directions, chunk=256):
    """-> depths f32 [R], normals f32 [R,3], ids i32 [R], second-best t f32 [R] (for tie margins)."""
    vertices = np.asarray(vertices, np.float32)
    faces = np.asarray(faces, np.int64)
    origins = np.asarray(origins, np.float32)
    directions = np.asarray(directions, np.float32)
    A, B, C = vertices[faces[:, 0]], vertices[faces[:, 1]], vertices[faces[:, 2]]
    R = origins.shape[0]
    depths = np.full(R, MAX_DIST, np.float32)
    second = np.full(R, FLT_MAX, np.float32)
    ids = np.full(R, -1, np.int32)
    for r0 in range(0, R, chunk):
        t = ray_triangle_t(origins[r0:r0 + chunk], directions[r0:r0 + chunk], A, B, C)
        t = np.where(np.isnan(t), FLT_MAX, t)
        order = np.argsort(t, axis=1, kind="stable")[:, :2]
        rows = np.arange(t.shape[0])
        t0 = t[rows, order[:, 0]]
        hit = t0 < MAX_DIST
        depths[r0:r0 + chunk] = np.where(hit, t0, MAX_DIST)
        ids[r0:r0 + chunk] = np.where(hit, order[:, 0], -1)
        if t.shape[1] > 1:
            second[r0:r0 + chunk] = t[rows, order[:, 1]]
    nrm = np.zeros((R, 3), np.float32)
    h = ids >= 0
    n = _cross(B[ids[h]] - A[ids[h]], C[ids[h]] - A[ids[h]])
    with np.errstate(divide="ignore", invalid="ignore"):
        ln = np.sqrt((n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1]) + n[:, 2] * n[:, 2]).astype(np.float32)
        nrm[h] = n / ln[:, None]
    return depths, nrm, ids, second
