"""Restatement of the reference's PLY writer, prim3d::save_mesh_as_ply
(src/prim3d/Utility/marching_cubes.cu:307-352).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference writes, in this order:
  :318-330  an ASCII header (`std::endl` after the two element lines is a plain newline),
  :336-339  per vertex: 3 float32 coordinates, then 3 uint8 colour components (15 bytes, little endian),
  :341-349  the faces as one block of int32 rows [3, i, j, k].
This file produces the same bytes with numpy so that the product's packed writer (csrc/bindings.cpp) can be compared
with it byte for byte.  Like the reference it takes float32 vertices, int32 faces and uint8 colours and nothing else
(`data_ptr<float>()` etc. throw on other dtypes).
"""
import numpy as np


def reference_ply_bytes(vertices: np.ndarray, faces: np.ndarray, colors: np.ndarray) -> bytes:
    assert vertices.dtype == np.float32 and faces.dtype == np.int32 and colors.dtype == np.uint8
    assert vertices.ndim == 2 and vertices.shape[1] == 3 and faces.ndim == 2 and faces.shape[1] == 3
    assert colors.shape == vertices.shape
    nv, nf = vertices.shape[0], faces.shape[0]
    head = ("ply\n"
            "format binary_little_endian 1.0\n"
            f"element vertex {nv}\n"
            "property float x\n"
            "property float y\n"
            "property float z\n"
            "property uchar red\n"
            "property uchar green\n"
            "property uchar blue\n"
            f"element face {nf}\n"
            "property list int int vertex_index\n"
            "end_header\n").encode("ascii")
    rec = np.zeros(nv, dtype=[("p", "<f4", 3), ("c", "u1", 3)])   # :336-339, 15 bytes per vertex
    rec["p"] = vertices
    rec["c"] = colors
    padded = np.concatenate([np.full((nf, 1), 3, dtype="<i4"), faces.astype("<i4")], axis=1)   # :341-345
    return head + rec.tobytes() + np.ascontiguousarray(padded).tobytes()
