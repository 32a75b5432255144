"""ctypes loader for oracle/libmc_oracle.so (the C restatement, oracle/mc_oracle.c).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
import ctypes
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB = None


def build_oracle(force: bool = False) -> Path:
    so = _DIR / "libmc_oracle.so"
    src = _DIR / "mc_oracle.c"
    inc = _DIR.parent / "primitive3d_amd" / "csrc" / "tri_table_packed.inc"
    if force or not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, inc.stat().st_mtime):
        subprocess.check_call(["make", "-C", str(_DIR), "-B", "libmc_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(str(build_oracle()))
        i64, f32 = ctypes.c_int64, ctypes.c_float
        p = ctypes.c_void_p
        lib.p3d_oracle_count.argtypes = [p, i64, i64, i64, f32, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        lib.p3d_oracle_count.restype = ctypes.c_int
        lib.p3d_oracle_extract.argtypes = [p, i64, i64, i64, f32, p, p, p, p, p]
        lib.p3d_oracle_extract.restype = ctypes.c_int
        lib.p3d_oracle_extract_mt.argtypes = [p, i64, i64, i64, f32, p, p, p, p, p, ctypes.c_int]
        lib.p3d_oracle_extract_mt.restype = ctypes.c_int
        lib.p3d_oracle_tri_table.argtypes = [p]
        lib.p3d_oracle_tri_table.restype = None
        _LIB = lib
    return _LIB


def _as_f32_grid(grid) -> np.ndarray:
    g = np.ascontiguousarray(np.asarray(grid), dtype=np.float32)
    assert g.ndim == 3
    return g


def oracle_tri_table() -> np.ndarray:
    out = np.zeros((256, 16), dtype=np.int8)
    _lib().p3d_oracle_tri_table(out.ctypes.data)
    return out


def oracle_count(grid, thresh: float):
    """(V, F) exactly as count_vertices_faces_kernel + host driver compute them
    (marching_cubes.cu:4-68, :251-252)."""
    g = _as_f32_grid(grid)
    v, f3 = ctypes.c_int64(0), ctypes.c_int64(0)
    rc = _lib().p3d_oracle_count(g.ctypes.data, g.shape[0], g.shape[1], g.shape[2],
                                 np.float32(thresh), ctypes.byref(v), ctypes.byref(f3))
    assert rc == 0
    return v.value, f3.value // 3


def oracle_extract(grid, thresh: float, lower=None, upper=None, threads: int = 1, counts=None, want_keys=True):
    """Returns (vertices f32 [V,3], faces i32 [F,3], vkeys i64 [V]); deterministic order
    (vertices by (voxel, axis), faces by (cell, table slot)).  threads > 1 (0 = all host cores) runs the OpenMP
    driver p3d_oracle_extract_mt, whose output is identical; `counts` = (V, F) skips the counting pass."""
    g = _as_f32_grid(grid)
    if lower is None:
        lower = [0.0, 0.0, 0.0]
    if upper is None:
        upper = [float(s) for s in g.shape]
    lo = np.asarray(lower, dtype=np.float32)
    up = np.asarray(upper, dtype=np.float32)
    nv, nf = counts if counts is not None else oracle_count(g, thresh)
    verts = np.zeros((nv, 3), dtype=np.float32)
    keys = np.zeros((nv,), dtype=np.int64) if want_keys else None
    faces = np.zeros((nf, 3), dtype=np.int32)
    kp = keys.ctypes.data if want_keys else None
    if threads == 1:
        rc = _lib().p3d_oracle_extract(g.ctypes.data, g.shape[0], g.shape[1], g.shape[2],
                                       np.float32(thresh), lo.ctypes.data, up.ctypes.data,
                                       verts.ctypes.data, kp, faces.ctypes.data)
        assert rc == 0
    else:
        rc = _lib().p3d_oracle_extract_mt(g.ctypes.data, g.shape[0], g.shape[1], g.shape[2],
                                          np.float32(thresh), lo.ctypes.data, up.ctypes.data,
                                          verts.ctypes.data, kp, faces.ctypes.data, int(threads))
        assert rc > 0
        oracle_extract.last_threads = rc
    return verts, faces, keys


def canonical_mesh(verts, faces, vkeys):
    """Order-independent form: vertices sorted by edge key; faces rewritten as triples of edge keys
    (slot order kept -- no rotation, winding matters) and sorted lexicographically.
    Returns (keys_sorted i64 [V], verts_sorted f32 [V,3], face_keys_sorted i64 [F,3])."""
    verts = np.asarray(verts)
    faces = np.asarray(faces).astype(np.int64)
    vkeys = np.asarray(vkeys).astype(np.int64)
    order = np.argsort(vkeys, kind="stable")
    fk = vkeys[faces] if faces.size else np.zeros((0, 3), dtype=np.int64)
    if fk.size:
        fo = np.lexsort((fk[:, 2], fk[:, 1], fk[:, 0]))
        fk = fk[fo]
    return vkeys[order], verts[order], fk
