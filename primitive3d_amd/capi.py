"""ctypes binding of the C ABI (include/p3d_mc.h) -- the same entry points the pybind adapter
(csrc/bindings.cpp) calls, exposed so tests/bench can drive the library directly on raw device
pointers.  Loading fails loudly if libp3dmc.so has not been built: there is no fallback path.
"""
import ctypes
from ctypes import POINTER, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

from ._build import capi_path

P3D_F32, P3D_F16 = 0, 1
P3D_OK = 0
P3D_ERANGE = -2

# every symbol include/p3d_mc.h declares
SYMBOLS = ("p3d_mc_abi_version", "p3d_last_error", "p3d_mc_workspace_bytes", "p3d_mc_count", "p3d_mc_count_scan",
           "p3d_mc_read_counts", "p3d_mc_read_counts_ex", "p3d_mc_emit", "p3d_mc_plane_records", "p3d_mc_export_plane_records", "p3d_mc_profile_enable",
           "p3d_mc_profile_read", "p3d_mc_profile_stage_name", "p3d_mc_extract_fused", "p3d_mc_debug_layout",
           "p3d_mc_workspace_bytes_batched", "p3d_mc_extract_fused_batched", "p3d_mc_reload_tuning",
           "p3d_mc_debug_counters", "p3d_mc_release_stream", "p3d_mc_shutdown", "p3d_mc_dev_hooks")


class Slab(ctypes.Structure):
    """p3d_mc_slab (include/p3d_mc.h)."""
    _fields_ = [("halo_last_plane", c_int32), ("part", c_int32), ("vertex_id_base", c_int64),
                ("halo_vertex_id_base", c_int64), ("x_origin", c_int64), ("split_plane", c_int64),
                ("rank_counts", c_void_p), ("rank", c_int32), ("rank_counts_stride", c_int32),
                ("export_first_plane_to", c_void_p), ("defer_totals", c_int32), ("reserved", c_int32),
                ("region_first_rows", c_void_p)]


class P3DError(RuntimeError):
    code = 0   # the P3D_E* value the call returned


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = capi_path()
        if not path.exists():
            raise ImportError(f"{path} is missing: build it with `python primitive3d_amd/_build.py` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        L = ctypes.CDLL(str(path))
        L.p3d_mc_abi_version.restype = c_int
        L.p3d_last_error.restype = c_char_p
        L.p3d_mc_workspace_bytes.argtypes = [c_int64, c_int64, c_int64, POINTER(c_size_t)]
        L.p3d_mc_count.argtypes = [c_void_p, c_int, c_int64, c_int64, c_int64, c_float, POINTER(Slab), c_void_p, c_void_p]
        L.p3d_mc_count_scan.argtypes = L.p3d_mc_count.argtypes
        L.p3d_mc_read_counts.argtypes = [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int32), c_void_p]
        L.p3d_mc_read_counts_ex.argtypes = [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int32), POINTER(c_int64 * 32),
                                            c_void_p]
        L.p3d_mc_emit.argtypes = [c_void_p, c_int, c_int64, c_int64, c_int64, c_float, POINTER(c_float * 3),
                                  POINTER(c_float * 3), POINTER(c_int64 * 3), POINTER(Slab), c_void_p, c_void_p,
                                  c_int64, c_void_p, c_int64, c_void_p, c_void_p]
        L.p3d_mc_plane_records.argtypes = [c_void_p, c_int64, c_int64, c_int64, c_int64, POINTER(c_void_p),
                                           POINTER(c_size_t)]
        L.p3d_mc_extract_fused.argtypes = [c_void_p, c_int, c_int64, c_int64, c_int64, c_float, POINTER(c_float * 3),
                                           POINTER(c_float * 3), POINTER(c_int64 * 3), POINTER(Slab), c_void_p,
                                           c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p]
        L.p3d_mc_workspace_bytes_batched.argtypes = [c_int64, c_int64, c_int64, c_int64, POINTER(c_size_t)]
        L.p3d_mc_extract_fused_batched.argtypes = [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_float,
                                                   POINTER(c_float * 3), POINTER(c_float * 3), c_void_p, c_void_p,
                                                   c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]
        L.p3d_mc_debug_layout.argtypes = [c_int64, c_int64, c_int64, POINTER(c_size_t), POINTER(c_size_t),
                                          POINTER(c_int64), POINTER(c_int32)]
        L.p3d_mc_debug_counters.argtypes = [POINTER(c_int64), c_int]
        L.p3d_mc_release_stream.argtypes = [c_void_p]
        L.p3d_mc_profile_enable.argtypes = [c_int]
        L.p3d_mc_profile_read.argtypes = [POINTER(c_float), c_int]
        L.p3d_mc_profile_stage_name.argtypes = [c_int]
        L.p3d_mc_profile_stage_name.restype = c_char_p
        for name in SYMBOLS:
            getattr(L, name)  # raises AttributeError if the .so lacks a declared symbol
            if name not in ("p3d_last_error", "p3d_mc_profile_stage_name"):
                getattr(L, name).restype = getattr(L, name).restype or c_int
        _LIB = L
    return _LIB


def _check(rc, what):
    if rc != P3D_OK:
        e = P3DError(f"{what} failed ({rc}): {lib().p3d_last_error().decode()}")
        e.code = rc
        raise e


def workspace_bytes(rx, ry, rz) -> int:
    n = c_size_t(0)
    _check(lib().p3d_mc_workspace_bytes(rx, ry, rz, byref(n)), "p3d_mc_workspace_bytes")
    return n.value


def _dtype_code(t):
    import torch
    if t.dtype == torch.float32:
        return P3D_F32
    if t.dtype == torch.float16:
        return P3D_F16
    raise TypeError(f"unsupported grid dtype {t.dtype}")


def _stream_ptr(t):
    import torch
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _on_device_of(t):
    """The library keys its per-device state (cursor ring, result mailbox) by the CURRENT device: make the tensor's
    device current around every call, as the pybind adapter does with a device guard."""
    import torch
    return torch.cuda.device(t.device)


def count(grid, thresh, ws, slab=None, scan=False):
    """p3d_mc_count on a contiguous device tensor [rx,ry,rz] (one pass, the one-pass kernels in count-only form);
    scan=True: p3d_mc_count_scan (deterministic dense ids by prefix scan: the renumbering fallback and the parity tests'
    keyed path)."""
    assert grid.is_cuda and grid.is_contiguous() and grid.dim() == 3
    rx, ry, rz = grid.shape
    fn, what = (lib().p3d_mc_count_scan, "p3d_mc_count_scan") if scan else (lib().p3d_mc_count, "p3d_mc_count")
    with _on_device_of(grid):
        _check(fn(c_void_p(grid.data_ptr()), _dtype_code(grid), rx, ry, rz, c_float(thresh),
                  byref(slab) if slab is not None else None, c_void_p(ws.data_ptr()), _stream_ptr(grid)), what)


def read_counts(ws, with_flags=False):
    v, f, o = c_int64(0), c_int64(0), c_int32(0)
    with _on_device_of(ws):
        _check(lib().p3d_mc_read_counts(c_void_p(ws.data_ptr()), byref(v), byref(f), byref(o), _stream_ptr(ws)),
               "p3d_mc_read_counts")
    # flags: bit 0 = a scratch region overflowed (rewrite the vertices with emit), bit 1 = a region outgrew its 2^26
    # ids (renumber with count, then emit) -- include/p3d_mc.h
    return (v.value, f.value, int(o.value)) if with_flags else (v.value, f.value)


def read_counts_ex(ws):
    """p3d_mc_read_counts_ex: (V, F, flags, [32 region totals of the streaming kernel])."""
    nv, nf, over = c_int64(0), c_int64(0), c_int32(0)
    reg = (c_int64 * 32)()
    with _on_device_of(ws):
        _check(lib().p3d_mc_read_counts_ex(c_void_p(ws.data_ptr()), byref(nv), byref(nf), byref(over), byref(reg), _stream_ptr(ws)),
               "p3d_mc_read_counts_ex")
    return nv.value, nf.value, over.value, list(reg)


def region_layout(region_totals, spill_rows=None, extra=None):
    """41 ascending rows for p3d_mc_slab.region_first_rows from the region totals of an earlier call: every region gets exactly
    its total (+ `extra[r]`, tests), the eight spill areas behind them `spill_rows` rows in all (default: a tenth of the total
    + 4096).  Returns (ctypes uint32 array to keep alive, rows the vertex buffer needs)."""
    first = [0]
    for r, n in enumerate(region_totals):
        first.append(first[-1] + max(0, int(n) + (int(extra[r]) if extra is not None else 0)))
    total = first[-1]
    spill = total // 10 + 4096 if spill_rows is None else int(spill_rows)
    for g in range(8):
        first.append(first[-1] + (spill + 7 - g) // 8)
    return (ctypes.c_uint32 * 41)(*first), first[-1]


def emit(grid, thresh, lower, upper, ws, vertices, faces, vertex_keys=None, slab=None, full_res=None):
    rx, ry, rz = grid.shape
    lo = (c_float * 3)(*[float(v) for v in lower])
    up = (c_float * 3)(*[float(v) for v in upper])
    fr = (c_int64 * 3)(*[int(v) for v in full_res]) if full_res is not None else None
    capv = vertices.shape[0] if vertices is not None else 0
    capf = faces.shape[0] if faces is not None else 0
    with _on_device_of(grid):
        _check(lib().p3d_mc_emit(c_void_p(grid.data_ptr()), _dtype_code(grid), rx, ry, rz, c_float(thresh),
                                 byref(lo), byref(up), byref(fr) if fr is not None else None,
                                 byref(slab) if slab is not None else None, c_void_p(ws.data_ptr()),
                                 c_void_p(vertices.data_ptr()) if capv else None, capv,
                                 c_void_p(faces.data_ptr()) if capf else None, capf,
                                 c_void_p(vertex_keys.data_ptr()) if vertex_keys is not None and capv else None,
                                 _stream_ptr(grid)), "p3d_mc_emit")


def release_stream(stream_ptr):
    """p3d_mc_release_stream: free what the library keeps for a stream (a raw hipStream_t as an int) on the current device."""
    _check(lib().p3d_mc_release_stream(c_void_p(stream_ptr)), "p3d_mc_release_stream")


def shutdown():
    _check(lib().p3d_mc_shutdown(), "p3d_mc_shutdown")


def debug_counters():
    """p3d_mc_debug_counters: what the library has launched since it was loaded."""
    out = (c_int64 * 7)()
    n = lib().p3d_mc_debug_counters(out, 7)
    assert n == 7, n
    return {"streaming_launches": out[0], "layout_passes": out[1], "streaming_passes": out[2], "count_emit_calls": out[3],
            "emissions_without_a_pass": out[4], "stream_rings": out[5], "ring_bytes": out[6]}


def reload_tuning():
    """Developer / test hook: the library re-reads its P3D_* launch-shape knobs from the environment."""
    _check(lib().p3d_mc_reload_tuning(), "p3d_mc_reload_tuning")


def profile_enable(mode: int):
    _check(lib().p3d_mc_profile_enable(mode), "p3d_mc_profile_enable")


def profile_read():
    """{stage name: ms} for the stages recorded in the most recent count/emit pair."""
    buf = (c_float * 16)()
    n = lib().p3d_mc_profile_read(buf, 16)
    if n < 0:
        _check(n, "p3d_mc_profile_read")
    return {lib().p3d_mc_profile_stage_name(i).decode(): buf[i] for i in range(n) if buf[i] >= 0}


def scratch_rows_for(cap_vertices: int) -> int:
    """Rows of vertex scratch for an expected vertex count: 32 regions, 25 % + 256 rows of slack each."""
    per = (cap_vertices + 31) // 32
    return 32 * max(per + per // 4 + 256, min(cap_vertices, 8192))  # small outputs may land in very few regions


def extract_fused_raw(grid, thresh, lower, upper, ws, vertices, faces, slab=None, full_res=None, scratch=None):
    """p3d_mc_extract_fused: one pass over the field, writes at most the capacities of the two buffers.
    `scratch` ([rows,3] f32) is required when `vertices` is given; allocated here if omitted.  (Parts 3 and 4 of an
    extraction in several calls -- p3d_mc_slab.part -- take the scratch without a vertex buffer.)"""
    import torch
    if vertices is not None and vertices.shape[0] and scratch is None and not (slab is not None and slab.region_first_rows):
        scratch = torch.empty((scratch_rows_for(vertices.shape[0]), 3), dtype=torch.float32, device=grid.device)
    rx, ry, rz = grid.shape
    lo = (c_float * 3)(*[float(v) for v in lower])
    up = (c_float * 3)(*[float(v) for v in upper])
    fr = (c_int64 * 3)(*[int(v) for v in full_res]) if full_res is not None else None
    capv = vertices.shape[0] if vertices is not None else 0
    capf = faces.shape[0] if faces is not None else 0
    with _on_device_of(grid):
        _check(lib().p3d_mc_extract_fused(c_void_p(grid.data_ptr()), _dtype_code(grid), rx, ry, rz, c_float(thresh),
                                          byref(lo), byref(up), byref(fr) if fr is not None else None,
                                          byref(slab) if slab is not None else None, c_void_p(ws.data_ptr()),
                                          c_void_p(vertices.data_ptr()) if capv else None, capv,
                                          c_void_p(scratch.data_ptr()) if scratch is not None else None,
                                          scratch.shape[0] if scratch is not None else 0,
                                          c_void_p(faces.data_ptr()) if capf else None, capf, _stream_ptr(grid)),
               "p3d_mc_extract_fused")


class FusedCaller:
    """p3d_mc_extract_fused / p3d_mc_emit / p3d_mc_read_counts / p3d_mc_export_plane_records on ONE grid and workspace with
    everything that does not change between the parts of an extraction marshalled once: the grid, the box, the workspace, the
    stream (the current one when this object is made) -- the multi-GPU path makes four to six calls per extraction, and
    building the ctypes arguments (two `torch.cuda.current_stream` look-ups, three arrays, a device guard) cost more host
    time per call than the library spends in it.  The caller keeps the tensors alive and stays on the device and stream it
    was on (SlabExtractor does)."""

    def __init__(self, grid, thresh, lower, upper, ws, full_res=None):
        import torch
        assert grid.is_cuda and grid.is_contiguous() and grid.dim() == 3
        self.L = lib()
        self.dev = grid.device
        self.shape = tuple(int(n) for n in grid.shape)
        self.grid_p, self.ws_p, self.dtype = c_void_p(grid.data_ptr()), c_void_p(ws.data_ptr()), _dtype_code(grid)
        self.thresh = c_float(thresh)
        self.lo = (c_float * 3)(*[float(v) for v in lower])
        self.up = (c_float * 3)(*[float(v) for v in upper])
        self.fr = (c_int64 * 3)(*[int(v) for v in full_res]) if full_res is not None else None
        self.stream = c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        self._keep = (grid, ws)

    def _guard(self):
        import torch
        return torch.cuda.device(self.dev) if torch.cuda.current_device() != self.dev.index else _NoGuard

    def fused(self, slab, vertices, scratch, faces):
        capv = vertices.shape[0] if vertices is not None else 0
        capf = faces.shape[0] if faces is not None else 0
        rx, ry, rz = self.shape
        with self._guard():
            _check(self.L.p3d_mc_extract_fused(self.grid_p, self.dtype, rx, ry, rz, self.thresh, byref(self.lo), byref(self.up),
                                               byref(self.fr) if self.fr is not None else None,
                                               byref(slab) if slab is not None else None, self.ws_p,
                                               c_void_p(vertices.data_ptr()) if capv else None, capv,
                                               c_void_p(scratch.data_ptr()) if scratch is not None else None,
                                               scratch.shape[0] if scratch is not None else 0,
                                               c_void_p(faces.data_ptr()) if capf else None, capf, self.stream),
                   "p3d_mc_extract_fused")

    def emit(self, slab, vertices, faces):
        capv = vertices.shape[0] if vertices is not None else 0
        capf = faces.shape[0] if faces is not None else 0
        rx, ry, rz = self.shape
        with self._guard():
            _check(self.L.p3d_mc_emit(self.grid_p, self.dtype, rx, ry, rz, self.thresh, byref(self.lo), byref(self.up),
                                      byref(self.fr) if self.fr is not None else None,
                                      byref(slab) if slab is not None else None, self.ws_p,
                                      c_void_p(vertices.data_ptr()) if capv else None, capv,
                                      c_void_p(faces.data_ptr()) if capf else None, capf, None, self.stream), "p3d_mc_emit")

    def read_counts(self):
        v, f, o = c_int64(0), c_int64(0), c_int32(0)
        with self._guard():
            _check(self.L.p3d_mc_read_counts(self.ws_p, byref(v), byref(f), byref(o), self.stream), "p3d_mc_read_counts")
        return v.value, f.value, int(o.value)

    def export_plane_records(self, plane, out):
        rx, ry, rz = self.shape
        with self._guard():
            _check(self.L.p3d_mc_export_plane_records(self.ws_p, rx, ry, rz, plane, c_void_p(out.data_ptr()), self.stream),
                   "p3d_mc_export_plane_records")
        return out


class _NoGuardType:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NoGuard = _NoGuardType()


def workspace_bytes_batched(nitems, rx, ry, rz) -> int:
    n = c_size_t(0)
    _check(lib().p3d_mc_workspace_bytes_batched(nitems, rx, ry, rz, byref(n)), "p3d_mc_workspace_bytes_batched")
    return n.value


def extract_fused_batched_raw(grids, thresh, lower, upper, ws, vertices, scratch, faces, item_offsets):
    """p3d_mc_extract_fused_batched on a contiguous [B, rx, ry, rz] device tensor (f32 or f16); item_offsets is a
    [2*(B+1)] int64 device tensor."""
    assert grids.is_cuda and grids.is_contiguous() and grids.dim() == 4
    B, rx, ry, rz = grids.shape
    lo = (c_float * 3)(*[float(v) for v in lower])
    up = (c_float * 3)(*[float(v) for v in upper])
    capv = vertices.shape[0] if vertices is not None else 0
    capf = faces.shape[0] if faces is not None else 0
    with _on_device_of(grids):
        _check(lib().p3d_mc_extract_fused_batched(
            c_void_p(grids.data_ptr()), _dtype_code(grids), B, rx, ry, rz, c_float(thresh), byref(lo), byref(up),
            c_void_p(ws.data_ptr()), c_void_p(vertices.data_ptr()) if capv else None, capv,
            c_void_p(scratch.data_ptr()) if capv else None, scratch.shape[0] if capv else 0,
            c_void_p(faces.data_ptr()) if capf else None, capf, c_void_p(item_offsets.data_ptr()),
            _stream_ptr(grids)), "p3d_mc_extract_fused_batched")


def debug_layout(rx, ry, rz):
    ob, orr, nu, ncz = c_size_t(0), c_size_t(0), c_int64(0), c_int32(0)
    _check(lib().p3d_mc_debug_layout(rx, ry, rz, byref(ob), byref(orr), byref(nu), byref(ncz)), "p3d_mc_debug_layout")
    return {"off_bits": ob.value, "off_records": orr.value, "num_units": nu.value, "chunks_per_row": ncz.value}


def extract_fused(grid, thresh, lower=None, upper=None, cap_vertices=None, cap_faces=None, return_ws=False):
    """One-pass extraction with capacity guess + exact re-emit on overflow (what the pybind adapter does)."""
    import torch
    rx, ry, rz = grid.shape
    lower = [0.0, 0.0, 0.0] if lower is None else lower
    upper = [rx, ry, rz] if upper is None else upper
    nvox = rx * ry * rz
    capv = max(1024, nvox // 16) if cap_vertices is None else cap_vertices
    capf = 2 * capv if cap_faces is None else cap_faces
    ws = torch.empty(workspace_bytes(rx, ry, rz), dtype=torch.uint8, device=grid.device)
    verts = torch.empty((capv, 3), dtype=torch.float32, device=grid.device)
    faces = torch.empty((capf, 3), dtype=torch.int32, device=grid.device)
    extract_fused_raw(grid, thresh, lower, upper, ws, verts, faces)
    nv, nf, over = read_counts(ws, with_flags=True)
    if nv > capv or nf > capf or over:
        if over & 2:  # ambiguous one-pass ids: dense ids by the scan-numbered counting call
            count(grid, thresh, ws, scan=True)
            nv, nf = read_counts(ws)
        verts = torch.empty((nv, 3), dtype=torch.float32, device=grid.device)
        faces = torch.empty((nf, 3), dtype=torch.int32, device=grid.device)
        emit(grid, thresh, lower, upper, ws, verts, faces)
    out = (verts[:nv], faces[:nf])
    return out + (ws,) if return_ws else out


def plane_records(ws, rx, ry, rz, plane):
    ptr, n = c_void_p(0), c_size_t(0)
    _check(lib().p3d_mc_plane_records(c_void_p(ws.data_ptr()), rx, ry, rz, plane, byref(ptr), byref(n)),
           "p3d_mc_plane_records")
    return ptr.value, n.value


def export_plane_records(ws, rx, ry, rz, plane, out):
    """Dense vertex-id records of one plane -> `out` (uint8 tensor of bytes_per_plane bytes)."""
    with _on_device_of(ws):
        _check(lib().p3d_mc_export_plane_records(c_void_p(ws.data_ptr()), rx, ry, rz, plane,
                                                 c_void_p(out.data_ptr()), _stream_ptr(ws)),
               "p3d_mc_export_plane_records")
    return out


def extract(grid, thresh, lower=None, upper=None, with_keys=False, return_ws=False):
    """Whole two-phase call through the C ABI on a device tensor -- p3d_mc_count -> p3d_mc_read_counts -> exactly sized
    tensors -> p3d_mc_emit, the literal binding of INTEGRATION.md; returns (vertices, faces[, keys][, ws]).  with_keys:
    the scan-numbered pair (p3d_mc_count_scan + the gather emitter), which can write a vertex's edge key next to it."""
    import torch
    rx, ry, rz = grid.shape
    lower = [0.0, 0.0, 0.0] if lower is None else lower
    upper = [rx, ry, rz] if upper is None else upper
    ws = torch.empty(workspace_bytes(rx, ry, rz), dtype=torch.uint8, device=grid.device)
    count(grid, thresh, ws, scan=with_keys)
    nv, nf, over = read_counts(ws, with_flags=True)
    if over & 2 and not with_keys:   # a region numbered more than 2^26 vertices: renumber (include/p3d_mc.h)
        count(grid, thresh, ws, scan=True)
        nv, nf = read_counts(ws)
    verts = torch.empty((nv, 3), dtype=torch.float32, device=grid.device)
    faces = torch.empty((nf, 3), dtype=torch.int32, device=grid.device)
    keys = torch.empty((nv,), dtype=torch.int64, device=grid.device) if with_keys else None
    emit(grid, thresh, lower, upper, ws, verts, faces, keys)
    out = (verts, faces, keys) if with_keys else (verts, faces)
    return out + (ws,) if return_ws else out
