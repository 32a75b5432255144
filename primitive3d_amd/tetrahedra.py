"""`marching_tetrahedras` with the reference's signature and results (prim3d/utility/marching_tetrahedras.py:89-235),
computed by the HIP library libp3dmt.so (include/p3d_mt.h) instead of a chain of ~25 PyTorch ops.

Same behaviour as the reference:
  * `tets` is corrected IN PLACE (corners 0 and 1 of negatively oriented tetrahedra are swapped, :147-148);
  * vertices come in the order of torch.unique's sorted edge rows, faces are int64 with the one-triangle tetrahedra
    first, `return_tet_idx=True` adds the tetrahedron of every face;
  * the result is differentiable w.r.t. `vertices` and `sdf`: when either requires a gradient the interpolation
    (:178-190) is re-done with torch ops on the endpoint pairs the library returns (the topology itself carries no
    gradient in the reference either: it is computed under no_grad, :150).
Inputs on the CPU are moved to the GPU and the results moved back (the reference runs its torch ops wherever the
tensors live, examples/sphere_tetrahedra.py:15-22); there is no CPU implementation here, and no fallback.
"""
import ctypes
from ctypes import POINTER, byref, c_char_p, c_int, c_int64, c_size_t, c_void_p
from typing import Tuple

import torch

from ._build import mt_path

_LIB = None
SYMBOLS = ("p3d_mt_abi_version", "p3d_mt_last_error", "p3d_mt_workspace_bytes", "p3d_mt_prepare", "p3d_mt_emit")


def lib():
    global _LIB
    if _LIB is None:
        path = mt_path()
        if not path.exists():
            raise ImportError(f"{path} is missing: build it with `python primitive3d_amd/_build.py`; there is no fallback")
        L = ctypes.CDLL(str(path))
        L.p3d_mt_abi_version.restype = c_int
        L.p3d_mt_last_error.restype = c_char_p
        L.p3d_mt_workspace_bytes.argtypes = [c_int64, c_int64, POINTER(c_size_t)]
        L.p3d_mt_prepare.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, POINTER(c_int64),
                                     POINTER(c_int64), c_void_p]
        L.p3d_mt_emit.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        for name in SYMBOLS:
            getattr(L, name)
        _LIB = L
    return _LIB


def _check(rc, what):
    if rc == -4:   # P3D_MT_EINDEX: the reference's `vertices[tets]` raises IndexError (marching_tetrahedras.py:60)
        raise IndexError(f"{what}: {lib().p3d_mt_last_error().decode()}")
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().p3d_mt_last_error().decode()}")


def marching_tetrahedras(vertices: torch.Tensor, tets: torch.Tensor, sdf: torch.Tensor,
                         return_tet_idx: bool = False) -> Tuple[torch.Tensor]:
    """vertices [N,3] float32, tets [T,4] int64 (corrected in place), sdf [N] float32 ->
    (verts [V,3] float32, faces [F,3] int64[, tet_idx [F] int64])."""
    if vertices.dim() != 2 or vertices.shape[1] != 3 or tets.dim() != 2 or tets.shape[1] != 4 or sdf.dim() != 1:
        raise ValueError("expected vertices [N,3], tets [T,4], sdf [N]")
    if vertices.dtype != torch.float32 or sdf.dtype != torch.float32:
        raise TypeError("the HIP marching tetrahedra take float32 vertices and sdf")
    if tets.dtype != torch.int64:
        raise TypeError("tets must be int64 (torch.long), as in the reference's example")
    if not torch.cuda.is_available():
        raise RuntimeError("marching_tetrahedras needs a GPU: there is no CPU implementation in this build")
    home = vertices.device
    dev = home if home.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
    v = vertices.detach().to(dev).contiguous()
    s = sdf.detach().to(dev).contiguous()
    t = tets if (tets.is_cuda and tets.is_contiguous() and tets.device == dev) else tets.to(dev).contiguous()
    n, nt = v.shape[0], t.shape[0]
    L = lib()
    with torch.cuda.device(dev):
        stream = c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        nbytes = c_size_t(0)
        _check(L.p3d_mt_workspace_bytes(n, nt, byref(nbytes)), "p3d_mt_workspace_bytes")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        nv, nf = c_int64(0), c_int64(0)
        _check(L.p3d_mt_prepare(c_void_p(v.data_ptr()), n, c_void_p(t.data_ptr()), nt, c_void_p(s.data_ptr()),
                                c_void_p(ws.data_ptr()), byref(nv), byref(nf), stream), "p3d_mt_prepare")
        verts = torch.empty((nv.value, 3), dtype=torch.float32, device=dev)
        pairs = torch.empty((nv.value, 2), dtype=torch.int64, device=dev)
        faces = torch.empty((nf.value, 3), dtype=torch.int64, device=dev)
        tet_idx = torch.empty((nf.value,), dtype=torch.int64, device=dev) if return_tet_idx else None
        _check(L.p3d_mt_emit(c_void_p(v.data_ptr()), c_void_p(t.data_ptr()), c_void_p(s.data_ptr()),
                             c_void_p(ws.data_ptr()), c_void_p(verts.data_ptr()), c_void_p(pairs.data_ptr()),
                             c_void_p(faces.data_ptr()), c_void_p(tet_idx.data_ptr()) if return_tet_idx else None, stream),
               "p3d_mt_emit")
    if t is not tets:  # the orientation fix reaches the caller's tensor, as in the reference (:148)
        tets.copy_(t.to(tets.device))
    if vertices.requires_grad or sdf.requires_grad:
        # the reference's differentiable part (:178-190) on the endpoint pairs, where the inputs live
        p = pairs.to(home)
        e = vertices[p]
        es = sdf[p].clone()
        es[:, -1] *= -1
        den = es.sum(1, keepdim=True)
        verts = (e * (torch.flip(es, [1]) / den)[..., None]).sum(1)
    else:
        verts = verts.to(home)
    faces = faces.to(home)
    if return_tet_idx:
        return verts, faces, tet_idx.to(home)
    return verts, faces
