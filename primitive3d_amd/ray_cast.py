"""`create_raycaster(vertices, faces)` as in the reference (prim3d/utility/ray_cast.py:6-27): returns the native
`RayCaster`, whose `invoke(origins, directions, depths, normals, primitives_ids)` fills three caller-allocated CUDA
tensors (nearest hit within 10 units; depth 10 / normal 0 / id -1 on a miss).

This build has no OptiX (`libPrim3D.enable_optix` is False), so of the reference wrapper's two placements only one
exists: the BVH is built on the host, hence a mesh living on the GPU is brought to the CPU first (:21-25).  Native side:
csrc/p3d_rc.hip behind include/p3d_rc.h."""
import torch

from . import libPrim3D as _C


def _on_host(t: torch.Tensor) -> torch.Tensor:
    return t.cpu() if t.is_cuda else t


def create_raycaster(vertices: torch.Tensor, faces: torch.Tensor) -> _C.RayCaster:
    assert not _C.enable_optix, "this build has no OptiX path"
    return _C.create_raycaster(_on_host(vertices), _on_host(faces))
