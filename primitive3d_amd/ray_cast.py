"""`create_raycaster` with the reference's signature (prim3d/utility/ray_cast.py:6-27): the BVH build takes host tensors
(this build has no OptiX: `enable_optix` is False, so vertices and faces are moved to the CPU exactly as the reference's
wrapper does, :21-25) and returns the native `RayCaster`, whose `invoke(origins, directions, depths, normals,
primitives_ids)` fills the three caller-allocated CUDA tensors (nearest hit within 10 units; depth 10 / normal 0 /
id -1 on a miss).  Native side: csrc/p3d_rc.hip behind include/p3d_rc.h."""
import torch

from . import libPrim3D as _C


def create_raycaster(vertices: torch.Tensor, faces: torch.Tensor) -> _C.RayCaster:
    if _C.enable_optix:  # never true in this build; kept so that the function reads like the reference's
        vertices = vertices.cuda() if not vertices.is_cuda else vertices
        faces = faces.cuda() if not faces.is_cuda else faces
    else:
        vertices = vertices.cpu() if vertices.is_cuda else vertices
        faces = faces.cpu() if faces.is_cuda else faces
    return _C.create_raycaster(vertices, faces)
