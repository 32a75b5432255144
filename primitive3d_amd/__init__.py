"""primitive3d_amd -- MI355X-native build of Primitive3D's marching-cubes hot path (and its sibling extractor,
marching tetrahedra).

Public surface mirrors the reference package for this path (prim3d/__init__.py:4-16):
`marching_cubes`, `save_mesh`, `marching_tetrahedras`, `create_raycaster`, `Timer`, `ENABLE_OPTIX`, `__version__`; the native module keeps
the reference's name `libPrim3D`.  Importing this package REQUIRES the built native artefacts
(libp3dmc.so + libPrim3D*.so, see _build.py); there is no Python or CPU fallback.
"""
from ._build import capi_path, pybind_path

if not capi_path().exists() or not pybind_path().exists():
    raise ImportError(
        "primitive3d_amd native libraries are not built "
        f"({capi_path().name}, {pybind_path().name}); run `python primitive3d_amd/_build.py` "
        "or `__graft_entry__.build()`. There is no fallback path.")

from . import libPrim3D  # noqa: E402  (pybind adapter over the C ABI)
from .marching_cubes import marching_cubes, marching_cubes_batched, save_mesh, scale_to_bound  # noqa: E402
from .ray_cast import create_raycaster  # noqa: E402
from .tetrahedra import marching_tetrahedras  # noqa: E402  (HIP library libp3dmt.so, loaded on first use)
from .misc import Timer  # noqa: E402

__version__ = "0.1.0"
ENABLE_OPTIX = libPrim3D.enable_optix

__all__ = ["__version__", "ENABLE_OPTIX", "Timer", "marching_cubes", "marching_cubes_batched", "save_mesh",
           "marching_tetrahedras", "create_raycaster", "scale_to_bound", "libPrim3D"]
