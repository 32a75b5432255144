"""Drop-in alias: `import prim3d` resolves to the MI355X build (primitive3d_amd), so callers written
against lzhnb/Primitive3D (`prim3d.marching_cubes`, `prim3d.save_mesh`, `prim3d.Timer`,
`prim3d.libPrim3D`, `prim3d.marching_tetrahedras`) run unchanged for the marching-cubes path and marching
tetrahedra and the BVH ray caster (the OptiX ray caster is NVIDIA-specific and outside this build)."""
import sys

import primitive3d_amd as _impl
from primitive3d_amd import (ENABLE_OPTIX, Timer, __version__, create_raycaster, libPrim3D,  # noqa: F401
                             marching_cubes, marching_tetrahedras, save_mesh)

sys.modules[__name__ + ".libPrim3D"] = libPrim3D


__all__ = ["__version__", "ENABLE_OPTIX", "Timer", "create_raycaster", "marching_cubes", "save_mesh",
           "marching_tetrahedras"]
