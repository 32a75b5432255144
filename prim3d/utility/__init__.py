"""`prim3d.utility` of the reference (prim3d/utility/__init__.py:2-13): the same four names, served by the MI355X build."""
from primitive3d_amd import create_raycaster, marching_cubes, marching_tetrahedras, save_mesh  # noqa: F401

__all__ = ["create_raycaster", "marching_cubes", "save_mesh", "marching_tetrahedras"]
