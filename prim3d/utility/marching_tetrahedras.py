"""`prim3d.utility.marching_tetrahedras` of the reference (module path kept for callers that import it directly)."""
from primitive3d_amd.tetrahedra import marching_tetrahedras  # noqa: F401
