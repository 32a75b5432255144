"""`prim3d.utility.ray_cast` of the reference (module path kept for callers that import it directly)."""
from primitive3d_amd.ray_cast import create_raycaster  # noqa: F401
