"""`prim3d.utility.marching_cubes` of the reference (module path kept for callers that import it directly)."""
from primitive3d_amd.marching_cubes import marching_cubes, save_mesh, scale_to_bound  # noqa: F401
