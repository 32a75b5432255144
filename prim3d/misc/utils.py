"""`prim3d.misc.utils` of the reference: Timer, TimerError and the (there unused) copy of scale_to_bound."""
from primitive3d_amd.marching_cubes import scale_to_bound  # noqa: F401
from primitive3d_amd.misc import Timer, TimerError  # noqa: F401
