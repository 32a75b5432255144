"""`prim3d.misc` of the reference (prim3d/misc/__init__.py:2-4)."""
from primitive3d_amd.misc import Timer, TimerError  # noqa: F401

__all__ = ["Timer"]
