"""Sphere SDF, the input of the reference's examples/sphere.py: a 200^3 integer grid holding
(x-50)^2 + (y-50)^2 + (z-50)^2 - 25^2, iso value 0.  The reference's kernels give V=11766, F=23528 on it."""
import numpy as np

from _common import run_example

N, CENTRE, RADIUS = 200, 50, 25

if __name__ == "__main__":
    axis = np.arange(N, dtype=np.int64) - CENTRE
    sdf = (axis[:, None, None] ** 2 + axis[None, :, None] ** 2 + axis[None, None, :] ** 2) - RADIUS ** 2
    run_example("sphere", sdf, 0, expected=(11766, 23528))
