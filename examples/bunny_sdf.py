"""Stanford-bunny SDF, the input of the reference's examples/bunny_sdf.py (its examples/data/bunny.npy, a 66^3 float
grid, kept here as tests/golden/bunny66.npy), iso value 0.  The reference's kernels give V=13282, F=26560 on it."""
import numpy as np

from _common import REPO, run_example

if __name__ == "__main__":
    sdf = np.load(REPO / "tests" / "golden" / "bunny66.npy")
    run_example("bunny", sdf, 0, expected=(13282, 26560))
