"""Seeded input grids shared by the CPU (oracle/golden) and GPU (parity) tests."""
import numpy as np
import torch

from primitive3d_amd.fields import perlin_grid, sphere_grid


def noise(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)


def plateau(shape, seed):
    """Many samples exactly equal to the iso value (inside test is strict >) and repeated values."""
    g = np.random.default_rng(seed).integers(-2, 3, size=shape).astype(np.float32)
    return g


def with_nans(shape, seed):
    g = noise(shape, seed)
    rng = np.random.default_rng(seed + 1)
    g[rng.random(shape) < 0.05] = np.nan
    return g


def with_infs(shape, seed):
    g = noise(shape, seed)
    rng = np.random.default_rng(seed + 2)
    m = rng.random(shape)
    g[m < 0.03] = np.inf
    g[m > 0.97] = -np.inf
    return g


# name -> (grid fp32 ndarray, thresh, lower, upper)   (lower/upper None = reference default)
def small_cases():
    c = {}
    c["sphere32"] = (sphere_grid(32).astype(np.float32), 0.0, None, None)
    c["sphere64"] = (sphere_grid(64).astype(np.float32), 0.0, None, None)
    c["noise_2x2x2"] = (noise((2, 2, 2), 1), 0.1, None, None)
    c["noise_5x7x9_box"] = (noise((5, 7, 9), 2), 0.0, [-1.0, 0.5, 2.0], [3.0, 4.5, 11.0])  # pins the :295 quirk
    c["noise_3x4x70"] = (noise((3, 4, 70), 3), -0.2, None, None)
    c["noise_9x5x129"] = (noise((9, 5, 129), 4), 0.3, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    c["noise_6x6x64"] = (noise((6, 6, 64), 5), 0.0, None, None)
    c["noise_6x6x65"] = (noise((6, 6, 65), 6), 0.0, None, None)
    c["noise_4x3x128"] = (noise((4, 3, 128), 7), 0.0, None, None)
    c["noise_2x300x2"] = (noise((2, 300, 2), 8), 0.0, None, None)
    c["noise_40x2x3"] = (noise((40, 2, 3), 9), 0.0, None, None)
    c["noise_33x17x200"] = (noise((33, 17, 200), 10), 0.5, [0.5, 0.5, 0.5], [1.5, 2.5, 3.5])
    c["plateau_12x11x70"] = (plateau((12, 11, 70), 11), 0.0, None, None)
    c["nan_10x9x66"] = (with_nans((10, 9, 66), 12), 0.0, None, None)
    c["inf_10x9x66"] = (with_infs((10, 9, 66), 13), 0.0, None, None)
    c["all_inside_4x4x4"] = (np.ones((4, 4, 4), np.float32), 0.0, None, None)
    c["all_outside_4x4x4"] = (np.zeros((4, 4, 4), np.float32), 0.0, None, None)
    # thresholds that no value exceeds / every value exceeds / that compare false with everything (cu:25: d > thresh)
    c["thresh_plus_inf"] = (noise((6, 5, 70), 14), float("inf"), None, None)
    c["thresh_minus_inf"] = (noise((6, 5, 70), 15), float("-inf"), None, None)
    c["thresh_nan"] = (noise((6, 5, 70), 16), float("nan"), None, None)
    # subnormal samples and a subnormal threshold: compares, differences and the IEEE divide must not flush to zero
    c["subnormal_7x6x66"] = ((noise((7, 6, 66), 17).astype(np.float64) * 1e-41).astype(np.float32), 0.0, None, None)
    c["subnormal_thresh_7x6x66"] = ((noise((7, 6, 66), 18).astype(np.float64) * 1e-41).astype(np.float32), 2e-42, None,
                                    None)
    # huge magnitudes: d1 - d0 overflows to inf on many edges (dt = finite / inf = 0)
    c["huge_6x6x66"] = (np.clip(noise((6, 6, 66), 19).astype(np.float64) * 3e38, -3.3e38, 3.3e38).astype(np.float32), 0.0,
                        None, None)
    c["perlin48"] = (perlin_grid(48, period=16, seed=3).numpy(), 0.0, None, None)
    c["perlin_40x24x96_thr"] = (perlin_grid((40, 24, 96), period=16, seed=4).numpy(), 0.05, [0, 0, 0], [2.0, 2.0, 2.0])
    return c
