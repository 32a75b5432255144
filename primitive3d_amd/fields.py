"""Synthetic scalar fields for tests and bench.  torch ops only, so the same
code runs on the CPU (tests, oracle inputs) and on the device (bench); values are fp32.

  sphere_grid(n)            examples/sphere.py:8-9 recipe at size n (centre n/4, radius n/8), int64
  perlin_grid(n, ...)       C3/C4: single-octave Perlin gradient noise, lattice period 64 voxels,
                            quintic fade, unit gradients from numpy default_rng(seed), samples at
                            (i+0.5)/period; generated per axis-0 slab so a rank can synthesise only
                            its own planes [x0, x1)
"""
import numpy as np
import torch


def sphere_grid(n: int = 200) -> np.ndarray:
    x, y, z = np.mgrid[:n, :n, :n]
    c, r = n // 4, n // 8
    return (x - c) ** 2 + (y - c) ** 2 + (z - c) ** 2 - r ** 2


def _fade(t):
    return t * t * t * (t * (t * 6.0 - 15.0) + 10.0)


def perlin_lattice(n_cells, seed: int = 0) -> torch.Tensor:
    """Unit gradient vectors on an (n_cells+1)^3 lattice (fp32, CPU)."""
    if isinstance(n_cells, int):
        n_cells = (n_cells,) * 3
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n_cells[0] + 1, n_cells[1] + 1, n_cells[2] + 1, 3))
    g /= np.linalg.norm(g, axis=-1, keepdims=True)
    return torch.from_numpy(g.astype(np.float32))


def perlin_grid(shape, period: int = 64, seed: int = 0, octaves: int = 1, persistence: float = 0.5,
                device="cpu", x0: int = 0, x1: int = None, slab: int = 32, dtype=torch.float32) -> torch.Tensor:
    """[x1-x0, ry, rz] planes of the Perlin field of full size `shape` (so slabs of one field agree)."""
    if isinstance(shape, int):
        shape = (shape,) * 3
    rx, ry, rz = shape
    x1 = rx if x1 is None else x1
    dev = torch.device(device)
    out = torch.empty((x1 - x0, ry, rz), dtype=dtype, device=dev)
    amp, per = 1.0, float(period)
    first = True
    for o in range(octaves):
        cells = tuple(int(np.ceil(s / per)) + 1 for s in shape)
        lat = perlin_lattice(cells, seed + o).to(dev)
        py = (torch.arange(ry, device=dev, dtype=torch.float32) + 0.5) / per
        pz = (torch.arange(rz, device=dev, dtype=torch.float32) + 0.5) / per
        iy, iz = py.floor().long(), pz.floor().long()
        fy, fz = (py - iy)[None, :, None], (pz - iz)[None, None, :]
        for s0 in range(x0, x1, slab):
            s1 = min(s0 + slab, x1)
            px = (torch.arange(s0, s1, device=dev, dtype=torch.float32) + 0.5) / per
            ix = px.floor().long()
            fx = (px - ix)[:, None, None]
            acc = None
            for dx in (0, 1):
                wx = _fade(fx) if dx else 1.0 - _fade(fx)
                for dy in (0, 1):
                    wy = _fade(fy) if dy else 1.0 - _fade(fy)
                    for dz in (0, 1):
                        wz = _fade(fz) if dz else 1.0 - _fade(fz)
                        g = lat[(ix + dx)[:, None, None], (iy + dy)[None, :, None], (iz + dz)[None, None, :]]
                        dot = g[..., 0] * (fx - dx) + g[..., 1] * (fy - dy) + g[..., 2] * (fz - dz)
                        term = dot * (wx * wy * wz)
                        acc = term if acc is None else acc + term
            blk = out[s0 - x0:s1 - x0]
            if first:
                blk.copy_((acc * amp).to(dtype))
            else:
                blk.add_((acc * amp).to(dtype))
        first = False
        amp *= persistence
        per /= 2.0
    return out
