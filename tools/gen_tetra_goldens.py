#!/usr/bin/env python3
"""Golden vectors for marching tetrahedra, produced by RUNNING THE REFERENCE in the build container.

The reference's prim3d/utility/marching_tetrahedras.py:89-235 is pure PyTorch: it is imported here by path (the rest
of the `prim3d` package needs the CUDA extension and is not imported) and run on the CPU.  Only inputs and outputs are
written (tests/golden/tetra_*.npz); no reference source travels.  Inputs:
  * the reference's own example data, examples/data/tetrahedra/{points,sdfs,tetrahedras}.npy
    (examples/sphere_tetrahedra.py:11-13) -- copied as data fixtures (MIT, attributed in tests/golden/README.md);
  * the docstring example (:118-136);
  * seeded random tetrahedral meshes (scipy Delaunay of random points, random orientation of the first two corners so
    that the reference's orientation fix, :147-148, has work to do) with smooth and with noisy scalar values.
Run:  python tools/gen_tetra_goldens.py
"""
import importlib.util
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
OUT = ROOT / "tests" / "golden"


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_marching_tetrahedras",
                                                  REF / "prim3d" / "utility" / "marching_tetrahedras.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def random_mesh(seed, npts, noisy):
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-1.0, 1.0, size=(npts, 3)).astype(np.float32)
    tets = Delaunay(pts.astype(np.float64)).simplices.astype(np.int64)
    # drop slivers: the orientation test is a float32 determinant, only its sign on well-shaped cells is portable
    p = pts[tets].astype(np.float64)
    vol = np.abs(np.linalg.det(p[:, 1:] - p[:, :1])) / 6.0
    tets = tets[vol > 1e-4]
    swap = rng.random(len(tets)) < 0.5
    tets[swap] = tets[swap][:, [1, 0, 2, 3]]
    r = np.linalg.norm(pts, axis=1)
    sdf = (r - 0.6).astype(np.float32) if not noisy else rng.standard_normal(npts).astype(np.float32)
    return pts, tets, sdf


def sliver_mesh(seed, jitter):
    """A mesh that KEEPS its slivers: Delaunay (qhull QJ) of a regular 12^3 lattice whose points are jittered by
    `jitter` (cells of the lattice are cospherical: thousands of nearly flat tetrahedra whose determinant is a few ulps
    of rounding noise) plus 500 random points.  Written as tests/golden/tetraslivers_*.npz: on these the reference's
    float32 `torch.det` (:50-65) and a float64 determinant may orient a nearly flat tet differently."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    lat = np.stack(np.meshgrid(*[np.linspace(-0.5, 0.5, 12)] * 3, indexing="ij"), -1).reshape(-1, 3)
    pts = (lat + rng.uniform(-jitter, jitter, size=lat.shape)).astype(np.float32)
    pts = np.concatenate([pts, rng.uniform(-1.0, 1.0, size=(500, 3)).astype(np.float32)])
    tets = Delaunay(pts.astype(np.float64), qhull_options="QJ").simplices.astype(np.int64)
    swap = rng.random(len(tets)) < 0.5
    tets[swap] = tets[swap][:, [1, 0, 2, 3]]
    sdf = (np.linalg.norm(pts, axis=1) - 0.45 + 0.05 * np.sin(7.0 * pts[:, 0])).astype(np.float32)
    return pts, tets, sdf


def run(ref, pts, tets, sdf):
    t = torch.from_numpy(tets.copy())
    v, f, ti = ref.marching_tetrahedras(torch.from_numpy(pts), t, torch.from_numpy(sdf), True)
    return dict(points=pts, tets=tets, sdf=sdf, tets_after=t.numpy(), verts=v.numpy(), faces=f.numpy(), tet_idx=ti.numpy())


def main():
    ref = load_reference()
    cases = {}
    d = REF / "examples" / "data" / "tetrahedra"
    ex = (np.load(d / "points.npy"), np.load(d / "tetrahedras.npy").astype(np.int64), np.load(d / "sdfs.npy"))
    cases["example_sphere"] = run(ref, *ex)
    cases["docstring"] = run(ref, np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32),
                             np.array([[0, 1, 2, 3]], np.int64), np.array([-1.0, -1.0, 0.5, 0.5], np.float32))
    cases["delaunay_smooth_400"] = run(ref, *random_mesh(1, 400, False))
    cases["delaunay_noisy_300"] = run(ref, *random_mesh(2, 300, True))
    cases["delaunay_noisy_2000"] = run(ref, *random_mesh(3, 2000, True))
    cases["all_outside"] = run(ref, ex[0][:50], np.array([[0, 1, 2, 3], [4, 5, 6, 7]], np.int64),
                               -np.ones(50, np.float32))
    for name, c in cases.items():
        np.savez_compressed(OUT / f"tetra_{name}.npz", **c)
        print(name, "V", c["verts"].shape, "F", c["faces"].shape, "flipped", int((c["tets_after"] != c["tets"]).any(1).sum()))
    # meshes with their slivers kept (separate file prefix: compared modulo the nearly flat tets, tests/test_tetra_cpu.py)
    for name, (seed, jitter) in {"jitter_1e-6": (5, 1e-6), "jitter_3e-8": (5, 3e-8)}.items():
        c = run(ref, *sliver_mesh(seed, jitter))
        np.savez_compressed(OUT / f"tetraslivers_{name}.npz", **c)
        p = c["points"][c["tets"]].astype(np.float64)
        det = np.linalg.det(p[:, 1:] - p[:, :1])
        print("slivers", name, "T", len(c["tets"]), "|det| < 1e-7:", int((np.abs(det) < 1e-7).sum()), "V", c["verts"].shape,
              "F", c["faces"].shape)
    return 0


if __name__ == "__main__":
    sys.exit(main())
