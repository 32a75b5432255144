"""Shared driver of the example scripts: extract on the GPU through the `prim3d` package, export a PLY, check the
counts against a recorded answer of the reference, and -- only if PyMCubes happens to be installed -- against it."""
import argparse
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:  # run from a source checkout without installing anything
    sys.path.insert(0, str(REPO))


def run_example(name, grid, iso, expected=None):
    """grid: numpy array (any dtype, as the reference's scripts pass it); expected: (V, F) or None."""
    import torch

    import prim3d

    ap = argparse.ArgumentParser(description=f"{name}: marching cubes on the MI355X build")
    ap.add_argument("--out", default=f"{name}.ply", help="PLY file to write ('' = skip the export)")
    ap.add_argument("--repeat", type=int, default=1, help="extract this many times (the later ones show steady state)")
    args = ap.parse_args()

    field = torch.tensor(grid).cuda()
    print(f"{name}: grid {tuple(field.shape)} {field.dtype}, iso {iso}")
    for it in range(args.repeat):
        with prim3d.Timer("extraction on the GPU: {:.6f}s"):
            verts, tris = prim3d.marching_cubes(field, iso, verbose=(it == 0))
            torch.cuda.synchronize()
    if args.out:
        with prim3d.Timer("PLY export: {:.6f}s"):
            prim3d.save_mesh(verts, tris, filename=args.out)
    got = (int(verts.shape[0]), int(tris.shape[0]))
    if expected is not None:
        assert got == tuple(expected), f"{name}: got {got}, the reference produces {tuple(expected)}"
        print(f"{name}: counts match the reference's recorded answer {tuple(expected)}")
    try:
        import mcubes  # the reference's CPU path; not a dependency of this package
    except ImportError:
        print("PyMCubes is not installed: no CPU cross-check")
    else:
        with prim3d.Timer("PyMCubes on the CPU: {:.6f}s"):
            cv, cf = mcubes.marching_cubes(grid, iso)
        assert got == (cv.shape[0], cf.shape[0]), (got, cv.shape, cf.shape)
    return verts, tris
