#!/usr/bin/env python3
"""Capture the behaviour of the REFERENCE Python wrapper (prim3d/utility/marching_cubes.py) as data.

Runs in the build container only: it imports the reference file from /root/reference with a stub
`prim3d.libPrim3D` in sys.modules (the real module needs CUDA) and records, for a list of inputs,
what the wrapper hands to `_C.marching_cubes` / which exception it raises / what it prints.  The
output (tests/golden/wrapper_cases.json) is committed; the reference source is not.
"""
import contextlib
import importlib.util
import io
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference/prim3d/utility/marching_cubes.py")
OUT = Path(__file__).resolve().parents[1] / "tests" / "golden" / "wrapper_cases.json"


def load_reference():
    calls = []
    stub = types.ModuleType("prim3d.libPrim3D")

    def marching_cubes(grid, thresh, lower, upper):
        calls.append({"dtype": str(grid.dtype), "shape": list(grid.shape), "contiguous": bool(grid.is_contiguous()),
                      "thresh": thresh, "thresh_type": type(thresh).__name__,
                      "lower": [float(v) for v in lower], "upper": [float(v) for v in upper],
                      "lower_types": [type(v).__name__ for v in lower], "upper_types": [type(v).__name__ for v in upper]})
        return torch.zeros((0, 3)), torch.zeros((0, 3), dtype=torch.int32)

    stub.marching_cubes = marching_cubes
    stub.save_mesh_as_ply = lambda *a: calls.append({"save": [str(a[0]), str(a[1].dtype), str(a[2].dtype), str(a[3].dtype),
                                                                 a[3].flatten()[:3].tolist()]})
    pkg = types.ModuleType("prim3d")
    pkg.libPrim3D = stub
    sys.modules["prim3d"] = pkg
    sys.modules["prim3d.libPrim3D"] = stub
    spec = importlib.util.spec_from_file_location("ref_mc", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, calls


def jsonable(v):
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    return v


def main():
    mod, calls = load_reference()
    # the GPU branch must be reachable without a GPU: fake availability and make .cuda() the identity
    torch.cuda.is_available = lambda: True
    torch.Tensor.cuda = lambda self, *a, **k: self

    out = {"scale_to_bound": [], "marching_cubes": [], "save_mesh": []}

    scale_inputs = [
        ("float", "2.0"), ("int", "2"), ("list3", "[1.0, 2.0, 3.0]"), ("tuple2_floats", "(0.5, 1.5)"),
        ("pair_of_triples", "([0, 0, 0], [1, 2, 3])"), ("len4", "[1.0, 2.0, 3.0, 4.0]"), ("str", "'abc'"),
        ("np_f32_pair", "(np.float32(0), np.float32(1))"), ("ndarray3", "np.array([1.0, 2.0, 3.0])"),
        ("tensor3", "torch.tensor([1.0, 2.0, 3.0])"), ("tuple2_ints", "(0, 1)"), ("none_in_list", "[None]"),
    ]
    for name, expr in scale_inputs:
        rec = {"name": name, "expr": expr}
        try:
            lo, up = mod.scale_to_bound(eval(expr))
            rec["lower"] = [float(v) for v in lo]
            rec["upper"] = [float(v) for v in up]
        except Exception as e:  # noqa: BLE001
            rec["raises"] = type(e).__name__
        out["scale_to_bound"].append(rec)

    mc_inputs = [
        ("int64_ndarray", "np.zeros((4, 5, 6), dtype=np.int64)", "0", {}),
        ("f64_tensor_scale_float", "torch.zeros((4, 5, 6), dtype=torch.float64)", "0.5", {"scale": 2.0}),
        ("f32_tensor_scale_box", "torch.zeros((3, 3, 3))", "0.25", {"scale": ([0.0, 1.0, 2.0], [3.0, 4.0, 5.0])}),
        ("permuted_noncontiguous", "torch.zeros((4, 5, 6)).permute(2, 1, 0)", "0", {}),
        ("dim_lt_2", "torch.zeros((1, 5, 6))", "0", {}),
        ("verbose", "torch.zeros((2, 2, 2))", "0", {"verbose": True}),
        ("cpu_mode_no_mcubes", "torch.zeros((2, 2, 2))", "0", {"cpu": True}),
        ("scale_int_scalar", "torch.zeros((2, 2, 2))", "0", {"scale": 2}),
        ("f16_tensor", "torch.zeros((2, 3, 4), dtype=torch.float16)", "1", {}),
    ]
    for name, gexpr, texpr, kw in mc_inputs:
        rec = {"name": name, "grid": gexpr, "thresh": texpr, "kwargs": jsonable(kw)}
        calls.clear()
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                v, f = mod.marching_cubes(eval(gexpr), eval(texpr), **kw)
            rec["call"] = calls[0] if calls else None
        except Exception as e:  # noqa: BLE001
            rec["raises"] = type(e).__name__
            rec["message"] = str(e)
        rec["stdout"] = buf.getvalue()
        out["marching_cubes"].append(rec)

    sm_inputs = [
        ("defaults", {"filename": "a.ply"}),
        ("path_and_verbose", {"filename": "Path('b.ply')", "verbose": True}),
        ("not_ply", {"filename": "c.obj"}),
        ("np_inputs_colors", {"filename": "d.ply", "np": True}),
    ]
    for name, kw in sm_inputs:
        calls.clear()
        rec = {"name": name, "kwargs": dict(kw)}
        buf = io.StringIO()
        try:
            v = torch.zeros((2, 3))
            f = torch.zeros((1, 3), dtype=torch.int64)
            args = {}
            if kw.get("np"):
                v, f = v.numpy(), f.numpy()
                args["colors"] = np.full((2, 3), 200.7)
            fn = eval(kw["filename"]) if kw["filename"].startswith("Path") else kw["filename"]
            with contextlib.redirect_stdout(buf):
                mod.save_mesh(v, f, filename=fn, verbose=kw.get("verbose", False), **args)
            rec["call"] = calls[0] if calls else None
        except Exception as e:  # noqa: BLE001
            rec["raises"] = type(e).__name__
        rec["stdout"] = buf.getvalue()
        out["save_mesh"].append(rec)

    OUT.parent.mkdir(parents=True, exist_ok=True)
    OUT.write_text(json.dumps(out, indent=1))
    print("wrote", OUT)


if __name__ == "__main__":
    from pathlib import Path  # noqa: F811  (used by eval above)
    main()
