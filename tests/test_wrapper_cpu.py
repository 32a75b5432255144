"""The Python wrapper (primitive3d_amd/marching_cubes.py) against the behaviour captured from the
reference wrapper (tests/golden/wrapper_cases.json <- tools/capture_wrapper_cases.py): same lower/
upper, dtype coercion, contiguity pass-through, exceptions and prints.  No GPU: `_C` is replaced by
a recorder and CUDA availability is faked, exactly as in the capture."""
import contextlib
import io
import json
from pathlib import Path

import numpy as np  # noqa: F401  (used by eval of the captured expressions)
import pytest
import torch

CASES = json.loads((Path(__file__).resolve().parent / "golden" / "wrapper_cases.json").read_text())


@pytest.fixture()
def wrapper(built, monkeypatch):
    import importlib
    w = importlib.import_module("primitive3d_amd.marching_cubes")  # (the package re-exports a function of that name)
    calls = []

    class Rec:
        @staticmethod
        def marching_cubes(grid, thresh, lower, upper):
            calls.append({"dtype": str(grid.dtype), "shape": list(grid.shape), "contiguous": bool(grid.is_contiguous()),
                          "thresh": thresh, "thresh_type": type(thresh).__name__,
                          "lower": [float(v) for v in lower], "upper": [float(v) for v in upper],
                          "lower_types": [type(v).__name__ for v in lower], "upper_types": [type(v).__name__ for v in upper]})
            return torch.zeros((0, 3)), torch.zeros((0, 3), dtype=torch.int32)

        @staticmethod
        def save_mesh_as_ply(fn, v, f, c):
            calls.append({"save": [str(fn), str(v.dtype), str(f.dtype), str(c.dtype), c.flatten()[:3].tolist()]})

    monkeypatch.setattr(w, "_C", Rec)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    return w, calls


@pytest.mark.parametrize("rec", CASES["scale_to_bound"], ids=lambda r: r["name"])
def test_scale_to_bound(wrapper, rec):
    w, _ = wrapper
    if "raises" in rec:
        with pytest.raises(eval(rec["raises"])):
            w.scale_to_bound(eval(rec["expr"]))
    else:
        lo, up = w.scale_to_bound(eval(rec["expr"]))
        assert [float(v) for v in lo] == rec["lower"] and [float(v) for v in up] == rec["upper"]


@pytest.mark.parametrize("rec", CASES["marching_cubes"], ids=lambda r: r["name"])
def test_marching_cubes_wrapper(wrapper, rec):
    w, calls = wrapper
    kw = dict(rec["kwargs"])
    if isinstance(kw.get("scale"), list) and len(kw["scale"]) == 2 and isinstance(kw["scale"][0], list):
        kw["scale"] = (kw["scale"][0], kw["scale"][1])
    buf = io.StringIO()
    if "raises" in rec:
        with pytest.raises(eval(rec["raises"])) as ei, contextlib.redirect_stdout(buf):
            w.marching_cubes(eval(rec["grid"]), eval(rec["thresh"]), **kw)
        assert str(ei.value) == rec["message"]
    else:
        with contextlib.redirect_stdout(buf):
            v, f = w.marching_cubes(eval(rec["grid"]), eval(rec["thresh"]), **kw)
        want = dict(rec["call"])
        if rec["name"] == "f16_tensor":
            # the ONE deliberate difference at the native boundary: the reference up-casts a float16 grid before its C++
            # entry sees it (marching_cubes.py:87); here it stays float16 -- the HIP library reads fp16 natively and
            # classifies exactly like the up-cast (tests/test_gpu_parity.py::test_fp16_grid_through_the_wrapper), so the
            # result is the same mesh without the copy
            assert want["dtype"] == "torch.float32"
            want["dtype"] = "torch.float16"
        assert calls[0] == want
    assert buf.getvalue() == rec["stdout"]


@pytest.mark.parametrize("rec", CASES["save_mesh"], ids=lambda r: r["name"])
def test_save_mesh_wrapper(wrapper, rec):
    from pathlib import Path  # noqa: F811
    w, calls = wrapper
    kw = rec["kwargs"]
    v, f, args = torch.zeros((2, 3)), torch.zeros((1, 3), dtype=torch.int64), {}
    if kw.get("np"):
        v, f = v.numpy(), f.numpy()
        args["colors"] = np.full((2, 3), 200.7)
    fn = eval(kw["filename"]) if kw["filename"].startswith("Path") else kw["filename"]
    buf = io.StringIO()
    if "raises" in rec:
        with pytest.raises(eval(rec["raises"])):
            w.save_mesh(v, f, filename=fn, verbose=kw.get("verbose", False), **args)
    else:
        with contextlib.redirect_stdout(buf):
            w.save_mesh(v, f, filename=fn, verbose=kw.get("verbose", False), **args)
        assert calls[0] == rec["call"]
    assert buf.getvalue() == rec["stdout"]


def test_ply_writer_bytes(built, tmp_path):
    """save_mesh_as_ply layout (marching_cubes.cu:318-349): header text, xyz f32 + rgb u8 per vertex,
    [3, i, j, k] int32 per face."""
    v = torch.tensor([[0.0, 1.0, 2.0], [3.5, -4.0, 5.25]])
    f = torch.tensor([[0, 1, 1]], dtype=torch.int32)
    c = torch.tensor([[1, 2, 3], [250, 251, 252]], dtype=torch.uint8)
    p = tmp_path / "m.ply"
    built.libPrim3D.save_mesh_as_ply(str(p), v, f, c)
    data = p.read_bytes()
    head = (b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty float x\nproperty float y\n"
            b"property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nelement face 1\n"
            b"property list int int vertex_index\nend_header\n")
    assert data.startswith(head)
    body = data[len(head):]
    expect = b"".join(np.float32(v[i].numpy()).tobytes() + bytes(c[i].tolist()) for i in range(2)) \
        + np.array([3, 0, 1, 1], dtype=np.int32).tobytes()
    assert body == expect


def test_ply_writer_matches_the_restated_reference_writer(built, tmp_path):
    """Random mesh through save_mesh (default colours and given colours) == oracle/ply_oracle.py byte for byte."""
    from oracle.ply_oracle import reference_ply_bytes
    rng = np.random.default_rng(5)
    v = rng.standard_normal((257, 3)).astype(np.float32)
    f = rng.integers(0, 257, size=(511, 3)).astype(np.int32)
    c = rng.integers(0, 256, size=(257, 3)).astype(np.uint8)
    p = tmp_path / "a.ply"
    built.save_mesh(v, f, c, filename=p)
    assert p.read_bytes() == reference_ply_bytes(v, f, c)
    built.save_mesh(torch.from_numpy(v), torch.from_numpy(f).long(), filename=str(p))   # wrapper: faces.int(), grey
    assert p.read_bytes() == reference_ply_bytes(v, f, np.full_like(c, 127))


def test_ply_writer_rejects_other_dtypes_like_the_reference(built, tmp_path):
    """marching_cubes.cu:334-347 reads the buffers with data_ptr<float>() / <int32_t>() / <uint8_t>(): other dtypes
    raise there, and here (no silent conversion); non-contiguous inputs raise the CHECK_CONTIGUOUS message."""
    C = built.libPrim3D
    v = torch.zeros(4, 3)
    f = torch.zeros(2, 3, dtype=torch.int32)
    c = torch.zeros(4, 3, dtype=torch.uint8)
    name = str(tmp_path / "x.ply")
    with pytest.raises(RuntimeError, match="expected scalar type Float"):
        C.save_mesh_as_ply(name, v.double(), f, c)
    with pytest.raises(RuntimeError, match="expected scalar type Int"):
        C.save_mesh_as_ply(name, v, f.long(), c)
    with pytest.raises(RuntimeError, match="expected scalar type Byte"):
        C.save_mesh_as_ply(name, v, f, c.float())
    with pytest.raises(RuntimeError, match="must be contiguous"):
        C.save_mesh_as_ply(name, torch.zeros(3, 4).t(), f, c)
    with pytest.raises(NotImplementedError):
        built.save_mesh(v, f, filename="mesh.obj")


def test_reference_module_paths_resolve(built):
    """Callers that import the reference's sub-modules directly (prim3d.utility.*, prim3d.misc.*, prim3d.version) keep
    working against the alias package."""
    import prim3d
    from prim3d.misc import Timer
    from prim3d.misc.utils import TimerError, scale_to_bound
    from prim3d.utility import create_raycaster, marching_cubes, marching_tetrahedras, save_mesh
    from prim3d.utility.marching_cubes import marching_cubes as mc2
    from prim3d.utility.marching_tetrahedras import marching_tetrahedras as mt2
    from prim3d.utility.ray_cast import create_raycaster as rc2
    from prim3d.version import __version__
    assert marching_cubes is prim3d.marching_cubes is mc2 and save_mesh is prim3d.save_mesh
    assert marching_tetrahedras is prim3d.marching_tetrahedras is mt2 and create_raycaster is prim3d.create_raycaster is rc2
    assert Timer is prim3d.Timer and issubclass(TimerError, Exception) and __version__ == prim3d.__version__
    assert scale_to_bound(2.0) == ([0.0, 0.0, 0.0], [2.0, 2.0, 2.0])
    assert set(prim3d.__all__) == {"__version__", "ENABLE_OPTIX", "Timer", "create_raycaster", "marching_cubes", "save_mesh",
                                   "marching_tetrahedras"}   # prim3d/__init__.py:12-16 of the reference


def test_package_surface(built):
    import prim3d
    assert prim3d.ENABLE_OPTIX is False and prim3d.__version__
    assert prim3d.libPrim3D is built.libPrim3D
    for name in ("marching_cubes", "save_mesh_as_ply", "test", "create_raycaster", "RayCaster", "enable_optix"):
        assert hasattr(prim3d.libPrim3D, name)
    with prim3d.Timer("took {:.6f}s"):
        pass
