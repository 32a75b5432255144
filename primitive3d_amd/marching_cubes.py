"""Python wrapper of the marching-cubes path: same signature, defaults, coercions, exceptions and
prints as the reference's prim3d/utility/marching_cubes.py:10-141, dispatching to the HIP build of
`libPrim3D` (csrc/bindings.cpp).  Nothing here computes; there is no fallback if the native module
is missing (import fails)."""
from pathlib import Path
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import libPrim3D as _C


def scale_to_bound(scale: Union[float, Sequence]) -> Tuple[List[float]]:
    """reference: prim3d/utility/marching_cubes.py:10-31 (python float -> cube, len-3 -> upper,
    len-2 -> (lower, upper) scalars or triples; anything else, including int scalars, TypeError)."""
    if isinstance(scale, float):
        return [0.0, 0.0, 0.0], [scale, scale, scale]
    if not isinstance(scale, (list, tuple, np.ndarray, torch.Tensor)):
        raise TypeError()
    n = len(scale)
    if n == 3:
        return [0.0, 0.0, 0.0], [v for v in scale]
    if n != 2:
        raise TypeError()
    if isinstance(scale[0], float):
        return [scale[0]] * 3, [scale[1]] * 3
    assert len(scale[0]) == len(scale[1]) == 3
    return [v for v in scale[0]], [v for v in scale[1]]


def _bounding_box(shape, scale):
    """(lower, upper) of the box the grid spans: [0, shape] when no scale is given (upper keeps Python ints then, as
    in the reference, :59-62), else whatever scale_to_bound makes of `scale`."""
    if scale is not None:
        return scale_to_bound(scale)
    return [0.0, 0.0, 0.0], [shape[0], shape[1], shape[2]]


def _via_pymcubes(density_grid, thresh, lower, upper):
    """The reference's CPU branch (:66-81): third-party PyMCubes plus its rescaling, including the division by the
    voxel size (:78).  PyMCubes is not a dependency here: without it the reference's own ImportError is raised."""
    try:
        import mcubes
    except:  # noqa: E722  (any failure of the import counts, as in the reference, :69)
        raise ImportError("the cpu mode cumcubes is the wrapper of `mcubes`, please install the mcubes")
    host_grid = density_grid.detach().cpu().numpy()
    verts, tris = mcubes.marching_cubes(host_grid, thresh)
    voxel = (np.array(upper) - np.array(lower)) / np.array(host_grid.shape)
    return torch.tensor(verts / voxel + np.array(lower)), torch.tensor(tris.astype(np.int64))


def _on_device_as_float32(density_grid):
    """ndarray -> tensor -> device -> float32 (:84-87; a non-contiguous tensor stays non-contiguous and is rejected
    by the native module, like in the reference); grids thinner than 2 samples along any axis: bare ValueError.
    A float16 grid stays float16: the native module reads it as it is and classifies it exactly like its up-cast
    (`float(v) > thresh`, the conversion is exact), so the mesh is the one the reference's `.to(torch.float32)` gives --
    without the copy and with half the bytes read."""
    if isinstance(density_grid, np.ndarray):
        density_grid = torch.tensor(density_grid)
    density_grid = density_grid.cuda()
    if density_grid.dtype != torch.float16:
        density_grid = density_grid.to(torch.float32)
    if min(density_grid.shape[0], density_grid.shape[1], density_grid.shape[2]) < 2:
        raise ValueError()
    return density_grid


def marching_cubes(
    density_grid: Union[torch.Tensor, np.ndarray],
    thresh: float,
    scale: Optional[Union[float, Sequence]] = None,
    verbose: bool = False,
    cpu: bool = False,
) -> Tuple[torch.Tensor]:
    """Same call as the reference's wrapper (prim3d/utility/marching_cubes.py:34-98): returns
    (vertices f32 [V,3], faces i32 [F,3]) on the device; `scale` as in scale_to_bound; `verbose` prints the two
    counts; `cpu=True` (or no GPU) takes the PyMCubes branch, which yields f64 vertices / i64 faces on the host."""
    lower, upper = _bounding_box(density_grid.shape, scale)
    if cpu or not torch.cuda.is_available():
        vertices, faces = _via_pymcubes(density_grid, thresh, lower, upper)
    else:
        vertices, faces = _C.marching_cubes(_on_device_as_float32(density_grid), thresh, lower, upper)
    if verbose:
        print(f"#vertices={vertices.shape[0]}")
        print(f"#triangles={faces.shape[0]}")
    return vertices, faces


def save_mesh(
    vertices: Union[torch.Tensor, np.ndarray],
    faces: Union[torch.Tensor, np.ndarray],
    colors: Optional[Union[torch.Tensor, np.ndarray]] = None,
    filename: Union[str, Path] = "temp.ply",
    verbose: bool = False,
) -> None:
    """Same call as the reference's save_mesh (:100-141): int32 faces, uint8 colours (mid grey when none are given),
    only the .ply format exists."""
    as_tensor = lambda a: torch.tensor(a) if isinstance(a, np.ndarray) else a  # noqa: E731
    name = str(filename) if isinstance(filename, Path) else filename
    vertices = as_tensor(vertices)
    faces = as_tensor(faces).int()
    colors = (torch.ones_like(vertices) * 127 if colors is None else as_tensor(colors)).to(torch.uint8)
    if not name.endswith(".ply"):
        raise NotImplementedError()
    _C.save_mesh_as_ply(name, vertices, faces, colors)
    if verbose:
        print(f"save as {name} successfully!")


def marching_cubes_batched(density_grids, thresh: float, scale=None):
    """Per-frame extraction of a batch of grids (BASELINE.json config 5: NeRF-style density grids).

    NOT in the reference (its C++ entry rejects 4-D input, marching_cubes.cu:219, and its wrapper
    up-casts to fp32, marching_cubes.py:87): `density_grids` is a [B, rx, ry, rz] CUDA tensor of
    float32 **or float16**; fp16 grids are read as fp16 (half the HBM bytes) and compared in fp32,
    which is exactly the reference applied to `density_grids[b].float()` (the conversion is exact).
    The whole batch is ONE call of the native library (include/p3d_mc.h: p3d_mc_extract_fused_batched: one
    streaming launch, one counting launch and one face launch for all items).
    Returns (vertices [sumV,3] f32, faces [sumF,3] i32 with per-item LOCAL vertex ids,
    vertex_offsets [B+1] i64, face_offsets [B+1] i64 -- all four ON THE DEVICE, like the meshes: the call does not
    synchronise; slicing with the offsets (`v[vo[b]:vo[b+1]]`) reads them back and so waits for the call's GPU work).
    A batch whose total vertex or face count exceeds int32 is extracted item by item (the limit applies per item).
    """
    from . import capi
    if isinstance(density_grids, np.ndarray):
        density_grids = torch.tensor(density_grids)
    density_grids = density_grids.cuda()
    if density_grids.dtype not in (torch.float16, torch.float32):
        density_grids = density_grids.to(torch.float32)
    if density_grids.dim() != 4 or min(density_grids.shape[1:]) < 2:
        raise ValueError()
    density_grids = density_grids.contiguous()
    B = density_grids.shape[0]
    shape = tuple(density_grids.shape[1:])
    if scale is None:
        lower, upper = [0.0, 0.0, 0.0], [shape[0], shape[1], shape[2]]
    else:
        lower, upper = scale_to_bound(scale)
    dev = density_grids.device
    if B == 0:
        z = torch.zeros(1, dtype=torch.int64, device=dev)
        return (torch.empty((0, 3), dtype=torch.float32, device=dev), torch.empty((0, 3), dtype=torch.int32, device=dev),
                z, z.clone())
    nvox = B * shape[0] * shape[1] * shape[2]
    ws = torch.empty(capi.workspace_bytes_batched(B, *shape), dtype=torch.uint8, device=dev)
    offs = torch.empty(2 * (B + 1), dtype=torch.int64, device=dev)
    key = (dev.index, B) + shape
    hint = _BATCH_HINTS.get(key)
    if hint is not None and hint[0] == "per_item":   # (an earlier call found the batch's totals beyond int32)
        # the marker expires: after _PER_ITEM_CALLS calls the one-launch path is tried again -- later batches of the shape
        # may be sparser (it used to stay for the life of the process)
        if hint[1] > 1:
            _BATCH_HINTS[key] = ("per_item", hint[1] - 1)
        else:
            _BATCH_HINTS.pop(key, None)
        return _batched_item_by_item(density_grids, thresh, lower, upper)
    capv, capf, slack = hint if hint is not None else (max(4096, nvox // 16), 2 * max(4096, nvox // 16), 5)
    for attempt in range(3):
        per_region = (capv + 32 * B - 1) // (32 * B)
        rows = 32 * B * max(per_region * slack // 4 + 256, min(capv, 2048))
        v = torch.empty((capv, 3), dtype=torch.float32, device=dev)
        f = torch.empty((capf, 3), dtype=torch.int32, device=dev)
        scratch = torch.empty((rows, 3), dtype=torch.float32, device=dev)
        capi.extract_fused_batched_raw(density_grids, thresh, lower, upper, ws, v, scratch, f, offs)
        try:
            nv, nf, flags = capi.read_counts(ws, with_flags=True)
        except capi.P3DError as e:
            if e.code != capi.P3D_ERANGE:
                raise
            # the TOTALS of the batch exceed int32 although face ids are local to an item: item by item, every item
            # is checked against the limit on its own -- and later calls on this shape go there directly instead of
            # allocating and streaming the whole batch first
            _remember_batch(key, ("per_item", _PER_ITEM_CALLS))
            break
        fitted = nv <= capv and nf <= capf and not flags
        _remember_batch(key, (nv + nv // 8 + 4096, nf + nf // 8 + 4096, min(32, 2 * slack) if flags & 1 else slack))
        if fitted:  # (the offsets stay on the device like the meshes: no synchronisation inside the call)
            return v[:nv], f[:nf], offs[:B + 1], offs[B + 1:]
        if flags & 2:
            break  # an item numbered more than 2^26 vertices in one region: only the per-item path can renumber
        capv, capf, slack = nv, nf, min(32, max(8, 2 * slack))   # exact sizes are known now: stream once more
    return _batched_item_by_item(density_grids, thresh, lower, upper)


_BATCH_HINTS = {}   # (device, B, rx, ry, rz) -> (vertex capacity, face capacity, scratch slack in quarters) | ("per_item", calls left)
_PER_ITEM_CALLS = 16


def _remember_batch(key, hint):
    """At most 256 shapes; the oldest entry goes when a new shape arrives (dicts keep insertion order)."""
    _BATCH_HINTS.pop(key, None)
    if len(_BATCH_HINTS) >= 256:
        _BATCH_HINTS.pop(next(iter(_BATCH_HINTS)))
    _BATCH_HINTS[key] = hint


def _batched_item_by_item(density_grids, thresh, lower, upper):
    """Last resort of marching_cubes_batched (pathological region imbalance, id-space overflow): one single-grid call of
    the C ABI per item, exact re-emission when a guess was too small."""
    from . import capi
    B = density_grids.shape[0]
    vs, fs = [], []
    for b in range(B):
        v, f = capi.extract_fused(density_grids[b], thresh, lower, upper)
        vs.append(v)
        fs.append(f)
    dev = density_grids.device
    voff = torch.tensor([0] + [v.shape[0] for v in vs], dtype=torch.int64).cumsum(0).to(dev)
    foff = torch.tensor([0] + [f.shape[0] for f in fs], dtype=torch.int64).cumsum(0).to(dev)
    return torch.cat(vs), torch.cat(fs), voff, foff
