"""dev: randomized parity fuzz of the one-pass call, the scan-numbered pair, the two-pass pair (p3d_mc_count +
p3d_mc_emit) and the one-pass call with a predicted region layout (p3d_mc_slab.region_first_rows: the regions' true totals
with a random error each -- exact, too few rows, too many, none) against the oracle (small random shapes)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from oracle import oracle_extract
from tests.test_gpu_parity import _assert_same_mesh, _hip_extract, _hip_extract_fused, _hip_extract_pair
from oracle import canonical_mesh
from primitive3d_amd import capi
from tests.test_gpu_layout import _layout_call, _mesh, _same
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
n_flag4 = 0


def layout_leg(g, thresh, ref, dt):
    """the same mesh through a layout made from the true region totals, each off by a random amount"""
    global n_flag4
    t = torch.from_numpy(np.ascontiguousarray(g)).to(gpu).to(dt)
    up = [float(v) for v in t.shape]
    # the mesh's size and the region totals from the count-only pass (no scratch whose regions an uneven field could overflow)
    ws0 = torch.empty(capi.workspace_bytes(*t.shape), dtype=torch.uint8, device=gpu)
    capi.count(t, thresh, ws0)
    nv, nf, fl0, regions = capi.read_counts_ex(ws0)
    assert fl0 == 0 and sum(regions) == nv, (fl0, nv, sum(regions))
    if os.environ.get("DEBUG_FIRST"):   # what a scratch-mode call sized 2 x voxels reports on the same field
        capv = max(4096, 2 * t.numel())
        v = torch.empty((capv, 3), device=gpu); f = torch.empty((2 * capv, 3), dtype=torch.int32, device=gpu)
        capi.extract_fused_raw(t, thresh, [0.0] * 3, up, ws0, v, f)
        r = capi.read_counts_ex(ws0)
        if r[2] or sum(r[3]) != r[0]:
            print("scratch-mode call:", tuple(t.shape), "V", r[0], "capv", capv, "flags", r[2], "region max", max(r[3]), "sum", sum(r[3]))
    mode = int(rng.integers(0, 4))
    err = {0: 0.0, 1: 0.02, 2: 0.3, 3: 1.0}[mode]   # exact / a slowly changing field / a different one / anything
    extra = [int(round(r * err * rng.uniform(-1, 1))) for r in regions]
    extra = [max(-r, e) for r, e in zip(regions, extra)]
    spill = int(rng.choice([64, nv // 8 + 4096, 2 * nv + 4096]))
    first, rows = capi.region_layout(regions, spill, extra)
    ws, v, f, nv2, nf2, flags, regions2 = _layout_call(capi, t, thresh, [0.0] * 3, up, first, rows, nf, guard=16)
    assert (v[rows:] == -7.0).all()
    if flags & 4:   # a spill area overflowed: the counts are right all the same; the caller re-emits
        n_flag4 += 1
        assert (nv2, nf2) == (nv, nf)
        v = torch.empty((nv, 3), device=gpu); f = torch.empty((max(nf, 1), 3), dtype=torch.int32, device=gpu)
        if nv and nf:
            capi.emit(t, thresh, [0.0] * 3, up, ws, v, f[:nf], None)
    else:
        assert (nv2, nf2, flags) == (nv, nf, 0) and list(regions2) == list(regions)
    if nv and nf:
        _same(_mesh(capi, ws, tuple(t.shape), v[:nv], f[:nf]), canonical_mesh(*ref))

gpu = torch.device("cuda", 0)
n_ok = 0
for it in range(int(os.environ.get("N", "60"))):
    rx, ry = int(rng.integers(2, 40)), int(rng.integers(2, 40))
    rz = int(rng.choice([rng.integers(2, 70), rng.integers(60, 140), rng.integers(129, 257), rng.integers(250, 700),
                         rng.integers(2040, 2200)]))
    if os.environ.get("RZ"):   # e.g. RZ=129,257: only rows of that length range
        lo, hi = (int(v) for v in os.environ["RZ"].split(","))
        rz = int(rng.integers(lo, hi))
    kind = rng.integers(0, 3)
    if kind == 0:
        g = rng.standard_normal((rx, ry, rz)).astype(np.float32)
    elif kind == 1:
        g = rng.integers(-2, 3, size=(rx, ry, rz)).astype(np.float32)
    else:
        x, y, z = np.meshgrid(np.arange(rx), np.arange(ry), np.arange(rz), indexing="ij")
        g = (np.sin(x * 0.7) + np.cos(y * 0.5) + np.sin(z * 0.11) + rng.standard_normal() * 0.3).astype(np.float32)
    thresh = float(rng.uniform(-0.5, 0.5))
    dt = torch.float32
    if os.environ.get("DTYPE") == "f16":   # fp16 grid: the oracle sees the up-cast values (marching_cubes.py:87)
        g = g.astype(np.float16)
        ref = oracle_extract(g.astype(np.float32), thresh)
        dt = torch.float16
    else:
        ref = oracle_extract(g, thresh)
    _assert_same_mesh(_hip_extract_fused(gpu, g, thresh, None, None, dtype=dt), ref)
    _assert_same_mesh(_hip_extract(gpu, g, thresh, None, None, dtype=dt), ref)
    _assert_same_mesh(_hip_extract_pair(gpu, g, thresh, None, None, dtype=dt), ref)   # (count-only pass + second streaming pass)
    layout_leg(g, thresh, ref, dt)
    n_ok += 1
print("fuzz ok:", n_ok, "cases (each also through a predicted region layout;", n_flag4, "of them overflowed a spill area and were re-emitted)")
