"""Host logic of the predicted region layout (p3d_mc_slab.region_first_rows, include/p3d_mc.h): capi.region_layout turns the 32
region totals of an earlier call into the 41 ascending rows the library validates -- no GPU needed (the kernels' side of it is
tests/test_gpu_layout.py; the reference has no counterpart: its slot IS the final row, marching_cubes.cu:104-109)."""
import numpy as np

from primitive3d_amd import capi


def _rows(arr):
    return [int(v) for v in arr]


def test_regions_get_exactly_their_totals_and_eight_spill_areas_follow():
    rng = np.random.default_rng(0)
    totals = [int(v) for v in rng.integers(0, 200000, size=32)]
    arr, rows = capi.region_layout(totals)
    first = _rows(arr)
    assert len(first) == 41 and first[0] == 0 and first[40] == rows
    assert [first[r + 1] - first[r] for r in range(32)] == totals          # no slack: an unchanged field moves nothing
    spill = [first[33 + g] - first[32 + g] for g in range(8)]
    assert sum(spill) == sum(totals) // 10 + 4096 and max(spill) - min(spill) <= 1
    assert all(a <= b for a, b in zip(first, first[1:]))                    # ascending: what p3d_mc_extract_fused checks


def test_explicit_spill_and_per_region_errors():
    totals = [1000] * 32
    arr, rows = capi.region_layout(totals, spill_rows=64, extra=[-2000 if r == 3 else 5 for r in range(32)])
    first = _rows(arr)
    sizes = [first[r + 1] - first[r] for r in range(32)]
    assert sizes[3] == 0 and all(s == 1005 for r, s in enumerate(sizes) if r != 3)   # (a region is never given negative rows)
    assert rows == sum(sizes) + 64 and [first[33 + g] - first[32 + g] for g in range(8)] == [8] * 8


def test_an_empty_mesh_still_has_a_valid_table():
    arr, rows = capi.region_layout([0] * 32)
    first = _rows(arr)
    assert first[:33] == [0] * 33 and rows == 4096 and first[40] == 4096
