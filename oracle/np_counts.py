"""Independent numpy count of vertices / faces: a second, vectorised
statement of count_vertices_faces_kernel (marching_cubes.cu:4-68) used to cross-check the C oracle.
TEST INFRASTRUCTURE ONLY."""
import re
from pathlib import Path

import numpy as np

_INC = Path(__file__).resolve().parents[1] / "primitive3d_amd" / "csrc" / "tri_table_packed.inc"


def tri_counts() -> np.ndarray:
    """#triangles per corner mask, from the packed table (nibble 0xF terminates a row)."""
    text = _INC.read_text()
    words = [int(w, 16) for w in re.findall(r"0x([0-9a-f]{16})ull", text)]
    assert len(words) == 256
    out = np.zeros(256, dtype=np.int64)
    for m, w in enumerate(words):
        n = 0
        while n < 15 and ((w >> (4 * n)) & 0xF) != 0xF:
            n += 3
        out[m] = n // 3
    return out


def np_count(grid, thresh):
    g = np.asarray(grid).astype(np.float32)
    inside = g > np.float32(thresh)  # strict >, NaN -> outside (marching_cubes.cu:25)
    v = int((inside[1:] != inside[:-1]).sum() + (inside[:, 1:] != inside[:, :-1]).sum()
            + (inside[:, :, 1:] != inside[:, :, :-1]).sum())
    c = inside.astype(np.int64)
    corners = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]  # :49-57
    sx, sy, sz = (s - 1 for s in g.shape)
    mask = np.zeros((max(sx, 0), max(sy, 0), max(sz, 0)), dtype=np.int64)
    for bit, (dx, dy, dz) in enumerate(corners):
        mask |= c[dx:dx + sx, dy:dy + sy, dz:dz + sz] << bit
    f = int(tri_counts()[mask].sum())
    active = int(((mask != 0) & (mask != 255)).sum())
    return v, f, active
