#!/usr/bin/env python3
"""Write tests/golden/small_meshes.npz: canonical meshes of the seeded small cases (tests/cases.py)
from the CPU oracle.  Deterministic: rerunning must reproduce the file byte-for-byte in content."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import canonical_mesh, oracle_extract  # noqa: E402
from tests.cases import small_cases  # noqa: E402


def main():
    out = {}
    for name, (g, thresh, lower, upper) in sorted(small_cases().items()):
        k, v, f = canonical_mesh(*oracle_extract(g, thresh, lower, upper))
        out[name + "__keys"] = k
        out[name + "__verts"] = v
        out[name + "__faces"] = f
    np.savez_compressed(ROOT / "tests" / "golden" / "small_meshes.npz", **out)
    print("wrote", len(out) // 3, "cases")


if __name__ == "__main__":
    main()
