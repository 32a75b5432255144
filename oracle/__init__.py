"""CPU oracle for the marching-cubes hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product path (primitive3d_amd/) must never import it.
"""
from .oracle import (canonical_mesh, oracle_count, oracle_extract, oracle_tri_table,  # noqa: F401
                     build_oracle)
